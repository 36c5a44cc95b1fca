// ebc_trace.h — wave timeline instrument (tools/wave_timeline.py; `make trace` builds
// libebcsim_trace.so with it).  Absent from the product library: without EBC_WAVE_TRACE every
// hook below is empty.  Each one-wave workgroup records when it started and ended (s_memrealtime:
// a constant 100 MHz clock shared by the whole device; s_memtime: shader cycles), where it ran,
// shader-cycle marks at points of interest and event counts, in the row its kernel tag and block
// index select.
#pragma once

#include <hip/hip_runtime.h>

namespace ebc {

#define EBC_TRACE_ROW 16  // u64 per wave: r0 r1 c0 c1 hw_id | marks 5..10 | counts 11..15

#ifdef EBC_WAVE_TRACE
__device__ unsigned long long *g_wave_trace;
__device__ unsigned g_wave_trace_blocks;
__device__ int g_withhold_human = -1;  // ebc_debug_withhold: the ORCA group of this flat human index never publishes
__device__ __forceinline__ unsigned long long *wave_trace_row(int tag) {
  if (!g_wave_trace || blockIdx.x >= g_wave_trace_blocks) return nullptr;
  return g_wave_trace + ((size_t)tag * g_wave_trace_blocks + blockIdx.x) * EBC_TRACE_ROW;
}
// marks and counts collect in LDS (one wave per workgroup) and leave with the wave: a global
// read-modify-write inside linearProgram2's loop would change what is being measured
__device__ __forceinline__ unsigned *wave_trace_lds() {
  __shared__ unsigned box[16];
  return box;
}
__device__ __forceinline__ bool wave_first_lane() {
  return (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1;
}
__device__ __forceinline__ void wave_mark(int k) {
  const unsigned c = (unsigned)__builtin_amdgcn_s_memtime();
  if (wave_first_lane()) wave_trace_lds()[k] = c;
}
__device__ __forceinline__ void wave_count(int k) {
  if (wave_first_lane()) wave_trace_lds()[8 + k] += 1;
}
struct WaveTrace {
  int tag;
  // nothing is kept in registers between the two stamps (the step kernel runs at its register
  // budget: two live 64-bit values would spill): the start goes straight to the row
  __device__ __forceinline__ explicit WaveTrace(int t) : tag(t) {
#if EBC_WAVE_TRACE > 1
    if (threadIdx.x < 16) wave_trace_lds()[threadIdx.x] = 0;
#endif
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long *o = wave_trace_row(t);
    if (o && threadIdx.x == 0) {
      o[0] = r0;
      o[2] = c0;
    }
  }
  __device__ __forceinline__ ~WaveTrace() {
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    unsigned long long *o = wave_trace_row(tag);
    if (o && threadIdx.x == 0) {
      o[1] = r1;
      o[3] = c1;
#if EBC_WAVE_TRACE > 1
      o[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
      const unsigned *box = wave_trace_lds();
      const unsigned c0 = (unsigned)o[2];
      for (int k = 0; k < 6; ++k) o[5 + k] = box[k] ? o[2] + (unsigned)(box[k] - c0) : 0;  // marks, shader cycles
      const unsigned long long prev_lp3 = o[12];  // the previous launch's count: how persistent is "hard"?
      for (int k = 0; k < 4; ++k) o[11 + k] = box[8 + k];
      o[15] = prev_lp3;
#endif
    }
  }
};
#if EBC_WAVE_TRACE > 1  // marks and counts inside the ORCA waves: they slow what they measure
#define EBC_MARK(k) ::ebc::wave_mark(k)
#define EBC_COUNT(k) ::ebc::wave_count(k)
#else
#define EBC_MARK(k)
#define EBC_COUNT(k)
#endif
#else
struct WaveTrace {
  __device__ __forceinline__ explicit WaveTrace(int) {}
};
#define EBC_MARK(k)
#define EBC_COUNT(k)
#endif

}  // namespace ebc
