// ebcsim_vn_stream.hip — launch of the streamed value-network blocks (ebc_vn_stream.h) for the shapes they are built
// for, and the two small kernels of a decision's selection side (ebc_decision_rank, ebc_decision_apply).  Its own
// translation unit: seconds to compile beside the general blocks' minutes.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ebc_vn_stream.h"
#include "ebc_vn_stream_api.h"

namespace ebc_host {

namespace {

template <int TI, int TH, int TO, int KIN, int KH, bool GROUP, bool FINAL, bool ROWS = false>
int launch(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int O, float *y,
           const ebc::MlpExtra &ex, int relu_out, const float *x = nullptr, int K0 = 0) {
  constexpr int NW = 8;
  constexpr size_t lds = 3 * (size_t)TH * 4096 + ((size_t)TH * 32 + 2 * (size_t)TO * 32) * 4 +
                         (GROUP ? (size_t)NW * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH : 0) + (FINAL ? 0 : (size_t)NW * 32 * EBC_VN_XROW)
#ifdef EBC_VNS_TRACE
                         + (size_t)NW * (TI + TO) * 4 * 8
#endif
      ;
  static bool raised[64] = {false};  // more than the 64 KB a launch gets by default; a function attribute is per device
  if (lds > 65536 && !raised[device & 63]) {
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_stream_kernel<TI, TH, TO, NW, KIN, KH, GROUP, FINAL, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised[device & 63] = true;
  }
  const dim3 grid((unsigned)((M + 32 * NW - 1) / (32 * NW))), block(64 * NW);
  hipLaunchKernelGGL((ebc::mlp2_stream_kernel<TI, TH, TO, NW, KIN, KH, GROUP, FINAL, ROWS>), grid, block, lds, st, M, L1, L2, y, O, ex, relu_out, x, K0);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

template <int TO, bool GROUP, bool FINAL>
int launch_k(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int K0, int H, int O, float *y,
             const ebc::MlpExtra &ex, int relu_out) {
  const bool kin = K0 <= 32 * 7 - 16, kh = H <= 32 * 7 - 16;
  if (kin && kh) return launch<7, 7, TO, 1, 1, GROUP, FINAL>(device, st, M, L1, L2, O, y, ex, relu_out);
  if (kin) return launch<7, 7, TO, 1, 0, GROUP, FINAL>(device, st, M, L1, L2, O, y, ex, relu_out);
  if (kh) return launch<7, 7, TO, 0, 1, GROUP, FINAL>(device, st, M, L1, L2, O, y, ex, relu_out);
  return launch<7, 7, TO, 0, 0, GROUP, FINAL>(device, st, M, L1, L2, O, y, ex, relu_out);
}

}  // namespace

int vn_stream_launch(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int K0, int H, int O,
                     float *y, const ebc::MlpExtra &ex, int relu_out, const float *x) {
  static const bool off = [] { const char *e = getenv("EBCSIM_VN_STREAM"); return e && atoi(e) == 0; }();  // measurements: the general block
  if (off) return EBC_VN_STREAM_NA;
  // `mlp1` as SarlValueNet calls it: observation rows in, its output as the consumers' fragments and as masked pair sums
  if (!ex.frag_in && x && L1.in_tiles == 1 && K0 > 16 && L1.out_tiles == 10 && L2.out_tiles == 7 && !ex.row_bias && !ex.final_w &&
      ex.partial && !y && !ex.store_y && ex.seg_rows >= 16 && (O & 3) == 0) {
    if (H <= 32 * 10 - 16) return launch<1, 10, 7, 0, 1, false, false, true>(device, st, M, L1, L2, O, y, ex, relu_out, x, K0);
    return launch<1, 10, 7, 0, 0, false, false, true>(device, st, M, L1, L2, O, y, ex, relu_out, x, K0);
  }
  if (!ex.frag_in || ex.frag_out) return EBC_VN_STREAM_NA;
  if (L1.in_tiles != 7 || L1.out_tiles != 7) return EBC_VN_STREAM_NA;
  // the attention block as SarlValueNet calls it: fragment input, the pair's group term, a one-output third layer
  if (ex.row_bias && ex.final_w && !ex.partial && y && L2.out_tiles == 7) {
    if ((ex.H & 3) || ex.group_rows <= 0 || 31 / ex.group_rows + 2 > EBC_VN_GROUPS) return EBC_VN_STREAM_NA;  // group terms parked in LDS
    return launch_k<7, true, true>(device, st, M, L1, L2, K0, H, O, y, ex, relu_out);
  }
  // `mlp2` as SarlValueNet calls it: fragment input, the attention-weighted sums of its rows, the rows never written
  if (!ex.row_bias && !ex.final_w && ex.partial && !y && !ex.store_y && L2.out_tiles == 4 && ex.seg_rows >= 16 && (O & 3) == 0)
    return launch_k<4, false, false>(device, st, M, L1, L2, K0, H, O, y, ex, relu_out);
  return EBC_VN_STREAM_NA;
}

}  // namespace ebc_host

// ---- the selection side of a decision: values, each env's actions by value, the size of its near-best set ------------
namespace ebc {

// One workgroup per env.  Ranking by counting (A <= 1024 values in LDS: action i's rank = the number of actions with a
// larger value, or the same value and a lower index): a full sort of every env's row, a dozen element-wise launches and a
// sort in torch before.
__global__ __launch_bounds__(128) void decision_rank_kernel(const float *v, const double *reward, double discount, double bound, int A,
                                                             double *values, int *order, int *count) {
  __shared__ double vals[1024];
  __shared__ int near;
  const int e = blockIdx.x;
  if (threadIdx.x == 0) near = 0;
  for (int i = threadIdx.x; i < A; i += blockDim.x) {
    const double x = reward[(size_t)e * A + i] + discount * (double)v[(size_t)e * A + i];
    vals[i] = x;
    values[(size_t)e * A + i] = x;
    order[(size_t)e * A + i] = i;  // (a NaN among the values leaves ranks unassigned: every entry stays a valid action)
  }
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < A; i += blockDim.x) {
    const double x = vals[i];
    int rank = 0;
    for (int j = 0; j < A; ++j) rank += (vals[j] > x || (vals[j] == x && j < i)) ? 1 : 0;
    order[(size_t)e * A + rank] = i;
  }
  // the best value is the one of rank 0: found again by a max over the row (cheap), then the near-best count
  double best = vals[0];
  for (int j = 1; j < A; ++j) best = vals[j] > best ? vals[j] : best;
  const double floor_ = best - bound;
  for (int i = threadIdx.x; i < A; i += blockDim.x) mine += vals[i] >= floor_ ? 1 : 0;
  atomicAdd(&near, mine);
  __syncthreads();
  if (threadIdx.x == 0) count[e] = near;
}

// values[env][act] = reward + discount * exact for the re-evaluated candidates; the largest |exact - coarse| among them
// (non-negative floats order like their bit patterns: an atomic max on the bits; a NaN lands above every bound)
__global__ __launch_bounds__(256) void decision_apply_kernel(const float *exact, const float *v, const long long *env, const long long *act,
                                                              const double *reward, double discount, int A, int n, double *values,
                                                              unsigned *worst_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float d = 0.0f;
  if (i < n) {
    const size_t at = (size_t)env[i] * A + (size_t)act[i];
    const float x = exact[i];
    d = fabsf(x - v[at]);
    values[at] = reward[at] + discount * (double)x;
  }
  unsigned bits = __float_as_uint(d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = __shfl_xor(bits, o, 64);
    bits = other > bits ? other : bits;
  }
  if ((threadIdx.x & 63) == 0 && bits) atomicMax(worst_bits, bits);
}

}  // namespace ebc

extern "C" int ebc_decision_apply(void *stream, const float *exact, const float *v, const long long *env, const long long *act,
                                  const double *reward, double discount, int A, int n, double *values, float *worst) {
  if (!exact || !v || !env || !act || !reward || !values || !worst || A <= 0 || n < 0)
    return ebc_host::fail(EBC_ERR_INVALID, "decision_apply arguments");
  if (n == 0) return EBC_OK;
  hipLaunchKernelGGL(ebc::decision_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, exact, v, env, act, reward,
                     discount, A, n, values, reinterpret_cast<unsigned *>(worst));
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

extern "C" int ebc_decision_rank(void *stream, const float *v, const double *reward, double discount, double bound, int E, int A,
                                 double *values, int32_t *order, int32_t *count) {
  if (!v || !reward || !values || !order || !count || E < 0 || A <= 0 || A > 1024)
    return ebc_host::fail(EBC_ERR_INVALID, "decision_rank arguments (1 <= A <= 1024)");
  if (E == 0) return EBC_OK;
  hipLaunchKernelGGL(ebc::decision_rank_kernel, dim3((unsigned)E), dim3(128), 0, (hipStream_t)stream, v, reward, discount, bound, A, values,
                     order, count);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

