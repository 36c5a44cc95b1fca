// ebcsim.hip — C ABI of libebcsim.so (gfx950 / MI355X): handle, device buffers, launches.
// Kernels: ebc_kernels.h; device arithmetic: ebc_device.h, ebc_orca_group.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ebc_host.h"
#include "ebc_kernels.h"

namespace {

using ebc::DevState;
using ebc::LookIO;
using ebc::StepIO;

using ebc_host::fail;
using ebc_host::g_err;

struct Handle {
  int device = 0;
  EbcParams p;
  DevState s;
  int T = 13;
  std::vector<void *> allocs;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool has_reset = false;
  double *act_scratch = nullptr;  // ebc_step_k: the ORCA robot's action when the caller keeps none
  int *robot_sim_rows = nullptr;
  ebc::RobotSim robot_sim = {nullptr, nullptr, nullptr};  // ebc_robot_orca_sim: the demonstrator's persistent rvo2 simulators
  bool faulted = false;  // a mailbox wait timed out and was reported: only ebc_reset re-arms the handle
  int orca_gs = 16;  // lanes per human of the ORCA waves
  int step_form = 3;  // the fused ORCA step: 3 = four roles, ENV with four lanes per env (default); 1 = ENV with a lane per human slot (rounds 1-2); 2 = orca_step2_kernel (experiment: ORCA groups commit their own human)
  unsigned epoch = 0;  // fused ORCA steps launched so far (StepGrid::epoch)
  std::vector<void *> pool_allocs;   // pool arrays (re-allocated by ebc_set_scene_pool)
  uint64_t *pool_grid_alloc = nullptr;
  // staging for host-location calls
  void *stage = nullptr;
  size_t stage_bytes = 0;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  double ms_sum = 0;
  int64_t launches = 0;
};

template <typename Tp>
int dev_alloc(Handle *h, Tp **out, size_t count) {
  void *ptr = nullptr;
  static const char *uncached = getenv("EBCSIM_UNCACHED_STATE");  // experiment: state the L2s do not hold
  if (uncached && uncached[0] == '1')
    HIP_TRY(hipExtMallocWithFlags(&ptr, count * sizeof(Tp) + 16, hipDeviceMallocUncached));
  else
    HIP_TRY(hipMalloc(&ptr, count * sizeof(Tp) + 16));
  HIP_TRY(hipMemset(ptr, 0, count * sizeof(Tp) + 16));
  h->allocs.push_back(ptr);
  *out = (Tp *)ptr;
  return EBC_OK;
}

int ensure_stage(Handle *h, size_t bytes) {
  if (bytes <= h->stage_bytes) return EBC_OK;
  if (h->stage) HIP_TRY(hipFree(h->stage));
  h->stage = nullptr;
  h->stage_bytes = 0;
  HIP_TRY(hipMalloc(&h->stage, bytes));
  h->stage_bytes = bytes;
  return EBC_OK;
}

// Mailboxes as every launch expects to find them (empty), and the fault word cleared.
int arm_mailboxes(Handle *h) {
  const size_t EN = (size_t)h->s.E * h->s.N, E = (size_t)h->s.E;
  // epoch tags of 0 match no launch (the epoch counter skips 0)
  HIP_TRY(hipMemsetAsync(h->s.vel, 0, EN * 16, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.env_done, 0, E * 8, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.rows_loaded, 0, E * 4, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.robot_ready, 0, E * 4, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.fault, 0, 4, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.frame, 0, E * 64, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.ract, 0, E * 64, h->stream));
  HIP_TRY(hipMemsetAsync(h->s.committed, 0, EN * 4, h->stream));
  return EBC_OK;
}

// Called with the stream idle and the fault word's value in hand: a set word is reported ONCE
// (EBC_ERR_DEVICE), cleared, and the handle refuses further steps until ebc_reset re-arms it.
int report_fault(Handle *h, unsigned fault) {
  if (!fault) return EBC_OK;
  h->faulted = true;
  (void)hipMemset(h->s.fault, 0, 4);
  return fail(EBC_ERR_DEVICE, "ORCA step: a mailbox wait timed out; the state is undefined until ebc_reset");
}

// A bump allocator over the staging buffer for one host-location call.
struct Stager {
  Handle *h;
  size_t off = 0;
  struct Out { void *host; void *dev; size_t bytes; };
  std::vector<Out> outs;
  template <typename Tp>
  Tp *out(Tp *host, size_t count) {
    if (!host) return nullptr;
    Tp *d = (Tp *)((char *)h->stage + off);
    outs.push_back({(void *)host, (void *)d, count * sizeof(Tp)});
    off += (count * sizeof(Tp) + 255) & ~(size_t)255;
    return d;
  }
  template <typename Tp>
  int in(const Tp *host, size_t count, const Tp **dev) {
    *dev = nullptr;
    if (!host) return EBC_OK;
    Tp *d = (Tp *)((char *)h->stage + off);
    off += (count * sizeof(Tp) + 255) & ~(size_t)255;
    HIP_TRY(hipMemcpyAsync(d, host, count * sizeof(Tp), hipMemcpyHostToDevice, h->stream));
    *dev = d;
    return EBC_OK;
  }
  int finish() {
    for (auto &o : outs)
      HIP_TRY(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, h->stream));
    unsigned fault = 0;  // rides with the copy-back: a host-location call never returns data of a broken step
    HIP_TRY(hipMemcpyAsync(&fault, h->s.fault, sizeof(fault), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return report_fault(h, fault);
  }
};

size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

int check_handle(void *handle, Handle **out) {
  if (!handle) return fail(EBC_ERR_INVALID, "null handle");
  *out = (Handle *)handle;
  hipError_t e = hipSetDevice((*out)->device);
  if (e != hipSuccess) return fail(EBC_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return EBC_OK;
}

int orca_blocks(const Handle *h) {
  const long humans = (long)h->s.E * h->s.N;
  const int hpw = EBC_WAVE / h->orca_gs;
  return (int)((humans + hpw - 1) / hpw);
}

int launch_orca(Handle *h) {
  const int blocks = orca_blocks(h);
#define OK_(GS) hipLaunchKernelGGL((ebc::orca_kernel<GS>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s)
#ifdef EBC_DEV_GS  // quick development builds: one group size, one row width (make dev)
  if (h->orca_gs != EBC_DEV_GS) return fail(EBC_ERR_UNSUPPORTED, "development build: one ORCA group size only");
  OK_(EBC_DEV_GS);
#else
  switch (h->orca_gs) {
    case 2: OK_(2); break;
    case 3: OK_(3); break;
    case 4: OK_(4); break;
    case 5: OK_(5); break;
    case 6: OK_(6); break;
    case 7: OK_(7); break;
    case 8: OK_(8); break;
    case 9: OK_(9); break;
    case 10: OK_(10); break;
    case 12: OK_(12); break;
    case 16: OK_(16); break;
    case 21: OK_(21); break;
    default: OK_(32); break;
  }
#endif
#undef OK_
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

template <int POLICY>
int launch_service(Handle *h, const StepIO &io) {
  const int epb = EBC_WAVE / h->s.N;
  const int blocks = (h->s.E + epb - 1) / epb;
  if (h->T == 17)
    hipLaunchKernelGGL((ebc::step_kernel<POLICY, 17>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
  else
    hipLaunchKernelGGL((ebc::step_kernel<POLICY, 13>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, io);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

template <int GS>
int launch_orca_step_gs(Handle *h, const StepIO &io, unsigned blocks, const ebc::StepGrid &g) {
  if (h->T == 17)
    hipLaunchKernelGGL((ebc::orca_step_kernel<GS, 17>), dim3((blocks + EBC_STEP_WPB - 1) / EBC_STEP_WPB), dim3(EBC_WAVE * EBC_STEP_WPB), 0, h->stream,
                       g.env_blocks, g.orca_blocks, h->s.E, h->s.N, h->s.n_magic, h->s.n_shift, (const float4 *)h->s.tile,
                       (const int *)h->s.n_humans, h->s.vel, g.epoch, h->step_form == 3 ? 1u : 0u, h->p, h->s, io, g);
  else
    hipLaunchKernelGGL((ebc::orca_step_kernel<GS, 13>), dim3((blocks + EBC_STEP_WPB - 1) / EBC_STEP_WPB), dim3(EBC_WAVE * EBC_STEP_WPB), 0, h->stream,
                       g.env_blocks, g.orca_blocks, h->s.E, h->s.N, h->s.n_magic, h->s.n_shift, (const float4 *)h->s.tile,
                       (const int *)h->s.n_humans, h->s.vel, g.epoch, h->step_form == 3 ? 1u : 0u, h->p, h->s, io, g);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int launch_orca_step_sized(Handle *h, const StepIO &io, unsigned blocks, const ebc::StepGrid &g);

template <int GS>
int launch_orca_step2_gs(Handle *h, const StepIO &io, const ebc::Step2Grid &g) {
  if (h->T == 17)
    hipLaunchKernelGGL((ebc::orca_step2_kernel<GS, 17>), dim3(g.total), dim3(EBC_WAVE), 0, h->stream, g.env1_blocks, g.orca_blocks,
                       h->s.E, h->s.N, h->s.n_magic, h->s.n_shift, (const float4 *)h->s.tile, (const int *)h->s.n_humans, g.epoch,
                       0u, 0u, 0u, h->p, h->s, io, g);
  else
    hipLaunchKernelGGL((ebc::orca_step2_kernel<GS, 13>), dim3(g.total), dim3(EBC_WAVE), 0, h->stream, g.env1_blocks, g.orca_blocks,
                       h->s.E, h->s.N, h->s.n_magic, h->s.n_shift, (const float4 *)h->s.tile, (const int *)h->s.n_humans, g.epoch,
                       0u, 0u, 0u, h->p, h->s, io, g);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

// The second form of the fused step (ebc_kernels.h: orca_step2_kernel): ENV1 (lane = env), ORCA (commits its human),
// ENV2 (lane = human slot).  The launch leaves the humans' next positions / velocities / float tile and the robots'
// next state in the second buffers: they are the current ones from here on.
int launch_orca_step2(Handle *h, const StepIO &io) {
  const int epb = EBC_WAVE / h->s.N;
  ebc::Step2Grid g;
  g.env1_blocks = (unsigned)((h->s.E + EBC_WAVE - 1) / EBC_WAVE);
  g.orca_blocks = (unsigned)orca_blocks(h);
  g.env2_blocks = (unsigned)((h->s.E + epb - 1) / epb);
  const unsigned long long blocks = (unsigned long long)g.env1_blocks + g.orca_blocks + g.env2_blocks;
  if (blocks >= 2147483648ull) return fail(EBC_ERR_UNSUPPORTED, "ORCA step grid >= 2^31 workgroups");
  if (++h->epoch == 0) h->epoch = 1;
  g.epoch = h->epoch;
  g.total = (unsigned)blocks;
  int rc;
#ifdef EBC_DEV_GS
  if (h->orca_gs != EBC_DEV_GS) return fail(EBC_ERR_UNSUPPORTED, "development build: one ORCA group size only");
  rc = launch_orca_step2_gs<EBC_DEV_GS>(h, io, g);
#else
  switch (h->orca_gs) {
    case 2: rc = launch_orca_step2_gs<2>(h, io, g); break;
    case 3: rc = launch_orca_step2_gs<3>(h, io, g); break;
    case 4: rc = launch_orca_step2_gs<4>(h, io, g); break;
    case 5: rc = launch_orca_step2_gs<5>(h, io, g); break;
    case 6: rc = launch_orca_step2_gs<6>(h, io, g); break;
    case 7: rc = launch_orca_step2_gs<7>(h, io, g); break;
    case 8: rc = launch_orca_step2_gs<8>(h, io, g); break;
    case 9: rc = launch_orca_step2_gs<9>(h, io, g); break;
    case 10: rc = launch_orca_step2_gs<10>(h, io, g); break;
    case 12: rc = launch_orca_step2_gs<12>(h, io, g); break;
    case 16: rc = launch_orca_step2_gs<16>(h, io, g); break;
    case 21: rc = launch_orca_step2_gs<21>(h, io, g); break;
    default: rc = launch_orca_step2_gs<32>(h, io, g); break;
  }
#endif
  if (rc == EBC_OK) {
    std::swap(h->s.robot, h->s.robot_n);
    std::swap(h->s.px, h->s.px_n);
    std::swap(h->s.py, h->s.py_n);
    std::swap(h->s.vx, h->s.vx_n);
    std::swap(h->s.vy, h->s.vy_n);
    std::swap(h->s.tile, h->s.tile_n);
  }
  return rc;
}

int launch_orca_step(Handle *h, const StepIO &io) {
  if (h->step_form == 2) return launch_orca_step2(h, io);
  const int epb = EBC_WAVE / h->s.N;
  const int R = h->s.N + h->s.S;
  ebc::StepGrid g;
  const unsigned state_blocks = (unsigned)((h->s.E + epb - 1) / epb);
  // form 3: the ENV role with four lanes per ENV (16 envs per wave) instead of a lane per human slot
  constexpr int envs_per_wave3 = EBC_WAVE / EBC_ENV_LANES;
  g.env_blocks = h->step_form == 3 ? (unsigned)((h->s.E + envs_per_wave3 - 1) / envs_per_wave3) : state_blocks;
  g.orca_blocks = (unsigned)orca_blocks(h);
  g.rows_epw = R <= EBC_WAVE ? (unsigned)(EBC_WAVE / R) : 1u;
  g.rows_blocks = (io.ob || io.obs_rotated) ? (unsigned)((h->s.E + g.rows_epw - 1) / g.rows_epw) : 0u;
  const unsigned long long blocks = (unsigned long long)g.env_blocks + state_blocks + g.orca_blocks + g.rows_blocks;
  if (blocks >= 2147483648ull) return fail(EBC_ERR_UNSUPPORTED, "ORCA step grid >= 2^31 workgroups");
  if (++h->epoch == 0) h->epoch = 1;  // every mailbox word is tagged with the epoch of the launch that wrote it; 0 = never
  g.epoch = h->epoch;
  g.total = (unsigned)blocks;
  const int rc = launch_orca_step_sized(h, io, blocks, g);
  // the launch leaves the robots' next state in robot_n: that is the current state from here on
  if (rc == EBC_OK) std::swap(h->s.robot, h->s.robot_n);
  return rc;
}

int launch_orca_step_sized(Handle *h, const StepIO &io, unsigned blocks, const ebc::StepGrid &g) {
#ifdef EBC_DEV_GS
  if (h->orca_gs != EBC_DEV_GS) return fail(EBC_ERR_UNSUPPORTED, "development build: one ORCA group size only");
  return launch_orca_step_gs<EBC_DEV_GS>(h, io, blocks, g);
#else
  switch (h->orca_gs) {
    case 2: return launch_orca_step_gs<2>(h, io, blocks, g);
    case 3: return launch_orca_step_gs<3>(h, io, blocks, g);
    case 4: return launch_orca_step_gs<4>(h, io, blocks, g);
    case 5: return launch_orca_step_gs<5>(h, io, blocks, g);
    case 6: return launch_orca_step_gs<6>(h, io, blocks, g);
    case 7: return launch_orca_step_gs<7>(h, io, blocks, g);
    case 8: return launch_orca_step_gs<8>(h, io, blocks, g);
    case 9: return launch_orca_step_gs<9>(h, io, blocks, g);
    case 10: return launch_orca_step_gs<10>(h, io, blocks, g);
    case 12: return launch_orca_step_gs<12>(h, io, blocks, g);
    case 16: return launch_orca_step_gs<16>(h, io, blocks, g);
    case 21: return launch_orca_step_gs<21>(h, io, blocks, g);
    default: return launch_orca_step_gs<32>(h, io, blocks, g);
  }
#endif
}

int launch_step(Handle *h, const StepIO &io, int policy) {
  if (policy == EBC_HUMAN_ORCA) return launch_orca_step(h, io);
  if (policy == EBC_HUMAN_LINEAR) return launch_service<EBC_HUMAN_LINEAR>(h, io);
  return launch_service<EBC_HUMAN_EXTERNAL>(h, io);
}

template <int POLICY>
int launch_lookahead(Handle *h, const LookIO &io) {
  const size_t lds = (size_t)io.A * h->s.N * 8 + (io.rows ? (size_t)EBC_LA_THREADS * h->T * 4 : 0);
  if (h->T == 17)
    hipLaunchKernelGGL((ebc::lookahead_kernel<POLICY, 17>), dim3(h->s.E), dim3(EBC_LA_THREADS), lds, h->stream, h->p, h->s, io);
  else
    hipLaunchKernelGGL((ebc::lookahead_kernel<POLICY, 13>), dim3(h->s.E), dim3(EBC_LA_THREADS), lds, h->stream, h->p, h->s, io);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

// Destination arrays of a scene upload: the live state or the scene pool.
struct SceneDst {
  int *n_humans;
  double *px, *py, *vx, *vy, *gx, *gy, *radius, *v_pref;
  uint8_t *type;
  int *n_static;
  double *spx, *spy, *sradius;
  uint64_t *grid;  // destination rows, written when the scene has a grid (or zeroed when `zero_grid`)
  double *robot;
};

// Pool arrays hold E reset slots + P custom scenes.  Re-allocation (ebc_set_scene_pool) keeps the
// reset slots.
int alloc_pool(Handle *h, int P) {
  ebc::ScenePool old = h->s.pool;
  std::vector<void *> old_allocs = h->pool_allocs;
  uint64_t *old_grid = h->pool_grid_alloc;
  h->pool_allocs.clear();
  ebc::ScenePool &pl = h->s.pool;
  const int E = h->s.E;
  const size_t slots = (size_t)E + P;
  const size_t PN = slots * h->s.N, PS = slots * (h->s.S ? h->s.S : 1);
  const size_t EN = (size_t)E * h->s.N, ES = (size_t)E * (h->s.S ? h->s.S : 1);
  const bool keep = !old_allocs.empty();
  auto get = [&](auto **out, size_t count, const void *from, size_t keep_count) -> int {
    void *ptr = nullptr;
    const size_t bytes = count * sizeof(**out) + 16;
    HIP_TRY(hipMalloc(&ptr, bytes));
    HIP_TRY(hipMemset(ptr, 0, bytes));
    if (keep && from) HIP_TRY(hipMemcpy(ptr, from, keep_count * sizeof(**out), hipMemcpyDeviceToDevice));
    h->pool_allocs.push_back(ptr);
    *out = reinterpret_cast<std::remove_reference_t<decltype(*out)>>(ptr);
    return EBC_OK;
  };
  int rc = EBC_OK;
#define G_(f, c, k) if (rc == EBC_OK) rc = get(&pl.f, (c), old.f, (k))
  G_(n_humans, slots, E); G_(px, PN, EN); G_(py, PN, EN); G_(vx, PN, EN); G_(vy, PN, EN); G_(gx, PN, EN);
  G_(gy, PN, EN); G_(radius, PN, EN); G_(v_pref, PN, EN); G_(type, PN, EN); G_(n_static, slots, E);
  G_(spx, PS, ES); G_(spy, PS, ES); G_(sradius, PS, ES); G_(robot, slots * 9, (size_t)E * 9);
  G_(pref, PN, EN);
#undef G_
  if (rc == EBC_OK) rc = get(&h->pool_grid_alloc, slots * h->s.G * 2, old_grid, (size_t)E * h->s.G * 2);
  pl.grid = (keep && old.grid) ? h->pool_grid_alloc : nullptr;
  pl.P = P;
  pl.stride = 0;
  pl.cursor = old.cursor;
  for (void *ptr : old_allocs) (void)hipFree(ptr);
  return rc;
}

// Copy rows of `sc` to rows `ids` (or 0..n-1) of `d`.
int upload_scene(Handle *h, const EbcScene *sc, const int32_t *ids, const SceneDst &d, bool with_grid,
                 hipMemcpyKind kind = hipMemcpyHostToDevice) {
  const int n = sc->n, N = h->s.N, S = h->s.S, G = h->s.G;
  bool contiguous = true;
  if (ids)
    for (int r = 0; r < n; ++r)
      if (ids[r] != ids[0] + r) contiguous = false;
  auto up = [&](void *dst_base, const void *src, size_t row_bytes) -> int {
    if (!src) return EBC_OK;
    if (contiguous) {
      const int e0 = ids ? ids[0] : 0;
      HIP_TRY(hipMemcpy((char *)dst_base + (size_t)e0 * row_bytes, src, (size_t)n * row_bytes, kind));
    } else {
      for (int r = 0; r < n; ++r)
        HIP_TRY(hipMemcpy((char *)dst_base + (size_t)ids[r] * row_bytes, (const char *)src + (size_t)r * row_bytes,
                          row_bytes, kind));
    }
    return EBC_OK;
  };
  int rc;
  const size_t rowN = (size_t)N * sizeof(double);
#define UP_(dst, src, bytes) if ((rc = up((void *)(dst), (const void *)(src), (bytes))) != EBC_OK) return rc
  UP_(d.n_humans, sc->n_humans, sizeof(int));
  UP_(d.px, sc->px, rowN); UP_(d.py, sc->py, rowN); UP_(d.vx, sc->vx, rowN); UP_(d.vy, sc->vy, rowN);
  UP_(d.gx, sc->gx, rowN); UP_(d.gy, sc->gy, rowN); UP_(d.radius, sc->radius, rowN);
  UP_(d.v_pref, sc->v_pref, rowN); UP_(d.type, sc->type, (size_t)N);
  UP_(d.robot, sc->robot, 9 * sizeof(double));
  if (S > 0) {
    UP_(d.n_static, sc->n_static, sizeof(int));
    UP_(d.spx, sc->spx, (size_t)S * sizeof(double)); UP_(d.spy, sc->spy, (size_t)S * sizeof(double));
    UP_(d.sradius, sc->sradius, (size_t)S * sizeof(double));
  }
  const size_t grow = (size_t)G * 2 * sizeof(uint64_t);
  if (sc->grid) {
    UP_(d.grid, sc->grid, grow);
  } else if (with_grid) {  // these rows get a free map
    const hipMemcpyKind scene_kind = kind;
    kind = hipMemcpyHostToDevice;
    std::vector<uint64_t> zeros((size_t)n * G * 2, 0);
    UP_(d.grid, zeros.data(), grow);
    kind = scene_kind;
  }
#undef UP_
  return EBC_OK;
}

int validate_scene(const Handle *h, const EbcScene *sc, const int32_t *ids, int id_limit) {
  const int N = h->s.N, S = h->s.S;
  if (!sc || sc->struct_size != sizeof(EbcScene)) return fail(EBC_ERR_INVALID, "EbcScene.struct_size");
  if (sc->n <= 0) return fail(EBC_ERR_INVALID, "scene.n");
  if (!sc->n_humans || !sc->px || !sc->py || !sc->vx || !sc->vy || !sc->gx || !sc->gy || !sc->radius ||
      !sc->v_pref || !sc->type || !sc->robot)
    return fail(EBC_ERR_INVALID, "scene array missing");
  if (S > 0 && (!sc->n_static || !sc->spx || !sc->spy || !sc->sradius))
    return fail(EBC_ERR_INVALID, "static rows missing");
  for (int r = 0; r < sc->n; ++r) {  // what the kernels assume, checked on the host
    const int e = ids ? ids[r] : r;
    if (e < 0 || e >= id_limit) return fail(EBC_ERR_INVALID, "env id out of range");
    if (sc->n_humans[r] < 0 || sc->n_humans[r] > N) return fail(EBC_ERR_INVALID, "n_humans > max_humans");
    if (S > 0 && (sc->n_static[r] < 0 || sc->n_static[r] > S)) return fail(EBC_ERR_INVALID, "n_static > max_static");
    for (int i = 0; i < sc->n_humans[r]; ++i)
      if (sc->type[(size_t)r * N + i] > EBC_CHILD) return fail(EBC_ERR_INVALID, "human type must be 0..2");
  }
  return EBC_OK;
}

}  // namespace

extern "C" {

int ebc_abi_version(void) { return EBC_ABI_VERSION; }

const char *ebc_last_error(void) { return g_err.c_str(); }

int ebc_params_default(EbcParams *p) {
  if (!p) return fail(EBC_ERR_INVALID, "null params");
  memset(p, 0, sizeof(*p));
  p->struct_size = sizeof(EbcParams);
  p->robot_kinematics = EBC_HOLONOMIC;
  p->time_step = 0.25;
  p->time_limit = 25;
  p->map_size_m = 9.0;
  p->map_resolution = 0.1;
  p->time_max = NAN;
  p->time_good = 10.0;
  p->max_goal_distance = NAN;
  p->success_reward = 1.0;
  for (int i = 0; i < 4; ++i) p->collision_penalty[i] = NAN;
  for (int i = 0; i < 3; ++i) {
    p->discomfort_dist[i] = 0.1;
    p->discomfort_factor[i] = 0.5;
  }
  p->orca_neighbor_dist = 10.0f;
  p->orca_time_horizon = 5.0f;
  p->orca_max_neighbors = 10;
  return EBC_OK;
}

int ebc_create(int device_id, int n_envs, int max_humans, int max_static, const EbcParams *params,
               void **handle_out) {
  if (!params || !handle_out) return fail(EBC_ERR_INVALID, "null argument");
  if (params->struct_size != sizeof(EbcParams))
    return fail(EBC_ERR_INVALID, "EbcParams.struct_size does not match this library");
  if (n_envs <= 0 || max_humans <= 0 || max_static < 0) return fail(EBC_ERR_INVALID, "bad dimensions");
  if ((double)n_envs * max_humans >= 2147483648.0) return fail(EBC_ERR_UNSUPPORTED, "n_envs * max_humans >= 2^31");
  if (max_humans - 1 + (params->robot_visible ? 1 : 0) > 32)
    return fail(EBC_ERR_UNSUPPORTED, "more than 32 other agents per human (ORCA group of 32 lanes)");
  if (max_humans + max_static > EBC_LA_MAX_ROWS)
    return fail(EBC_ERR_UNSUPPORTED, "max_humans + max_static > 128 observation rows");
  if (params->orca_max_neighbors > EBC_MAXNB || params->orca_max_neighbors < 0)
    return fail(EBC_ERR_UNSUPPORTED, "orca_max_neighbors > 10");
  if (params->robot_kinematics != EBC_HOLONOMIC && params->robot_kinematics != EBC_UNICYCLE)
    return fail(EBC_ERR_INVALID, "robot_kinematics");
  const int G = (int)std::nearbyint(params->map_size_m / params->map_resolution);
  if (G <= 0 || G > 128) return fail(EBC_ERR_UNSUPPORTED, "occupancy grid wider than 128 cells");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(EBC_ERR_DEVICE, "no HIP device: libebcsim has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(EBC_ERR_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  Handle *h = new Handle();
  h->device = device_id;
  h->p = *params;
  h->T = params->with_agent_type ? 17 : 13;
  DevState &s = h->s;
  memset(&s, 0, sizeof(s));
  s.E = n_envs;
  s.N = max_humans;
  s.S = max_static;
  s.G = G;
  // rvo2 holds these as floats (Agent.cpp: invTimeHorizon, invTimeStep, rangeSq)
  s.inv_time_horizon = 1.0f / params->orca_time_horizon;
  s.inv_time_step = 1.0f / (float)params->time_step;
  s.range_sq = params->orca_neighbor_dist * params->orca_neighbor_dist;
  {  // h / N for h < 2^31 as mulhi(h, magic) >> shift (N >= 2; N == 1 is the identity)
    unsigned sh = 0;
    while ((1u << sh) < (unsigned)max_humans) ++sh;
    s.n_magic = max_humans > 1 ? (unsigned)((1ull << (31 + sh)) / (unsigned)max_humans + 1) : 0;
    s.n_shift = max_humans > 1 ? sh - 1 : 0;
  }
  const size_t EN = (size_t)n_envs * max_humans, ES = (size_t)n_envs * (max_static ? max_static : 1);
  int rc = EBC_OK;
#define A_(field, cnt) if (rc == EBC_OK) rc = dev_alloc(h, &s.field, (cnt))
  A_(n_humans, n_envs); A_(px, EN); A_(py, EN); A_(vx, EN); A_(vy, EN); A_(gx, EN); A_(gy, EN);
  A_(radius, EN); A_(v_pref, EN); A_(type, EN); A_(n_static, n_envs); A_(spx, ES); A_(spy, ES);
  A_(sradius, ES); A_(robot, (size_t)n_envs * 9); A_(robot_n, (size_t)n_envs * 9); A_(time, n_envs); A_(arrival, EN);
  A_(tile, EN * 2);
  A_(done, n_envs); A_(hact, EN * 2);
  A_(vel, EN); A_(env_done, n_envs); A_(rows_loaded, n_envs); A_(robot_ready, n_envs); A_(fault, 1);  // zeroed: tag 0 = no launch
  A_(px_n, EN); A_(py_n, EN); A_(vx_n, EN); A_(vy_n, EN); A_(tile_n, EN * 2);
  A_(frame, (size_t)n_envs * 4); A_(ract, (size_t)n_envs * 4); A_(committed, EN);
#undef A_
  if (rc == EBC_OK) rc = dev_alloc(h, &s.pool.cursor, n_envs);
  if (rc == EBC_OK) rc = dev_alloc(h, &s.grid_scene, n_envs);
  if (rc == EBC_OK) rc = alloc_pool(h, 0);
  if (rc != EBC_OK) {
    for (void *ptr : h->allocs) (void)hipFree(ptr);
    delete h;
    return rc;
  }
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    for (void *ptr : h->allocs) (void)hipFree(ptr);
    delete h;
    return fail(EBC_ERR_DEVICE, "hipStreamCreate failed");
  }
  h->stream = h->own_stream;
  {
    const int others = max_humans - 1 + (params->robot_visible ? 1 : 0);
    // lanes per human = its number of others, rounded up to a size that is instantiated and never
    // to one that fits fewer humans in a wave (64 / GS): 9 others -> 9 lanes, 7 humans per wave
    static const int sizes[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
    for (int g : sizes)
      if (g >= others) { h->orca_gs = g; break; }
    const char *form = getenv("EBCSIM_STEP_FORM");  // measurements only: 1 = the four-role launch of rounds 1-2
    if (form && atoi(form) >= 1 && atoi(form) <= 3) h->step_form = atoi(form);
    const char *force = getenv("EBCSIM_ORCA_GROUP");  // measurements only: a larger group size
    if (force && atoi(force) >= h->orca_gs)
      for (int g : sizes)
        if (g >= atoi(force)) { h->orca_gs = g; break; }
  }
  *handle_out = h;
  return EBC_OK;
}

int ebc_destroy(void *handle) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  (void)hipStreamSynchronize(h->stream);
  for (void *ptr : h->allocs) (void)hipFree(ptr);
  for (void *ptr : h->pool_allocs) (void)hipFree(ptr);
  if (h->stage) (void)hipFree(h->stage);
  for (hipEvent_t ev : h->ev) (void)hipEventDestroy(ev);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return EBC_OK;
}

int ebc_set_stream(void *handle, void *hip_stream) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)hip_stream;  // NULL = the HIP null stream (torch's default stream)
  return EBC_OK;
}

int ebc_synchronize(void *handle) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned fault = 0;  // a role of the fused ORCA step gave up waiting for another (never expected)
  HIP_TRY(hipMemcpy(&fault, h->s.fault, sizeof(fault), hipMemcpyDeviceToHost));
  return report_fault(h, fault);
}

int ebc_dims(void *handle, int32_t out[5]) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  out[0] = h->s.E; out[1] = h->s.N; out[2] = h->s.S; out[3] = h->T; out[4] = h->s.G;
  return EBC_OK;
}

// env.reset of the listed envs from `sc`, whose arrays are on the host (ebc_reset) or on the device (ebc_generate_reset)
static int reset_impl(Handle *h, const int32_t *env_ids, const EbcScene *sc, hipMemcpyKind kind) {
  int rc;
  DevState &s = h->s;
  const int n = sc->n, N = s.N;
  if (n > s.E) return fail(EBC_ERR_INVALID, "scene.n out of range");
  HIP_TRY(hipStreamSynchronize(h->stream));
  ebc::ScenePool &pl = s.pool;
  if (sc->grid) pl.grid = h->pool_grid_alloc;
  // the live state ...
  const SceneDst live = {s.n_humans, s.px, s.py, s.vx, s.vy, s.gx, s.gy, s.radius, s.v_pref, s.type,
                         s.n_static, s.spx, s.spy, s.sradius, nullptr, s.robot};
  EbcScene no_grid = *sc;
  no_grid.grid = nullptr;
  if ((rc = upload_scene(h, &no_grid, env_ids, live, false, kind)) != EBC_OK) return rc;
  // ... and reset slot e of the pool (auto-reset source, and the home of env e's occupancy grid)
  const SceneDst slot = {pl.n_humans, pl.px, pl.py, pl.vx, pl.vy, pl.gx, pl.gy, pl.radius, pl.v_pref,
                         pl.type, pl.n_static, pl.spx, pl.spy, pl.sradius, h->pool_grid_alloc, pl.robot};
  if ((rc = upload_scene(h, sc, env_ids, slot, pl.grid != nullptr, kind)) != EBC_OK) return rc;
  // per-env scalars: global_time = 0, arrival = 0, done = 0 (env.py:149-151); grid and restart slot
  bool contiguous = true;
  for (int r = 0; r < n && env_ids; ++r)
    if (env_ids[r] != env_ids[0] + r) contiguous = false;
  std::vector<int> ids(n);
  for (int r = 0; r < n; ++r) ids[r] = env_ids ? env_ids[r] : r;
  const size_t rowN = (size_t)N * sizeof(double);
  if (contiguous) {
    const int e0 = ids[0];
    HIP_TRY(hipMemset(s.time + e0, 0, (size_t)n * sizeof(double)));
    HIP_TRY(hipMemset(s.arrival + (size_t)e0 * N, 0, (size_t)n * rowN));
    HIP_TRY(hipMemset(s.done + e0, 0, (size_t)n));
    HIP_TRY(hipMemcpy(s.grid_scene + e0, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    if (pl.P == 0) HIP_TRY(hipMemcpy(pl.cursor + e0, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  } else {
    for (int r = 0; r < n; ++r) {
      const int e = ids[r];
      HIP_TRY(hipMemset(s.time + e, 0, sizeof(double)));
      HIP_TRY(hipMemset(s.arrival + (size_t)e * N, 0, rowN));
      HIP_TRY(hipMemset(s.done + e, 0, 1));
      HIP_TRY(hipMemcpy(s.grid_scene + e, &e, sizeof(int), hipMemcpyHostToDevice));
      if (pl.P == 0) HIP_TRY(hipMemcpy(pl.cursor + e, &e, sizeof(int), hipMemcpyHostToDevice));
    }
  }
  {  // the float tile the ORCA waves read (orca.py:110-140), for the whole batch, and the preferred
     // velocities of the reset slots
    const size_t EN = (size_t)s.E * N;
    hipLaunchKernelGGL(ebc::pool_pref_kernel, dim3((unsigned)((EN + 255) / 256)), dim3(256), 0, h->stream, h->s, (size_t)0, EN);
    hipLaunchKernelGGL(ebc::tile_kernel, dim3((unsigned)((EN + 255) / 256)), dim3(256), 0, h->stream, h->p, h->s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  // between launches every mailbox is empty; after a reported fault they may not be: re-arm
  if (h->faulted) {
    if ((rc = arm_mailboxes(h)) != EBC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->faulted = false;
  }
  h->has_reset = true;
  return EBC_OK;
}

int ebc_reset(void *handle, const int32_t *env_ids, const EbcScene *sc) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if ((rc = validate_scene(h, sc, env_ids, h->s.E)) != EBC_OK) return rc;
  return reset_impl(h, env_ids, sc, hipMemcpyHostToDevice);
}

static int pool_impl(Handle *h, const EbcScene *sc, int stride, hipMemcpyKind kind) {
  int rc;
  DevState &s = h->s;
  if (stride < 0) return fail(EBC_ERR_INVALID, "stride");
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int P = sc->n, E = s.E;
  if (h->has_reset) {  // running episodes keep their maps: into the envs' own slots before the old slots are freed
    hipLaunchKernelGGL(ebc::rehome_grid_kernel, dim3((unsigned)E), dim3(256), 0, h->stream, h->s, h->pool_grid_alloc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if ((rc = alloc_pool(h, P)) != EBC_OK) return rc;  // keeps the E reset slots
  ebc::ScenePool &pl = s.pool;
  if (sc->grid) pl.grid = h->pool_grid_alloc;
  const size_t N = s.N, S = s.S ? s.S : 1;
  const SceneDst slot = {pl.n_humans + E, pl.px + E * N, pl.py + E * N, pl.vx + E * N, pl.vy + E * N,
                         pl.gx + E * N, pl.gy + E * N, pl.radius + E * N, pl.v_pref + E * N, pl.type + E * N,
                         pl.n_static + E, pl.spx + E * S, pl.spy + E * S, pl.sradius + E * S,
                         h->pool_grid_alloc + (size_t)E * s.G * 2, pl.robot + (size_t)E * 9};
  if ((rc = upload_scene(h, sc, nullptr, slot, false, kind)) != EBC_OK) return rc;
  {
    const size_t first = (size_t)E * N, count = (size_t)P * N;
    hipLaunchKernelGGL(ebc::pool_pref_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, h->s, first, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  pl.stride = stride % P;
  std::vector<int> cur(E);
  for (int e = 0; e < E; ++e) cur[e] = E + e % P;
  HIP_TRY(hipMemcpy(pl.cursor, cur.data(), (size_t)E * sizeof(int), hipMemcpyHostToDevice));
  return EBC_OK;
}

int ebc_set_scene_pool(void *handle, const EbcScene *sc, int stride) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if ((rc = validate_scene(h, sc, nullptr, sc ? sc->n : 0)) != EBC_OK) return rc;
  return pool_impl(h, sc, stride, hipMemcpyHostToDevice);
}

// ---- scenes generated on the device (ebc_scene_gen.h)
namespace {
struct GenBatch {  // n generated scenes in device memory, freed with the object
  std::vector<void *> allocs;
  EbcScene sc;  // device pointers
  ~GenBatch() {
    for (void *ptr : allocs) (void)hipFree(ptr);
  }
};

int validate_gen(const Handle *h, const EbcSceneGen *g, int n) {
  if (!g || g->struct_size != sizeof(EbcSceneGen)) return fail(EBC_ERR_INVALID, "EbcSceneGen.struct_size");
  if (n <= 0) return fail(EBC_ERR_INVALID, "number of scenes");
  long total = 0;
  for (int t = 0; t < 3; ++t) {
    if (g->count[t] < 0) return fail(EBC_ERR_INVALID, "EbcSceneGen.count");
    total += g->count[t];
    if (!g->count[t]) continue;
    const int r = g->rule[t];
    if (r != EBC_RULE_CIRCLE_CROSSING && r != EBC_RULE_SQUARE_CROSSING && r != EBC_RULE_SQUARE_CROSSING_OLD)
      return fail(EBC_ERR_INVALID, "EbcSceneGen.rule");
    // what the reference itself cannot run (scene_generator.py:449-457 raises for children on the circle; the
    // old square rule exists for bicycles only, :459-498)
    if (r == EBC_RULE_CIRCLE_CROSSING && t == EBC_CHILD) return fail(EBC_ERR_INVALID, "circle_crossing is not defined for children");
    if (r == EBC_RULE_SQUARE_CROSSING_OLD && t != EBC_BICYCLE) return fail(EBC_ERR_INVALID, "square_crossing_old is defined for bicycles only");
  }
  if (total > h->s.N) return fail(EBC_ERR_INVALID, "the generated scenes have more humans than max_humans");
  if (g->num_circles < 0 || g->num_walls < 0) return fail(EBC_ERR_INVALID, "EbcSceneGen.num_circles / num_walls");
  if (g->num_walls > 0 && (g->min_wall_length < 1 || g->max_wall_length < g->min_wall_length))
    return fail(EBC_ERR_INVALID, "EbcSceneGen.min_wall_length / max_wall_length");
  if (!(g->map_resolution > 0) || (int)std::nearbyint(g->map_size_m / g->map_resolution) != h->s.G)
    return fail(EBC_ERR_INVALID, "EbcSceneGen map size differs from the handle's");
  if (g->num_circles + g->num_walls > 0 && h->s.S == 0)
    return fail(EBC_ERR_INVALID, "a map with obstacles needs max_static > 0");
  return EBC_OK;
}

int generate_batch(Handle *h, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int n, GenBatch &b) {
  int rc = validate_gen(h, gen, n);
  if (rc) return rc;
  const int N = h->s.N, S = h->s.S, G = h->s.G;
  const size_t nN = (size_t)n * N, nS = (size_t)n * (S ? S : 1);
  auto get = [&](auto **out, size_t count) -> int {
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, count * sizeof(**out) + 16));
    b.allocs.push_back(ptr);
    *out = reinterpret_cast<std::remove_reference_t<decltype(*out)>>(ptr);
    return EBC_OK;
  };
  ebc::SceneRow d = {};
  uint32_t *mt = nullptr, *dseeds = nullptr;
  int *status = nullptr;
  const bool with_grid = gen->num_circles + gen->num_walls > 0;
#define G_(f, c) if (rc == EBC_OK) rc = get(&d.f, (c))
  G_(n_humans, n); G_(px, nN); G_(py, nN); G_(vx, nN); G_(vy, nN); G_(gx, nN); G_(gy, nN); G_(radius, nN); G_(v_pref, nN);
  G_(type, nN); G_(n_static, n); G_(spx, nS); G_(spy, nS); G_(sradius, nS); G_(robot, (size_t)n * 9);
  if (with_grid) G_(grid, (size_t)n * G * 2);
#undef G_
  if (rc == EBC_OK) rc = get(&mt, (size_t)n * 624);
  if (rc == EBC_OK) rc = get(&status, 1);
  if (rc == EBC_OK && seeds) rc = get(&dseeds, n);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), h->stream));
  if (seeds) HIP_TRY(hipMemcpyAsync(dseeds, seeds, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(ebc::scene_gen_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, *gen, (const uint32_t *)dseeds,
                     seed0, n, N, S, G, d, mt, status);
  HIP_TRY(hipGetLastError());
  int st = 0;
  HIP_TRY(hipMemcpyAsync(&st, status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (st & EBC_GEN_STATIC_OVERFLOW) return fail(EBC_ERR_INVALID, "a generated map has more observation rows than max_static");
  EbcScene &sc = b.sc;
  memset(&sc, 0, sizeof(sc));
  sc.struct_size = sizeof(EbcScene);
  sc.n = n;
  sc.n_humans = d.n_humans;
  sc.px = d.px; sc.py = d.py; sc.vx = d.vx; sc.vy = d.vy; sc.gx = d.gx; sc.gy = d.gy; sc.radius = d.radius; sc.v_pref = d.v_pref;
  sc.type = d.type;
  sc.n_static = d.n_static;
  sc.spx = d.spx; sc.spy = d.spy; sc.sradius = d.sradius;
  sc.grid = d.grid;
  sc.robot = d.robot;
  return EBC_OK;
}
}  // namespace

int ebc_generate_scenes(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int n, EbcSceneOut *out) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!out || out->struct_size != sizeof(EbcSceneOut)) return fail(EBC_ERR_INVALID, "EbcSceneOut.struct_size");
  GenBatch b;
  if ((rc = generate_batch(h, gen, seed0, seeds, n, b)) != EBC_OK) return rc;
  const hipMemcpyKind kind = out->location == EBC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const size_t N = h->s.N, S = h->s.S ? h->s.S : 1, G = h->s.G;
  const EbcScene &sc = b.sc;
#define OUT_(f, bytes) if (out->f && sc.f) HIP_TRY(hipMemcpy(out->f, sc.f, (size_t)n * (bytes), kind))
  OUT_(n_humans, sizeof(int)); OUT_(px, N * 8); OUT_(py, N * 8); OUT_(vx, N * 8); OUT_(vy, N * 8); OUT_(gx, N * 8); OUT_(gy, N * 8);
  OUT_(radius, N * 8); OUT_(v_pref, N * 8); OUT_(type, N); OUT_(n_static, sizeof(int)); OUT_(spx, S * 8); OUT_(spy, S * 8);
  OUT_(sradius, S * 8); OUT_(robot, 72); OUT_(grid, G * 16);
#undef OUT_
  if (out->grid && !sc.grid) {  // free maps
    if (out->location == EBC_DEVICE) HIP_TRY(hipMemset(out->grid, 0, (size_t)n * G * 16));
    else memset(out->grid, 0, (size_t)n * G * 16);
  }
  return EBC_OK;
}

int ebc_generate_reset(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int first, int n) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (first < 0 || n <= 0 || (long)first + n > h->s.E) return fail(EBC_ERR_INVALID, "env range");
  GenBatch b;
  if ((rc = generate_batch(h, gen, seed0, seeds, n, b)) != EBC_OK) return rc;
  std::vector<int32_t> ids(n);
  for (int r = 0; r < n; ++r) ids[r] = first + r;
  return reset_impl(h, ids.data(), &b.sc, hipMemcpyDeviceToDevice);
}

int ebc_generate_pool(void *handle, const EbcSceneGen *gen, uint32_t seed0, const uint32_t *seeds, int n, int stride) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  GenBatch b;
  if ((rc = generate_batch(h, gen, seed0, seeds, n, b)) != EBC_OK) return rc;
  return pool_impl(h, &b.sc, stride, hipMemcpyDeviceToDevice);
}

int ebc_set_human_actions(void *handle, int location, const double *act) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!act) return fail(EBC_ERR_INVALID, "null actions");
  const size_t bytes = (size_t)h->s.E * h->s.N * 2 * sizeof(double);
  HIP_TRY(hipMemcpyAsync(h->s.hact, act, bytes,
                         location == EBC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                         h->stream));
  if (location != EBC_DEVICE) HIP_TRY(hipStreamSynchronize(h->stream));
  return EBC_OK;
}

}  // extern "C"

namespace {

// orca_robot_kernel for every env -> d_act [E][2] (device pointer), on the handle's stream
int launch_robot_orca(Handle *h, double safety_space, double *d_act) {
  const int others = h->s.N + h->s.S;  // rows of the observation
  static const int sizes[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
  int gs = 32;
  for (int g : sizes)
    if (g >= others) { gs = g; break; }
  const int epw = EBC_WAVE / gs;
  const unsigned blocks = (unsigned)((h->s.E + epw - 1) / epw);
#define RK_(GS) hipLaunchKernelGGL((ebc::orca_robot_kernel<GS>), dim3(blocks), dim3(EBC_WAVE), 0, h->stream, h->p, h->s, safety_space, d_act, h->robot_sim)
#ifdef EBC_DEV_GS
  if (gs != 21) return fail(EBC_ERR_UNSUPPORTED, "development build: robot ORCA for 21-lane groups only");
  RK_(21);
#else
  switch (gs) {
    case 2: RK_(2); break;
    case 3: RK_(3); break;
    case 4: RK_(4); break;
    case 5: RK_(5); break;
    case 6: RK_(6); break;
    case 7: RK_(7); break;
    case 8: RK_(8); break;
    case 9: RK_(9); break;
    case 10: RK_(10); break;
    case 12: RK_(12); break;
    case 16: RK_(16); break;
    case 21: RK_(21); break;
    default: RK_(32); break;
  }
#endif
#undef RK_
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int check_robot_orca(const Handle *h, double safety_space) {
  if (!(safety_space >= 0.0)) return fail(EBC_ERR_INVALID, "safety_space");
  if (h->p.robot_kinematics != EBC_HOLONOMIC) return fail(EBC_ERR_UNSUPPORTED, "ORCA returns ActionXY: holonomic robots only");
  if (h->s.N + h->s.S > 32) return fail(EBC_ERR_UNSUPPORTED, "robot ORCA: more than 32 observation rows");
  return EBC_OK;
}

int launch_observe(Handle *h, double *d_ob, float *d_obs) {
  const size_t rows = (size_t)h->s.E * (h->s.N + h->s.S);
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  if (h->T == 17)
    hipLaunchKernelGGL((ebc::observe_kernel<17>), dim3(blocks), dim3(256), 0, h->stream, h->p, h->s, d_ob, d_obs);
  else
    hipLaunchKernelGGL((ebc::observe_kernel<13>), dim3(blocks), dim3(256), 0, h->stream, h->p, h->s, d_ob, d_obs);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

}  // namespace

extern "C" {

int ebc_robot_orca(void *handle, double safety_space, int location, double *action) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_robot_orca before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_robot_orca: the handle reported a mailbox fault; ebc_reset re-arms it");
  if (!action) return fail(EBC_ERR_INVALID, "null action");
  if ((rc = check_robot_orca(h, safety_space)) != EBC_OK) return rc;
  double *d_act = action;
  Stager st{h};
  if (location != EBC_DEVICE) {
    if ((rc = ensure_stage(h, pad256((size_t)h->s.E * 2 * 8) + 1024)) != EBC_OK) return rc;
    d_act = st.out(action, (size_t)h->s.E * 2);
  }
  if ((rc = launch_robot_orca(h, safety_space, d_act)) != EBC_OK) return rc;
  if (location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_robot_orca_sim(void *handle, int enable) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  const size_t E = h->s.E, R = (size_t)h->s.N + h->s.S;
  if (!enable) {  // the arrays stay allocated (freed with the handle); the kernels get no simulator
    h->robot_sim.rows = nullptr;
    return EBC_OK;
  }
  if (!h->robot_sim.radius) {
    int *rows = nullptr;
    if ((rc = dev_alloc(h, &rows, E)) != EBC_OK) return rc;
    if ((rc = dev_alloc(h, &h->robot_sim.radius, E * R)) != EBC_OK) return rc;
    if ((rc = dev_alloc(h, &h->robot_sim.self, E)) != EBC_OK) return rc;
    h->robot_sim_rows = rows;
  }
  h->robot_sim.rows = h->robot_sim_rows;
  HIP_TRY(hipMemset(h->robot_sim.rows, 0xff, E * sizeof(int)));  // -1: no simulator yet (a fresh policy object)
  return EBC_OK;
}

int ebc_robot_orca_sim_state(void *handle, int location, int set, int32_t *rows, float *radius, float *self) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!h->robot_sim.rows) return fail(EBC_ERR_STATE, "ebc_robot_orca_sim_state: no persistent simulator (ebc_robot_orca_sim)");
  if (!rows || !radius || !self) return fail(EBC_ERR_INVALID, "null argument");
  const size_t E = h->s.E, R = (size_t)h->s.N + h->s.S;
  HIP_TRY(hipStreamSynchronize(h->stream));
  const hipMemcpyKind kind = location == EBC_DEVICE ? hipMemcpyDeviceToDevice : (set ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  if (set) {
    HIP_TRY(hipMemcpy(h->robot_sim.rows, rows, E * sizeof(int), kind));
    HIP_TRY(hipMemcpy(h->robot_sim.radius, radius, E * R * sizeof(float), kind));
    HIP_TRY(hipMemcpy(h->robot_sim.self, self, E * 2 * sizeof(float), kind));
  } else {
    HIP_TRY(hipMemcpy(rows, h->robot_sim.rows, E * sizeof(int), kind));
    HIP_TRY(hipMemcpy(radius, h->robot_sim.radius, E * R * sizeof(float), kind));
    HIP_TRY(hipMemcpy(self, h->robot_sim.self, E * 2 * sizeof(float), kind));
  }
  return EBC_OK;
}

// A step's launch carries per-call state (the double-buffered robot, the launch counter of the mailboxes): replayed
// from a HIP graph it would run with the arguments of the capture.  Refused where it would be recorded.
static int refuse_capture(Handle *h, const char *what) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
    return fail(EBC_ERR_UNSUPPORTED, std::string(what) + ": the stream is being captured into a HIP graph; a step must be launched, not replayed");
  return EBC_OK;
}

int ebc_step_k(void *handle, const EbcStepKArgs *a) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!a || a->struct_size != sizeof(EbcStepKArgs)) return fail(EBC_ERR_INVALID, "EbcStepKArgs.struct_size");
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_step_k before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_step_k: the handle reported a mailbox fault; ebc_reset re-arms it");
  if ((rc = refuse_capture(h, "ebc_step_k")) != EBC_OK) return rc;
  if (a->K < 1) return fail(EBC_ERR_INVALID, "K");
  if (a->human_policy != EBC_HUMAN_ORCA && a->human_policy != EBC_HUMAN_LINEAR && a->human_policy != EBC_HUMAN_EXTERNAL)
    return fail(EBC_ERR_INVALID, "human_policy (a cached look-ahead holds for one step only)");
  if (a->robot_policy != EBC_ROBOT_EXTERNAL && a->robot_policy != EBC_ROBOT_LINEAR && a->robot_policy != EBC_ROBOT_ORCA)
    return fail(EBC_ERR_INVALID, "robot_policy");
  if (a->robot_policy == EBC_ROBOT_EXTERNAL && !a->robot_action) return fail(EBC_ERR_INVALID, "robot_action is NULL");
  if (a->robot_policy == EBC_ROBOT_LINEAR && h->p.robot_kinematics != EBC_HOLONOMIC)
    return fail(EBC_ERR_UNSUPPORTED, "the linear robot policy is holonomic (simulator/policy/linear.py:11)");
  if (a->robot_policy == EBC_ROBOT_ORCA && (rc = check_robot_orca(h, a->robot_safety_space)) != EBC_OK) return rc;
  if (a->flags & EBC_FLAG_BORDER) return fail(EBC_ERR_UNSUPPORTED, "ebc_step_k: no border");
  const DevState &s = h->s;
  const size_t E = s.E, R = s.N + s.S, T = h->T, K = (size_t)a->K;
  const double *d_act_in = a->robot_action;
  float *d_state = a->state_rotated, *d_obs = a->obs_rotated;
  long long *d_rows = a->n_rows;
  double *d_act_out = a->robot_action_out, *d_reward = a->reward, *d_dmin = a->dmin, *d_goal = a->dist_to_goal;
  uint8_t *d_done = a->done, *d_info = a->info;
  Stager st{h};
  if (a->location != EBC_DEVICE) {
    const size_t need = K * (pad256(E * 2 * 8) * 2 + pad256(E * 8) * 3 + pad256(E) * 2 + pad256(E * 3 * 8) +
                             pad256(E * R * T * 4) * 2) + pad256(E * 2 * 8) + 8192;
    if ((rc = ensure_stage(h, need)) != EBC_OK) return rc;
    if ((rc = st.in(a->robot_policy == EBC_ROBOT_EXTERNAL ? a->robot_action : nullptr, K * E * 2, &d_act_in)) != EBC_OK) return rc;
    d_state = st.out(a->state_rotated, K * E * R * T); d_rows = st.out(a->n_rows, K * E);
    d_act_out = st.out(a->robot_action_out, K * E * 2); d_reward = st.out(a->reward, K * E);
    d_done = st.out(a->done, K * E); d_info = st.out(a->info, K * E); d_dmin = st.out(a->dmin, K * E * 3);
    d_goal = st.out(a->dist_to_goal, K * E); d_obs = st.out(a->obs_rotated, K * E * R * T);
  }
  // the ORCA robot's action of step k needs a home when the caller does not ask for the actions
  double *orca_act = nullptr;
  if (a->robot_policy == EBC_ROBOT_ORCA && !d_act_out) {
    if (a->location != EBC_DEVICE) {
      orca_act = (double *)((char *)h->stage + st.off);
      st.off += pad256(E * 2 * 8);
    } else {
      if (!h->act_scratch) {
        void *ptr = nullptr;
        HIP_TRY(hipMalloc(&ptr, E * 2 * 8));
        h->allocs.push_back(ptr);
        h->act_scratch = (double *)ptr;
      }
      orca_act = h->act_scratch;
    }
  }
  for (size_t k = 0; k < K; ++k) {
    if (d_state && (rc = launch_observe(h, nullptr, d_state + k * E * R * T)) != EBC_OK) return rc;
    if (d_rows) {
      hipLaunchKernelGGL(ebc::row_counts_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, h->stream, h->s, d_rows + k * E);
      HIP_TRY(hipGetLastError());
    }
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.auto_reset = (a->flags & EBC_FLAG_AUTO_RESET) ? 1 : 0;
    io.robot_policy = a->robot_policy == EBC_ROBOT_LINEAR ? EBC_ROBOT_LINEAR : EBC_ROBOT_EXTERNAL;
    if (a->robot_policy == EBC_ROBOT_ORCA) {
      double *act = d_act_out ? d_act_out + k * E * 2 : orca_act;
      if ((rc = launch_robot_orca(h, a->robot_safety_space, act)) != EBC_OK) return rc;
      io.robot_action = act;
    } else if (a->robot_policy == EBC_ROBOT_EXTERNAL) {
      io.robot_action = d_act_in + k * E * 2;
    }
    io.reward = d_reward ? d_reward + k * E : nullptr;
    io.done = d_done ? d_done + k * E : nullptr;
    io.info = d_info ? d_info + k * E : nullptr;
    io.dmin = d_dmin ? d_dmin + k * E * 3 : nullptr;
    io.dist_to_goal = d_goal ? d_goal + k * E : nullptr;
    // with EBC_ROBOT_ORCA the step reads the action from where robot ORCA left it (already robot_action_out[k])
    io.robot_action_out = (a->robot_policy != EBC_ROBOT_ORCA && d_act_out) ? d_act_out + k * E * 2 : nullptr;
    io.obs_rotated = d_obs ? d_obs + k * E * R * T : nullptr;
    if ((rc = launch_step(h, io, a->human_policy)) != EBC_OK) return rc;
  }
  if (a->location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_step(void *handle, const EbcStepArgs *a) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!a || a->struct_size != sizeof(EbcStepArgs)) return fail(EBC_ERR_INVALID, "EbcStepArgs.struct_size");
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_step before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_step: the handle reported a mailbox fault; ebc_reset re-arms it");
  if ((rc = refuse_capture(h, "ebc_step")) != EBC_OK) return rc;
  if (a->human_policy < EBC_HUMAN_EXTERNAL || a->human_policy > EBC_HUMAN_CACHED)
    return fail(EBC_ERR_INVALID, "human_policy");
  if (a->robot_policy == EBC_ROBOT_LINEAR && h->p.robot_kinematics != EBC_HOLONOMIC)
    return fail(EBC_ERR_UNSUPPORTED, "the linear robot policy is holonomic (simulator/policy/linear.py:11)");
  if (a->robot_policy == EBC_ROBOT_EXTERNAL && !a->robot_action)
    return fail(EBC_ERR_INVALID, "robot_action is NULL");
  if (a->robot_policy != EBC_ROBOT_EXTERNAL && a->robot_policy != EBC_ROBOT_LINEAR)
    return fail(EBC_ERR_INVALID, "robot_policy");
  if ((a->flags & EBC_FLAG_BORDER) && !a->border) return fail(EBC_ERR_INVALID, "border is NULL");
  const DevState &s = h->s;
  const size_t E = s.E, N = s.N, R = s.N + s.S, T = h->T;
  StepIO io;
  memset(&io, 0, sizeof(io));
  io.robot_policy = a->robot_policy;
  io.auto_reset = (a->flags & EBC_FLAG_AUTO_RESET) ? 1 : 0;
  io.has_border = (a->flags & EBC_FLAG_BORDER) ? 1 : 0;
  if (io.has_border) memcpy(io.border, a->border, sizeof(io.border));
  Stager st{h};
  if (a->location == EBC_DEVICE) {
    io.robot_action = a->robot_action;
    io.reward = a->reward; io.done = a->done; io.info = a->info; io.dmin = a->dmin;
    io.dist_to_goal = a->dist_to_goal; io.robot_action_out = a->robot_action_out;
    io.human_action = a->human_action; io.ob = a->ob; io.obs_rotated = a->obs_rotated;
  } else {
    const size_t need = pad256(E * 2 * 8) * 2 + pad256(E * 8) * 2 + pad256(E) * 2 + pad256(E * 3 * 8) +
                        pad256(E * N * 2 * 8) + pad256(E * R * 5 * 8) + pad256(E * R * T * 4) + 4096;
    if ((rc = ensure_stage(h, need)) != EBC_OK) return rc;
    if ((rc = st.in(a->robot_policy == EBC_ROBOT_EXTERNAL ? a->robot_action : nullptr, E * 2,
                    &io.robot_action)) != EBC_OK)
      return rc;
    io.reward = st.out(a->reward, E); io.done = st.out(a->done, E); io.info = st.out(a->info, E);
    io.dmin = st.out(a->dmin, E * 3); io.dist_to_goal = st.out(a->dist_to_goal, E);
    io.robot_action_out = st.out(a->robot_action_out, E * 2);
    io.human_action = st.out(a->human_action, E * N * 2); io.ob = st.out(a->ob, E * R * 5);
    io.obs_rotated = st.out(a->obs_rotated, E * R * T);
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->timing) {
    if (h->ev_used + 2 > h->ev.size()) {
      for (int q = 0; q < 2; ++q) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        h->ev.push_back(ev);
      }
    }
    e0 = h->ev[h->ev_used++];
    e1 = h->ev[h->ev_used++];
    HIP_TRY(hipEventRecord(e0, h->stream));
  }
  if ((rc = launch_step(h, io, a->human_policy)) != EBC_OK) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(e1, h->stream));
  if (a->location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_observe(void *handle, int location, double *ob, float *obs_rotated) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_observe before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_observe: the handle reported a mailbox fault; ebc_reset re-arms it");
  const DevState &s = h->s;
  const size_t rows = (size_t)s.E * (s.N + s.S), T = h->T;
  double *d_ob = ob;
  float *d_obs = obs_rotated;
  Stager st{h};
  if (location != EBC_DEVICE) {
    if ((rc = ensure_stage(h, pad256(rows * 5 * 8) + pad256(rows * T * 4) + 1024)) != EBC_OK) return rc;
    d_ob = st.out(ob, rows * 5);
    d_obs = st.out(obs_rotated, rows * T);
  }
  if ((rc = launch_observe(h, d_ob, d_obs)) != EBC_OK) return rc;
  if (location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_lookahead(void *handle, const EbcLookaheadArgs *a) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!a || a->struct_size != sizeof(EbcLookaheadArgs))
    return fail(EBC_ERR_INVALID, "EbcLookaheadArgs.struct_size");
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_lookahead before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_lookahead: the handle reported a mailbox fault; ebc_reset re-arms it");
  if (!a->actions || a->n_actions <= 0) return fail(EBC_ERR_INVALID, "actions");
  if (a->n_actions > EBC_LA_MAX_ACTIONS) return fail(EBC_ERR_UNSUPPORTED, "more than 128 actions");
  if (a->human_policy != EBC_HUMAN_ORCA && a->human_policy != EBC_HUMAN_LINEAR &&
      a->human_policy != EBC_HUMAN_EXTERNAL && a->human_policy != EBC_HUMAN_CACHED)
    return fail(EBC_ERR_INVALID, "human_policy");
  if ((a->flags & EBC_FLAG_BORDER) && !a->border) return fail(EBC_ERR_INVALID, "border is NULL");
  const DevState &s = h->s;
  const size_t E = s.E, R = s.N + s.S, T = h->T, A = a->n_actions;
  LookIO io;
  memset(&io, 0, sizeof(io));
  io.A = a->n_actions;
  io.has_border = (a->flags & EBC_FLAG_BORDER) ? 1 : 0;
  if (io.has_border) memcpy(io.border, a->border, sizeof(io.border));
  Stager st{h};
  if (a->location == EBC_DEVICE) {
    io.actions = a->actions;
    io.reward = a->reward; io.done = a->done; io.info = a->info; io.dmin = a->dmin;
    io.next_ob = a->next_ob; io.rows = a->rows_rotated;
  } else {
    const size_t need = pad256(A * 2 * 8) + pad256(E * A * 8) + pad256(E * A) * 2 + pad256(E * A * 3 * 8) +
                        pad256(E * R * 5 * 8) + (a->rows_rotated ? pad256(E * A * R * T * 4) : 0) + 4096;
    if ((rc = ensure_stage(h, need)) != EBC_OK) return rc;
    if ((rc = st.in(a->actions, A * 2, &io.actions)) != EBC_OK) return rc;
    io.reward = st.out(a->reward, E * A); io.done = st.out(a->done, E * A);
    io.info = st.out(a->info, E * A); io.dmin = st.out(a->dmin, E * A * 3);
    io.next_ob = st.out(a->next_ob, E * R * 5); io.rows = st.out(a->rows_rotated, E * A * R * T);
  }
  if (a->human_policy == EBC_HUMAN_ORCA && (rc = launch_orca(h)) != EBC_OK) return rc;  // -> hact
  rc = a->human_policy == EBC_HUMAN_LINEAR ? launch_lookahead<EBC_HUMAN_LINEAR>(h, io)
                                           : launch_lookahead<EBC_HUMAN_EXTERNAL>(h, io);
  if (rc != EBC_OK) return rc;
  if (a->location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_row_counts(void *handle, int location, long long *n_rows) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!h->has_reset) return fail(EBC_ERR_STATE, "ebc_row_counts before ebc_reset");
  if (h->faulted) return fail(EBC_ERR_STATE, "ebc_row_counts: the handle reported a mailbox fault; ebc_reset re-arms it");
  if (!n_rows) return fail(EBC_ERR_INVALID, "null n_rows");
  long long *d = n_rows;
  Stager st{h};
  if (location != EBC_DEVICE) {
    if ((rc = ensure_stage(h, pad256((size_t)h->s.E * 8) + 1024)) != EBC_OK) return rc;
    d = st.out(n_rows, (size_t)h->s.E);
  }
  hipLaunchKernelGGL(ebc::row_counts_kernel, dim3((unsigned)((h->s.E + 255) / 256)), dim3(256), 0, h->stream, h->s, d);
  HIP_TRY(hipGetLastError());
  if (location != EBC_DEVICE) return st.finish();
  return EBC_OK;
}

int ebc_il_targets(void *stream, const double *reward, const uint8_t *done, const uint8_t *info, int K, int E,
                   double gamma_bar, double *values, uint8_t *keep) {
  if (!reward || !done || !info || !values || !keep || K <= 0 || E <= 0) return fail(EBC_ERR_INVALID, "il_targets arguments");
  hipLaunchKernelGGL(ebc::il_targets_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reward, done,
                     info, K, E, gamma_bar, values, keep);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_get_state(void *handle, const EbcStateView *v) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  if (!v || v->struct_size != sizeof(EbcStateView)) return fail(EBC_ERR_INVALID, "EbcStateView.struct_size");
  const DevState &s = h->s;
  const size_t E = s.E, EN = (size_t)s.E * s.N;
  const hipMemcpyKind kind = v->location == EBC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
#define CP_(dst, src, bytes) if (dst) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), kind, h->stream))
  CP_(v->px, s.px, EN * 8); CP_(v->py, s.py, EN * 8); CP_(v->vx, s.vx, EN * 8); CP_(v->vy, s.vy, EN * 8);
  CP_(v->gx, s.gx, EN * 8); CP_(v->gy, s.gy, EN * 8); CP_(v->radius, s.radius, EN * 8);
  CP_(v->v_pref, s.v_pref, EN * 8); CP_(v->type, s.type, EN); CP_(v->n_humans, s.n_humans, E * 4);
  CP_(v->robot, s.robot, E * 9 * 8); CP_(v->global_time, s.time, E * 8);
  CP_(v->arrival_time, s.arrival, EN * 8); CP_(v->done, s.done, E);
#undef CP_
  if (v->location != EBC_DEVICE) HIP_TRY(hipStreamSynchronize(h->stream));
  return EBC_OK;
}

int ebc_timing(void *handle, int enable) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  h->timing = enable != 0;
  return EBC_OK;
}

int ebc_timing_read(void *handle, int reset, double *avg_ms, int64_t *launches) {
  Handle *h;
  int rc = check_handle(handle, &h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t q = 0; q + 1 < h->ev_used; q += 2) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev[q], h->ev[q + 1]));
    h->ms_sum += ms;
    h->launches += 1;
  }
  h->ev_used = 0;
  if (avg_ms) *avg_ms = h->launches ? h->ms_sum / (double)h->launches : 0.0;
  if (launches) *launches = h->launches;
  if (reset) {
    h->ms_sum = 0;
    h->launches = 0;
  }
  return EBC_OK;
}

}  // extern "C"

#ifdef EBC_WAVE_TRACE
// tools only (libebcsim_trace.so): where the kernels leave their wave timeline, [4][blocks][EBC_TRACE_ROW] u64
extern "C" int ebc_debug_wave_trace(void *device_buffer, unsigned blocks) {
  unsigned long long *b = (unsigned long long *)device_buffer;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ebc::g_wave_trace), &b, sizeof(b)) != hipSuccess) return EBC_ERR_DEVICE;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ebc::g_wave_trace_blocks), &blocks, sizeof(blocks)) != hipSuccess) return EBC_ERR_DEVICE;
  return EBC_OK;
}
// tests only: from the next fused ORCA step on, the ORCA group of flat human index `human` (e * N + i) does
// not publish its velocity, so its consumers run into EBC_SPIN_LIMIT (lowered in this build); -1 = off
// tests / debugging only: the hand-off records of the second step form, copied to the host
extern "C" int ebc_debug_read(void *handle, int which, void *dst, size_t bytes) {
  Handle *h = (Handle *)handle;
  const void *src = which == 0 ? (const void *)h->s.frame : which == 1 ? (const void *)h->s.ract : (const void *)h->s.committed;
  (void)hipDeviceSynchronize();
  return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? EBC_OK : EBC_ERR_DEVICE;
}
extern "C" int ebc_debug_withhold(int human) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(ebc::g_withhold_human), &human, sizeof(human)) != hipSuccess) return EBC_ERR_DEVICE;
  return EBC_OK;
}
#endif
