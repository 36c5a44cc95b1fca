// ebcsim_value_net.hip — the value-network blocks of libebcsim.so (ebc_mlp2_* / ebc_pair_*): a translation unit of
// its own, compiled beside ebcsim.hip.  Kernels: ebc_value_net.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ebc_host.h"
#include "ebc_value_net.h"
#include "ebc_vn_stream_api.h"

using ebc_host::fail;

// ------------------------------------------------------------------------- value-network layers
namespace {

uint16_t bf16_rn(float v) {  // round to nearest even, as v_cvt_pk_bf16_f32
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);  // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_to_float(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float v;
  memcpy(&v, &u, 4);
  return v;
}

struct Mlp2 {
  int device = 0, K0 = 0, H = 0, O = 0;
  ebc::PackedLayer L1{}, L2{};
  ebc::F32Layer F1{}, F2{};  // the same weights transposed in float32 (ebc_mlp2_forward_f32)
  int in_frag = 0;           // EBC_MLP_IN_FRAGMENTS: the first layer packed in accumulator k order (input = fragments)
  float *final_w = nullptr;  // optional third layer with one output
  float final_b = 0.0f;
  std::vector<void *> allocs;
};

// W [out][in] (torch Linear layout) -> A fragments of mfma_f32_32x32x16_bf16, split in hi / lo.
// acc_order: the layer's input is an accumulator tile of the layer before (fragment element j of lane
// half h in k-step s is row 16 s + 8 (j >> 2) + 4 h + (j & 3)); otherwise natural order 16 s + 8 h + j.
int pack_layer(Mlp2 *m, const float *W, const float *b, int out, int in, bool acc_order, ebc::PackedLayer *L) {
  const int To = (out + 31) / 32, Ti = (in + 31) / 32;
  std::vector<uint16_t> frag((size_t)To * Ti * 2 * 2 * 64 * 8);
  for (int t = 0; t < To; ++t)
    for (int u = 0; u < Ti; ++u)
      for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int i = lane & 31, hh = lane >> 5;
            const int k = acc_order ? 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) : 16 * s + 8 * hh + j;
            const int row = t * 32 + i, colk = u * 32 + k;
            const float v = (row < out && colk < in) ? W[(size_t)row * in + colk] : 0.0f;
            const uint16_t hi = bf16_rn(v), lo = bf16_rn(v - bf16_to_float(hi));
            const size_t base = ((((size_t)t * Ti + u) * 2 + s) * 2) * 64 * 8;
            frag[base + (size_t)lane * 8 + j] = hi;
            frag[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
          }
  std::vector<float> bias((size_t)To * 2 * 16, 0.0f);
  for (int t = 0; t < To; ++t)
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) {
        const int unit = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (unit < out) bias[((size_t)t * 2 + hh) * 16 + r] = b[unit];
      }
  void *df = nullptr, *db = nullptr;
  HIP_TRY(hipMalloc(&df, frag.size() * 2));
  m->allocs.push_back(df);
  HIP_TRY(hipMalloc(&db, bias.size() * 4));
  m->allocs.push_back(db);
  HIP_TRY(hipMemcpy(df, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  L->frag = (const uint4 *)df;
  L->bias = (const float *)db;
  L->in_tiles = Ti;
  L->out_tiles = To;
  return EBC_OK;
}

// W [out][in] -> transposed [in][out padded to 64], float32 as it is
int pack_f32(Mlp2 *m, const float *W, const float *b, int out, int in, ebc::F32Layer *L) {
  const int out_pad = (out + 63) / 64 * 64;
  if (out_pad > 64 * EBC_F32_MAXU) return fail(EBC_ERR_UNSUPPORTED, "mlp2: more than 320 units in a layer");
  std::vector<float> wt((size_t)in * out_pad, 0.0f), bias((size_t)out_pad, 0.0f);
  for (int o = 0; o < out; ++o) {
    bias[o] = b[o];
    for (int k = 0; k < in; ++k) wt[(size_t)k * out_pad + o] = W[(size_t)o * in + k];
  }
  void *dw = nullptr, *db = nullptr;
  HIP_TRY(hipMalloc(&dw, wt.size() * 4));
  m->allocs.push_back(dw);
  HIP_TRY(hipMalloc(&db, bias.size() * 4));
  m->allocs.push_back(db);
  HIP_TRY(hipMemcpy(dw, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  L->wt = (const float *)dw;
  L->b = (const float *)db;
  L->in = in;
  L->out = out;
  L->out_pad = out_pad;
  return EBC_OK;
}

template <int TI, int TO, int NW, int LEAN, int KIN>
int launch_mlp2_kin(const Mlp2 *m, hipStream_t st, const float *x, int M, int relu_out, float *y, const ebc::MlpExtra &ex) {
  const size_t weights = LEAN == 2 ? (size_t)(TI + TO) * 4096 : LEAN ? (size_t)(2 * TI + TO) * 4096 : 2 * (size_t)(TI + TO) * 4096;
  const size_t lds = (weights + 31) / 32 * 32 + (size_t)m->L1.out_tiles * 32 * 4 +
                     (ex.row_bias ? (size_t)NW * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH : 0);  // + the waves' parked group terms
  static size_t raised_dev[64][2] = {{0}};  // more than the 64 KB a launch gets by default; a function attribute is per device
  size_t &raised = raised_dev[m->device & 63][ex.row_bias ? 1 : 0];
  if (lds > 65536 && lds > raised) {
    if (ex.row_bias)
      HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_split_wg_kernel<TI, TO, NW, true, LEAN, KIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else
      HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_split_wg_kernel<TI, TO, NW, false, LEAN, KIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised = lds;
  }
  constexpr int rows = 32 * NW;
  {  // the row-group sums live in the coalesced-output epilogue (ebc_value_net.h): the block must be one that takes it
    constexpr size_t a_size = (size_t)TI * 256, b_size = (size_t)TO * 256;
    constexpr size_t per_u = LEAN == 2 ? (a_size + b_size + 1) / 2 : LEAN ? (2 * a_size + b_size + 1) / 2 : a_size + b_size;
    if (ex.partial && !((m->O & 3) == 0 && (size_t)NW * 32 * EBC_VN_XROW <= per_u * 2 * 16))
      return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_reduce: this block shape has no tile epilogue");
  }
  const dim3 grid((unsigned)((M + rows - 1) / rows)), block(64 * NW);
  if (ex.row_bias)
    hipLaunchKernelGGL((ebc::mlp2_split_wg_kernel<TI, TO, NW, true, LEAN, KIN>), grid, block, lds, st, x, M, m->K0, m->L1, m->L2, relu_out, y, m->O, ex);
  else
    hipLaunchKernelGGL((ebc::mlp2_split_wg_kernel<TI, TO, NW, false, LEAN, KIN>), grid, block, lds, st, x, M, m->K0, m->L1, m->L2, relu_out, y, m->O, ex);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

// 200-wide inputs (the h1 rows of mlp2 and of the attention stack) end in the first half of their seventh tile
template <int TI, int TO, int NW, int LEAN>
int launch_mlp2_shape(const Mlp2 *m, hipStream_t st, const float *x, int M, int relu_out, float *y, const ebc::MlpExtra &ex) {
  if constexpr (TI == 7) {
    if (m->K0 <= 32 * TI - 16) return launch_mlp2_kin<TI, TO, NW, LEAN, 1>(m, st, x, M, relu_out, y, ex);
  }
  return launch_mlp2_kin<TI, TO, NW, LEAN, 0>(m, st, x, M, relu_out, y, ex);
}

template <int TI, int TO>
int launch_mlp2_to(const Mlp2 *m, hipStream_t st, const float *x, int M, int relu_out, float *y, const ebc::MlpExtra &ex) {
  // Workgroup shape by LDS: the full layout (both halves of the weights double-buffered) twice per CU where it
  // fits (4 waves each); else the lean layout (one slot for the L2 halves) twice per CU where THAT fits: two
  // independent 4-wave workgroups fill each other's waits, which one lock-stepped 8-wave workgroup cannot; else
  // 8 waves sharing one full layout (two waves per SIMD with 256 registers each: measured faster than 4 waves
  // with 512 for the widest shape, spills included).
  constexpr size_t full = 2 * (size_t)(TI + TO) * 4096, lean = (size_t)(2 * TI + TO) * 4096, half_cu = 78 * 1024;
#ifdef EBC_MLP_NO_LEAN
  constexpr bool use_lean = false;
#else
  constexpr bool use_lean = full > half_cu && lean <= half_cu;
#endif
  if (use_lean) return launch_mlp2_shape<TI, TO, 4, 1>(m, st, x, M, relu_out, y, ex);
#ifdef EBC_MLP_LEAN2  // A/B: one slot per half for shapes whose lean layout does not fit twice (7 + 7 tiles)
  constexpr size_t lean2 = (size_t)(TI + TO) * 4096;
  if (full > half_cu && lean2 + 16 * 1024 <= half_cu) return launch_mlp2_shape<TI, TO, 4, 2>(m, st, x, M, relu_out, y, ex);
#endif
  return launch_mlp2_shape<TI, TO, (full > 80 * 1024 ? 8 : 4), 0>(m, st, x, M, relu_out, y, ex);
}

template <int TI>
int launch_mlp2(const Mlp2 *m, hipStream_t st, const float *x, int M, int relu_out, float *y, const ebc::MlpExtra &ex) {
  switch (m->L2.out_tiles) {
    case 1: return launch_mlp2_to<TI, 1>(m, st, x, M, relu_out, y, ex);
    case 2: return launch_mlp2_to<TI, 2>(m, st, x, M, relu_out, y, ex);
    case 3: return launch_mlp2_to<TI, 3>(m, st, x, M, relu_out, y, ex);
    case 4: return launch_mlp2_to<TI, 4>(m, st, x, M, relu_out, y, ex);
    case 5: return launch_mlp2_to<TI, 5>(m, st, x, M, relu_out, y, ex);
    case 6: return launch_mlp2_to<TI, 6>(m, st, x, M, relu_out, y, ex);
    case 7: return launch_mlp2_to<TI, 7>(m, st, x, M, relu_out, y, ex);
    default: return fail(EBC_ERR_UNSUPPORTED, "mlp2: more than 224 outputs");
  }
}

int mlp2_dispatch(const Mlp2 *m, hipStream_t st, const float *x, int M, int relu_out, float *y, const ebc::MlpExtra &ex) {
#ifdef EBC_DEV_GS
  return fail(EBC_ERR_UNSUPPORTED, "development build: no value-network blocks");
#else
  switch (m->L1.in_tiles) {
    case 1: return launch_mlp2<1>(m, st, x, M, relu_out, y, ex);
    case 2: return launch_mlp2<2>(m, st, x, M, relu_out, y, ex);
    case 3: return launch_mlp2<3>(m, st, x, M, relu_out, y, ex);
    case 4: return launch_mlp2<4>(m, st, x, M, relu_out, y, ex);
    case 5: return launch_mlp2<5>(m, st, x, M, relu_out, y, ex);
    case 6: return launch_mlp2<6>(m, st, x, M, relu_out, y, ex);
    default: return launch_mlp2<7>(m, st, x, M, relu_out, y, ex);
  }
#endif
}

}  // namespace

extern "C" {

int ebc_mlp2_create(int device_id, int K0, int H, int O, const float *w1, const float *b1, const float *w2,
                    const float *b2, const float *w3, const float *b3, void **out) {
  return ebc_mlp2_create_ex(device_id, K0, H, O, w1, b1, w2, b2, w3, b3, 0, out);
}

int ebc_mlp2_create_ex(int device_id, int K0, int H, int O, const float *w1, const float *b1, const float *w2,
                       const float *b2, const float *w3, const float *b3, int flags, void **out) {
  if (!w1 || !b1 || !w2 || !b2 || !out) return fail(EBC_ERR_INVALID, "null argument");
  if (K0 <= 0 || H <= 0 || O <= 0 || K0 > 224 || O > 224 || H > 320) return fail(EBC_ERR_UNSUPPORTED, "mlp2 dimensions");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(EBC_ERR_DEVICE, "no HIP device: libebcsim has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(EBC_ERR_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  Mlp2 *m = new Mlp2();
  m->device = device_id; m->K0 = K0; m->H = H; m->O = O;
  m->in_frag = (flags & EBC_MLP_IN_FRAGMENTS) ? 1 : 0;
  int rc = pack_layer(m, w1, b1, H, K0, m->in_frag != 0, &m->L1);
  if (rc == EBC_OK) rc = pack_layer(m, w2, b2, O, H, true, &m->L2);
  if (rc == EBC_OK) rc = pack_f32(m, w1, b1, H, K0, &m->F1);
  if (rc == EBC_OK) rc = pack_f32(m, w2, b2, O, H, &m->F2);
  if (rc == EBC_OK && w3) {
    void *dw = nullptr;
    if (hipMalloc(&dw, (size_t)O * 4) != hipSuccess || hipMemcpy(dw, w3, (size_t)O * 4, hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(EBC_ERR_DEVICE, "mlp2: third layer upload failed");
    if (dw) m->allocs.push_back(dw);
    m->final_w = (float *)dw;
    m->final_b = b3 ? b3[0] : 0.0f;
  }
  if (rc != EBC_OK) {
    for (void *ptr : m->allocs) (void)hipFree(ptr);
    delete m;
    return rc;
  }
  *out = m;
  return EBC_OK;
}

int ebc_mlp2_forward(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                     int group_rows, float *y) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m || !x || !y || M < 0) return fail(EBC_ERR_INVALID, "mlp2 forward arguments");
  if (row_bias && group_rows <= 0) return fail(EBC_ERR_INVALID, "mlp2: group_rows");
  if (M == 0) return EBC_OK;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = (hipStream_t)stream;
  const ebc::MlpExtra ex = {row_bias, group_rows, m->H, m->final_w, m->final_b, nullptr, nullptr, 0, 1};
  return mlp2_dispatch(m, st, x, M, relu_out, y, ex);
}

int ebc_mlp2_update(void *mlp, void *stream, const float *w1, const float *b1, const float *w2, const float *b2,
                    const float *w3, const float *b3) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m || !w1 || !b1 || !w2 || !b2) return fail(EBC_ERR_INVALID, "mlp2 update arguments");
  if ((m->final_w != nullptr) != (w3 != nullptr)) return fail(EBC_ERR_INVALID, "mlp2 update: third layer given / missing");
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = (hipStream_t)stream;
  struct { const float *W, *b; int out, in, acc; ebc::PackedLayer *L; ebc::F32Layer *F; } layers[2] = {
      {w1, b1, m->H, m->K0, m->in_frag, &m->L1, &m->F1}, {w2, b2, m->O, m->H, 1, &m->L2, &m->F2}};
  for (auto &l : layers) {
    const size_t To = (l.out + 31) / 32, Ti = (l.in + 31) / 32;
    size_t n = To * Ti * 2 * 64 * 8;
    if ((size_t)l.in * l.F->out_pad > n) n = (size_t)l.in * l.F->out_pad;
    hipLaunchKernelGGL(ebc::mlp2_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, l.W, l.b, l.out, l.in, l.acc,
                       (unsigned short *)const_cast<uint4 *>(l.L->frag), const_cast<float *>(l.L->bias),
                       const_cast<float *>(l.F->wt), const_cast<float *>(l.F->b), l.F->out_pad);
    HIP_TRY(hipGetLastError());
  }
  if (w3) {
    HIP_TRY(hipMemcpyAsync(m->final_w, w3, (size_t)m->O * 4, hipMemcpyDeviceToDevice, st));
    float fb = 0.0f;  // the one scalar of the block that lives in the handle: read back (a few bytes, once per update)
    if (b3) {
      HIP_TRY(hipMemcpyAsync(&fb, b3, 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    m->final_b = fb;
  }
  return EBC_OK;
}

int ebc_mlp2_forward_f32(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                         int group_rows, float *y) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m || !x || !y || M < 0) return fail(EBC_ERR_INVALID, "mlp2 forward arguments");
  if (row_bias && group_rows <= 0) return fail(EBC_ERR_INVALID, "mlp2: group_rows");
  if (M == 0) return EBC_OK;
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = (hipStream_t)stream;
  // few rows: a workgroup per 32-row tile, its waves sharing the tile (the form a decision's re-evaluation gets); many
  // rows: four tiles per workgroup walking the hidden tiles together on weights staged in LDS.  EBCSIM_F32_FORM=1/2
  // forces the many-/few-row form (measurements).
  static const int form = [] { const char *e = getenv("EBCSIM_F32_FORM"); return e ? atoi(e) : 0; }();
  if (form == 2 || (form == 0 && M <= EBC_F32_TILE_ROWS)) {
    constexpr int NW = 8;
    const size_t tl = ((size_t)((m->K0 + 3) & ~3) * 32 + (size_t)((m->H + 31) / 32) * 1024 + NW * 32) * 4;
    static size_t raised_tile[64] = {0};
    if (tl > 65536 && tl > raised_tile[m->device & 63]) {
      HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_tile_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl));
      raised_tile[m->device & 63] = tl;
    }
    hipLaunchKernelGGL((ebc::mlp2_f32_tile_kernel<NW>), dim3((unsigned)((M + 31) / 32)), dim3(NW * 64), tl, st, x, M, m->F1, m->F2,
                       relu_out, y, row_bias, group_rows, (const float *)m->final_w, m->final_b);
    HIP_TRY(hipGetLastError());
    return EBC_OK;
  }
  const size_t lds = ((size_t)((m->K0 + 3) & ~3) * 32 + (size_t)32 * m->F2.out_pad) * 4;  // a hidden tile's W1 columns + W2 rows
  const dim3 grid((unsigned)((M + 127) / 128)), block(256);
  static size_t raised_dev[64] = {0};  // more than the 64 KB a launch gets by default (per device)
  if (lds > 65536 && lds > raised_dev[m->device & 63]) {
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_f32_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised_dev[m->device & 63] = lds;
  }
#define F32_(T2) hipLaunchKernelGGL((ebc::mlp2_f32_kernel<T2>), grid, block, lds, st, x, M, m->F1, m->F2, relu_out, y, row_bias, \
                                    group_rows, (const float *)m->final_w, m->final_b)
  switch ((m->O + 31) / 32) {
    case 1: F32_(1); break;
    case 2: F32_(2); break;
    case 3: F32_(3); break;
    case 4: F32_(4); break;
    case 5: F32_(5); break;
    case 6: F32_(6); break;
    default: F32_(7); break;
  }
#undef F32_
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_mlp2_forward_reduce(void *mlp, void *stream, const float *x, int M, int relu_out, const float *row_bias,
                            int group_rows, float *y, int seg_rows, const float *row_weight, double *partial) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m || !x || !partial || M < 0) return fail(EBC_ERR_INVALID, "mlp2 forward_reduce arguments");
  if (row_bias && group_rows <= 0) return fail(EBC_ERR_INVALID, "mlp2: group_rows");
  if (m->final_w) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_reduce: a block with a one-output third layer has no [M][O] rows");
  if (seg_rows < 16) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_reduce: groups of fewer than 16 rows (a tile would touch more than three)");
  if (m->O & 3) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_reduce: O must be a multiple of 4");
  if (M == 0) return EBC_OK;
  HIP_TRY(hipSetDevice(m->device));
  const ebc::MlpExtra ex = {row_bias, group_rows, m->H, nullptr, 0.0f, row_weight, partial, seg_rows, y ? 1 : 0};
  return mlp2_dispatch(m, (hipStream_t)stream, x, M, relu_out, y, ex);
}

int ebc_mlp2_forward_ex(void *mlp, void *stream, const EbcMlpArgs *a) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m || !a || a->struct_size != sizeof(EbcMlpArgs)) return fail(EBC_ERR_INVALID, "EbcMlpArgs.struct_size");
  if (a->M < 0 || (!a->x && !a->frag_in)) return fail(EBC_ERR_INVALID, "mlp2 forward_ex: no input");
  if ((a->frag_in != nullptr) != (m->in_frag != 0))
    return fail(EBC_ERR_INVALID, "mlp2 forward_ex: fragment input needs a block created with EBC_MLP_IN_FRAGMENTS, row input one without");
  if (a->row_bias && a->group_rows <= 0) return fail(EBC_ERR_INVALID, "mlp2: group_rows");
  if (a->partial) {
    if (m->final_w) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_ex: a block with a one-output third layer has no [M][O] rows");
    if (a->seg_rows < 16) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_ex: groups of fewer than 16 rows");
  }
  if ((a->partial || a->frag_out) && (m->O & 3)) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_ex: O must be a multiple of 4");
  if (a->frag_out && m->final_w) return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_ex: a one-output block has no tiles to hand on");
  if (!a->y && !a->partial && !a->frag_out) return fail(EBC_ERR_INVALID, "mlp2 forward_ex: no output");
  if (a->M == 0) return EBC_OK;
  HIP_TRY(hipSetDevice(m->device));
  // the tile epilogue writes rows only when asked to (store_y) once partial sums or fragments are wanted
  const int store_y = a->y ? 1 : 0;
  ebc::MlpExtra ex = {a->row_bias, a->group_rows, m->H, m->final_w, m->final_b, a->row_weight, a->partial, a->seg_rows, store_y,
                      (const uint4 *)a->frag_in, (uint4 *)a->frag_out};
  if (!a->partial && a->frag_out) {  // fragments without sums: the sums' machinery with nothing to add up to
    return fail(EBC_ERR_UNSUPPORTED, "mlp2 forward_ex: frag_out comes with partial sums (the tile epilogue)");
  }
  if (!(a->flags & EBC_MLP_GENERAL_KERNEL)) {  // the attention block at its widest: the streamed kernel (ebc_vn_stream.h)
    const int rc = ebc_host::vn_stream_launch(m->device, (hipStream_t)stream, a->M, m->L1, m->L2, m->K0, m->H, m->O, a->y, ex, a->relu_out, a->x);
    if (rc != EBC_VN_STREAM_NA) return rc;
  }
  return mlp2_dispatch(m, (hipStream_t)stream, a->x, a->M, a->relu_out, a->y, ex);
}

int ebc_pair_weights(void *stream, const float *scores, const long long *n_valid, int B, int R, float *w) {
  if (!scores || !w || B < 0 || R <= 0) return fail(EBC_ERR_INVALID, "pair_weights arguments");
  if (B == 0) return EBC_OK;
  hipLaunchKernelGGL(ebc::pair_weights_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, n_valid, B, R, w);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_pair_mask(void *stream, const long long *n_valid, int B, int R, float *w) {
  if (!n_valid || !w || B < 0 || R <= 0) return fail(EBC_ERR_INVALID, "pair_mask arguments");
  if (B == 0) return EBC_OK;
  const size_t n = (size_t)B * R;
  hipLaunchKernelGGL(ebc::pair_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_valid, B, R, w);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_pair_combine(void *stream, const double *partial, const long long *n_valid, int B, int R, int O, int mean, float *out) {
  if (!partial || !out || B < 0 || R < 16 || R > 32 || O <= 0) return fail(EBC_ERR_INVALID, "pair_combine arguments (16 <= R <= 32)");
  if ((O & 3) || (((size_t)partial | (size_t)out) & 15)) return fail(EBC_ERR_INVALID, "pair_combine: O a multiple of 4, 16-byte aligned buffers");
  if (B == 0) return EBC_OK;
  hipLaunchKernelGGL(ebc::pair_combine_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, partial, n_valid, B, R, O,
                     mean, out);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_pair_mean(void *stream, const float *h, const long long *n_valid, int B, int R, int H, float *g) {
  if (!h || !g || B < 0 || R <= 0 || H <= 0) return fail(EBC_ERR_INVALID, "pair_mean arguments");
  if (B == 0) return EBC_OK;
  if (((size_t)h | (size_t)g) & 15 || (H & 3)) {
    if (H & 3) return fail(EBC_ERR_UNSUPPORTED, "pair_mean: H must be a multiple of 4");
    return fail(EBC_ERR_INVALID, "pair_mean: 16-byte aligned buffers");
  }
  hipLaunchKernelGGL(ebc::pair_mean_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, n_valid, B, R, H, g);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_pair_attend(void *stream, const float *scores, const float *feat, const long long *n_valid, int B, int R, int F,
                    float *out) {
  if (!scores || !feat || !out || B < 0 || R <= 0 || F <= 0) return fail(EBC_ERR_INVALID, "pair_attend arguments");
  if (B == 0) return EBC_OK;
  if ((F & 3) || F > 256) return fail(EBC_ERR_UNSUPPORTED, "pair_attend: F must be a multiple of 4, at most 256");
  if (((size_t)feat | (size_t)out) & 15) return fail(EBC_ERR_INVALID, "pair_attend: 16-byte aligned buffers");
  hipLaunchKernelGGL(ebc::pair_attend_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, feat,
                     n_valid, B, R, F, out);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

int ebc_mlp2_destroy(void *mlp) {
  Mlp2 *m = (Mlp2 *)mlp;
  if (!m) return EBC_OK;
  for (void *ptr : m->allocs) (void)hipFree(ptr);
  delete m;
  return EBC_OK;
}

}  // extern "C"

