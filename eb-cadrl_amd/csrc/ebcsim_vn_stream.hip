// ebcsim_vn_stream.hip — launch of the streamed attention block (ebc_vn_stream.h) for the shapes it is built for.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ebc_vn_stream.h"
#include "ebc_vn_stream_api.h"

namespace ebc_host {

namespace {

template <int TI, int TH, int TO, int KIN, int KH>
int launch(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int O, float *y,
           const ebc::MlpExtra &ex) {
  constexpr int NW = 8;
  constexpr size_t lds = 3 * (size_t)TH * 4096 + ((size_t)TH * 32 + 2 * (size_t)TO * 32) * 4 +
                         (size_t)NW * EBC_VN_GROUPS * EBC_VN_GROUP_PITCH
#ifdef EBC_VNS_TRACE
                         + (size_t)NW * (TI + TO) * 4 * 8
#endif
      ;
  static bool raised[64] = {false};  // more than the 64 KB a launch gets by default; a function attribute is per device
  if (lds > 65536 && !raised[device & 63]) {
    HIP_TRY(hipFuncSetAttribute((const void *)ebc::mlp2_stream_kernel<TI, TH, TO, NW, KIN, KH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised[device & 63] = true;
  }
  const dim3 grid((unsigned)((M + 32 * NW - 1) / (32 * NW))), block(64 * NW);
  hipLaunchKernelGGL((ebc::mlp2_stream_kernel<TI, TH, TO, NW, KIN, KH>), grid, block, lds, st, M, L1, L2, y, O, ex);
  HIP_TRY(hipGetLastError());
  return EBC_OK;
}

}  // namespace

int vn_stream_launch(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int K0, int H, int O,
                     float *y, const ebc::MlpExtra &ex) {
  static const bool off = [] { const char *e = getenv("EBCSIM_VN_STREAM"); return e && atoi(e) == 0; }();  // measurements: the general block
  if (off) return EBC_VN_STREAM_NA;
  // the attention block as SarlValueNet calls it: fragment input, the pair's group term, a one-output third layer
  if (!ex.frag_in || !ex.row_bias || !ex.final_w || ex.partial || ex.frag_out || !y) return EBC_VN_STREAM_NA;
  if ((ex.H & 3) || ex.group_rows <= 0 || 31 / ex.group_rows + 2 > EBC_VN_GROUPS) return EBC_VN_STREAM_NA;  // group terms parked in LDS
  if (L1.in_tiles != 7 || L1.out_tiles != 7 || L2.out_tiles != 7) return EBC_VN_STREAM_NA;
  const bool kin = K0 <= 32 * 7 - 16, kh = H <= 32 * 7 - 16;
  if (kin && kh) return launch<7, 7, 7, 1, 1>(device, st, M, L1, L2, O, y, ex);
  if (kin) return launch<7, 7, 7, 1, 0>(device, st, M, L1, L2, O, y, ex);
  if (kh) return launch<7, 7, 7, 0, 1>(device, st, M, L1, L2, O, y, ex);
  return launch<7, 7, 7, 0, 0>(device, st, M, L1, L2, O, y, ex);
}

}  // namespace ebc_host
