// ebc_vn_stream_api.h — host entry of the streamed value-network blocks (ebcsim_vn_stream.hip; its own translation unit
// so that it compiles beside the others).
#pragma once

#include "ebc_host.h"
#include "ebc_vn_common.h"

namespace ebc_host {

// EBC_OK: launched.  EBC_VN_STREAM_NA: this block / call is not one the streamed kernel takes (the caller launches the
// general block); anything else: an error (ebc_last_error()).
#define EBC_VN_STREAM_NA (-1000)
int vn_stream_launch(int device, hipStream_t st, int M, const ebc::PackedLayer &L1, const ebc::PackedLayer &L2, int K0, int H, int O,
                     float *y, const ebc::MlpExtra &ex, int relu_out, const float *x);

}  // namespace ebc_host
