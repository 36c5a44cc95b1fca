// ebc_host.h — host-side helpers shared by the translation units of libebcsim.so (ebcsim.hip: the simulation path;
// ebcsim_value_net.hip: the value-network blocks; ebcsim_vn_stream.hip: the streamed blocks — units that compile side
// by side).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ebcsim.h"

namespace ebc_host {

inline thread_local std::string g_err;  // ebc_last_error(): one per thread, shared by the units (C++17 inline variable)

inline int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

}  // namespace ebc_host

#define HIP_TRY(x)                                                                                    \
  do {                                                                                                \
    hipError_t err__ = (x);                                                                           \
    if (err__ != hipSuccess)                                                                          \
      return ebc_host::fail(EBC_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(err__));       \
  } while (0)
