// mvp_host.h -- host-side glue shared by the C-ABI entry points (no torch, no allocation, no global state).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mvp_abi.h"

namespace mvp {
// hipGetLastError() after a launch: 0 == MVP_OK, otherwise the (positive) hipError_t value.
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MVP_OK : (int)e;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
}  // namespace mvp
