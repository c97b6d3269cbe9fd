// Shared helpers for liblossyless_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lossyless_amd.h"

namespace lla {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return LLA_EHIP;
}

// Launch-error check: picks up invalid configuration / missing code object.
inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LLA_OK : hip_fail(e);
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace lla
