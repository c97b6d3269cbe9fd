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

// Largest dynamic LDS allocation `kernel` may be launched with on the CURRENT device (<= 160 KiB), after
// opting the kernel in to it there; cached per (device, kernel) -- vit.hip.
unsigned dynamic_lds_limit(const void *kernel);

// Two tower lanes (vit.hip): what a tower handle (lla_tower_create) holds -- two non-blocking HIP streams of
// one device, on which the slices of a batch alternate so that one slice's kernel tails and HBM-bound kernels
// run beside the other's GEMMs.  Owned by the caller through the handle: the library keeps no lane state.
struct Lanes {
  hipStream_t st[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
  int device = -1;
  int next = 0;         // lane of the next deferred slice
  bool dirty = false;   // deferred passes have been queued since the last join
};
int tower_lanes();                           // 2, or 1 with LLA_VIT_STREAMS=1
int lanes_create(Lanes **out);               // on the current device
void lanes_destroy(Lanes *ln);
int lanes_fork(Lanes *ln, hipStream_t caller);   // both lanes wait for what `caller` has queued so far
int lanes_join(Lanes *ln, hipStream_t caller);   // `caller` waits for both lanes

}  // namespace lla
