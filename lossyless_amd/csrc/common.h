// Shared helpers for liblossyless_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

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

// The product library reads NO environment variable and carries no A/B switch site: switch values are link-time functions
// (switches.h: constants in the product, the environment in the tools/ builds), alternative GEMM launchers and retired kernels
// live under ablation/ and are compiled only by `make ablation` / `make probes` (round 6).
constexpr int kWave = 64;  // gfx950 wavefront
constexpr int kLnxWaitDefault = 24000;  // ~12-17 us (tools/lnx_wait_sweep.py: 6000 loses 0.4 % to the row tiles it leaves to the clean-up kernel, 12000 .. unbounded are level); a sibling one round later never arrives

// Largest dynamic LDS allocation `kernel` may be launched with on the CURRENT device (<= 160 KiB), after
// opting the kernel in to it there; cached per (device, kernel) -- tower.hip.
unsigned dynamic_lds_limit(const void *kernel);

#ifdef __HIPCC__
// Cross-lane primitives on the VALU (DPP controls, v_permlane32_swap, v_readlane) instead of __shfl*, which
// hipcc lowers to ds_bpermute_b32: an LDS-unit instruction with an lgkmcnt round trip per step.  A 64-lane sum
// is four DPP adds and four v_readlane here against six ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes, the same value in every lane (all 64 lanes must be active)
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm:[1,0,3,2]   lane ^ 1
  v += dpp_f32<0x4E>(v);    // quad_perm:[2,3,0,1]   lane ^ 2
  v += dpp_f32<0x141>(v);   // row_half_mirror       the other quad of the 8
  v += dpp_f32<0x140>(v);   // row_mirror            the other half of the 16
  // every lane of a 16-lane row now holds its row's sum: add the four rows through SGPRs
  const int u = __builtin_bit_cast(int, v);   // (the builtin is typed int: a float argument would be CONVERTED)
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48));
  return (r0 + r1) + (r2 + r3);
}
// {value of lane (l & 31), value of lane (l | 32)} in every lane l: one v_permlane32_swap (gfx950)
__device__ __forceinline__ void half_wave_pair_f32(float v, float &lower, float &upper) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  unsigned w = u;
  asm volatile("" : "+v"(w));   // a second REGISTER holding v: swapping a register with itself just exchanges its halves
  const auto r = __builtin_amdgcn_permlane32_swap(u, w, false, false);   // r[0] = {lo, lo}, r[1] = {hi, hi}
  const unsigned lo = r[0], hi = r[1];   // (scalars first: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
  lower = __builtin_bit_cast(float, lo);
  upper = __builtin_bit_cast(float, hi);
}
#endif

#ifdef __HIPCC__
// Compile-time A/B for DESIGN.md 5.4 (make variant DEFS=-DLLA_KERNEL_ACQUIRE=1 / -DLLA_KERNEL_RELEASE=1): every tower
// kernel opens with an agent-scope acquire (buffer_inv sc1) / closes with an agent-scope release (buffer_wbl2 sc1) of its
// own, on top of what the kernel boundary does.  Off in the product: one process per GPU needs neither.
#ifndef LLA_KERNEL_ACQUIRE
#define LLA_KERNEL_ACQUIRE 0
#endif
#ifndef LLA_KERNEL_RELEASE
#define LLA_KERNEL_RELEASE 0
#endif
// Round 6, the one decisive A/B on 5.9 (make variant DEFS=...; tools/ab.sh soak): which of {dispatch overlap, L2 write-back,
// L1 / L2 invalidate} at the ONE boundary residual GEMM -> lnx_cleanup_kernel removes the two-process mismatch.
//   LLA_LNX_SYNC=1   the host waits for the stream between launch_q4(EPI_RESID_LNX) and lnx_cleanup_kernel (tower.hip)
//   LLA_LNX_FENCE&1  agent-scope release (vmcnt(0); buffer_wbl2 sc1; vmcnt(0)) at the end of that GEMM only
//   LLA_LNX_FENCE&2  agent-scope acquire (buffer_inv sc1) at the top of lnx_cleanup_kernel only
#ifndef LLA_LNX_SYNC
#define LLA_LNX_SYNC 0
#endif
#ifndef LLA_LNX_FENCE
#define LLA_LNX_FENCE 0
#endif
__device__ __forceinline__ void kernel_acquire() {
#if LLA_KERNEL_ACQUIRE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ void kernel_release() {
#if LLA_KERNEL_RELEASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}
#endif

// Two tower lanes (tower.hip): what a tower handle (lla_tower_create) holds -- two non-blocking HIP streams of
// one device, on which the slices of a batch alternate so that one slice's kernel tails and HBM-bound kernels
// run beside the other's GEMMs.  Owned by the caller through the handle: the library keeps no lane state.
struct Lanes {
  hipStream_t st[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
  int device = -1;
  int next = 0;         // lane of the next deferred slice
  bool dirty = false;   // deferred passes have been queued since the last join
  // lla_tower_set_option: LayerNorm in the residual GEMMs' epilogues (gemm_q4.hip EPI_RESID_LNX) on / off, and the
  // shader cycles a column tile waits for its siblings there (< 0: never -- every row tile takes the clean-up kernel)
  int lnx = 1;
  int lnx_wait = kLnxWaitDefault;
};
int tower_lanes();                           // 2, or 1 with LLA_VIT_STREAMS=1
int lanes_create(Lanes **out);               // on the current device
void lanes_destroy(Lanes *ln);
int lanes_fork(Lanes *ln, hipStream_t caller);   // both lanes wait for what `caller` has queued so far
int lanes_join(Lanes *ln, hipStream_t caller);   // `caller` waits for both lanes

}  // namespace lla
