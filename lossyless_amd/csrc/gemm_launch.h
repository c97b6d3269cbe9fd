// The tower's GEMM entry point across translation units: launch_gemm_rt(epilogue, operand mode, ...) picks the kernel for the
// shape.  Defined by gemm_pp.hip in the product library and by ablation/gemm_select.hip (the same selection + every A/B switch)
// in the tools/ builds; called by tower.hip, gemm_api.hip.
#pragma once
#include "gemm_common.h"

namespace lla {

int launch_gemm_rt(int epi, int amode, const GemmParams &p, hipStream_t st, Profiler *prof = nullptr);

}  // namespace lla

// (epilogue, operand mode) pairs the library launches -> the defining file's `launch_gemm<EPI, AMODE>` template
#define LLA_DEFINE_LAUNCH_GEMM                                                                                  \
  int launch_gemm_rt(int epi, int amode, const GemmParams &p, hipStream_t st, Profiler *prof) {                 \
    if (amode == A_PLAIN) {                                                                                     \
      switch (epi) {                                                                                            \
        case EPI_F16: return launch_gemm<EPI_F16, A_PLAIN>(p, st, prof);                                        \
        case EPI_QGELU: return launch_gemm<EPI_QGELU, A_PLAIN>(p, st, prof);                                    \
        case EPI_RESID: return launch_gemm<EPI_RESID, A_PLAIN>(p, st, prof);                                    \
        case EPI_RELU: return launch_gemm<EPI_RELU, A_PLAIN>(p, st, prof);                                      \
        case EPI_ADDRELU: return launch_gemm<EPI_ADDRELU, A_PLAIN>(p, st, prof);                                \
        default: return LLA_EINVAL;                                                                             \
      }                                                                                                         \
    }                                                                                                           \
    if (amode == A_PATCH_NHWC && epi == EPI_PATCH) return launch_gemm<EPI_PATCH, A_PATCH_NHWC>(p, st, prof);    \
    if (amode == A_PATCH_NCHW && epi == EPI_PATCH) return launch_gemm<EPI_PATCH, A_PATCH_NCHW>(p, st, prof);    \
    if (amode == A_CONV3 && epi == EPI_RELU) return launch_gemm<EPI_RELU, A_CONV3>(p, st, prof);                \
    return LLA_EINVAL;                                                                                          \
  }
