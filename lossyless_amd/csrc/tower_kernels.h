// The tower's memory-bound kernels across translation units (layernorm_attention.hip), as tower.hip launches them.
#pragma once
#include "gemm_common.h"

namespace lla {

// LayerNorm over 768 of `rows` fp32 rows (row stride in elements) -> fp16; rev: walk the rows last to first
int layernorm_impl(const float *x, size_t row_stride, const float *w, const float *b, void *y16, int rows, hipStream_t st,
                   Profiler *prof, int rev = 0);
// softmax(q k^T / 8) v over the 50 tokens of B images, 12 heads: qkv fp16 [B * 50][2304] -> o fp16 [B * 50][768]
int attention_impl(const void *qkv, void *o, int B, hipStream_t st, Profiler *prof, int rev = 0);
// token assembly (class token, positional embedding) + ln_pre (fp32, in place) + ln_1 of block 0 (fp16)
int ln_pre_ln1_impl(float *x, const float *cls, const float *pos, const float *wpre, const float *bpre, const float *w1,
                    const float *b1, void *h16, int rows, hipStream_t st, Profiler *prof);
// behind an EPI_RESID_LNX GEMM (gemm_q4.hip): LayerNorm of the row tiles whose `done` words do not carry the launch's epoch
int lnx_cleanup_impl(const float *x, const unsigned *done, const float *w, const float *b, void *y16, int tiles_m, int rev,
                     unsigned epoch, hipStream_t st, Profiler *prof);

}  // namespace lla
