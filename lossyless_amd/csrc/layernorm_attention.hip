// The CLIP ViT-B/32 tower's memory-bound kernels (gfx950): LayerNorm and attention.  fp32 statistics, fp16 storage.
//
// Stand in for ln_pre / ln_1 / ln_2 / ln_post and nn.MultiheadAttention's softmax(q k^T / sqrt(64)) v inside
// `z = self.clip(X)` (hub/compressor.py:93; clip==1.0 VisionTransformer.forward, ResidualAttentionBlock):
//   layernorm768_kernel  one wave per 768-wide row, float4 loads, the one-pass statistics and fixed summation tree of
//                        gemm_common.h (ln_finish / ln_affine): the same bits as the residual GEMMs' EPI_RESID_LNX epilogue.
//   lnx_cleanup_kernel   the row tiles such a GEMM left to it (siblings late): the same arithmetic from x.
//   ln_pre_ln1_kernel    class-token insert + positional embedding + ln_pre (in place, fp32) + block 0's ln_1.
//   attention50_kernel   one wave per (image, head): S^T = K Q^T and O^T = V^T P^T on MFMA; the softmax row of a query lives
//                        in two lanes, and the probabilities feed the second MFMA without leaving registers.
// Split out of vit.hip in round 6, kernels verbatim.
#include "tower_kernels.h"

namespace lla {
namespace {

// ---------------------------------------------------------------------------
// LayerNorm over 768 (one wave per row)
// ---------------------------------------------------------------------------
// (cross-lane sums: wave_sum_f32 in common.h -- DPP + readlane)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_f32(v); }

struct Row768 {
  float4 v[3];
};

// One residual-stream row (768 fp32), 3 x 16 bytes per lane, read with `sc0 sc1` (missing in this CU's vector L1).
//
// Why (round 3, docs/history/DESIGN_rounds_1-5.md 5.3): with TWO tower lanes (two hardware queues; opt-in since round 3) the tower was not
// bit-reproducible: 1-5 embeddings per 10^6 images differed by up to 3e-3 from run to run, never on one stream.
// The largest contributor was here: x is updated in place by the out-proj / c_proj GEMMs and read by the LayerNorm
// that follows in the same stream, and with plain loads a LayerNorm wave now and then still saw a line of x as it was
// BEFORE the update (same-box A/B with the residual stream snapshotted around every kernel,
// tools/snapshot_probe.py: 5 / 5 / 17 wrong rows per 1500 passes with plain loads, 0 / 0 / 0 / 0 with `sc0`, `sc1`
// or both -- `sc0` alone suffices, so the stale copy sat in the CU's vector L1).  It is NOT the whole story: a
// second, rarer contributor (about 1 embedding per 10^6 images) remains in two-lane mode and was not pinned down,
// which is why one stream is the default.  On one stream these loads change nothing (0 differing embeddings in
// 15 M images either way); they cost nothing measurable.
// LLA_LN_LOAD (compile time, A/B only): 0 = plain loads, 1 = `sc1`, 3 = `sc0`, 2 = `sc0 sc1`.
// Round 5: PLAIN again (0).  The two-lane mode these loads were for is gone from the product, and with a second PROCESS on
// the GPU it is exactly the loads on the device-scope path that read stale lines (`sc1` loads 100-1000 x more
// exposed than plain ones; the clean-up kernel of 5.8 read x with `sc0 sc1` and turned ~60 differing records per 10^6
// images into 26 000 when most row tiles went through it).
#ifndef LLA_LN_LOAD
#define LLA_LN_LOAD 0
#endif
__device__ __forceinline__ Row768 load_row768(const float *row, int lane) {
  Row768 in;
  const float4 *src = reinterpret_cast<const float4 *>(row) + lane;
#if LLA_LN_LOAD == 0
#pragma unroll
  for (int i = 0; i < 3; ++i) in.v[i] = src[64 * i];
#elif LLA_LN_LOAD == 1
  asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
               "global_load_dwordx4 %1, %3, off offset:1024 sc1\n\t"
               "global_load_dwordx4 %2, %3, off offset:2048 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(in.v[0]), "=&v"(in.v[1]), "=&v"(in.v[2]) : "v"(src) : "memory");
#elif LLA_LN_LOAD == 3
  asm volatile("global_load_dwordx4 %0, %3, off sc0\n\t"
               "global_load_dwordx4 %1, %3, off offset:1024 sc0\n\t"
               "global_load_dwordx4 %2, %3, off offset:2048 sc0\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(in.v[0]), "=&v"(in.v[1]), "=&v"(in.v[2]) : "v"(src) : "memory");
#else
  asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
               "global_load_dwordx4 %1, %3, off offset:1024 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %3, off offset:2048 sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(in.v[0]), "=&v"(in.v[1]), "=&v"(in.v[2]) : "v"(src) : "memory");
#endif
  return in;
}

// (mean, rstd) of a row in the canonical arithmetic of gemm_common.h (ln_finish): lane l holds columns
// 256 i + 4 l .. + 3, so wave_sum_f32 of the lane's quad of block i IS column tile i's partial sum t_i
__device__ __forceinline__ void row_stats(const Row768 &x, float &mean, float &rstd) {
  float t[3], u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t[i] = wave_sum((x.v[i].x + x.v[i].y) + (x.v[i].z + x.v[i].w));
    u[i] = wave_sum((x.v[i].x * x.v[i].x + x.v[i].y * x.v[i].y) + (x.v[i].z * x.v[i].z + x.v[i].w * x.v[i].w));
  }
  ln_finish((t[0] + t[1]) + t[2], (u[0] + u[1]) + u[2], mean, rstd);
}

__device__ __forceinline__ Row768 row_affine(const Row768 &x, float mean, float rstd,
                                             const float *__restrict__ w,
                                             const float *__restrict__ b, int lane) {
  Row768 y;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 g = reinterpret_cast<const float4 *>(w)[lane + 64 * i];
    const float4 o = reinterpret_cast<const float4 *>(b)[lane + 64 * i];
    y.v[i].x = ln_affine(x.v[i].x, mean, rstd, g.x, o.x);
    y.v[i].y = ln_affine(x.v[i].y, mean, rstd, g.y, o.y);
    y.v[i].z = ln_affine(x.v[i].z, mean, rstd, g.z, o.z);
    y.v[i].w = ln_affine(x.v[i].w, mean, rstd, g.w, o.w);
  }
  return y;
}

__device__ __forceinline__ void store_row_f16(f16 *dst, const Row768 &y, int lane) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f16x4 h;
    h[0] = (f16)y.v[i].x; h[1] = (f16)y.v[i].y; h[2] = (f16)y.v[i].z; h[3] = (f16)y.v[i].w;
    reinterpret_cast<f16x4 *>(dst)[lane + 64 * i] = h;
  }
}

__global__ __launch_bounds__(256) void layernorm768_kernel(const float *__restrict__ x,
                                                           size_t row_stride,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ b,
                                                           f16 *__restrict__ y, int rows, int rev) {
  kernel_acquire();
  const int blk = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;   // (rev: last rows first -- see GemmParams::rev)
  const int row = blk * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const Row768 in = load_row768(x + (size_t)row * row_stride, lane);
  float mean, rstd;
  row_stats(in, mean, rstd);
  const Row768 out = row_affine(in, mean, rstd, w, b, lane);
  store_row_f16(y + (size_t)row * kWidth, out, lane);
  kernel_release();
}

#ifndef LLA_LNX_CLEANUP_SPLIT
#define LLA_LNX_CLEANUP_SPLIT 8   // workgroups per row tile of lnx_cleanup_kernel (A/B: make variant DEFS=-DLLA_LNX_CLEANUP_SPLIT=n; 1, 2, 4, 8 or 16)
#endif
static_assert(64 % LLA_LNX_CLEANUP_SPLIT == 0 && (64 / LLA_LNX_CLEANUP_SPLIT) % 4 == 0,
              "lnx_cleanup_kernel: a wave takes 64 / split rows, four at a time");
// Behind every EPI_RESID_LNX GEMM (gemm_q4.hip): the row tiles whose three column tiles did not ALL normalise their
// chunk in the GEMM's epilogue (a sibling tile was late: another round of the persistent grid, a busy CU) get their
// LayerNorm here, from x, in the same arithmetic (gemm_common.h ln_finish / ln_affine: same bits either way).
// grid = tiles_m x split workgroups of 4 waves (split = 8: 8 rows per wave); a workgroup whose row tile is complete -- with the
// row tiles walked in triples (gemm_q4.hip) nearly all of them -- exits at once: 31 us per full-size launch, most of it looking.
__global__ __launch_bounds__(256) void lnx_cleanup_kernel(const float *__restrict__ x, const unsigned *__restrict__ done,
                                                          const float *__restrict__ w, const float *__restrict__ b,
                                                          f16 *__restrict__ y, int rev, unsigned epoch, int split) {
  kernel_acquire();
#if LLA_LNX_FENCE & 2
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  const int blk = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int rt = blk / split;
  // (agent-scope loads: the words were written through by other CUs in the kernel before)
  const bool complete = __hip_atomic_load(done + rt * 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch &&
                        __hip_atomic_load(done + rt * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch &&
                        __hip_atomic_load(done + rt * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
  if (complete) return;
  const int lane = threadIdx.x & 63;
  const int per_wave = 64 / split;            // split = workgroups per row tile: 1 (64 rows per wave) or 8 (8 rows per wave)
  const int row0 = rt * 256 + (blk - rt * split) * (256 / split) + (threadIdx.x >> 6) * per_wave;
  // four rows in flight per wave (12 loads of 16 bytes per lane before the first use: the kernel runs on the ~9 % of
  // row tiles that straddle two rounds, a latency-bound loop of one row at a time took 110 us per launch)
  for (int r0 = 0; r0 < per_wave; r0 += 4) {
    f32x4 raw[4][3];      // (native vectors: the asm writes them itself and the wait ties all twelve)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 *src = reinterpret_cast<const float4 *>(x + (size_t)(row0 + r0 + r) * kWidth) + lane;
      asm volatile("global_load_dwordx4 %0, %3, off\n\t"            // (plain loads: see LLA_LN_LOAD)
                   "global_load_dwordx4 %1, %3, off offset:1024\n\t"
                   "global_load_dwordx4 %2, %3, off offset:2048"
                   : "=&v"(raw[r][0]), "=&v"(raw[r][1]), "=&v"(raw[r][2]) : "v"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[0][2]), "+v"(raw[1][0]), "+v"(raw[1][1]),
                 "+v"(raw[1][2]), "+v"(raw[2][0]), "+v"(raw[2][1]), "+v"(raw[2][2]), "+v"(raw[3][0]), "+v"(raw[3][1]),
                 "+v"(raw[3][2])::"memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Row768 in;
#pragma unroll
      for (int i = 0; i < 3; ++i) in.v[i] = make_float4(raw[r][i][0], raw[r][i][1], raw[r][i][2], raw[r][i][3]);
      float mean, rstd;
      row_stats(in, mean, rstd);
      store_row_f16(y + (size_t)(row0 + r0 + r) * kWidth, row_affine(in, mean, rstd, w, b, lane), lane);
    }
  }
  kernel_release();
}

// Token assembly + ln_pre (fp32, in place) + ln_1 of block 0 (fp16 out).
// Patch rows already hold conv + pos (EPI_PATCH); class rows are built here.
__global__ __launch_bounds__(256) void ln_pre_ln1_kernel(
    float *__restrict__ x, const float *__restrict__ cls, const float *__restrict__ pos,
    const float *__restrict__ wpre, const float *__restrict__ bpre, const float *__restrict__ w1,
    const float *__restrict__ b1, f16 *__restrict__ h, int rows) {
  kernel_acquire();
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4 *xr = reinterpret_cast<float4 *>(x + (size_t)row * kWidth);
  Row768 in;
  if (row % kTokens == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 c = reinterpret_cast<const float4 *>(cls)[lane + 64 * i];
      const float4 q = reinterpret_cast<const float4 *>(pos)[lane + 64 * i];
      in.v[i] = make_float4(c.x + q.x, c.y + q.y, c.z + q.z, c.w + q.w);
    }
  } else {
    in = load_row768(x + (size_t)row * kWidth, lane);
  }
  float mean, rstd;
  row_stats(in, mean, rstd);
  const Row768 t = row_affine(in, mean, rstd, wpre, bpre, lane);
#pragma unroll
  for (int i = 0; i < 3; ++i) xr[lane + 64 * i] = t.v[i];
  row_stats(t, mean, rstd);
  const Row768 u = row_affine(t, mean, rstd, w1, b1, lane);
  store_row_f16(h + (size_t)row * kWidth, u, lane);
  kernel_release();
}


// ---------------------------------------------------------------------------
// Attention over 50 tokens, 12 heads of 64.  One wave per (image, head).
// ---------------------------------------------------------------------------
constexpr int kVPitch = 72;  // halfs; 144-byte rows keep 16-byte alignment and spread banks
// LLA_ATTN_LOAD (compile time, A/B only): 0 = plain loads of qkv (default), 2 = `sc1`, 3 = `sc0 sc1`
#ifndef LLA_ATTN_LOAD
#define LLA_ATTN_LOAD 0
#endif
#if LLA_ATTN_LOAD == 2
#define LLA_ATTN_SC " sc1"
#elif LLA_ATTN_LOAD == 3
#define LLA_ATTN_SC " sc0 sc1"
#else
#define LLA_ATTN_SC ""
#endif

// 4 waves per SIMD (<= 128 VGPRs: 119 used, no spills): 4 workgroups per CU instead of 3, 60 -> 58 us
__global__ __launch_bounds__(256, 4) void attention50_kernel(const f16 *__restrict__ qkv,
                                                          f16 *__restrict__ o, int B, int rev) {
  kernel_acquire();
  __shared__ __attribute__((aligned(16))) f16 lds[4][64 * kVPitch];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r32 = lane & 31, hk = lane >> 5;
  const int blk = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;   // (rev: last images first -- see GemmParams::rev)
  const int b = blk / 3;
  const int head = (blk - b * 3) * 4 + wid;
  f16 *vs = lds[wid];
  const f16 *base = qkv + (size_t)b * kTokens * (3 * kWidth) + head * kHeadDim;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  f16x8 kf[2][4], qf[2][4];
#if LLA_ATTN_LOAD
  // (A/B build, round 5) qkv read past this CU's vector L1 (`sc1` / `sc0 sc1`): the buffer is rewritten by every
  // layer's QKV and c_fc GEMMs.  All loads unconditional from clamped rows (an asm output merged with a zero on
  // another path is copied before the data arrives), one counted wait tied to every destination, then the masks.
  {
    f16x8 vv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = lane + 64 * it, j = id >> 3, dc = id & 7;
      const f16 *src = base + (size_t)(j < kTokens ? j : 0) * (3 * kWidth) + 2 * kWidth + dc * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" LLA_ATTN_SC : "=v"(vv[it]) : "v"(src) : "memory");
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 32 * t + r32;
      const f16 *rp = base + (size_t)(row < kTokens ? row : 0) * (3 * kWidth) + 8 * hk;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        asm volatile("global_load_dwordx4 %0, %1, off" LLA_ATTN_SC : "=v"(qf[t][s]) : "v"(rp + 16 * s) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" LLA_ATTN_SC : "=v"(kf[t][s]) : "v"(rp + kWidth + 16 * s) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]),
                   "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]),
                   "+v"(qf[1][2]), "+v"(qf[1][3]), "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]),
                   "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3])
                 :: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = lane + 64 * it, j = id >> 3, dc = id & 7;
      *reinterpret_cast<f16x8 *>(vs + j * kVPitch + dc * 8) = j < kTokens ? vv[it] : zero8;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (32 + r32 >= kTokens) { qf[1][s] = zero8; kf[1][s] = zero8; }
  }
#else
  // V -> LDS, row major [key][d], keys 50..63 zero (0 * garbage must stay 0)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int id = lane + 64 * it, j = id >> 3, dc = id & 7;
    f16x8 v = zero8;
    if (j < kTokens)
      v = *reinterpret_cast<const f16x8 *>(base + (size_t)j * (3 * kWidth) + 2 * kWidth + dc * 8);
    *reinterpret_cast<f16x8 *>(vs + j * kVPitch + dc * 8) = v;
  }

  // K and Q fragments straight from global in MFMA operand layout:
  // operand row = lane & 31, k-slots = 8 consecutive d at 16 s + 8 (lane >> 5)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 32 * t + r32;
    const bool ok = row < kTokens;
    const f16 *rp = base + (size_t)(ok ? row : 0) * (3 * kWidth) + 8 * hk;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[t][s] = ok ? *reinterpret_cast<const f16x8 *>(rp + 16 * s) : zero8;
      kf[t][s] = ok ? *reinterpret_cast<const f16x8 *>(rp + kWidth + 16 * s) : zero8;
    }
  }
#endif

  // S^T[j][i] = K[j] . Q[i]  ->  lane holds query i = 32 it + (lane & 31),
  // keys j = 32 jt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  f32x16 sT[2][2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[jt][it][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        sT[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[jt][s], qf[it][s], sT[jt][it], 0, 0, 0);
    }

  // softmax over keys: 32 of a query's 64 key slots are in this lane, the rest in lane ^ 32
  f16x8 pf[2][2][2];  // [it][jt][s'] : B operand of O^T = V^T P^T, k-slot e <-> r = 8 s' + e
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hk;
        const float s = j < kTokens ? sT[jt][it][r] * 0.125f : -3.0e38f;
        sT[jt][it][r] = s;
        mx = fmaxf(mx, s);
      }
    {
      float lo, hi;
      half_wave_pair_f32(mx, lo, hi);   // (one v_permlane32_swap: common.h)
      mx = fmaxf(lo, hi);
    }
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hk;
        const float e = j < kTokens ? __expf(sT[jt][it][r] - mx) : 0.f;
        sT[jt][it][r] = e;
        sum += e;
      }
    {
      float lo, hi;
      half_wave_pair_f32(sum, lo, hi);
      sum = lo + hi;
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[it][jt][sp][e] = (f16)(sT[jt][it][8 * sp + e] * inv);
  }

  __syncthreads();  // V tile visible

  // O^T[d][i] = sum_j V[j][d] P[i][j].  A operand: row d = 32 dt + (lane & 31), k-slot e of
  // step (jt, s') is key j = 32 jt + 16 s' + (e & 3) + 8 (e >> 2) + 4 (lane >> 5): the same
  // slot->key map the probabilities already have, so P never moves between lanes.
  f32x16 oT[2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) oT[dt][it][r] = 0.f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f16x8 vf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = 32 * jt + 16 * sp + (e & 3) + 8 * (e >> 2) + 4 * hk;
          vf[e] = vs[j * kVPitch + 32 * dt + r32];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
          oT[dt][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[it][jt][sp], oT[dt][it], 0, 0, 0);
      }
    }

  __syncthreads();  // all V reads done; reuse the tile for O
  // lane holds query i = 32 it + (lane & 31), d = 32 dt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 q4;
#pragma unroll
        for (int e = 0; e < 4; ++e) q4[e] = (f16)oT[dt][it][4 * g + e];
        *reinterpret_cast<f16x4 *>(vs + (32 * it + r32) * kVPitch + 32 * dt + 8 * g + 4 * hk) = q4;
      }
  __syncthreads();
  f16 *ob = o + (size_t)b * kTokens * kWidth + head * kHeadDim;
#pragma unroll
  for (int it = 0; it < 7; ++it) {
    const int id = lane + 64 * it, i = id >> 3, dc = id & 7;
    if (i < kTokens)
      *reinterpret_cast<f16x8 *>(ob + (size_t)i * kWidth + dc * 8) =
          *reinterpret_cast<const f16x8 *>(vs + i * kVPitch + dc * 8);
  }
  kernel_release();
}

}  // namespace

int layernorm_impl(const float *x, size_t row_stride, const float *w, const float *b, void *y16, int rows, hipStream_t st,
                   Profiler *prof, int rev) {
  if (rows < 0 || row_stride < (size_t)kWidth || (row_stride & 3u)) return LLA_EINVAL;
  if (rows == 0) return LLA_OK;
  if (!x || !w || !b || !y16) return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_LAYERNORM, (double)rows * kWidth * 6.0);
  layernorm768_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, row_stride, w, b, reinterpret_cast<f16 *>(y16), rows, rev);
  return check_launch();
}

int attention_impl(const void *qkv, void *o, int B, hipStream_t st, Profiler *prof, int rev) {
  if (B < 0) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!qkv || !o) return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_ATTENTION, (double)B * 12 * 4.0 * kTokens * kTokens * kHeadDim);
  attention50_kernel<<<B * 3, 256, 0, st>>>(reinterpret_cast<const f16 *>(qkv), reinterpret_cast<f16 *>(o), B, rev);
  return check_launch();
}

int ln_pre_ln1_impl(float *x, const float *cls, const float *pos, const float *wpre, const float *bpre, const float *w1,
                    const float *b1, void *h16, int rows, hipStream_t st, Profiler *prof) {
  ProfScope scope(prof, st, LLA_PROF_LAYERNORM, (double)rows * kWidth * 10.0);
  ln_pre_ln1_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, cls, pos, wpre, bpre, w1, b1, reinterpret_cast<f16 *>(h16), rows);
  return check_launch();
}

int lnx_cleanup_impl(const float *x, const unsigned *done, const float *w, const float *b, void *y16, int tiles_m, int rev,
                     unsigned epoch, hipStream_t st, Profiler *prof) {
  ProfScope scope(prof, st, LLA_PROF_LAYERNORM, 0.0);
  const int split = LLA_LNX_CLEANUP_SPLIT;   // (one workgroup per row tile measured: 74 us per launch against 31 -- the few row tiles that DO need it decide)
  lnx_cleanup_kernel<<<tiles_m * split, 256, 0, st>>>(x, done, w, b, reinterpret_cast<f16 *>(y16), rev, epoch, split);
  return check_launch();
}

}  // namespace lla

using namespace lla;

extern "C" {

int lla_layernorm768(const float *x, size_t row_stride, const float *w, const float *b, void *y16, int rows, void *stream) {
  return layernorm_impl(x, row_stride, w, b, y16, rows, as_stream(stream), nullptr);
}

int lla_attention50(const void *qkv, void *o, int B, void *stream) {
  return attention_impl(qkv, o, B, as_stream(stream), nullptr);
}

}  // extern "C"
