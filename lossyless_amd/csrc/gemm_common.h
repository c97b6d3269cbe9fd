// GEMM building blocks shared by the tower's GEMM kernels (gemm_kernels.h, gemm_q4_kernel.h, gemm_w8_kernel.h):
// operand / epilogue enums, GemmParams, store helpers and the three epilogue families.  Moved out of vit.hip
// verbatim in round 4 so that the new kernel compiles as its own translation unit.
#pragma once
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

namespace lla {

// ---------------------------------------------------------------------------
// optional event profiler (see include/lossyless_amd.h)
// ---------------------------------------------------------------------------
struct Profiler {
  struct Rec { hipEvent_t a, b; int cls; double work; };
  std::vector<Rec> pool;
  size_t used = 0;
};
struct ProfScope {
  Profiler *p; hipStream_t st; Profiler::Rec *r = nullptr;
  ProfScope(Profiler *p_, hipStream_t st_, int cls, double work) : p(p_), st(st_) {
    if (p && p->used < p->pool.size()) {
      r = &p->pool[p->used++];
      r->cls = cls; r->work = work;
      (void)hipEventRecord(r->a, st);
    }
  }
  ~ProfScope() { if (r) (void)hipEventRecord(r->b, st); }
};

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWidth = 768, kLayers = 12, kHeadDim = 64, kTokens = 50;  // 12 heads
constexpr int kPatches = 49, kPatchK = 3072, kMlp = 3072, kOut = 512;
constexpr int kImgElems = 224 * 224 * 3;


// ---------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int BM2 = 256, BN2 = 128;   // gemm256_f16_kernel's tile (gemm_kernels.h; lla_conv3x3_relu_f16 checks its shapes against it)
#ifndef LLA_W8_DEFAULT
#define LLA_W8_DEFAULT 1   // 1: the large fp16-output GEMMs (QKV, c_fc; M >= 9000, N % 256 == 0) run on gemm_w8.hip
#endif
constexpr int kGemmThreads = 256;

enum { EPI_F16 = 0, EPI_QGELU = 1, EPI_RESID = 2, EPI_PATCH = 3, EPI_RELU = 4, EPI_ADDRELU = 5,
       // (6 .. 8: the algebraic LayerNorm fusion of round 3, retired in round 6: docs/history/DESIGN_rounds_1-5.md 5.4)
       // round 5: EPI_RESID whose epilogue ALSO applies the LayerNorm that follows the residual add (ln_2 after out-proj,
       // ln_1 of the next block after c_proj) to its own 256 x 256 chunk of x and writes it as fp16: the three column
       // tiles of a row tile exchange exact per-row partial sums through memory (GemmParams::lnx_*, gemm_q4.hip)
       EPI_RESID_LNX = 9 };
// `sc0` (miss in this CU's vector L1, L2 hits allowed) on the loads that read buffers another kernel of the same
// stream rewrites in place (docs/history/DESIGN_rounds_1-5.md 5.3: with two tower lanes a LayerNorm wave was served stale L1 lines of the
// residual stream): LLA_DMA_SC0 = the GEMMs' LDS-DMA operand loads (activations; no reuse in L1 anyway), LLA_RMW_SC0 =
// the residual rows of the read-modify-write epilogue.  On by default as a precaution for the opt-in two-lane mode:
// neither changes the speed (same-box A/B: 98.6k / 98.7k img/s) nor, on one stream, the results.
// Values: 0 plain, 1 `sc0`, 2 `sc1`, 3 `sc0 sc1` (round 5: MI355X_MICROARCH.md says `sc0` loads hit L1 like plain and
// only `sc1` / `sc0 sc1` / `nt` bypass it; make variant DEFS="-DLLA_DMA_SC0=2 -DLLA_RMW_SC0=2 -DLLA_ATTN_LOAD=2").
#ifndef LLA_DMA_SC0
#define LLA_DMA_SC0 1
#endif
#ifndef LLA_RMW_SC0
#define LLA_RMW_SC0 1
#endif
#if LLA_DMA_SC0 == 1
#define LLA_DMA_SC " sc0"
#define LLA_DMA_AUX 1
#elif LLA_DMA_SC0 == 2
#define LLA_DMA_SC " sc1"
#define LLA_DMA_AUX 16
#elif LLA_DMA_SC0 == 3
#define LLA_DMA_SC " sc0 sc1"
#define LLA_DMA_AUX 17
#else
#define LLA_DMA_SC ""
#define LLA_DMA_AUX 0
#endif
#if LLA_RMW_SC0 == 1
#define LLA_RMW_SC " sc0"
#elif LLA_RMW_SC0 == 2
#define LLA_RMW_SC " sc1"
#elif LLA_RMW_SC0 == 3
#define LLA_RMW_SC " sc0 sc1"
#else
#define LLA_RMW_SC ""
#endif
constexpr int epi_base(int e) { return e == EPI_RESID_LNX ? EPI_RESID : e; }   // the epilogue family of a kernel instantiation

enum { A_PLAIN = 0, A_PATCH_NHWC = 1, A_PATCH_NCHW = 2, A_CONV3 = 3 };

// 128 bytes of zeros: the out-of-image taps of the implicit 3x3 convolution GEMM read their A chunk here
__device__ __attribute__((aligned(128))) f16 g_zero_line[64];

}  // namespace
// (external linkage: the launchers of gemm_pp.hip / gemm_q4.hip / gemm_w8.hip take it from tower.hip)
struct GemmParams {
  const f16 *A;
  const f16 *W;
  const float *bias;  // [N] or null
  void *C;
  const float *pos;   // EPI_PATCH: positional embedding [50][768]
  int M, N, K;
  int lda, ldc;       // elements
  unsigned long long *trace;  // LLA_GEMM_DEBUG=9 only: per-K-tile s_memtime stamps of 8 workgroups' wave 0
  const void *resid;          // EPI_ADDRELU: fp16 [M][ldr] added before the ReLU
  int ldr;
  int conv_h, conv_w, conv_cin;   // A_CONV3: image height / width / input channels (row m = (b, y, x); lda = channel pitch)
  int n_store;                // gemm_epilogue: columns >= n_store (a multiple of 32) are computed but not stored (0: all)
  // 1: walk the row tiles from the LAST row to the first.  The tower alternates the direction from kernel to kernel
  // (vit_forward_impl), so that a kernel starts on the rows its producer wrote last -- the ones still in the 256-MB
  // memory-side cache -- instead of the ones written first, which are long gone.  Order only: same results.
  int rev;
  // A_PATCH_* operands in pieces (lla_vit_b32_forward_gather): the image batch is a_chunk_images images per piece
  // (a multiple of 256: 256 images are 49 whole 256-row tiles, so no tile of the 256-row kernel straddles two pieces)
  // instead of one contiguous array at A.  0: contiguous.  Only the ping-pong kernel's 256-row instantiation reads it.
  int a_chunk_images;
  const f16 *a_chunk[64];
  // EPI_RESID_LNX (gemm_q4.hip, N == 768): LayerNorm of the updated rows in the epilogue.
  const float *lnx_g, *lnx_b;   // gamma / beta [768]
  f16 *lnx_h;                   // [M][768] LayerNorm output
  float *lnx_part;              // [M / 256][3][256] 16-byte granules {sum, sum of squares, epoch, epoch} of a column tile's 256 columns
  unsigned *lnx_flag;           // [M / 256][3] zeroed before the pass; == lnx_epoch: that tile's granules are published
  unsigned *lnx_done;           // [M / 256][3] zeroed before the pass; == lnx_epoch: that column tile wrote its chunk of h
  unsigned lnx_epoch;           // unique per launch (never 0): tags everything this launch publishes
  int lnx_wait;                 // shader cycles a workgroup waits for its two siblings before it leaves the row tile to
                                // lnx_cleanup_kernel (0: looks once; < 0: does not even look -- every row tile takes the clean-up path)
};
namespace {
constexpr int kLnSlots = 24;  // the slice buffers reserve kLnSlots x 8 bytes per row for the LayerNorm epilogues' exchange area (tower.hip: `part`)

// Element offset of logical K index kk (multiple of 8) inside one patch row.
template <int AMODE>
__device__ __forceinline__ int patch_koff(int kk) {
  if constexpr (AMODE == A_PATCH_NHWC) {
    const int kh = kk / 96;  // 32 pixels * 3 channels per patch row
    return kh * (224 * 3) + (kk - kh * 96);
  } else {
    const int c = kk >> 10, rem = kk & 1023;
    return c * (224 * 224) + (rem >> 5) * 224 + (rem & 31);
  }
}

// Element offset of patch row m = b*49 + py*7 + px.
template <int AMODE>
__device__ __forceinline__ size_t patch_rowoff(int m) {
  const int b = m / kPatches, p = m - b * kPatches;
  const int py = p / 7, px = p - py * 7;
  if constexpr (AMODE == A_PATCH_NHWC)
    return (size_t)b * kImgElems + (size_t)(py * 32) * (224 * 3) + px * 96;
  else
    return (size_t)b * kImgElems + (size_t)(py * 32) * 224 + px * 32;
}

// Bijective XCD-aware remap: hardware places workgroup id w on XCD w % 8; give each
// XCD one contiguous run of logical tiles so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// QuickGELU x * sigmoid(1.702 x) in fp32.  The product is laundered through an empty asm so that
// hipcc cannot fold it into the following fp16 conversion (v_fma_mixlo_f16 rounds once, mul + cvt
// twice, and which elements got which depended on register allocation): every GEMM path must
// round the same way for their outputs to be bit-identical.
__device__ __forceinline__ float quick_gelu(float x) {
  // exp(-1.702 x) = exp2(x * (-1.702 * log2 e)): one multiply feeding v_exp_f32
  float y = x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -2.45546696f));
  asm volatile("" : "+v"(y));
  return y;
}

// ---------------------------------------------------------------------------
// LayerNorm over 768: ONE arithmetic for every code path (layernorm768_kernel, ln_pre_ln1_kernel, the residual GEMMs'
// EPI_RESID_LNX epilogue, lnx_cleanup_kernel), so that a row's fp16 output does not depend on which of them ran --
// an image's embedding must not depend on the batch it travelled in (tests: tower at 8 / 257 / 1024 / 8704 images).
//   * a row's (sum, sum of squares) are added up in one fixed tree: quads of 4 consecutive columns as
//     (x0 + x1) + (x2 + x3); 32-column slots as ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)); 64 columns as
//     s0 + s1; 128 columns (a wave tile of the four-wave GEMM) as p0 + p1; 256 columns (a column tile) as w0 + w1;
//     the row as (t0 + t1) + t2.  wave_sum_f32 (common.h) adds 64 lanes holding one quad each in exactly this
//     order (DPP xor 1, xor 2, half mirror, row mirror, then (r0 + r1) + (r2 + r3)).
//   * mean = S / 768 (as a multiply), var = max(Q / 768 - mean^2, 0), rstd = 1 / sqrt(var + 1e-5): one pass over the
//     row instead of the two of rounds 1-4 (fp32: Q / 768 and mean^2 differ by many orders less than their 2^-24
//     resolution unless |mean| >> sigma, which a residual stream is not; tests/test_gpu_vit.py bounds it at 36 sigma).
//   * y = ((x - mean) * rstd) * gamma + beta, four separately rounded fp32 operations (-ffp-contract=off), then fp16.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ln_finish(float S, float Q, float &mean, float &rstd) {
  mean = S * (1.f / kWidth);
  const float var = fmaxf(Q * (1.f / kWidth) - mean * mean, 0.f);
  rstd = 1.f / sqrtf(var + 1e-5f);
}
__device__ __forceinline__ float ln_affine(float x, float mean, float rstd, float g, float b) {
  return (x - mean) * rstd * g + b;
}

// 16-byte global store hidden from hipcc's wait-count bookkeeping.  A store the compiler can see
// stays "pending VMEM" in its model across the persistent loop's back edge, and it then drains the
// whole vector-memory queue (s_waitcnt vmcnt(0)) -- i.e. the LDS-DMA ring -- at the top of the next
// K-tile.  The trailing s_nop keeps the next instruction off the data registers until the store has
// read them (cdna_hip_programming.md 5.7 item 1).
// Cache policy of the epilogues' output stores (A/B builds: make variant NAME=nt DEFS="-DLLA_ST_POLICY=1"):
// 0 plain, 1 `nt` (non-temporal: the line is not expected to be re-used from this L2), 2 `sc1 nt`, 3 `sc0 sc1 nt`,
// 4 `sc1` / 5 `sc0 sc1` (write-through: the store side of MI355X_MICROARCH.md hazard G16, round 5).
#ifndef LLA_ST_POLICY
#define LLA_ST_POLICY 0
#endif
#if LLA_ST_POLICY == 1
#define LLA_ST_SC " nt"
#elif LLA_ST_POLICY == 2
#define LLA_ST_SC " sc1 nt"
#elif LLA_ST_POLICY == 3
#define LLA_ST_SC " sc0 sc1 nt"
#elif LLA_ST_POLICY == 4
#define LLA_ST_SC " sc1"
#elif LLA_ST_POLICY == 5
#define LLA_ST_SC " sc0 sc1"
#else
#define LLA_ST_SC ""
#endif
template <typename V>
__device__ __forceinline__ void store16(void *dst, const V &v) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
  asm volatile("global_store_dwordx4 %0, %1, off" LLA_ST_SC "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
template <typename V>
__device__ __forceinline__ void store8(void *dst, const V &v) {
  static_assert(sizeof(V) == 8, "8-byte vectors only");
  asm volatile("global_store_dwordx2 %0, %1, off" LLA_ST_SC "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

// Epilogue shared by the GEMM kernels.  32x32 MFMA C/D layout with swapped operands: lane
// holds output row m = mw + 32 i + (lane & 31) and columns n = nw + 32 j + 8 g + 4 (lane >> 5)
// + e for register r = 4 g + e, i.e. 4 consecutive columns per register quad.
template <int EPI, int NI = 2, int NJ = 2, int COAL = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams &p, f32x16 (&acc)[NI][NJ], int mw,
                                              int nw, int r32, int hk) {
  f32x4 bias4[NJ][4];
  const int ncol = nw + 4 * hk;
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias4[j][g] = *reinterpret_cast<const f32x4 *>(p.bias + ncol + 32 * j + 8 * g);
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bias4[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = mw + 32 * i + r32;
    if (m >= p.M) continue;
    size_t row_off;
    const float *pos_row = nullptr;
    if constexpr (epi_base(EPI) == EPI_PATCH) {  // patch row b*49 + t -> token row b*50 + 1 + t, plus pos
      const int b = m / kPatches, t = m - b * kPatches;
      row_off = (size_t)(b * kTokens + 1 + t) * p.ldc;
      pos_row = p.pos + (1 + t) * kWidth;
    } else {
      row_off = (size_t)m * p.ldc;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (epi_base(EPI) == EPI_RELU || epi_base(EPI) == EPI_ADDRELU || epi_base(EPI) == EPI_F16) {
        if (nw + 32 * j >= p.n_store) continue;   // padding columns of a narrow convolution output: not stored
      }
      f32x4 old[4];  // residual / positional rows: 4 loads in flight per (i, j), then 4 stores
      if constexpr (epi_base(EPI) == EPI_RESID) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if constexpr (COAL) {  // ablation: lane-contiguous (wrong-element) addresses, same footprint
            const int lane = threadIdx.x & 63, q = (i * NJ + j) * 4 + g;
            old[g] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.C) +
                         (size_t)min(mw + 4 * q + (lane >> 4), p.M - 1) * p.ldc + nw + (lane & 15) * 4);
          } else
          old[g] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.C) + row_off +
                                                    ncol + 32 * j + 8 * g);
      } else if constexpr (epi_base(EPI) == EPI_PATCH) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          old[g] = *reinterpret_cast<const f32x4 *>(pos_row + ncol + 32 * j + 8 * g);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = ncol + 32 * j + 8 * g;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        v += bias4[j][g];
        if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU || epi_base(EPI) == EPI_RELU || epi_base(EPI) == EPI_ADDRELU) {
          if constexpr (epi_base(EPI) == EPI_QGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = quick_gelu(v[e]);
          }
          if constexpr (epi_base(EPI) == EPI_ADDRELU) {   // + identity branch (fp16 [M][ldr]), as the ResNet bottleneck does
            const f16x4 r4 = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const f16 *>(p.resid) + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
          }
          if constexpr (epi_base(EPI) == EPI_RELU || epi_base(EPI) == EPI_ADDRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          f16x4 h4;
#pragma unroll
          for (int e = 0; e < 4; ++e) h4[e] = (f16)v[e];
          if constexpr (COAL) {
            const int lane = threadIdx.x & 63, q = (i * NJ + j) * 4 + g;
            if constexpr (COAL == 2)  // 64 contiguous bytes per row, 8 rows per instruction
              *reinterpret_cast<f16x4 *>(reinterpret_cast<f16 *>(p.C) +
                  (size_t)min(mw + 8 * (q >> 1) + (lane >> 3), p.M - 1) * p.ldc + nw + 32 * (q & 1) + (lane & 7) * 4) = h4;
            else
            *reinterpret_cast<f16x4 *>(reinterpret_cast<f16 *>(p.C) +
                (size_t)min(mw + 4 * q + (lane >> 4), p.M - 1) * p.ldc + nw + (lane & 15) * 4) = h4;
          } else
          store8(reinterpret_cast<f16 *>(p.C) + row_off + n, h4);
        } else {
          if constexpr (COAL) {
            const int lane = threadIdx.x & 63, q = (i * NJ + j) * 4 + g;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.C) +
                (size_t)min(mw + 4 * q + (lane >> 4), p.M - 1) * p.ldc + nw + (lane & 15) * 4) = old[g] + v;
          } else {
            const f32x4 o = old[g] + v;
            store16(reinterpret_cast<float *>(p.C) + row_off + n, o);
            if constexpr (epi_base(EPI) == EPI_RESID) {
            }
          }
        }
      }
      if constexpr (epi_base(EPI) == EPI_RESID) {
      }
    }
  }
  // Every load issued above must be consumed on every path (rows beyond M skip the loop body):
  // a load hipcc still counts as pending at the end of this function makes it drain the whole
  // vector-memory queue -- the LDS-DMA ring -- at the top of the caller's next K-tile.
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(bias4[j][g]));
}

// Line-assembling epilogue for 32*NI x 64 wave tiles (NJ = 2) that lie fully inside M.  In the
// MFMA layout a store instruction touches 32 rows x 16 (fp16) or 32 (fp32) contiguous bytes and
// the L2 is left to assemble its lines from partial writes; address-only ablations
// (LLA_GEMM_DEBUG=3/5) priced that at 8 % of the layer and showed that 64 contiguous bytes per
// row are enough.  Here the raw fp32 accumulators go through a 2 KiB per-wave LDS scratch, 16 rows
// x 32 columns at a time (16-byte chunks XOR-swizzled by row pair), and come back
//   fp16 outputs: 8 lanes per row, 8 consecutive columns each -> one 16-byte store per lane,
//                 128 contiguous bytes per row, 8 rows per instruction;
//   fp32 outputs: 8 lanes per row, 4 consecutive columns each -> 128 contiguous bytes per row.
// Bias / QuickGELU / residual are applied in that layout, per element in the same order as
// gemm_epilogue (bit-identical results).  The scratch is private to the wave and the LDS executes
// one wave's operations in order: no barrier, only a compiler fence.
// NOLOAD (probe builds only, wrong results): the residual rows are not read -- what would the epilogue cost if they
// were already in registers?
template <int EPI, int NI, bool NOLOAD = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams &p, f32x16 (&acc)[NI][2],
                                                     int mw, int nw, int lane, unsigned char *scr) {
  constexpr bool kHalfOut = epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU || epi_base(EPI) == EPI_RELU ||
                            epi_base(EPI) == EPI_ADDRELU;
  const int r32 = lane & 31, hk = lane >> 5;
  const int r16 = r32 & 15, rhalf = r32 >> 4;
  unsigned char *wrow = scr + r16 * 128;
  const int wswz = r16 >> 1;
  auto wave_fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto stage = [&](int i, int j, int half) {  // this half's 16 rows x 32 columns -> scratch
    if (rhalf == half) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        *reinterpret_cast<f32x4 *>(wrow + (((2 * g + hk) ^ wswz) << 4)) = v;
      }
    }
    wave_fence();
  };
  if constexpr (kHalfOut) {
    // fp16 outputs are finished (bias, QuickGELU, conversion) in the MFMA layout and staged as fp16:
    // 16 rows x 64 columns = 2 KiB per pass, half the LDS bytes and half the passes of staging fp32
    // (all 8 waves run their epilogues at once; a K-tile-level trace priced the fp32 staging at
    // ~8 000 cycles per tile, mostly LDS bandwidth).
    f32x4 bias4[2][4];
    const int ncol = nw + 4 * hk;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bias4[j][g] = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + ncol + 32 * j + 8 * g)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
    // read side: 8 rows x 8 lanes x 16 bytes per instruction = whole 128-byte lines (a 40 MB burst of
    // such stores from all CUs lands in 4.7k cycles per tile, 32-byte pieces in 7.5k: tools/ubench/store_rate)
    const int row = lane >> 3, q = lane & 7;
    const unsigned char *rd0 = scr + row * 128 + ((q ^ (row >> 1)) << 4);
    const unsigned char *rd1 = scr + (row + 8) * 128 + ((q ^ ((row + 8) >> 1)) << 4);
    f16 *crow = reinterpret_cast<f16 *>(p.C) + (size_t)(mw + row) * p.ldc + nw + 8 * q;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f16x4 h[2][4];
      if constexpr (epi_base(EPI) == EPI_ADDRELU) {   // identity branch (fp16 [M][ldr]): all 8 loads of the unit first
        const f16 *rrow = reinterpret_cast<const f16 *>(p.resid) + (size_t)(mw + 32 * i + r32) * p.ldr + ncol;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) h[j][g] = *reinterpret_cast<const f16x4 *>(rrow + 32 * j + 8 * g);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          v += bias4[j][g];
          if constexpr (epi_base(EPI) == EPI_QGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
          }
          if constexpr (epi_base(EPI) == EPI_ADDRELU) {   // same order as gemm_epilogue: (acc + bias) + identity
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)h[j][g][e];
          }
          if constexpr (epi_base(EPI) == EPI_RELU || epi_base(EPI) == EPI_ADDRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) h[j][g][e] = (f16)v[e];
        }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (rhalf == half) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)  // columns 32 j + 8 g + 4 hk .. +3: 16-byte chunk 4 j + g, half hk
              *reinterpret_cast<f16x4 *>(wrow + (((4 * j + g) ^ wswz) << 4) + 8 * hk) = h[j][g];
        }
        wave_fence();
        const f16x8 o0 = *reinterpret_cast<const f16x8 *>(rd0);
        const f16x8 o1 = *reinterpret_cast<const f16x8 *>(rd1);
        wave_fence();
        f16 *dst = crow + (size_t)(32 * i + 16 * half) * p.ldc;
        store16(dst, o0);
        store16(dst + (size_t)8 * p.ldc, o1);
      }
    }
  } else {
    const int rrow = lane >> 3, rch = lane & 7;  // read side: 8 rows x 8 column quads, twice
    const unsigned char *rd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 8 * u + rrow;
      rd[u] = scr + row * 128 + ((rch ^ (row >> 1)) << 4);
    }
    // Units of 16 rows (i, half).  ALL vector-memory traffic of this epilogue is issued from inline asm
    // and waited for with hand-counted vmcnt: (a) the wave's vector-memory queue is in order, and the
    // residual / positional rows of unit k+1 are requested BEFORE unit k is transposed and stored, so
    // that using them only requires `vmcnt(8)` (4 stores of unit k + 4 loads of unit k+2 may stay in
    // flight) instead of a store round trip per unit; (b) loads and stores hipcc can see stay "pending"
    // in its model across the persistent loop's back edge and it then drains the LDS-DMA ring with
    // `vmcnt(0)` in front of every K-tile (see store16).  LDS-DMA pieces still in flight are OLDER than
    // everything here and only make the waits shorter-than-counted, never wrong.
    // (UNCONDITIONAL load -- no bias: a line of zeros.  Behind a branch hipcc merges an asm output with the other
    // path's value through register copies made right after the asm, i.e. before the data has arrived, and may then
    // reuse the destination registers, which the load overwrites later: seen in round 4 as wild addresses in a
    // four-quad version of this code; two quads happened to survive.)
    f32x4 bias_t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float *bp = p.bias ? p.bias + nw + 32 * j + 4 * rch : reinterpret_cast<const float *>(g_zero_line) + 4 * rch;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias_t[j]) : "v"(bp) : "memory");
    }
    unsigned coff[2][2];  // element offsets into C (32-bit: registers are scarce here)
    f32x4 old[2][4];
    float *const cbase = reinterpret_cast<float *>(p.C);
    auto request = [&](int k) {
      const int i = k >> 1, half = k & 1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = mw + 32 * i + 16 * half + 8 * u + rrow;
        const float *prow;
        if constexpr (epi_base(EPI) == EPI_PATCH) {  // patch row b*49 + t -> token row b*50 + 1 + t, plus pos
          const int bimg = m / kPatches, t = m - bimg * kPatches;
          coff[k & 1][u] = (unsigned)(m + bimg + 1) * (unsigned)p.ldc + (unsigned)(nw + 4 * rch);
          prow = p.pos + (1 + t) * kWidth + nw + 4 * rch;
        } else {
          coff[k & 1][u] = (unsigned)m * (unsigned)p.ldc + (unsigned)(nw + 4 * rch);
          prow = cbase + coff[k & 1][u];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (NOLOAD) asm volatile("v_mov_b32 %0, 0" : "=v"(old[k & 1][2 * j + u][0]) : "v"(prow + 32 * j) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off" LLA_RMW_SC : "=v"(old[k & 1][2 * j + u]) : "v"(prow + 32 * j) : "memory");
        }
      }
    };
    request(0);
    constexpr int kLoadsAhead = 4, kStoresBehind = 4;
#pragma unroll
    for (int k = 0; k < 2 * NI; ++k) {
      if (k + 1 < 2 * NI) request(k + 1);
      // rows of unit k have landed once only the younger operations are outstanding: 4 loads of unit
      // k+1 (if requested) + the stores of unit k-1 (if any); the two bias loads are older still
      const int younger = (k + 1 < 2 * NI ? kLoadsAhead : 0) + (k > 0 ? kStoresBehind : 0);
#define LLA_WAIT_OLD(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(old[k & 1][0]), "+v"(old[k & 1][1]), "+v"(old[k & 1][2]), "+v"(old[k & 1][3]), "+v"(bias_t[0]), "+v"(bias_t[1])::"memory")
      if (younger == 8) LLA_WAIT_OLD(8);
      else LLA_WAIT_OLD(4);
#undef LLA_WAIT_OLD
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        stage(k >> 1, j, k & 1);
        f32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = *reinterpret_cast<const f32x4 *>(rd[u]);
        wave_fence();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          v[u] += bias_t[j];
          const f32x4 o = old[k & 1][2 * j + u] + v[u];
          store16(cbase + coff[k & 1][u] + 32 * j, o);
        }
      }
    }
  }
}

// fp16 epilogue without the LDS round trips (full 32*NI x 64 wave tiles): after the swapped-operand
// 32x32 MFMA a row's packed fp16 outputs sit split across the half-waves (lane r: columns 8k .. 8k+3,
// lane r+32: columns 8k+4 .. 8k+7 of column group k).  One v_permlane32_swap per dword and group pair
// (k, k+1) hands the upper half's group-k data down and the lower half's group-(k+1) data up, so every
// lane owns 8 consecutive columns: one 16-byte store per lane and pair, 32 contiguous bytes per row and
// instruction, a row's 128 bytes within four consecutive instructions.  The LDS-staged variant makes
// longer row segments (64-128 B) but pays two LDS round trips per 16 rows: measured 7.7-10k cycles per
// 320x256 tile with all eight waves in it (VALU/LDS latency bound, not HBM: unchanged on half the CUs).
// Same fp32 arithmetic and roundings as gemm_epilogue -> bit-identical outputs.
template <int EPI, int NI>
__device__ __forceinline__ void gemm_epilogue_swap(const GemmParams &p, f32x16 (&acc)[NI][2], int mw,
                                                   int nw, int lane) {
  static_assert(epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU, "fp16 outputs only");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int r32 = lane & 31, hk = lane >> 5;
  f32x4 bias4[2][4];
  const int ncol = nw + 4 * hk;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bias4[j][g] = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + ncol + 32 * j + 8 * g)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
  // byte address of this lane's 16 bytes in row mw + r32, column group pair 0 of j = 0
  unsigned char *crow = reinterpret_cast<unsigned char *>(reinterpret_cast<f16 *>(p.C) + (size_t)(mw + r32) * p.ldc + nw) + 16 * hk;
  const size_t row_step = (size_t)32 * p.ldc * 2;
  auto pack4 = [&](int i, int j, int g, unsigned &lo, unsigned &hi) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
    v += bias4[j][g];
    if constexpr (epi_base(EPI) == EPI_QGELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
    }
    typedef f16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 a = {(f16)v[0], (f16)v[1]}, b = {(f16)v[2], (f16)v[3]};
    lo = __builtin_bit_cast(unsigned, a);
    hi = __builtin_bit_cast(unsigned, b);
  };
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        unsigned ax, ay, bx, by;
        pack4(i, j, k, ax, ay);
        pack4(i, j, k + 1, bx, by);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        const u32x4 out = {rx[0], ry[0], rx[1], ry[1]};
        store16(crow + i * row_step + (32 * j + 8 * k) * 2, out);
      }
  }
}


}  // namespace

// gemm_q4.hip: the four-wave 256 x 256 x 64 kernel (A_PLAIN operands; epi = EPI_F16 / EPI_QGELU / EPI_RESID).
// Returns LLA_EINVAL for shapes it does not take (the caller then uses the ping-pong kernel).
int launch_q4(int epi, const GemmParams &p, hipStream_t st);
// gemm_w8.hip: the eight-wave 256 x 256 x 64 kernel on v_mfma_f32_16x16x32_f16 (A_PLAIN operands; EPI_F16 / EPI_QGELU; every M).
int launch_w8(int epi, const GemmParams &p, hipStream_t st);
int num_cus();
}  // namespace lla
