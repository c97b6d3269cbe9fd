// CLIP ViT-B/32 visual tower for gfx950 (MI355X): fp16 storage, fp32 accumulate,
// fp32 residual stream, fp32 LayerNorm / softmax statistics.
//
// Stands in for `z = self.clip(X)` at hub/compressor.py:93 (clip==1.0
// VisionTransformer.forward; recipe: SURVEY.md 8(a) row A10, section 9.3).
//
// Kernels (all hand-written for wave64 / MFMA 32x32x16 f16):
//   gemm_f16_kernel      C[M][N] = A[M][K] * W[N][K]^T, 128x128x64 workgroup tile,
//                        4 waves (2x2) of 64x64, LDS double buffer with an XOR
//                        swizzle that makes every ds_read_b128 conflict free,
//                        register-staged global prefetch one K-tile ahead, fused
//                        epilogues (bias / QuickGELU / residual / patch scatter+pos),
//                        A-operand addressing modes that read 32x32 patches straight
//                        out of NHWC or NCHW image batches (no im2col pass).
//   layernorm768_kernel  one wave per 768-wide row, float4 loads, fp32 two-pass.
//   ln_pre_ln1_kernel    class-token insert + ln_pre (in place, fp32) + layer-0 ln_1.
//   attention50_kernel   one wave per (image, head): S^T = K Q^T and O^T = V^T P^T on
//                        MFMA; the softmax row of a query lives in two lanes, and the
//                        probabilities feed the second MFMA without leaving registers.
#include "gemm_common.h"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace lla {
namespace {

// One K-tile (BK = 64 = 4 MFMA k-steps) of a 64x64 wave tile out of LDS, with the
// fragment reads of step s+1 issued BEFORE the MFMAs of step s (register double buffer):
// the two waves of a SIMD run in lock-step behind the workgroup barrier, so without this the
// LDS latency of every k-step is exposed for both of them at the same time.
// `late()` runs between the MFMAs of steps 2 and 3: the LDS-DMA refill is issued there,
// because hipcc models global_load_lds as a FLAT access that may touch LDS and from then on
// only emits `s_waitcnt lgkmcnt(0)` -- placed late, the counted waits of steps 0..2 survive.
template <typename Late>
__device__ __forceinline__ void wave_tile_k64(const f16 *sa_row, const f16 *sb_row, int hk, int swz,
                                              f32x16 (&acc)[2][2], Late late) {
  f16x8 af[2][2], bf[2][2];
  auto fetch = [&](int s, int buf) {
    const int chunk = ((2 * s + hk) ^ swz) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[buf][i] = *reinterpret_cast<const f16x8 *>(sa_row + i * 32 * BK + chunk);
      bf[buf][i] = *reinterpret_cast<const f16x8 *>(sb_row + i * 32 * BK + chunk);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < 3) fetch(s + 1, (s + 1) & 1);
    if (s == 3) late();
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs (hipcc sinks it)
    // operands swapped on purpose: D^T[n][m] puts 4 CONSECUTIVE output columns of one
    // output row in each lane's register quad -> 8/16-byte epilogue stores
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// GLDS = true : operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 16 B per
//               lane, no staging VGPRs, no ds_write pass); the XOR swizzle is applied to
//               the per-lane SOURCE address because the LDS destination of an LDS-DMA is
//               wave-base + lane * 16 (linear).
// GLDS = false: register-staged variant of the same layout (kept for A/B and as a
//               reference for the DMA path).
template <int EPI, int AMODE, bool GLDS>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f16_kernel(GemmParams p) {
  kernel_acquire();
  // [buffer][A|B][128 rows][64 halfs]; 16-byte chunk c of row r sits at chunk
  // c ^ ((r >> 1) & 7): 16 rows that differ mod 16 then cover all 16 slots of the
  // 256-byte bank row, which is what each ds_read_b128 lane group touches.
  __shared__ __attribute__((aligned(16))) f16 smem[2][2][BM * BK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int r32 = lane & 31, hk = lane >> 5;

  const int tiles_n = p.N / BN;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // staging assignment: thread owns chunk (tid & 7) of rows (tid >> 3) + 32 i, i.e. LDS
  // chunk index tid + 256 i -- linear in the lane id, as the LDS-DMA requires
  const int srow = tid >> 3, pc = tid & 7;
  const int lc = pc ^ ((srow >> 1) & 7);  // same for all four rows (32 i is 0 mod 16)
  const f16 *a_ptr[4];
  const f16 *b_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 32 * i;
    if (m >= p.M) m = p.M - 1;  // clamp: loaded, never stored
    if constexpr (AMODE == A_PLAIN)
      a_ptr[i] = p.A + (size_t)m * p.lda + lc * 8;
    else
      a_ptr[i] = p.A + patch_rowoff<AMODE>(m);
    b_ptr[i] = p.W + (size_t)(n0 + srow + 32 * i) * p.K + lc * 8;
  }

  f16x8 ra[4], rb[4];
  auto a_off = [&](int kt) {
    if constexpr (AMODE == A_PLAIN) return kt * BK; else return patch_koff<AMODE>(kt * BK + lc * 8);
  };
  auto gload = [&](int kt) {  // register-staged path
    const int aoff = a_off(kt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const f16x8 *>(a_ptr[i] + aoff);
      rb[i] = *reinterpret_cast<const f16x8 *>(b_ptr[i] + kt * BK);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      reinterpret_cast<f16x8 *>(smem[buf][0])[tid + 256 * i] = ra[i];
      reinterpret_cast<f16x8 *>(smem[buf][1])[tid + 256 * i] = rb[i];
    }
  };
  auto dma = [&](int kt, int buf) {  // LDS-DMA path: 8 x 1 KiB per wave per K-tile
    const int aoff = a_off(kt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_ptr[i] + aoff),
                                       (lptr_t)(smem[buf][0] + (wid * 64 + 256 * i) * 8), 16, 0, LLA_DMA_AUX);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_ptr[i] + kt * BK),
                                       (lptr_t)(smem[buf][1] + (wid * 64 + 256 * i) * 8), 16, 0, LLA_DMA_AUX);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int swz = (r32 >> 1) & 7;
  const int a_row_base = (wr * 64 + r32) * BK;
  const int b_row_base = (wc * 64 + r32) * BK;

  const int nk = p.K / BK;
  if constexpr (GLDS) {
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if constexpr (!GLDS) {
      if (kt + 1 < nk) gload(kt + 1);
    }
    wave_tile_k64(smem[cur][0] + a_row_base, smem[cur][1] + b_row_base, hk, swz, acc, [&] {
      if constexpr (GLDS) {
        if (kt + 1 < nk) dma(kt + 1, cur ^ 1);
      }
    });
    if constexpr (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (kt + 1 < nk) lstore(cur ^ 1);
    }
    __syncthreads();
  }

  gemm_epilogue<EPI>(p, acc, m0 + wr * 64, n0 + wc * 64, r32, hk);
  kernel_release();
}

// One K-tile of a 64x64 wave tile, hand-scheduled.  hipcc cannot emit counted LDS waits while
// an LDS-DMA is in flight (it models global_load_lds as a FLAT access that may touch LDS and
// degrades every `s_waitcnt lgkmcnt(N)` to N = 0), so the ds_read / wait / MFMA stream is
// written out: 12 fragment reads up front, the last 4 after the first MFMA group, counted
// waits (LDS returns in order) so that each k-step starts as soon as ITS four fragments are
// in.  Every fragment has its own registers (no reuse inside the block).
// Operand map: %0..%3 acc[i][j] (i major); %4+4s.. = af[s][0], af[s][1], bf[s][0], bf[s][1];
// %20+s = LDS byte address of A row/chunk for step s (i = 1 at +4096); %24+s likewise for B.
// MFMA operands are swapped (srcA = W fragment, srcB = activation fragment): see gemm_epilogue.
#define LLA_RD4(S, FA0, FA1, FB0, FB1, AA, BA)                                    \
  "ds_read_b128 " FA0 ", " AA "\n\t"                                              \
  "ds_read_b128 " FA1 ", " AA " offset:4096\n\t"                                  \
  "ds_read_b128 " FB0 ", " BA "\n\t"                                              \
  "ds_read_b128 " FB1 ", " BA " offset:4096\n\t"
#define LLA_MM4(FA0, FA1, FB0, FB1)                                               \
  "v_mfma_f32_32x32x16_f16 %0, " FB0 ", " FA0 ", %0\n\t"                          \
  "v_mfma_f32_32x32x16_f16 %1, " FB1 ", " FA0 ", %1\n\t"                          \
  "v_mfma_f32_32x32x16_f16 %2, " FB0 ", " FA1 ", %2\n\t"                          \
  "v_mfma_f32_32x32x16_f16 %3, " FB1 ", " FA1 ", %3\n\t"

__device__ __forceinline__ void wave_tile_k64_asm(unsigned a_addr, unsigned b_addr, int hk, int swz,
                                                  f32x16 (&acc)[2][2]) {
  unsigned aa[4], ba[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned c = (unsigned)(((2 * s + hk) ^ swz) * 16);
    aa[s] = a_addr + c;
    ba[s] = b_addr + c;
  }
  f16x8 f[16];
  asm volatile(
      LLA_RD4(0, "%4", "%5", "%6", "%7", "%20", "%24")
      LLA_RD4(1, "%8", "%9", "%10", "%11", "%21", "%25")
      LLA_RD4(2, "%12", "%13", "%14", "%15", "%22", "%26")
      "s_waitcnt lgkmcnt(8)\n\t"
      LLA_MM4("%4", "%5", "%6", "%7")
      LLA_RD4(3, "%16", "%17", "%18", "%19", "%23", "%27")
      "s_waitcnt lgkmcnt(8)\n\t"
      LLA_MM4("%8", "%9", "%10", "%11")
      "s_waitcnt lgkmcnt(4)\n\t"
      LLA_MM4("%12", "%13", "%14", "%15")
      "s_waitcnt lgkmcnt(0)\n\t"
      LLA_MM4("%16", "%17", "%18", "%19")
      : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]),
        "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]),
        "=&v"(f[7]), "=&v"(f[8]), "=&v"(f[9]), "=&v"(f[10]), "=&v"(f[11]), "=&v"(f[12]),
        "=&v"(f[13]), "=&v"(f[14]), "=&v"(f[15])
      : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba[0]), "v"(ba[1]), "v"(ba[2]),
        "v"(ba[3])
      : "memory");
}

// 256x128x64 workgroup tile, 8 waves (4 x 2) of 64x64, THREE LDS stages (3 x 48 KiB) fed by
// LDS-DMA two K-tiles ahead.  One raw s_barrier per K-tile; the DMA queue is never drained
// in the loop: `s_waitcnt vmcnt(6)` retires exactly the six 1-KiB pieces of the tile about
// to be read and leaves the next tile's six in flight across the barrier.
#ifndef LLA_GROUP_M
#define LLA_GROUP_M 4   // row tiles per group of the tile walk (A/B: make variant DEFS=-DLLA_GROUP_M=n)
#endif
constexpr int BM2 = 256, BN2 = 128, kStages = 3, kGroupM = LLA_GROUP_M;
constexpr int kStageHalfs = (BM2 + BN2) * BK;

template <int EPI, int AMODE, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256_f16_kernel(GemmParams p) {
  kernel_acquire();
  __shared__ __attribute__((aligned(16))) f16 smem[kStages * kStageHalfs];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int r32 = lane & 31, hk = lane >> 5;

  // Tile order inside an XCD's contiguous run: groups of kGroupM row-tiles swept across all
  // column-tiles with the row index fastest, so the ~32 tiles an XCD has in flight form a
  // (kGroupM x 8) patch whose A and W panels fit its 4 MiB L2 and are shared while hot.
  const int tiles_n = p.N / BN2;
  const int tiles_m = (p.M + BM2 - 1) / BM2;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int per_group = kGroupM * tiles_n;
  const int grp = logical / per_group;
  const int in_grp = logical - grp * per_group;
  const int gh = (tiles_m - grp * kGroupM) < kGroupM ? (tiles_m - grp * kGroupM) : kGroupM;
  const int tile_n = in_grp / gh;
  const int tile_m = grp * kGroupM + (in_grp - tile_n * gh);
  const int m0 = tile_m * BM2, n0 = tile_n * BN2;

  // staging: LDS chunk index of thread = tid + 512 i (A: i < 4, B: i < 2) -> row (tid >> 3) + 64 i
  const int srow = tid >> 3, pc = tid & 7;
  const int lc = pc ^ ((srow >> 1) & 7);
  const f16 *a_ptr[4];
  const f16 *b_ptr[2];
  int cy[4], cx[4];   // A_CONV3: pixel coordinates of this thread's four rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 64 * i;
    if (m >= p.M) m = p.M - 1;
    if constexpr (AMODE == A_PLAIN) {
      a_ptr[i] = p.A + (size_t)m * p.lda + lc * 8;
    } else if constexpr (AMODE == A_CONV3) {
      // implicit 3x3 / stride 1 / pad 1 convolution over NHWC [B][H][W][lda]: row m is output pixel
      // (b, y, x) and the K index runs over (kh, kw, c) -- the order the weights are packed in; a K-tile of
      // 64 channels lies inside one tap because cin % 64 == 0
      const int pix = p.conv_h * p.conv_w;
      const int b = m / pix, r = m - b * pix;
      cy[i] = r / p.conv_w;
      cx[i] = r - cy[i] * p.conv_w;
      // centre tap; with 32 input channels a K-tile holds TWO taps: chunks 0-3 the first, 4-7 the second
      a_ptr[i] = p.A + (size_t)m * p.lda + (p.conv_cin >= BK ? lc : (lc & 3)) * 8;
    } else {
      a_ptr[i] = p.A + patch_rowoff<AMODE>(m);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) b_ptr[i] = p.W + (size_t)(n0 + srow + 64 * i) * p.K + lc * 8;
  const int conv_cpt = (AMODE == A_CONV3 && p.conv_cin >= BK) ? p.conv_cin / BK : 1;   // K-tiles per tap

  auto dma = [&](int kt, int stage) {
    int aoff = 0;
    if constexpr (AMODE == A_PLAIN) aoff = kt * BK;
    else if constexpr (AMODE != A_CONV3) aoff = patch_koff<AMODE>(kt * BK + lc * 8);
    f16 *sa = smem + stage * kStageHalfs;
    f16 *sb = sa + BM2 * BK;
    if constexpr (AMODE == A_CONV3) {
      int tap, c0 = 0;
      if (p.conv_cin >= BK) { tap = kt / conv_cpt; c0 = (kt - tap * conv_cpt) * BK; }
      else tap = 2 * kt + (lc >> 2);          // (per lane; tap 9 = the zero padding of K = 288 -> 320)
      const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
      const int off = (dy * p.conv_w + dx) * p.lda + c0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool inside = tap < 9 && (unsigned)(cy[i] + dy) < (unsigned)p.conv_h &&
                            (unsigned)(cx[i] + dx) < (unsigned)p.conv_w;
        const f16 *src = inside ? a_ptr[i] + off : g_zero_line + lc * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + (wid * 64 + 512 * i) * 8), 16, 0, LLA_DMA_AUX);
      }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_ptr[i] + aoff),
                                       (lptr_t)(sa + (wid * 64 + 512 * i) * 8), 16, 0, LLA_DMA_AUX);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_ptr[i] + kt * BK),
                                       (lptr_t)(sb + (wid * 64 + 512 * i) * 8), 16, 0, LLA_DMA_AUX);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int swz = (r32 >> 1) & 7;
  const int a_row_base = (wr * 64 + r32) * BK;
  const int b_row_base = BM2 * BK + (wc * 64 + r32) * BK;

  const int nk = p.K / BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;  // LDS byte address of stage 0
  dma(0, 0);
  if (nk > 1) dma(1, 1);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed for THIS wave once at most the next tile's 6 pieces remain in flight
    // (s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14.)
    // The builtin, not inline asm: hipcc's own wait-count pass must SEE the lgkmcnt(0),
    // otherwise it keeps treating the LDS-DMA as an outstanding FLAT access and degrades every
    // counted LDS wait of the next K-tile to lgkmcnt(0).
    if (DBG == 1) __builtin_amdgcn_s_waitcnt(0x0070);
    else if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0076);  // vmcnt(6) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0) lgkmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // => landed for every wave; previous stage free for all
    asm volatile("" ::: "memory");
    if (DBG != 1 && kt + 2 < nk) {  // refill the stage every wave finished reading before the barrier
      int st2 = stage + 2;
      if (st2 >= kStages) st2 -= kStages;
      dma(kt + 2, st2);
    }
    const unsigned sbytes = lds_base + (unsigned)(stage * kStageHalfs * 2);
    if (DBG != 2) wave_tile_k64_asm(sbytes + a_row_base * 2, sbytes + b_row_base * 2, hk, swz, acc);
    if (++stage == kStages) stage = 0;
  }
  // MFMA results are read by VALU next: cover the XDL write -> VALU read hazard by hand
  // (hipcc pads nothing for instructions inside an asm statement)
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
  gemm_epilogue<EPI>(p, acc, m0 + wr * 64, n0 + wc * 64, r32, hk);
  kernel_release();
}

// ---------------------------------------------------------------------------
// Persistent GEMM: 256 x (128 | 256) x 64 tiles, 8 waves (2 x 4) of 128 x (32 | 64), one
// workgroup per CU walking its share of the tiles with the operand stream running ACROSS tile
// boundaries (no per-tile prologue bubble; the epilogue's stores drain under the next tile's
// first K-step).  Two 64-KiB LDS stages; the LDS-DMA pieces of K-tile t+1 are issued two at a
// time between the MFMA groups of K-tile t, so no wave sits in a burst of VMEM issue while
// its SIMD's matrix pipe idles.  The DMA is emitted as inline asm on purpose: hipcc then does
// not know an LDS-writing FLAT op is pending and keeps COUNTED lgkmcnt waits for the
// compiler-scheduled ds_read / MFMA stream (fragment reads of k-step s+1 issued before the
// MFMAs of step s, pinned with sched_barrier).
// ---------------------------------------------------------------------------


__device__ __forceinline__ void dma16(const f16 *gsrc, unsigned lds_dst_wave_base) {
  // LDS destination = M0 + lane * 16.  M0 is saved / restored: it belongs to the compiler.
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %2\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off" LLA_DMA_SC "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_wave_base)
               : "memory");
}

template <int EPI, int AMODE, int NJ, int KB, int STAGES, int DBG = 0, int NI = 4>
__global__ __launch_bounds__(512, 2) void gemm_persistent_kernel(GemmParams p) {
  kernel_acquire();
  // NI = 32-row MFMA tiles per wave along M: workgroup tile height PBM = 64 * NI (256 or 320;
  // 320 divides M = 51200 into 160 row-tiles, which balances 3-column-tile GEMMs on 256 CUs)
  constexpr int PBM = 64 * NI;
  // KB = K-extent of one LDS stage (32 or 64 halfs per row); STAGES-deep ring, the DMA runs
  // D = STAGES - 1 K-tiles ahead.  A loaded HBM/MALL round trip is ~4-5k cycles on this chip
  // while a 64-deep K-tile is 1-2k cycles of MFMA, so the ring has to cover several tiles:
  // KB = 32 buys twice the depth for the same LDS bytes.
  constexpr int PBN = 128 * NJ;
  constexpr int CH = KB / 8;                 // 16-byte chunks per LDS row
  constexpr int ROWS_I = 512 / CH;           // rows covered by one 512-thread DMA sweep
  constexpr int kAPieces = PBM / ROWS_I, kBPieces = PBN / ROWS_I, kPieces = kAPieces + kBPieces;
  constexpr int kABytes = PBM * KB * 2, kBBytes = PBN * KB * 2, kStageBytes = kABytes + kBBytes;
  constexpr int KSTEPS = KB / 16, D = STAGES - 1;
  static_assert(STAGES * kStageBytes <= 160 * 1024, "LDS ring too large");
  static_assert((D - 1) * kPieces <= 63, "vmcnt field");
  __shared__ __attribute__((aligned(16))) f16 smem[STAGES * kStageBytes / 2];
  // DBG 4 = direct (MFMA-layout) epilogue, DBG 3 = address-only coalescing ablation
  constexpr bool kStaged = NJ == 2 && DBG != 3 && DBG != 4 && DBG != 5;
  static_assert(!kStaged || STAGES * kStageBytes + 8 * 2048 <= 160 * 1024, "no room for the epilogue scratch");
  __shared__ __attribute__((aligned(16))) unsigned char epi_scr[kStaged ? 8 * 2048 : 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int r32 = lane & 31, hk = lane >> 5;

  // ---- which tiles are mine (XCD-contiguous logical range, grouped 4-row-tile order)
  const int tiles_n = p.N / PBN, tiles_m = (p.M + PBM - 1) / PBM;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int q = total >> 3, r = total & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int count = q + (xcd < r ? 1 : 0);
  const int n_my = slot < count ? (count - slot + nslots - 1) / nslots : 0;
  if (n_my == 0) return;
  auto tile_origin = [&](int j, int &m0, int &n0) {
    int logical = start + slot + j * nslots;
    if (p.rev) logical = total - 1 - logical;
    const int per_group = kGroupM * tiles_n;
    const int grp = logical / per_group;
    const int in_grp = logical - grp * per_group;
    const int gh = (tiles_m - grp * kGroupM) < kGroupM ? (tiles_m - grp * kGroupM) : kGroupM;
    const int tn = in_grp / gh;
    m0 = (grp * kGroupM + (in_grp - tn * gh)) * PBM;
    n0 = tn * PBN;
  };

  // ---- loader state: row pointers of the tile being streamed in.  LDS chunk index of a
  // thread = tid + 512 i  ->  row (tid / CH) + ROWS_I * i, physical chunk tid % CH; the XOR
  // swizzle goes on the SOURCE chunk (the DMA destination is lane-linear).
  const f16 *a_ptr[kAPieces];
  const f16 *b_ptr[kBPieces];
  auto set_load_tile = [&](int j) {
    // once per tile: the thread's row / chunk are recomputed from a laundered tid rather than kept
    // in registers across the K loop (they spilled, and a scratch reload here waits on vmcnt,
    // i.e. on the DMA pieces just issued)
    int lt = tid;
    asm volatile("" : "+v"(lt));
    const int srow = lt / CH, pc = lt % CH;
    const int lc = KB == 64 ? (pc ^ ((srow >> 1) & 7)) : (pc ^ ((srow >> 2) & 3));
    int m0, n0;
    tile_origin(j, m0, n0);
#pragma unroll
    for (int i = 0; i < kAPieces; ++i) {
      int m = m0 + srow + ROWS_I * i;
      if (m >= p.M) m = p.M - 1;
      if constexpr (AMODE == A_PLAIN) a_ptr[i] = p.A + (size_t)m * p.lda + lc * 8;
      else a_ptr[i] = p.A + patch_rowoff<AMODE>(m);
    }
#pragma unroll
    for (int i = 0; i < kBPieces; ++i)
      b_ptr[i] = p.W + (size_t)(n0 + srow + ROWS_I * i) * p.K + lc * 8;
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned wave_off = (unsigned)wid * 1024u;
  auto dma_piece = [&](int piece, int kt, int stage) {  // piece < kAPieces: A, else B
    const unsigned sb = lds_base + (unsigned)stage * kStageBytes + wave_off;
    if (piece < kAPieces) {
      int aoff;
      if constexpr (AMODE == A_PLAIN) {
        aoff = kt * KB;
      } else {  // patch gather: the K offset depends on the thread's chunk
        const int srow_p = tid / CH, pc_p = tid % CH;
        const int lc_p = KB == 64 ? (pc_p ^ ((srow_p >> 1) & 7)) : (pc_p ^ ((srow_p >> 2) & 3));
        aoff = patch_koff<AMODE>(kt * KB + lc_p * 8);
      }
      dma16(a_ptr[piece] + aoff, __builtin_amdgcn_readfirstlane(sb + (unsigned)piece * 8192u));
    } else {
      dma16(b_ptr[piece - kAPieces] + kt * KB,
            __builtin_amdgcn_readfirstlane(sb + kABytes + (unsigned)(piece - kAPieces) * 8192u));
    }
  };

  f32x16 acc[NI][NJ];
  auto zero_acc = [&] {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

  const int swz = KB == 64 ? ((r32 >> 1) & 7) : ((r32 >> 2) & 3);
  const int a_row_base = (wr * 32 * NI + r32) * KB;                   // halfs, within the A tile
  const int b_row_base = (kABytes / 2) + (wc * 32 * NJ + r32) * KB;   // halfs, within the stage

  const int nk = p.K / KB;
  const int total_iters = n_my * nk;
  int ld_j = 0, ld_kt = 0, ld_stage = 0, issued = 0;  // load cursor
  set_load_tile(0);
  auto advance_load = [&] {
    ++issued;
    if (++ld_stage == STAGES) ld_stage = 0;
    if (++ld_kt == nk) { ld_kt = 0; ++ld_j; if (ld_j < n_my) set_load_tile(ld_j); }
  };
  for (int d = 0; d < D && d < total_iters; ++d) {  // prologue: fill D stages
#pragma unroll
    for (int pce = 0; pce < kPieces; ++pce) dma_piece(pce, ld_kt, ld_stage);
    advance_load();
  }

  int cj = 0, ckt = 0, m0c, n0c, stage = 0;
  bool pend = false;  // a finished tile whose epilogue has not run yet
  int pm0 = 0, pn0 = 0;
  auto run_epilogue = [&] {
    // the lane index is laundered so that the epilogue's per-lane address arithmetic is redone
    // per tile instead of being hoisted out of the K loop (where it only adds register pressure)
    int el = lane;
    asm volatile("" : "+v"(el));
    if constexpr (kStaged) {
      if (pm0 + wr * 32 * NI + 32 * NI <= p.M) {  // wave-uniform; ragged last rows take the direct path
        gemm_epilogue_staged<EPI, NI>(p, acc, pm0 + wr * 32 * NI, pn0 + wc * 64, el, epi_scr + wid * 2048);
        return;
      }
    }
    gemm_epilogue<EPI, NI, NJ, DBG == 3 ? 1 : (DBG == 5 ? 2 : 0)>(p, acc, pm0 + wr * 32 * NI, pn0 + wc * 32 * NJ, el & 31, el >> 5);
  };
  tile_origin(0, m0c, n0c);
  for (int it = 0; it < total_iters; ++it) {
    // K-tile `it` has landed for this wave once only the younger tiles' pieces are in flight
    // (loads complete in order; any store still pending only makes this wait longer) ...
    unsigned long long t_w0 = 0, t_w1 = 0;
    if constexpr (DBG == 9) t_w0 = __builtin_amdgcn_s_memtime();
    const int ahead = issued - it - 1;  // tiles issued after `it`
    if (ahead >= 3 && D >= 4) __builtin_amdgcn_s_waitcnt(0x0070 | ((3 * kPieces) & 15) | (((3 * kPieces) >> 4) << 14));
    else if (ahead == 2 && D >= 3) __builtin_amdgcn_s_waitcnt(0x0070 | ((2 * kPieces) & 15) | (((2 * kPieces) >> 4) << 14));
    else if (ahead == 1 && D >= 2) __builtin_amdgcn_s_waitcnt(0x0070 | ((1 * kPieces) & 15) | (((1 * kPieces) >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    asm volatile("" ::: "memory");
    unsigned long long t_wm = 0;
    if constexpr (DBG == 9) t_wm = __builtin_amdgcn_s_memtime();   // own pieces landed; now the barrier
    __builtin_amdgcn_s_barrier();  // ... and for every wave; the stage read last iteration is free
    asm volatile("" ::: "memory");
    if constexpr (DBG == 9) t_w1 = __builtin_amdgcn_s_memtime();
    const bool more = issued < total_iters;
    // The finished tile's epilogue runs HERE, after the wait + barrier of the next K-tile and
    // before its MFMAs, not at the end of the tile: the wave has a single vmcnt, so stores
    // issued just before a wait would be waited for (a full store round trip per tile, and the
    // output traffic was measured to cost 26 % -- DESIGN.md); issued here they have a whole
    // K-tile of MFMA work to drain before the next wait.
    if (pend) {
      run_epilogue();
      if (DBG == 2) {
        zero_acc();
      } else {
        // the next tile's first k-step overwrites every accumulator (C = 0 operand): tell the register
        // allocator the old values are dead, so that the epilogue may reuse their registers as it
        // consumes them (without this the residual epilogue of the 320-row tile spilled 114 registers)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_nondeterministic_value(acc[i][j]);
      }
      pend = false;
    }

    const f16 *sbase = smem + stage * (kStageBytes / 2);
    const f16 *sa_row = sbase + a_row_base;
    const f16 *sb_row = sbase + b_row_base;
    f16x8 fa[2][NI], fb[2][NJ];
    auto fetch = [&](int s, int buf) {
      const int chunk = ((2 * s + hk) ^ swz) * 8;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        fa[buf][i] = *reinterpret_cast<const f16x8 *>(sa_row + i * 32 * KB + chunk);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        fb[buf][j] = *reinterpret_cast<const f16x8 *>(sb_row + j * 32 * KB + chunk);
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 1 < KSTEPS) fetch(s + 1, (s + 1) & 1);
      if (more && DBG != 1) {  // refill the stage freed by the barrier, a few pieces per k-step
#pragma unroll
        for (int pce = 0; pce < kPieces; ++pce)
          if (pce * KSTEPS / kPieces == s) dma_piece(pce, ld_kt, ld_stage);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (DBG != 2) {
        if (s == 0 && ckt == 0) {  // first k-step of an output tile: C = 0 as an inline operand, no zeroing pass
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0][j], fa[0][i], zero16, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s & 1][j], fa[s & 1][i], acc[i][j], 0, 0, 0);
        }
      } else {  // ablation: keep the fragment reads alive, skip the matrix pipe
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fa[s & 1][i]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(fb[s & 1][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DBG == 9) {
      if (p.trace && wid == 0 && lane == 0 && (blockIdx.x & 31) == 0 && it < 128) {
        unsigned long long *t = p.trace + ((size_t)(blockIdx.x >> 5) * 128 + it) * 4;
        t[0] = t_w0; t[1] = t_w1; t[2] = __builtin_amdgcn_s_memtime();
        t[3] = ((unsigned long long)__builtin_amdgcn_s_memrealtime() << 24) | ((t_wm - t_w0) << 8 & 0xffff00ull) | (unsigned long long)ckt;  // 100 MHz clock | vmcnt-wait cycles | K-tile
      }
    }
    if (more) advance_load();
    if (++stage == STAGES) stage = 0;
    if (++ckt == nk) {
      pend = true; pm0 = m0c; pn0 = n0c;
      ckt = 0;
      if (++cj < n_my) tile_origin(cj, m0c, n0c);
    }
  }
  if (pend) run_epilogue();
  kernel_release();
}

// ---------------------------------------------------------------------------
// Ping-pong persistent GEMM (round 2): (64 NI) x 256 x 64 tiles, 8 waves as 2 (M) x 4 (N), each
// wave a (32 NI) x 64 output tile -- the same tile and epilogues as gemm_persistent_kernel, but
// the K loop is organised so that the two waves of a SIMD work out of phase instead of running
// the same segment in lock-step:
//
//  * A K-tile is walked in NI phases, one 32-row A fragment each.  A phase has a MATRIX segment
//    (8 MFMAs, 32 x 64 x 64; the operand registers of k-step s are refilled from LDS behind the
//    MFMAs of k-step s+1 -- next A fragment, in the last phase the next K-tile's B fragments and
//    first A fragment) and a LOAD segment (the k-step-3 refill, the phase's LDS-DMA pieces, the
//    waits).  The wave's 64-column B operand stays in 32 VGPRs for the whole K-tile.
//  * ONE s_barrier per phase.  Between two barriers the upper wave row (wr = 0) runs
//    matrix(p), load(p+1) and the lower row load(p), matrix(p): the load segments sit under the
//    partner's MFMAs, and where the two matrix segments overlap the SIMD's matrix pipe takes MFMAs
//    from both waves (one wave alone issues a dependent-accumulator MFMA only every ~37 cycles).
//  * Fragment-major K-tiles free LDS progressively: the 64 rows of A fragment p (32 per wave row,
//    one 8 KiB DMA piece, each half re-filled by the wave row that reads it) are dead after phase p,
//    the B region after phase 0.  A piece is refilled one phase after its last read, B pieces from
//    phase 2 on, always with the K-tile AFTER the next one: with two 64/72 KiB stages the LDS-DMA runs
//    1-2 K-tiles ahead and one counted `s_waitcnt vmcnt` per K-tile never drains the queue.
//  * DMA addresses are SGPR base + one 32-bit VGPR offset per piece (global_load_lds ... saddr):
//    NI + 1 address VGPRs instead of 2 (NI + 4).
//
// Hazards.  Interval g = t NI + p runs between barriers g and g+1.  Slot (t, p) = load(t, p) is
// executed by the upper row in interval g-1 and by the lower row in interval g.
//   WAR  A piece q of K-tile t: last read in load(t, q) (k-step 3); its halves are rewritten by the
//        row that read them, in slot (t, q+1), after that row's lgkmcnt(0).  B of K-tile t: last read
//        in slot (t, 0), by the lower row in interval t NI, waited for before barrier t NI + 1; B
//        pieces are rewritten from slot (t, 2) on, i.e. not before interval t NI + 1.
//   RAW  K-tile t+1 is first read in matrix(t, NI-1), by the upper row in interval t NI + NI - 1.
//        Every wave confirms its own pieces of K-tile t+1 (counted vmcnt) in its last load segment
//        before barrier t NI + NI - 1: slot (t, NI-1) for the upper row, slot (t, NI-2) for the lower.
// ---------------------------------------------------------------------------
// A wave-uniform pointer the compiler can no longer prove uniform (it went through VALU integer
// division) back into an SGPR pair.
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *ptr) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const unsigned char *>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void dma16s(unsigned voff, const void *sbase, unsigned lds_dst_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %3\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2" LLA_DMA_SC "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst_wave_base)
               : "memory");
}

// DBG (ablation build only): 1 = no LDS-DMA in the loop, 2 = no MFMAs, 4 = no fragment reads in the
// loop, 5 = linear DMA source lanes (wrong data); TRACE: s_memtime stamps (sums written to p.trace).
// LLA_GEMM_DEBUG = 9 selects the traced plain kernel, 10 + d the traced ablation d.
template <int EPI, int AMODE, int NI, int DBG = 0, bool TRACE = false, bool SWAP_EPI = true>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
  kernel_acquire();
  constexpr int PBM = 64 * NI, PBN = 256;
  constexpr int kABytes = PBM * 128, kBBytes = PBN * 128, kStageBytes = kABytes + kBBytes;
  // slot of the K-tile walk in which B piece i of K-tile u is issued: (u-2, 2+i) while 2+i < NI, else (u-1, 2+i-NI)
  constexpr auto b_slot = [](int i) { return 2 + i < NI ? 2 + i : 2 + i - NI; };
  constexpr auto n_slot = [b_slot](int ph) { int n = 1; for (int i = 0; i < 4; ++i) n += b_slot(i) == ph; return n; };
  constexpr int kLastSlot = b_slot(3);   // slot of K-tile t that carries the last piece of K-tile t+1
  constexpr auto pieces_after = [n_slot](int from, int to) { int n = 0; for (int q = from; q <= to; ++q) n += n_slot(q); return n; };
  constexpr int kConfUpper = pieces_after(kLastSlot + 1, NI - 1), kConfLower = pieces_after(kLastSlot + 1, NI - 2);
  static_assert(2 * kStageBytes + 8 * 2048 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStageBytes + 8 * 2048];
  unsigned char *const epi_scr = smem + 2 * kStageBytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int r32 = lane & 31, hk = lane >> 5;

  // ---- my tiles: XCD-contiguous logical range in 4-row-tile groups (as gemm_persistent_kernel)
  const int tiles_n = p.N / PBN, tiles_m = (p.M + PBM - 1) / PBM;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int tq = total >> 3, tr = total & 7;
  const int start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int count = tq + (xcd < tr ? 1 : 0);
  const int n_my = slot < count ? (count - slot + nslots - 1) / nslots : 0;
  if (n_my == 0) return;
#ifdef LLA_ABLATION
  // per-workgroup span in 100 MHz ticks (tools/gemm_pp_trace.py "spans"): are some workgroups stragglers?
  if (!TRACE && DBG == 0 && p.trace && tid == 0) {
    p.trace[512 + 2 * bid] = __builtin_amdgcn_s_memrealtime();
    p.trace[1536 + 2 * bid] = __builtin_amdgcn_s_memtime();
  }
#endif
  auto tile_origin = [&](int j, int &m0, int &n0) {
    int logical = start + slot + j * nslots;
    if (p.rev) logical = total - 1 - logical;
    const int per_group = kGroupM * tiles_n;
    const int grp = logical / per_group;
    const int in_grp = logical - grp * per_group;
    const int gh = (tiles_m - grp * kGroupM) < kGroupM ? (tiles_m - grp * kGroupM) : kGroupM;
    const int tn = in_grp / gh;
    m0 = (grp * kGroupM + (in_grp - tn * gh)) * PBM;
    n0 = tn * PBN;
  };

  // ---- loader: thread owns chunk pc of piece row srow; LDS row 64 q + srow of the A region holds
  // tile row 32 q + srow (upper wave row, filled by waves 0-3) or 32 NI + 32 q + srow - 32 (lower,
  // waves 4-7): piece q = fragment q of both wave rows.  Source chunk is XOR-swizzled (the DMA
  // destination is lane-linear).  Two cursors (A pieces / B pieces) walk the K-tiles of my tiles.
  const int srow = tid >> 3, pc = tid & 7;
  const int lc = DBG == 5 ? pc : (pc ^ ((srow >> 1) & 7));
  unsigned voffA[NI];        // byte offset of this thread's 16 bytes of piece q, from sA
  const unsigned voffB = (unsigned)(srow * p.K + lc * 8) * 2u;
  const unsigned char *sA = nullptr, *sB = nullptr;   // wave-uniform bases of the cursors' tiles
  int la_j = 0, la_kt = 0, la_u = 0, lb_j = 0, lb_kt = 0, lb_u = 0;
  const int nk = p.K / 64;
  auto set_tile_a = [&](int j) {
    int m0, n0;
    tile_origin(j < n_my ? j : n_my - 1, m0, n0);
    int lt = srow;
    asm volatile("" : "+v"(lt));   // recomputed per tile, not kept live across the K loop
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      int m = m0 + (lt < 32 ? 32 * q + lt : 32 * NI + 32 * q + lt - 32);
      if (m >= p.M) m = p.M - 1;
      if constexpr (AMODE == A_PLAIN) {
        voffA[q] = (unsigned)((m - m0) * p.lda + lc * 8) * 2u;
      } else {
        const int b0 = m0 / kPatches;
        voffA[q] = (unsigned)(patch_rowoff<AMODE>(m) - (size_t)b0 * kImgElems) * 2u;
      }
    }
    if constexpr (AMODE == A_PLAIN)
      sA = reinterpret_cast<const unsigned char *>(p.A) + (size_t)m0 * p.lda * 2;
    else if (p.a_chunk_images) {   // the batch in pieces: this (256-row) tile's images lie inside one of them
      const int b0 = m0 / kPatches, ci = b0 / p.a_chunk_images;
      sA = reinterpret_cast<const unsigned char *>(p.a_chunk[ci]) + (size_t)(b0 - ci * p.a_chunk_images) * kImgElems * 2;
    } else
      sA = reinterpret_cast<const unsigned char *>(p.A) + (size_t)(m0 / kPatches) * kImgElems * 2;
    sA = uniform_ptr(sA);
  };
  auto set_tile_b = [&](int j) {
    int m0, n0;
    tile_origin(j < n_my ? j : n_my - 1, m0, n0);
    sB = uniform_ptr(reinterpret_cast<const unsigned char *>(p.W) + (size_t)n0 * p.K * 2);
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned wave_off = (unsigned)wid * 1024u;
  auto issue_a = [&](int q) {
    const unsigned sb = lds_base + (unsigned)(la_u & 1) * kStageBytes + wave_off;
    unsigned va = voffA[q];
    const unsigned char *a_base = sA;
    if constexpr (AMODE == A_PLAIN) a_base += (size_t)la_kt * 128;
    else va += (unsigned)patch_koff<AMODE>(la_kt * 64 + lc * 8) * 2u;
    dma16s(va, a_base, __builtin_amdgcn_readfirstlane(sb + (unsigned)q * 8192u));
  };
  auto issue_b = [&](int i) {
    const unsigned sb = lds_base + (unsigned)(lb_u & 1) * kStageBytes + wave_off + kABytes;
    dma16s(voffB, sB + (size_t)lb_kt * 128 + (size_t)i * 64 * p.K * 2,
           __builtin_amdgcn_readfirstlane(sb + (unsigned)i * 8192u));
  };
  // The cursors run one K-tile ahead at their advance points (slot 0 for A, slot b_slot(3) for B), so
  // they change tile exactly in the second-to-last K-tile of an output tile: WRAP is a compile-time
  // property of the K-tile body.  (As a run-time test the tile change put a taken branch over ~100
  // instructions on the straight-line path: ~100 cycles of instruction fetch per K-tile and cursor.)
  auto advance_a = [&](bool wrap) { ++la_u; ++la_kt; if (wrap) { la_kt = 0; ++la_j; set_tile_a(la_j); } };
  auto advance_b = [&](bool wrap) { ++lb_u; ++lb_kt; if (wrap) { lb_kt = 0; ++lb_j; set_tile_b(lb_j); } };
  // DMA pieces of slot ph of the K-tile walk (cursor order: A(NI-1) of K-tile t+1 in slot 0, then
  // A(ph-1) of K-tile t+2; B pieces by b_slot, B3 last)
  auto dma_slot = [&](int ph, bool wrap) {
    if (DBG == 1) return;
    if (ph == 0) { issue_a(NI - 1); advance_a(wrap); }
    else issue_a(ph - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (b_slot(i) == ph) { issue_b(i); if (i == 3) advance_b(wrap); }
  };

  f32x16 acc[NI][2];
  const int swz = (r32 >> 1) & 7;
  unsigned a_off[4], b_off[4];   // byte offsets inside a stage of this lane's fragment rows, k-step s
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned c = (unsigned)(((2 * s + hk) ^ swz) * 16);
    a_off[s] = (unsigned)((32 * wr + r32) * 128) + c;
    b_off[s] = (unsigned)kABytes + (unsigned)((wc * 64 + r32) * 128) + c;
  }
  f16x8 fb[2][4], fa[4];
  auto read_b = [&](const unsigned char *sbase, int s) {   // both 32-column B fragments, k-step s
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j][s] = *reinterpret_cast<const f16x8 *>(sbase + b_off[s] + j * 4096);
  };
  auto read_a = [&](const unsigned char *sbase, int frag, int s) {
    fa[s] = *reinterpret_cast<const f16x8 *>(sbase + a_off[s] + frag * 8192);
  };

  // ---- prologue: K-tile 0 completely, then of K-tile 1 what the slots of "K-tile -1" would have issued
  set_tile_a(0);
  set_tile_b(0);
  if (DBG != 1 || true) {
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_a(q);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_b(i);
    advance_a(nk == 1);
    advance_b(nk == 1);
#pragma unroll
    for (int q = 0; q < NI - 1; ++q) issue_a(q);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (2 + i < NI) issue_b(i);
  }
  {
    constexpr int kPro = (NI - 1) + (NI - 2 < 4 ? NI - 2 : 4);   // K-tile 1 pieces issued so far may stay in flight
    __builtin_amdgcn_s_waitcnt(0x0070 | (kPro & 15) | ((kPro >> 4) << 14));
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int s = 0; s < 4; ++s) { read_b(smem, s); read_a(smem, 0, s); }
  if (wr == 0) dma_slot(0, nk == 2);   // the upper row runs its load segments half a phase ahead
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");

  unsigned long long t_prev = 0, t_sum[2 * NI + 2] = {}, t_cyc0 = 0, t_real0 = 0, t_fine[4] = {};   // TRACE
  if constexpr (TRACE) { t_prev = t_cyc0 = __builtin_amdgcn_s_memtime(); t_real0 = __builtin_amdgcn_s_memrealtime(); }
  int it = 0;   // global K-tile counter (selects the LDS stage)

  // load segment of slot ph of the K-tile whose stage offset is so (ROW: 0 upper, 1 lower wave row;
  // READ3: the k-step-3 operand refill belongs to this slot)
  auto load_seg = [&](auto row_c, int ph, unsigned so, bool read3, bool wrap) {
    constexpr int ROW = decltype(row_c)::value;
    if (DBG != 4 && read3) {
      if (ph > 0) read_a(smem + so, ph, 3);
      else { read_b(smem + so, 3); read_a(smem + so, 0, 3); }
    }
    dma_slot(ph, wrap);
    // fragment reads done; in the row's last load segment before K-tile t+1 is first read also:
    // K-tile t+1 has landed for this wave (only pieces issued after its last one may be in flight)
    if (ROW == 0 && ph == NI - 1) __builtin_amdgcn_s_waitcnt(0x0070 | (kConfUpper & 15) | ((kConfUpper >> 4) << 14));
    else if (ROW == 1 && ph == NI - 2) __builtin_amdgcn_s_waitcnt(0x0070 | (kConfLower & 15) | ((kConfLower >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
  };

  // One K-tile.  FIRST: first K-tile of an output tile (the k-step-0 MFMAs take C = 0 as an inline
  // operand; its fragments were read after the previous epilogue); LAST: last K-tile of an output tile
  // (the next K-tile's fragments are read after the epilogue, so no fragment register is live across it).
  // WRAP_CUR / WRAP_NEXT: the cursors change tile in slots of this K-tile (the lower row's slots, and the
  // upper row's slots ph >= 1) / in slot 0 of the NEXT K-tile, which the upper row runs at the end of this one.
  auto ktile = [&](auto first_c, auto last_c, auto wrap_c, auto wrapn_c) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    constexpr bool WRAP_CUR = decltype(wrap_c)::value, WRAP_NEXT = decltype(wrapn_c)::value;
    unsigned so_cur = (unsigned)(it & 1) * kStageBytes, so_next = (unsigned)((it + 1) & 1) * kStageBytes;
#pragma unroll
    for (int ph = 0; ph < NI; ++ph) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+s"(so_cur), "+s"(so_next));   // addresses are formed per read, not kept live
      if constexpr (TRACE) {   // (phase sums over middle K-tiles only: tile boundaries are accounted separately)
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (!FIRST && !LAST) t_sum[2 * ph + 1] += t - t_prev;
        t_prev = t;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (wr == 1) load_seg(std::integral_constant<int, 1>{}, ph, so_cur, !(FIRST && ph == 0), WRAP_CUR);
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- matrix segment: 8 MFMAs.  The operand registers of k-step s are refilled ONE
      // k-step later (behind the MFMAs of step s+1; step 3 in the following load segment): writing a
      // register an MFMA issued just before still reads stalls the wave until that MFMA has drained
      // (measured: 400-cycle segments instead of 256 with immediate refills).
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr (DBG == 2) {
          asm volatile("" ::"v"(fa[s]));
          asm volatile("" ::"v"(fb[0][s]));
          asm volatile("" ::"v"(fb[1][s]));
          if (s == 0 && FIRST) acc[ph][0] = acc[ph][1] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        } else if (s == 0 && FIRST) {
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][0], fa[0], zero16, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][s], fa[s], acc[ph][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 4 && s >= 1) {
          if (ph < NI - 1) read_a(smem + so_cur, ph + 1, s - 1);
          else if (!LAST) { read_b(smem + so_next, s - 1); read_a(smem + so_next, 0, s - 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (TRACE) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (!FIRST && !LAST) t_sum[2 * ph] += t - t_prev;
        t_prev = t;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (wr == 0) {
        if (ph < NI - 1) load_seg(std::integral_constant<int, 0>{}, ph + 1, so_cur, true, WRAP_CUR);
        else if (!LAST) {
          if constexpr (TRACE) {   // the same segment, stamped inside (middle K-tiles)
            unsigned long long f0, f1, f2, f3;
            f0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
            read_b(smem + so_next, 3); read_a(smem + so_next, 0, 3);
            __builtin_amdgcn_sched_barrier(0); f1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
            dma_slot(0, WRAP_NEXT);
            __builtin_amdgcn_sched_barrier(0); f2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            asm volatile("" ::: "memory");
            f3 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!FIRST) { t_fine[0] += f1 - f0; t_fine[1] += f2 - f1; t_fine[2] += f3 - f2; t_fine[3] += f0 - t_prev; }
          } else {
            load_seg(std::integral_constant<int, 0>{}, 0, so_next, true, WRAP_NEXT);
          }
        }
      }
      asm volatile("" ::: "memory");
    }
    ++it;
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  // K-tile kt of an output tile: the cursors (one K-tile ahead) change tile in the slots of K-tile nk - 2;
  // the upper row runs slot 0 of K-tile kt + 1 at the end of K-tile kt.  nk >= 4 (K >= 256).
  for (int cj = 0; cj < n_my; ++cj) {
    ktile(T_{}, F_{}, F_{}, F_{});
    {
      int kt = 1;
      for (; kt + 1 < nk - 3; kt += 2) { ktile(F_{}, F_{}, F_{}, F_{}); ktile(F_{}, F_{}, F_{}, F_{}); }
      if (kt < nk - 3) ktile(F_{}, F_{}, F_{}, F_{});
    }
    ktile(F_{}, F_{}, F_{}, T_{});      // kt = nk - 3: its trailing slot 0 belongs to K-tile nk - 2
    ktile(F_{}, F_{}, T_{}, F_{});      // kt = nk - 2
    ktile(F_{}, T_{}, F_{}, F_{});
    // ---- output tile finished: both rows run their epilogues together
    asm volatile("" ::: "memory");
    int m0c, n0c;
    tile_origin(cj, m0c, n0c);
    int el = lane;
    asm volatile("" : "+v"(el));
    const int mw = m0c + wr * 32 * NI, nw = n0c + wc * 64;
    if (mw + 32 * NI <= p.M) {
      if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
        if (SWAP_EPI) gemm_epilogue_swap<EPI, NI>(p, acc, mw, nw, el);
        else gemm_epilogue_staged<EPI, NI>(p, acc, mw, nw, el, epi_scr + wid * 2048);
      } else {
        gemm_epilogue_staged<EPI, NI>(p, acc, mw, nw, el, epi_scr + wid * 2048);
      }
    } else {
      gemm_epilogue<EPI, NI, 2, 0>(p, acc, mw, nw, el & 31, el >> 5);
    }
    {
      // first K-tile of the next output tile (confirmed before the last matrix segment): its B fragments
      // and first A fragment.  Unconditional (after the last tile it reads LDS bytes nobody uses) so that
      // the old fragment values are dead on every path across the epilogue.
      const unsigned so = (unsigned)(it & 1) * kStageBytes;
#pragma unroll
      for (int s = 0; s < 4; ++s) { read_b(smem + so, s); read_a(smem + so, 0, s); }
      if (wr == 0) load_seg(std::integral_constant<int, 0>{}, 0, so, false, false);
      else __builtin_amdgcn_s_waitcnt(0xC07F);
      asm volatile("" ::: "memory");
    }
    if constexpr (TRACE) {   // everything between the last matrix segment and here = epilogue
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_sum[2 * NI] += t - t_prev; t_prev = t; t_sum[2 * NI + 1] += 1;
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(fa[s]), "v"(fb[0][s]), "v"(fb[1][s]));
  __builtin_amdgcn_s_waitcnt(0x0070);   // trailing (unused) DMA pieces must land before the LDS is released
#ifdef LLA_ABLATION
  if (!TRACE && DBG == 0 && p.trace && tid == 0) {
    p.trace[512 + 2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    p.trace[1536 + 2 * bid + 1] = __builtin_amdgcn_s_memtime();
  }
#endif
  if constexpr (TRACE) {
    if (p.trace && (wid & 3) == 0 && lane == 0 && (blockIdx.x & 31) == 0) {
      unsigned long long *t = p.trace + ((size_t)(blockIdx.x >> 5) * 2 + wr) * 32;
#pragma unroll
      for (int i = 0; i < 2 * NI + 2; ++i) t[i] = t_sum[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[16 + i] = t_fine[i];
      t[13] = __builtin_amdgcn_s_memrealtime() - t_real0;   // 100 MHz ticks over the same span as t[12]
      t[12] = __builtin_amdgcn_s_memtime() - t_cyc0;
      t[14] = (unsigned long long)n_my * (nk - 2); t[15] = NI;   // middle K-tiles traced
    }
  }
  kernel_release();
}

#ifdef LLA_PROBES   // measured alternatives that lost: tools/-only build, not in the product library
// ---------------------------------------------------------------------------
// Two-workgroups-per-CU GEMM ("duo").  gemm_pp_kernel keeps the matrix pipe busy inside the K loop,
// but all eight waves of a CU reach the epilogue together and the pipe then idles for 15-40 % of a
// tile (and the whole chip stores at once: the burst is HBM-write-bound).  Here a workgroup is FOUR
// waves (one per SIMD, 1 x 4 over N) computing a (32 NI) x 256 tile, and two workgroups share a CU:
// nothing synchronises them, so one workgroup's epilogue, barrier waits and load segments run under
// the other's MFMAs.  The per-wave program is the pp kernel's (fragment-major K-tiles, one barrier
// per phase, operand registers refilled one k-step late, piece-granular LDS-DMA ring), minus the row
// rotation.  What differs:
//   * A: two stages of NI 4-KiB pieces (32 rows; every wave DMAs 8 rows of a piece and reads all 32).
//   * B: a wave's 64 weight rows are read by that wave only, so they are PRIVATE: one 8-KiB region per
//     wave, single-buffered, refilled by its owner right after its last read (phase 0 of a K-tile, for
//     the next K-tile) and confirmed by its own vmcnt before its first read (phase NI-1): no barrier
//     is involved in B at all.
//   * one cursor: phase g of the K-tile walk issues A piece g + 2 NI - 1 and, in phase 0, the eight B
//     instructions of the next K-tile; it changes output tile in phase 0 of K-tile nk - 2.
//   * RAW on A: piece g + 2 is first read in the matrix segment of phase g + 1, so every wave confirms its
//     quarter of it in the load segment of phase g (counted vmcnt: the instructions issued after it are
//     2 NI - 3 A pieces plus the B groups of the phase-0 slots in between), then the barrier.
//     WAR on A: piece g is last read in the load segment of phase g (k-step 3), refilled after barrier g+1.
// LDS per workgroup: 2 x NI x 4 + 32 + 8 (epilogue scratch) = 80 KiB at NI = 5: two per CU exactly.
// ---------------------------------------------------------------------------
// DBG (ablation build only; wrong results): 1 = no LDS-DMA in the loop, 2 = no B DMA in the loop, 3 = no
// epilogue, 4 = B issued in phase 1 instead of 0 (one phase less lead)
template <int EPI, int AMODE, int NI, bool SWAP_EPI = true, int DBG = 0>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(GemmParams p) {
  constexpr int PBM = 32 * NI, PBN = 256;
  constexpr int kABytes = PBM * 128;   // one A stage
  constexpr int kBOff = 2 * kABytes, kScrOff = kBOff + 4 * 8192;
  static_assert(2 * (kScrOff + 4 * 2048) <= 160 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kScrOff + 4 * 2048];
  constexpr auto is0 = [](int x) { return ((x % NI) + NI) % NI == 0; };
  // instructions younger than A piece g + 2 at the end of the load segment of phase g (g = p mod NI)
  constexpr auto conf_a = [is0](int ph) { int n = 2 * NI - 3; for (int d = 0; d <= 2 * NI - 3; ++d) n += 8 * is0(ph - d); return n; };
  constexpr auto conf = [conf_a](int ph) { const int a = conf_a(ph); return (ph == NI - 1 && NI - 1 < a) ? NI - 1 : a; };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wc = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hk = lane >> 5;

  // ---- my tiles: XCD-contiguous logical range in kGroupM-row-tile groups (as gemm_pp_kernel)
  const int tiles_n = p.N / PBN, tiles_m = (p.M + PBM - 1) / PBM;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int tq = total >> 3, tr = total & 7;
  const int start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int count = tq + (xcd < tr ? 1 : 0);
  const int n_my = slot < count ? (count - slot + nslots - 1) / nslots : 0;
  if (n_my == 0) return;
#ifdef LLA_PROBES
  // tools/gemm_pp_trace.py "duo": per workgroup HW_ID / XCC_ID and the 100 MHz stamps of its start, of every
  // epilogue's start and end, and of its end: do the two workgroups of a CU run their epilogues together?
  unsigned long long *const tr_wg = (p.trace && tid == 0 && DBG == 0) ? p.trace + 4096 + (size_t)bid * 40 : nullptr;
  if (tr_wg) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tr_wg[0] = ((unsigned long long)xcc << 32) | hw;
    tr_wg[1] = __builtin_amdgcn_s_memrealtime();
    tr_wg[2] = (unsigned long long)n_my;
  }
#endif
  constexpr int kGroupD = 2 * kGroupM;   // same rows per group as the 64 NI-row tiles of the pp kernel
  auto tile_origin = [&](int j, int &m0, int &n0) {
    const int logical = start + slot + j * nslots;
    const int per_group = kGroupD * tiles_n;
    const int grp = logical / per_group;
    const int in_grp = logical - grp * per_group;
    const int gh = (tiles_m - grp * kGroupD) < kGroupD ? (tiles_m - grp * kGroupD) : kGroupD;
    const int tn = in_grp / gh;
    m0 = (grp * kGroupD + (in_grp - tn * gh)) * PBM;
    n0 = tn * PBN;
  };

  // ---- loader.  A piece q = tile rows 32 q .. 32 q + 31: this thread's 16 bytes are chunk pc of piece
  // row srow (LDS row-major, 128 B per row, chunk XOR-swizzled on the SOURCE side: the DMA destination is
  // lane-linear).  B instruction i = rows 8 i .. 8 i + 7 of the wave's 64 weight rows.
  const int lrow = lane >> 3, pc = lane & 7;
  const int srow = wc * 8 + lrow;
  const int lc = pc ^ ((srow >> 1) & 7);
  unsigned voffA[NI];
  const unsigned voffB0 = (unsigned)(lrow * p.K + (pc ^ (lrow >> 1)) * 8) * 2u;       // even i
  const unsigned voffB1 = (unsigned)(lrow * p.K + (pc ^ (lrow >> 1) ^ 4) * 8) * 2u;   // odd i: rows 8 i + lrow swizzle with bit 2 set
  const unsigned char *sA = nullptr, *sB = nullptr;   // wave-uniform bases of the cursor's tile
  int cu_j = 0, cu_kt = 0, cu_u = 0;
  const int nk = p.K / 64;
  auto set_tile = [&](int j) {
    int m0, n0;
    tile_origin(j < n_my ? j : n_my - 1, m0, n0);
    int lt = srow;
    asm volatile("" : "+v"(lt));   // recomputed per tile, not kept live across the K loop
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      int m = m0 + 32 * q + lt;
      if (m >= p.M) m = p.M - 1;
      if constexpr (AMODE == A_PLAIN) {
        voffA[q] = (unsigned)((m - m0) * p.lda + lc * 8) * 2u;
      } else {
        const int b0 = m0 / kPatches;
        voffA[q] = (unsigned)(patch_rowoff<AMODE>(m) - (size_t)b0 * kImgElems) * 2u;
      }
    }
    if constexpr (AMODE == A_PLAIN)
      sA = reinterpret_cast<const unsigned char *>(p.A) + (size_t)m0 * p.lda * 2;
    else
      sA = reinterpret_cast<const unsigned char *>(p.A) + (size_t)(m0 / kPatches) * kImgElems * 2;
    sA = uniform_ptr(sA);
    sB = uniform_ptr(reinterpret_cast<const unsigned char *>(p.W) + (size_t)(n0 + wc * 64) * p.K * 2);
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned wave_a = (unsigned)wc * 1024u, wave_b = (unsigned)kBOff + (unsigned)wc * 8192u;
  auto issue_a = [&](int q) {
    unsigned va = voffA[q];
    const unsigned char *a_base = sA;
    if constexpr (AMODE == A_PLAIN) a_base += (size_t)cu_kt * 128;
    else va += (unsigned)patch_koff<AMODE>(cu_kt * 64 + lc * 8) * 2u;
    dma16s(va, a_base, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(cu_u & 1) * kABytes + wave_a + (unsigned)q * 4096u));
  };
  auto issue_b = [&] {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      dma16s((i & 1) ? voffB1 : voffB0, sB + (size_t)cu_kt * 128 + (size_t)i * 8 * p.K * 2,
             __builtin_amdgcn_readfirstlane(lds_base + wave_b + (unsigned)i * 1024u));
  };
  auto advance = [&](bool wrap) { ++cu_u; ++cu_kt; if (wrap) { cu_kt = 0; ++cu_j; set_tile(cu_j); } };
  auto dma_slot = [&](int ph, bool wrap) {
    if (DBG == 1) return;
    if (DBG == 4) {   // (cursor advance kept in phase 0 for A: B one phase late reads K-tile t+2's columns: wrong data, same traffic)
      if (ph == 0) { issue_a(NI - 1); advance(wrap); }
      else { issue_a(ph - 1); if (ph == 1) issue_b(); }
      return;
    }
    if (ph == 0) { issue_a(NI - 1); if (DBG != 2) issue_b(); advance(wrap); }
    else issue_a(ph - 1);
  };

  f32x16 acc[NI][2];
  const int swz = (r32 >> 1) & 7;
  unsigned a_off[4], b_off[4];   // byte offsets of this lane's fragment rows, k-step s (A: inside a stage)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned c = (unsigned)(((2 * s + hk) ^ swz) * 16);
    a_off[s] = (unsigned)(r32 * 128) + c;
    b_off[s] = (unsigned)kBOff + (unsigned)((wc * 64 + r32) * 128) + c;
  }
  f16x8 fb[2][4], fa[4];
  auto read_b = [&](int s) {   // both 32-column B fragments, k-step s
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j][s] = *reinterpret_cast<const f16x8 *>(smem + b_off[s] + j * 4096);
  };
  auto read_a = [&](const unsigned char *sbase, int frag, int s) {
    fa[s] = *reinterpret_cast<const f16x8 *>(sbase + a_off[s] + frag * 4096);
  };

  // ---- prologue: K-tile 0 completely, then of K-tile 1 what phases 1 .. NI-1 of "K-tile -1" would have issued
  set_tile(0);
#pragma unroll
  for (int q = 0; q < NI; ++q) issue_a(q);
  issue_b();
  advance(nk == 1);
#pragma unroll
  for (int q = 0; q < NI - 1; ++q) issue_a(q);
  __builtin_amdgcn_s_waitcnt(0x0070 | ((NI - 1) & 15));
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int s = 0; s < 4; ++s) { read_b(s); read_a(smem, 0, s); }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
  int it = 0;   // global K-tile counter (selects the A stage)

  // One K-tile.  FIRST: the k-step-0 MFMAs take C = 0 as an inline operand and the fragments were read
  // after the previous epilogue; LAST: the next K-tile's fragments are read after the epilogue, so no
  // fragment register is live across it; WRAP: the cursor changes output tile in phase 0.
  auto ktile = [&](auto first_c, auto last_c, auto wrap_c) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, WRAP = decltype(wrap_c)::value;
    unsigned so_cur = (unsigned)(it & 1) * kABytes, so_next = (unsigned)((it + 1) & 1) * kABytes;
#pragma unroll
    for (int ph = 0; ph < NI; ++ph) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+s"(so_cur), "+s"(so_next));   // addresses are formed per read, not kept live
      // ---------------- load segment
      if (!(FIRST && ph == 0)) {
        if (ph > 0) read_a(smem + so_cur, ph, 3);
        else { read_b(3); read_a(smem + so_cur, 0, 3); }
      }
      if (ph == 0) {   // the B region is rewritten by the DMA issued next: this wave's reads of it are done
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" ::: "memory");
      }
      dma_slot(ph, WRAP);
      {
        constexpr int c0 = conf(0), c1 = conf(1), c2 = conf(2), c3 = conf(3), c4 = conf(4 < NI ? 4 : 0);
#define LLA_WAIT_VM(C) __builtin_amdgcn_s_waitcnt(0x0070 | ((C) & 15) | (((C) >> 4) << 14))
        if (ph == 0) LLA_WAIT_VM(c0);
        else if (ph == 1) LLA_WAIT_VM(c1);
        else if (ph == 2) LLA_WAIT_VM(c2);
        else if (ph == 3) LLA_WAIT_VM(c3);
        else LLA_WAIT_VM(c4);
#undef LLA_WAIT_VM
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- matrix segment: 8 MFMAs; the operand registers of k-step s are refilled one
      // k-step later (see gemm_pp_kernel)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s == 0 && FIRST) {
          const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][0], fa[0], zero16, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ph][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][s], fa[s], acc[ph][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 1) {
          if (ph < NI - 1) read_a(smem + so_cur, ph + 1, s - 1);
          else if (!LAST) { read_b(s - 1); read_a(smem + so_next, 0, s - 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("" ::: "memory");
    }
    ++it;
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  for (int cj = 0; cj < n_my; ++cj) {   // nk >= 4 (K >= 256)
    ktile(T_{}, F_{}, F_{});
    for (int kt = 1; kt < nk - 2; ++kt) ktile(F_{}, F_{}, F_{});
    ktile(F_{}, F_{}, T_{});            // kt = nk - 2: the cursor moves on to the next output tile
    ktile(F_{}, T_{}, F_{});
    asm volatile("" ::: "memory");
    int m0c, n0c;
    tile_origin(cj, m0c, n0c);
    int el = lane;
    asm volatile("" : "+v"(el));
    const int nw = n0c + wc * 64;
#ifdef LLA_PROBES
    if (tr_wg && cj < 16) tr_wg[4 + 2 * cj] = __builtin_amdgcn_s_memrealtime();
#endif
    if (DBG == 3) {   // keep the accumulators alive without storing them
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][0][e] + acc[i][1][e];
      if (t == 1.2345e30f) reinterpret_cast<f16 *>(p.C)[el] = (f16)t;
    } else if (m0c + PBM <= p.M) {
      if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
        if (SWAP_EPI) gemm_epilogue_swap<EPI, NI>(p, acc, m0c, nw, el);
        else gemm_epilogue_staged<EPI, NI>(p, acc, m0c, nw, el, smem + kScrOff + wc * 2048);
      } else {
        gemm_epilogue_staged<EPI, NI>(p, acc, m0c, nw, el, smem + kScrOff + wc * 2048);
      }
    } else {
      gemm_epilogue<EPI, NI, 2, 0>(p, acc, m0c, nw, el & 31, el >> 5);
    }
#ifdef LLA_PROBES
    if (tr_wg && cj < 16) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (trace only: the stores have left the wave's queue)
      tr_wg[5 + 2 * cj] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    {
      // first K-tile of the next output tile (A piece 0 confirmed before the last barrier, B by this wave's
      // own wait in the last load segment).  Unconditional: after the last tile it reads bytes nobody uses.
      const unsigned so = (unsigned)(it & 1) * kABytes;
#pragma unroll
      for (int s = 0; s < 4; ++s) { read_b(s); read_a(smem + so, 0, s); }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      asm volatile("" ::: "memory");
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(fa[s]), "v"(fb[0][s]), "v"(fb[1][s]));
  __builtin_amdgcn_s_waitcnt(0x0070);   // trailing (unused) DMA pieces must land before the LDS is released
#ifdef LLA_PROBES
  if (tr_wg) tr_wg[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------
// Quad GEMM (round 3, experimental: LLA_GEMM_QUAD=1): 256 x 256 x 64 tiles on FOUR waves (2 x 2), one wave per
// SIMD, each a 128 x 128 output tile = 16 accumulator tiles of 32x32 (256 accumulator registers per lane: the
// register file of a wave that has its SIMD to itself, arch + acc VGPRs).  The shape hipBLASLt's kernel for these
// GEMMs has (MT256x256x64, 256 threads): 8 fragment reads per 16 MFMAs instead of 7 per 10, one wave's worth of
// address arithmetic / waits / barriers per SIMD instead of two.  Same persistent tile walk, LDS
// layout, LDS-DMA ring (two 64-KiB stages) and deferred epilogue as gemm_persistent_kernel.
// ---------------------------------------------------------------------------
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(256, 1) void gemm_quad_kernel(GemmParams p) {
  constexpr int NI = 4, NJ = 4, KB = 64, STAGES = 2;
  constexpr int PBM = 256, PBN = 256;
  constexpr int CH = KB / 8;
  constexpr int ROWS_I = 256 / CH;           // 32 rows per 256-thread DMA sweep
  constexpr int kAPieces = PBM / ROWS_I, kBPieces = PBN / ROWS_I, kPieces = kAPieces + kBPieces;   // 8 + 8
  constexpr int kABytes = PBM * KB * 2, kBBytes = PBN * KB * 2, kStageBytes = kABytes + kBBytes;   // 64 KiB
  constexpr int KSTEPS = KB / 16;
  __shared__ __attribute__((aligned(16))) f16 smem[STAGES * kStageBytes / 2];
  __shared__ __attribute__((aligned(16))) unsigned char epi_scr[4 * 2048];

  // The accumulators fill the AccVGPRs and the epilogue wants most of the arch VGPRs for a moment, so NOTHING per-lane
  // is kept across a K-tile: the lane index is re-derived (v_mbcnt, from an SGPR mask the compiler cannot see through)
  // wherever it is needed, and every other loop-carried value is wave-uniform (SGPRs).  A spilled address costs more
  // than its reload here: the reload waits on vmcnt, i.e. on the LDS-DMA pieces in flight.
  const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  auto lane_now = [] {
    unsigned m = ~0u;
    asm volatile("" : "+s"(m));
    return (int)__builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u));
  };

  const int tiles_n = p.N / PBN, tiles_m = (p.M + PBM - 1) / PBM;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int q = total >> 3, r = total & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int count = q + (xcd < r ? 1 : 0);
  const int n_my = slot < count ? (count - slot + nslots - 1) / nslots : 0;
  if (n_my == 0) return;
  auto tile_origin = [&](int j, int &m0, int &n0) {
    int logical = start + slot + j * nslots;
    if (p.rev) logical = total - 1 - logical;
    const int per_group = kGroupM * tiles_n;
    const int grp = logical / per_group;
    const int in_grp = logical - grp * per_group;
    const int gh = (tiles_m - grp * kGroupM) < kGroupM ? (tiles_m - grp * kGroupM) : kGroupM;
    const int tn = in_grp / gh;
    m0 = (grp * kGroupM + (in_grp - tn * gh)) * PBM;
    n0 = tn * PBN;
  };

  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  int ld_m0 = 0, ld_n0 = 0;                  // origin of the tile being streamed in (uniform)
  // one K-tile's 16 LDS-DMA pieces: thread -> row tid / 8 (+ 32 per piece), source chunk (tid % 8) ^ swizzle.
  // dma_prepare() derives the lane's part once per K-tile; dma_piece() is issued BETWEEN the MFMAs of the K-tile
  // (one wave per SIMD: whatever is not under an MFMA is on the critical path).
  int dp_srow = 0, dp_col = 0;
  unsigned dp_sb = 0;
  auto dma_prepare = [&](int kt, int stage) {
    const int t = wid * 64 + lane_now();
    dp_srow = t / CH;
    dp_col = (((t % CH) ^ ((dp_srow >> 1) & 7)) * 8) + kt * KB;
    dp_sb = lds_base + (unsigned)stage * kStageBytes + (unsigned)wid * 1024u;
  };
  auto dma_piece = [&](int piece) {
    if (piece < kAPieces) {
      int m = ld_m0 + dp_srow + ROWS_I * piece;
      if (m >= p.M) m = p.M - 1;
      dma16(p.A + (size_t)m * p.lda + dp_col, __builtin_amdgcn_readfirstlane(dp_sb + (unsigned)piece * 4096u));
    } else {
      dma16(p.W + (size_t)(ld_n0 + dp_srow + ROWS_I * (piece - kAPieces)) * p.K + dp_col,
            __builtin_amdgcn_readfirstlane(dp_sb + kABytes + (unsigned)(piece - kAPieces) * 4096u));
    }
  };
  auto dma_tile = [&](int kt, int stage) {
    dma_prepare(kt, stage);
#pragma unroll
    for (int piece = 0; piece < kPieces; ++piece) dma_piece(piece);
  };

  f32x16 acc[2][NI][2];   // [column half][row tile][column tile in the half]: the staged epilogue takes a half
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;

  const int nk = p.K / KB;
  const int total_iters = n_my * nk;
  int ld_j = 0, ld_kt = 0, ld_stage = 0, issued = 0;
  tile_origin(0, ld_m0, ld_n0);
  auto advance_load = [&] {
    ++issued;
    ld_stage ^= 1;
    if (++ld_kt == nk) { ld_kt = 0; ++ld_j; if (ld_j < n_my) tile_origin(ld_j, ld_m0, ld_n0); }
  };
  dma_tile(0, 0);
  advance_load();

  int cj = 0, ckt = 0, m0c, n0c, stage = 0;
  bool pend = false;
  int pm0 = 0, pn0 = 0;
  auto run_epilogue = [&] {
    const int el = lane_now();
    const int mw = pm0 + wr * 128, nw = pn0 + wc * 128;
    if (DBG == 0 && mw + 128 <= p.M) {
      gemm_epilogue_staged<EPI, NI>(p, acc[0], mw, nw, el, epi_scr + wid * 2048);
      gemm_epilogue_staged<EPI, NI>(p, acc[1], mw, nw + 64, lane_now(), epi_scr + wid * 2048);
    } else {
      gemm_epilogue<EPI, NI, 2>(p, acc[0], mw, nw, el & 31, el >> 5);
      const int e2 = lane_now();
      gemm_epilogue<EPI, NI, 2>(p, acc[1], mw, nw + 64, e2 & 31, e2 >> 5);
    }
  };
  tile_origin(0, m0c, n0c);
  constexpr bool kTrace = DBG == 9 || DBG == 10;   // s_memtime stamps per K-tile (tools/quad_trace.py); 10 = without operand traffic
  for (int it = 0; it < total_iters; ++it) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if constexpr (kTrace) ts0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): K-tile `it` has landed (the next one is not out yet)
    asm volatile("" ::: "memory");
    if (DBG != 2) __builtin_amdgcn_s_barrier();   // DBG 2 (timing ablation, racy): no workgroup barrier
    asm volatile("" ::: "memory");
    if constexpr (kTrace) ts1 = __builtin_amdgcn_s_memtime();
    const bool more = DBG != 1 && DBG != 10 && issued < total_iters;   // DBG 1 (timing ablation, wrong results): no operand traffic after the first K-tile
    if (more) dma_prepare(ld_kt, ld_stage);   // the other stage is free since the barrier; pieces go out under the MFMAs
    if (pend) {
      run_epilogue();
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_nondeterministic_value(acc[h][i][j]);
      pend = false;
    }
    const int ln = lane_now();
    const int r32 = ln & 31, hk = ln >> 5, swz = (r32 >> 1) & 7;
    const f16 *sbase = smem + stage * (kStageBytes / 2);
    const f16 *sa_row = sbase + (wr * 128 + r32) * KB;
    const f16 *sb_row = sbase + (kABytes / 2) + (wc * 128 + r32) * KB;
    f16x8 fa[2][NI], fb[2][NJ];
    auto fetch = [&](int s, int buf) {
      const int chunk = ((2 * s + hk) ^ swz) * 8;
#pragma unroll
      for (int i = 0; i < NI; ++i) fa[buf][i] = *reinterpret_cast<const f16x8 *>(sa_row + i * 32 * KB + chunk);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[buf][j] = *reinterpret_cast<const f16x8 *>(sb_row + j * 32 * KB + chunk);
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 1 < KSTEPS) fetch(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (s == 0 && ckt == 0) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc[j >> 1][i][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0][j], fa[0][i], zero16, 0, 0, 0);
            if (more && ((i * NJ + j) & 1)) { dma_piece((i * NJ + j) >> 1); __builtin_amdgcn_sched_barrier(0); }
          }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            acc[j >> 1][i][j & 1] =
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s & 1][j], fa[s & 1][i], acc[j >> 1][i][j & 1], 0, 0, 0);
            // the next K-tile's 16 pieces: one behind every second MFMA of k-steps 0 and 1
            if (more && s < 2 && ((i * NJ + j) & 1)) { dma_piece(s * 8 + ((i * NJ + j) >> 1)); __builtin_amdgcn_sched_barrier(0); }
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kTrace) { if (s == 0) ts2 = __builtin_amdgcn_s_memtime(); }
    }
    if constexpr (kTrace) {
      if (p.trace && wid == 0 && (blockIdx.x & 31) == 0 && it < 128 && lane_now() == 0) {
        unsigned long long *t = p.trace + ((size_t)(blockIdx.x >> 5) * 128 + it) * 4;
        t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = __builtin_amdgcn_s_memtime();
      }
    }
    if (more) advance_load();
    stage ^= 1;
    if (++ckt == nk) {
      pend = true; pm0 = m0c; pn0 = n0c;
      ckt = 0;
      if (++cj < n_my) tile_origin(cj, m0c, n0c);
    }
  }
  if (pend) run_epilogue();
}

#endif  // LLA_PROBES


#ifdef LLA_PROBES   // (see gemm_quad_kernel)
template <int EPI>
int launch_quad(const GemmParams &p, hipStream_t st) {
  const int cus = num_cus();
  const int total = ((p.M + 255) / 256) * (p.N / 256);
  const int grid = total < cus ? total : cus;
  static const int direct = [] { const char *e = lla_getenv("LLA_GEMM_EPILOGUE"); return (e && e[0] == 'd') ? 1 : 0; }();
#ifdef LLA_PROBES
  static const int dbg = [] { const char *e = lla_getenv("LLA_QUAD_DBG"); return e ? std::atoi(e) : 0; }();
  if (dbg == 1) { gemm_quad_kernel<EPI, 1><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 2) { gemm_quad_kernel<EPI, 2><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 9) { gemm_quad_kernel<EPI, 9><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 10) { gemm_quad_kernel<EPI, 10><<<grid, 256, 0, st>>>(p); return check_launch(); }
#endif
  if (direct) gemm_quad_kernel<EPI, 4><<<grid, 256, 0, st>>>(p);
  else gemm_quad_kernel<EPI, 0><<<grid, 256, 0, st>>>(p);
  return check_launch();
}

#endif  // LLA_PROBES

template <int EPI, int AMODE, int NJ, int KB, int STAGES, int NI>
int launch_persistent_cfg(const GemmParams &p, hipStream_t st, int grid) {
  // LLA_GEMM_EPILOGUE=direct: MFMA-layout stores instead of the LDS-staged line-assembling epilogue
  static const int dbg = [] {
    const char *epi = lla_getenv("LLA_GEMM_EPILOGUE");
#ifdef LLA_PROBES
    if (const char *e = lla_getenv("LLA_GEMM_DEBUG")) return std::atoi(e);
#endif
    return (epi && epi[0] == 'd') ? 4 : 0;
  }();
#ifdef LLA_PROBES
  // Ablation / trace variants (wrong-element addresses, skipped pipes, s_memtime stamps): only in
  // the -DLLA_PROBES build that tools/ load explicitly; the shipped library ignores LLA_GEMM_DEBUG.
  if (dbg == 1) { gemm_persistent_kernel<EPI, AMODE, NJ, KB, STAGES, 1, NI><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 2) { gemm_persistent_kernel<EPI, AMODE, NJ, KB, STAGES, 2, NI><<<grid, 512, 0, st>>>(p); return check_launch(); }
#endif
#ifdef LLA_ABLATION
  if (dbg == 4) gemm_persistent_kernel<EPI, AMODE, NJ, KB, STAGES, 4, NI><<<grid, 512, 0, st>>>(p);
  else
#endif
  gemm_persistent_kernel<EPI, AMODE, NJ, KB, STAGES, 0, NI><<<grid, 512, 0, st>>>(p);
  (void)dbg;
  return check_launch();
}

// Rounds a persistent grid needs for `tiles` work items (the slowest workgroup's tile count).
inline int rounds_for(int tiles, int cus) { return (tiles + cus - 1) / cus; }

template <int EPI, int AMODE>
int launch_pp(const GemmParams &p_in, hipStream_t st) {
  GemmParams p = p_in;
  const int cus = num_cus();
  const int tiles_n = p.N / 256;
  const int t256 = ((p.M + 255) / 256) * tiles_n, t320 = ((p.M + 319) / 320) * tiles_n;
  static const int allow320 = [] { const char *e = lla_getenv("LLA_GEMM_TALL"); return e ? std::atoi(e) : 1; }();
  const bool tall = allow320 && !p.a_chunk_images && rounds_for(t320, cus) * 320 <= rounds_for(t256, cus) * 256;
  const int total = tall ? t320 : t256;
  int grid = total < cus ? total : cus;
  // Balanced persistent grid: the launch lasts rounds_for(total, cus) tiles per workgroup whatever happens, so
  // start only as many workgroups as that round count needs (rounded up to a multiple of the 8 XCDs) and leave
  // the other CUs to the other tower lane's kernels: 51 200 rows -> 1440 / 1920 / 480 tiles = exactly 6 / 8 / 2
  // rounds on 240 workgroups, against 5.625 / 7.5 / 1.875 (same duration) on 256.
  static const bool balanced = [] { const char *e = lla_getenv("LLA_GEMM_BALANCED"); return !(e && e[0] == '0'); }();
  if (balanced && total > cus) {
    const int rounds = rounds_for(total, cus);
    const int need = ((total + rounds - 1) / rounds + 7) & ~7;
    if (need < grid) grid = need;
  }
#ifdef LLA_PROBES
  static const int dbg = [] { const char *e = lla_getenv("LLA_GEMM_DEBUG"); return e ? std::atoi(e) : 0; }();
  static const int cap = [] { const char *e = lla_getenv("LLA_GEMM_GRID"); return e ? std::atoi(e) : 0; }();
  if (cap > 0 && grid > cap) grid = cap;   // experiment: fewer CUs (is the epilogue bandwidth-bound?)
#define LLA_PP_DBG(CODE, D, T)                                                 \
  if (dbg == CODE) {                                                           \
    if (tall) gemm_pp_kernel<EPI, AMODE, 5, D, T><<<grid, 512, 0, st>>>(p);    \
    else gemm_pp_kernel<EPI, AMODE, 4, D, T><<<grid, 512, 0, st>>>(p);         \
    return check_launch();                                                     \
  }
  LLA_PP_DBG(1, 1, false) LLA_PP_DBG(2, 2, false) LLA_PP_DBG(4, 4, false) LLA_PP_DBG(5, 5, false)
  LLA_PP_DBG(9, 0, true) LLA_PP_DBG(11, 1, true) LLA_PP_DBG(12, 2, true) LLA_PP_DBG(14, 4, true)
#undef LLA_PP_DBG
#endif
#ifdef LLA_ABLATION
  static const bool staged = [] { const char *e = lla_getenv("LLA_GEMM_EPILOGUE"); return e && e[0] == 's'; }();
  if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
    if (staged) {   // A/B: LDS-staged fp16 epilogue (bit-identical)
      if (tall) gemm_pp_kernel<EPI, AMODE, 5, 0, false, false><<<grid, 512, 0, st>>>(p);
      else gemm_pp_kernel<EPI, AMODE, 4, 0, false, false><<<grid, 512, 0, st>>>(p);
      return check_launch();
    }
  }
#endif
  if (tall) gemm_pp_kernel<EPI, AMODE, 5><<<grid, 512, 0, st>>>(p);
  else gemm_pp_kernel<EPI, AMODE, 4><<<grid, 512, 0, st>>>(p);
  return check_launch();
}

#ifdef LLA_PROBES   // (see gemm_duo_kernel)
template <int EPI, int AMODE>
int launch_duo(const GemmParams &p, hipStream_t st) {
  const int slots = 2 * num_cus();
  const int tiles_n = p.N / 256;
  const int t128 = ((p.M + 127) / 128) * tiles_n, t160 = ((p.M + 159) / 160) * tiles_n;
  static const int force = [] { const char *e = lla_getenv("LLA_GEMM_DUO_NI"); return e ? std::atoi(e) : 0; }();
  bool tall = rounds_for(t160, slots) * 160 <= rounds_for(t128, slots) * 128;
  if (force == 4) tall = false;
  if (force == 5) tall = true;
  const int total = tall ? t160 : t128;
  int grid = total < slots ? total : slots;
#ifdef LLA_PROBES
  static const int dbg = [] { const char *e = lla_getenv("LLA_GEMM_DEBUG"); return e ? std::atoi(e) : 0; }();
  static const int cap = [] { const char *e = lla_getenv("LLA_GEMM_GRID"); return e ? std::atoi(e) : 0; }();
  if (cap > 0 && grid > cap) grid = cap;
#define LLA_DUO_DBG(D)                                                                  \
  if (dbg == D) {                                                                       \
    if (tall) gemm_duo_kernel<EPI, AMODE, 5, true, D><<<grid, 256, 0, st>>>(p);         \
    else gemm_duo_kernel<EPI, AMODE, 4, true, D><<<grid, 256, 0, st>>>(p);              \
    return check_launch();                                                              \
  }
  if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) { LLA_DUO_DBG(1) LLA_DUO_DBG(2) LLA_DUO_DBG(3) LLA_DUO_DBG(4) }
#undef LLA_DUO_DBG
#endif
  static const bool staged = [] { const char *e = lla_getenv("LLA_GEMM_EPILOGUE"); return e && e[0] == 's'; }();
  if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
    if (staged) {   // A/B: LDS-staged fp16 epilogue (bit-identical)
      if (tall) gemm_duo_kernel<EPI, AMODE, 5, false><<<grid, 256, 0, st>>>(p);
      else gemm_duo_kernel<EPI, AMODE, 4, false><<<grid, 256, 0, st>>>(p);
      return check_launch();
    }
  }
  if (tall) gemm_duo_kernel<EPI, AMODE, 5><<<grid, 256, 0, st>>>(p);
  else gemm_duo_kernel<EPI, AMODE, 4><<<grid, 256, 0, st>>>(p);
  return check_launch();
}

#endif  // LLA_PROBES

template <int EPI, int AMODE, int NJ>
int launch_persistent(const GemmParams &p, hipStream_t st) {
  const int cus = num_cus();
  const int tiles_n = p.N / (128 * NJ);
  // tile height: 256 rows, or 320 when that shortens the critical path (cost ~ rounds x rows)
  const int t256 = ((p.M + 255) / 256) * tiles_n, t320 = ((p.M + 319) / 320) * tiles_n;
  static const int allow320 = [] { const char *e = lla_getenv("LLA_GEMM_TALL"); return e ? std::atoi(e) : 1; }();
  // (on a tie the taller tile wins: 10 % fewer operand bytes per flop; FC1 292 -> 287 us)
  const bool tall = allow320 && NJ == 2 &&
                    rounds_for(t320, cus) * 320 <= rounds_for(t256, cus) * 256;
  const int total = tall ? t320 : t256;
  static const int persist = [] { const char *e = lla_getenv("LLA_GEMM_PERSIST"); return e ? std::atoi(e) : 1; }();
  const int grid = (!persist || total < cus) ? total : cus;
  // KB = 32 (twice the ring depth) measured WORSE end to end (61k vs 72k img/s): 64-byte row
  // segments waste half of every 128-byte line fetched when the operands are not L2-warm.
  static const int kb = [] { const char *e = lla_getenv("LLA_GEMM_KB"); return e ? std::atoi(e) : 64; }();
  if (kb == 64) {
    if constexpr (NJ == 2) {
      if (tall) return launch_persistent_cfg<EPI, AMODE, 2, 64, 2, 5>(p, st, grid);
      return launch_persistent_cfg<EPI, AMODE, 2, 64, 2, 4>(p, st, grid);
    } else {
      return launch_persistent_cfg<EPI, AMODE, 1, 64, 3, 4>(p, st, grid);
    }
  }
#ifdef LLA_ABLATION
  if constexpr (NJ == 2) return launch_persistent_cfg<EPI, AMODE, 2, 32, 4, 4>(p, st, grid);
  else return launch_persistent_cfg<EPI, AMODE, 1, 32, 5, 4>(p, st, grid);
#else
  return LLA_EINVAL;   // (unreachable: kb is 64 in the product library)
#endif
}

inline int gemm_tile() {
  static const int v = [] {
    const char *e = lla_getenv("LLA_GEMM_TILE");
    return e ? std::atoi(e) : 1;
  }();
  return v;
}

inline bool use_glds() {
  static const bool v = [] {
    const char *e = lla_getenv("LLA_GEMM_GLDS");
    return !(e && e[0] == '0');
  }();
  return v;
}

#ifdef LLA_ABLATION
// Shadow execution (tools/shadow_probe.py): every kernel of the tower is run a second time into spare buffers and
// the two outputs are compared on the device; mismatches go to a log: log[0] = count, then per mismatch
// {tag = layer * 8 + kind, 16-byte index, first differing words of a and b}.
__global__ void shadow_compare_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, size_t n16, int tag,
                                      unsigned long long *__restrict__ log) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
      const unsigned long long k = atomicAdd(log, 1ull);
      if (k < 255) {
        log[1 + 4 * k] = (unsigned long long)tag;
        log[2 + 4 * k] = i;
        log[3 + 4 * k] = ((unsigned long long)x.y << 32) | x.x;
        log[4 + 4 * k] = ((unsigned long long)y.y << 32) | y.x;
      }
    }
  }
}
#endif

template <int EPI, int AMODE>
int launch_gemm(const GemmParams &p_in, hipStream_t st, Profiler *prof = nullptr) {
  // tools/gemm_trace.py: LLA_GEMM_TRACE = device address of a u64 [8][128][4] buffer (with LLA_GEMM_DEBUG=9)
  GemmParams p = p_in;
#ifdef LLA_ABLATION
  static unsigned long long *const trace = [] {
    const char *e = lla_getenv("LLA_GEMM_TRACE");
    return e ? reinterpret_cast<unsigned long long *>(std::strtoull(e, nullptr, 0)) : nullptr;
  }();
  p.trace = trace;
#endif
  if (p.M <= 0) return LLA_OK;
  if (p.N % BN || p.K % BK || !p.A || !p.W || !p.C) return LLA_EINVAL;
  if (p.a_chunk_images) {   // (the batch in pieces: patch embedding of a chip-filling pass on the ping-pong kernel only)
    static const int pp_on = [] { const char *e = lla_getenv("LLA_GEMM_PP"); return e ? std::atoi(e) : 1; }();
    if (AMODE == A_PLAIN || AMODE == A_CONV3 || epi_base(EPI) != EPI_PATCH || (p.a_chunk_images & 255) || p.M < 9000 ||
        gemm_tile() != 1 || !pp_on || p.N % 256 || p.N < 768 || p.K < 256)
      return LLA_EINVAL;
  }
  if (p.n_store <= 0 || p.n_store > p.N) p.n_store = p.N;
  // the fp32 epilogues address C with 32-bit element offsets (registers are scarce there)
  if ((epi_base(EPI) == EPI_RESID || epi_base(EPI) == EPI_PATCH) &&
      ((size_t)p.M + (size_t)p.M / kPatches + 2) * (size_t)p.ldc >= (1ull << 32))
    return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_GEMM, 2.0 * p.M * p.N * p.K);
  if constexpr (epi_base(EPI) == EPI_RELU || epi_base(EPI) == EPI_ADDRELU) {
    // ResNet-tower GEMMs.  1x1 convolutions whose output is a multiple of 256 channels wide
    // (every bottleneck's expanding convolution, the reducing ones of layer3 / layer4) run on the persistent 256-wide
    // kernel with the line-assembling epilogue: whole 128-byte lines instead of 16-byte pieces per row took the
    // add+ReLU convolution of layer1 (3.7 GB of activations per 1024 images) from 3.2 to 5.3 TB/s and the tower from
    // 32.7k to 36.1k img/s (LLA_RN_PERSIST=0: the round-2 selection; 1: 128-wide persistent tiles with the MFMA-layout
    // epilogue -- no gain, so the per-tile prologue bubble was not the problem, the partial-line stores were).
    // Narrow outputs (64 / 128 channels) and the implicit 3x3 convolutions stay on the one-tile-per-workgroup kernel.
    if constexpr (AMODE == A_PLAIN) {
      static const int persist = [] { const char *e = lla_getenv("LLA_RN_PERSIST"); return e ? std::atoi(e) : 2; }();
      // (3: the ping-pong kernel where its K loop has something to overlap -- K >= 256 and at least three column tiles)
#ifdef LLA_ABLATION
      if (persist >= 3 && p.M >= 9000 && p.N % 256 == 0 && p.N >= 768 && p.K >= 256 && p.n_store == p.N) return launch_pp<EPI, AMODE>(p, st);
#endif
      if (persist >= 2 && p.M >= 9000 && p.N % 256 == 0 && p.n_store == p.N) return launch_persistent<EPI, AMODE, 2>(p, st);
#ifdef LLA_ABLATION
      if (persist == 1 && p.M >= 9000) return launch_persistent<EPI, AMODE, 1>(p, st);
#endif
    }
    if (p.M > 128 || AMODE == A_CONV3) {
      const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2);
      gemm256_f16_kernel<EPI, AMODE><<<tiles2, 512, 0, st>>>(p);
    } else if constexpr (AMODE == A_CONV3) {
      return LLA_EINVAL;
    } else {
      const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
      gemm_f16_kernel<EPI, AMODE, true><<<tiles, kGemmThreads, 0, st>>>(p);
    }
    return check_launch();
  } else if constexpr (epi_ln_in(EPI) || epi_ln_out(EPI)) {
    // LayerNorm-fused variants exist for the default kernel selection only (vit_forward_impl asks ln_fused())
#ifdef LLA_PROBES
    if constexpr (AMODE == A_PLAIN) {
      static const int q4 = [] { const char *e = lla_getenv("LLA_GEMM_Q4"); return e ? std::atoi(e) : 1; }();
      if (q4 && p.M >= 9000 && p.ldc == p.N) {
        const int rc = launch_q4(EPI, p, st);
        if (rc != LLA_EINVAL) return rc;
      }
    }
#endif
    if (p.M >= 9000 && p.N % 256 == 0 && p.N >= 768 && p.K >= 256) return launch_pp<EPI, AMODE>(p, st);
    if (p.M > 128) {
      const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2);
      gemm256_f16_kernel<EPI, AMODE><<<tiles2, 512, 0, st>>>(p);
    } else {
      const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
      gemm_f16_kernel<EPI, AMODE, true><<<tiles, kGemmThreads, 0, st>>>(p);
    }
    return check_launch();
  } else {
  // Small problems (< ~9k rows: batches under ~190 images) do not fill 256 persistent workgroups
  // with 256-wide tiles; measured at batch 128: 40.6k img/s persistent vs 48.4k with the
  // one-tile-per-workgroup 256x128 kernel (more, smaller tiles), so those go there.
  const bool big_enough = p.M >= 9000;
  if constexpr (AMODE == A_PLAIN && (EPI == EPI_F16 || EPI == EPI_QGELU)) {
    // the eight-wave kernel on the small MFMA shape (gemm_w8.hip, round 6) takes the large fp16-output layers (QKV, c_fc),
    // ragged M included; same bits as every other path (tests/test_gpu_variants.py).  LLA_GEMM_W8=0 (tools/ build): the
    // round-5 selection, 2: at every M
    static const int w8 = [] { const char *e = lla_getenv("LLA_GEMM_W8"); return e ? std::atoi(e) : LLA_W8_DEFAULT; }();
    if (w8 && (big_enough || w8 == 2) && !p.xhat && !p.ln_stats && p.n_store == p.N) {
      const int rc = launch_w8(EPI, p, st);
      if (rc != LLA_EINVAL) return rc;
    }
  }
  if (gemm_tile() == 1 && !big_enough && p.M > 128) {
    const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2);
    gemm256_f16_kernel<EPI, AMODE><<<tiles2, 512, 0, st>>>(p);
    return check_launch();
  }
  if constexpr (AMODE == A_PLAIN && (EPI == EPI_F16 || EPI == EPI_QGELU || EPI == EPI_RESID)) {
    // the four-wave 256 x 256 kernel (gemm_q4.hip) takes the large layers whose M is a whole number of its tiles;
    // LLA_GEMM_Q4=0 keeps everything on the ping-pong kernel (A/B, bit-identical: tests/test_gpu_variants.py)
    static const int q4 = [] { const char *e = lla_getenv("LLA_GEMM_Q4"); return e ? std::atoi(e) : 1; }();
    if (q4 && big_enough && p.ldc == p.N && !p.xhat && !p.ln_stats) {
      const int rc = launch_q4(EPI, p, st);
      if (rc != LLA_EINVAL) return rc;
    }
  }
#ifdef LLA_PROBES
  if constexpr (AMODE == A_PLAIN && (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU || epi_base(EPI) == EPI_RESID)) {
    static const int quad = [] { const char *e = lla_getenv("LLA_GEMM_QUAD"); return e ? std::atoi(e) : 0; }();
    if (quad && p.M >= 9000 && p.N % 256 == 0 && p.K >= 128) return launch_quad<EPI>(p, st);
  }
#endif
  if (gemm_tile() == 1 && p.M > 128) {  // persistent kernels: wide tiles where N allows
    static const int pp = [] { const char *e = lla_getenv("LLA_GEMM_PP"); return e ? std::atoi(e) : 1; }();
#ifdef LLA_PROBES
    static const int duo = [] { const char *e = lla_getenv("LLA_GEMM_DUO"); return e ? std::atoi(e) : 0; }();
    if (duo && p.N % 256 == 0 && p.N >= 768 && p.K >= 256) return launch_duo<EPI, AMODE>(p, st);
#endif
    if (pp && p.N % 256 == 0 && p.N >= 768 && p.K >= 256) return launch_pp<EPI, AMODE>(p, st);
    static const int wide_min_n = [] { const char *e = lla_getenv("LLA_GEMM_WIDE_MIN_N"); return e ? std::atoi(e) : 768; }();
    if (p.N % 256 == 0 && p.N >= wide_min_n) return launch_persistent<EPI, AMODE, 2>(p, st);
    return launch_persistent<EPI, AMODE, 1>(p, st);
  }
  if (gemm_tile() == 256 && p.M > 128) {
    const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2);
#ifdef LLA_PROBES
    static const int dbg = [] { const char *e = lla_getenv("LLA_GEMM_DEBUG"); return e ? std::atoi(e) : 0; }();
    if (dbg == 1) gemm256_f16_kernel<EPI, AMODE, 1><<<tiles2, 512, 0, st>>>(p);
    else if (dbg == 2) gemm256_f16_kernel<EPI, AMODE, 2><<<tiles2, 512, 0, st>>>(p);
    else
#endif
    gemm256_f16_kernel<EPI, AMODE><<<tiles2, 512, 0, st>>>(p);
    return check_launch();
  }
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
#ifdef LLA_ABLATION
  if (!use_glds()) {
    gemm_f16_kernel<EPI, AMODE, false><<<tiles, kGemmThreads, 0, st>>>(p);
    return check_launch();
  }
#endif
  gemm_f16_kernel<EPI, AMODE, true><<<tiles, kGemmThreads, 0, st>>>(p);
  return check_launch();
  }
}

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
// Why (round 3): with TWO tower lanes (two hardware queues; opt-in since round 3) the tower was not
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

#ifdef LLA_PROBES
// Row statistics of the fused LayerNorm: the residual GEMMs' epilogues leave per-row partial (sum, sum of squares)
// over 32-column slots; this turns them into (mean, 1 / sqrt(var + eps)) per row.  One thread per row, 192 bytes in,
// 8 out: 10 MB per launch at 51 200 rows.
__global__ __launch_bounds__(256) void ln_stats_kernel(const float *__restrict__ part, float *__restrict__ stats,
                                                       int rows) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  const float4 *src = reinterpret_cast<const float4 *>(part + (size_t)m * kLnSlots * 2);
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnSlots / 2; ++i) {
    const float4 v = src[i];
    s += v.x + v.z;
    q += v.y + v.w;
  }
  const float mean = s * (1.f / kWidth);
  const float var = fmaxf(q * (1.f / kWidth) - mean * mean, 0.f);
  reinterpret_cast<float2 *>(stats)[m] = make_float2(mean, 1.f / sqrtf(var + 1e-5f));
}
#endif

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

// ---------------------------------------------------------------------------
// weight blob layout
// ---------------------------------------------------------------------------
constexpr size_t kAlign = 256;
constexpr size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

size_t param_bytes(int id) {
  switch (id) {
    case LLA_VIT_CONV1_NHWC:
    case LLA_VIT_CONV1_NCHW: return (size_t)kWidth * kPatchK * 2;
    case LLA_VIT_CLASS_EMB: return kWidth * 4;
    case LLA_VIT_POS_EMB: return (size_t)kTokens * kWidth * 4;
    case LLA_VIT_LN_PRE_W: case LLA_VIT_LN_PRE_B:
    case LLA_VIT_LN_POST_W: case LLA_VIT_LN_POST_B: return kWidth * 4;
    case LLA_VIT_PROJ_T: return (size_t)kOut * kWidth * 2;
    case LLA_VIT_LN1_W: case LLA_VIT_LN1_B: case LLA_VIT_LN2_W: case LLA_VIT_LN2_B:
    case LLA_VIT_OUT_B: case LLA_VIT_CPROJ_B: return kWidth * 4;
    case LLA_VIT_QKV_W: return (size_t)3 * kWidth * kWidth * 2;
    case LLA_VIT_QKV_B: return 3 * kWidth * 4;
    case LLA_VIT_OUT_W: return (size_t)kWidth * kWidth * 2;
    case LLA_VIT_FC_W: return (size_t)kMlp * kWidth * 2;
    case LLA_VIT_FC_B: return kMlp * 4;
    case LLA_VIT_CPROJ_W: return (size_t)kWidth * kMlp * 2;
    case LLA_VIT_QKV_WG: return (size_t)3 * kWidth * kWidth * 2;
    case LLA_VIT_QKV_C: case LLA_VIT_QKV_D: return 3 * kWidth * 4;
    case LLA_VIT_FC_WG: return (size_t)kMlp * kWidth * 2;
    case LLA_VIT_FC_C: case LLA_VIT_FC_D: return kMlp * 4;
    default: return (size_t)-1;
  }
}

size_t globals_bytes() {
  size_t t = 0;
  for (int id = 0; id < LLA_VIT_GLOBAL_COUNT; ++id) t += align_up(param_bytes(id));
  return t;
}
size_t layer_bytes() {
  size_t t = 0;
  for (int id = LLA_VIT_LN1_W; id < LLA_VIT_LAYER_END; ++id) t += align_up(param_bytes(id));
  return t;
}
size_t param_offset(int id, int layer) {
  if (id >= 0 && id < LLA_VIT_GLOBAL_COUNT) {
    size_t t = 0;
    for (int k = 0; k < id; ++k) t += align_up(param_bytes(k));
    return t;
  }
  if (id >= LLA_VIT_LN1_W && id < LLA_VIT_LAYER_END && layer >= 0 && layer < kLayers) {
    size_t t = globals_bytes() + (size_t)layer * layer_bytes();
    for (int k = LLA_VIT_LN1_W; k < id; ++k) t += align_up(param_bytes(k));
    return t;
  }
  return (size_t)-1;
}

struct Workspace {
  float *x;  // [chunk*50][768] fp32 residual stream
  f16 *h;    // [chunk*50][768]  LayerNorm output / attention output
  f16 *big;  // [chunk*50][3072] qkv (2304 wide) or MLP hidden
  f16 *xh;   // [chunk*50][768]  fp16 copy of the residual stream (A operand of the LayerNorm-fused GEMMs)
  float *part;   // [chunk*50][kLnSlots][2] row partial sums
  float *stats;  // [chunk*50][2] (mean, rstd)
};
size_t workspace_bytes(int chunk) {
  const size_t rows = (size_t)chunk * kTokens;
  return align_up(rows * kWidth * 4) + align_up(rows * kWidth * 2) + align_up(rows * kMlp * 2) +
         align_up(rows * kWidth * 2) + align_up(rows * kLnSlots * 2 * 4) + align_up(rows * 2 * 4);
}
// LLA_VIT_LN_FUSE=1: LayerNorm folded into the GEMMs around it.  Built and measured in round 3 under
// the one-stream pipeline: 88.9k img/s against 92.3k with LayerNorm as its own kernel -- the 22 LayerNorm launches
// it removes (0.8 ms per step) cost 1.1 ms in heavier epilogues (second output stream + row sums in the residual
// GEMMs, per-row / per-column corrections in the consumers at 256 VGPRs) and 22 small statistics kernels.  Off by
// default; same embeddings within 5e-4 of the fp32 oracle either way (tests/test_gpu_vit.py).
bool ln_fused() {
#ifndef LLA_PROBES
  return false;   // (the fused instantiations exist in the ablation build only: slower)
#endif
  static const bool v = [] {
    const char *e = lla_getenv("LLA_VIT_LN_FUSE");
    if (!(e && e[0] == '1')) return false;
    for (const char *k : {"LLA_GEMM_PP", "LLA_GEMM_DUO", "LLA_GEMM_TILE"})   // fused variants: default kernels only
      if (lla_getenv(k)) return false;
    return true;
  }();
  return v;
}

// The tower's kernels walk the rows in alternating directions (GemmParams::rev): a kernel starts on the rows its
// producer wrote LAST, which are still in the 256-MB memory-side cache (and partly in L2), instead of on the first
// ones, which every producer of more than 256 MB has long pushed out.  LLA_VIT_ZIGZAG=0: every kernel top-down (A/B).
int zigzag() {
  static const int v = [] { const char *e = lla_getenv("LLA_VIT_ZIGZAG"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

bool prune_last_block() {
  static const bool v = [] {
    const char *e = lla_getenv("LLA_VIT_PRUNE_LAST");
    return !(e && e[0] == '0');
  }();
  return v;
}

// images per tower slice are capped so that chunk * 50 * 768 element offsets fit 32 bits (fp32 epilogues)
constexpr int kMaxChunk = 65536;

int default_chunk() {
  static int v = [] {
    const char *e = lla_getenv("LLA_VIT_CHUNK");
    const int c = e ? std::atoi(e) : 0;
    // 4352 images = 680 row tiles of 320: 99.6 % full rounds of the persistent GEMMs on 256 CUs (1024 images: 160 row
    // tiles, 6 / 8 / 2 rounds on 240 of the 256 CUs) and 4x fewer launches: tower alone 94.9k img/s at 1024, 96.3k at 1088,
    // 99.5k at 4352, 101.0k at 8704 (tools/slice_probe.py); end to end 8704 gains nothing over 4352 (97.5k vs 97.9k)
    // round 4: 8704 (1700 row tiles of 256 for the four-wave kernel: 19.9 / 59.8 / 79.7 rounds): +1.0 % end to end over 4352
    // with whole-pass timed regions (99.3k vs 98.4k img/s, same box)
    return c > 0 ? c : 8704;
  }();
  return v;
}

int lane_split_min() {   // batches below this many images stay on one lane (their GEMMs are too small to share the chip)
  static const int v = [] {
    const char *e = lla_getenv("LLA_VIT_SPLIT_MIN");
    const int n = e ? std::atoi(e) : 640;
    return n >= 2 ? n : 2;
  }();
  return v;
}

}  // namespace

int num_cus() {
  static const int v = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n = prop.multiProcessorCount;
    }
    return n;
  }();
  return v;
}

namespace {
}  // namespace

// Two tower lanes.  A batch is cut into slices (<= chunk images) and the slices alternate between two
// library-owned HIP streams, each with its own slice buffers: the tail of one lane's persistent GEMM (the
// last, partly filled round of tiles) and its HBM-bound LayerNorm / attention kernels run beside the other
// lane's GEMMs instead of leaving CUs idle.  Images are independent, so the embeddings are bit-identical to
// the one-lane pass (tests/test_gpu_vit.py).  Measured on batch 1024: 93.0k -> 99.5k img/s for the tower
// alone (tools/two_stream_probe.py).  OPT-IN since round 3 (LLA_VIT_STREAMS=2): with two hardware queues active
// the tower is not bit-reproducible on this stack -- between one embedding per 10^6 and one per 10^8 images (box and build dependent) comes out a few fp16 ulps
// (<= 3e-3) different from run to run, i.e. a 1 M-image file differs from its own re-run (found
// by the 1 M-image sharding test) -- while one stream gave 0 differing embeddings in 15 M images.  Bit-exact
// records are this path's contract, so the default is ONE stream (-4 % img/s); profiled passes always use one.
int tower_lanes() {
#ifndef LLA_ABLATION
  return 1;   // product build: one stream.  Two lanes are not bit-reproducible and live in the ablation build
#endif
  static const int v = [] {
    const char *e = lla_getenv("LLA_VIT_STREAMS");
    const int n = e ? std::atoi(e) : 1;
    return n >= 2 ? 2 : 1;
  }();
  return v;
}
// A tower handle (lla_tower_create) owns the two lane streams of ONE device and their events; nothing about
// the lanes lives in the library itself.
int lanes_create(Lanes **out) {
  Lanes *l = new Lanes();
  hipError_t e = hipGetDevice(&l->device);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipStreamCreateWithFlags(&l->st[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l->join[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&l->fork, hipEventDisableTiming);
  if (e != hipSuccess) { lanes_destroy(l); return hip_fail(e); }
  *out = l;
  return LLA_OK;
}

void lanes_destroy(Lanes *l) {
  if (!l) return;
  for (int i = 0; i < 2; ++i) {
    if (l->st[i]) { (void)hipStreamSynchronize(l->st[i]); (void)hipStreamDestroy(l->st[i]); }
    if (l->join[i]) (void)hipEventDestroy(l->join[i]);
  }
  if (l->fork) (void)hipEventDestroy(l->fork);
  delete l;
}

unsigned dynamic_lds_limit(const void *kernel) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, unsigned> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 64u * 1024u;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, kernel);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int v = 64 * 1024;
  (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  const unsigned cap = (unsigned)v > 160u * 1024u ? 160u * 1024u : (unsigned)v;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
  cache.emplace(key, cap);
  return cap;
}

int lanes_fork(Lanes *ln, hipStream_t caller) {
  hipError_t e = hipEventRecord(ln->fork, caller);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamWaitEvent(ln->st[i], ln->fork, 0);
  return e == hipSuccess ? LLA_OK : hip_fail(e);
}

int lanes_join(Lanes *ln, hipStream_t caller) {
  for (int i = 0; i < 2; ++i) {
    hipError_t e = hipEventRecord(ln->join[i], ln->st[i]);
    if (e == hipSuccess) e = hipStreamWaitEvent(caller, ln->join[i], 0);
    if (e != hipSuccess) return hip_fail(e);
  }
  ln->dirty = false;
  return LLA_OK;
}

}  // namespace lla

using namespace lla;

extern "C" {

size_t lla_vit_b32_weights_bytes(void) { return globals_bytes() + (size_t)kLayers * layer_bytes(); }
size_t lla_vit_b32_param_offset(int param, int layer) { return param_offset(param, layer); }
size_t lla_vit_b32_param_bytes(int param) { return param_bytes(param); }
size_t lla_vit_b32_workspace_bytes(int chunk) {
  if (chunk <= 0) chunk = default_chunk();
  return (size_t)tower_lanes() * workspace_bytes(chunk > kMaxChunk ? kMaxChunk : chunk);   // one slice buffer per lane
}

int lla_patch_embed_f16(const void *images, int layout, int B, const void *conv_w, const float *pos,
                        float *x, void *stream) {
  if (B < 0 || (layout != LLA_LAYOUT_NHWC && layout != LLA_LAYOUT_NCHW)) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!images || !conv_w || !pos || !x) return LLA_EINVAL;
  GemmParams pe{};
  pe.A = reinterpret_cast<const f16 *>(images);
  pe.W = reinterpret_cast<const f16 *>(conv_w);
  pe.C = x;
  pe.pos = pos;
  pe.M = B * kPatches; pe.N = kWidth; pe.K = kPatchK; pe.lda = 0; pe.ldc = kWidth;
  hipStream_t st = as_stream(stream);
  if (layout == LLA_LAYOUT_NHWC) return launch_gemm<EPI_PATCH, A_PATCH_NHWC>(pe, st);
  return launch_gemm<EPI_PATCH, A_PATCH_NCHW>(pe, st);
}

int lla_gemm_f16(const void *A, const void *W, const float *bias, void *C, int M, int N, int K,
                 int epilogue, void *stream) {
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(A);
  p.W = reinterpret_cast<const f16 *>(W);
  p.bias = bias;
  p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
#ifdef LLA_ABLATION
  if (const char *e = lla_getenv("LLA_GEMM_DEBUG_LDA0")) {  // ablation: alias all A / C rows
    if (e[0] == '1' || e[0] == '3') p.lda = 0;
    if (e[0] == '2' || e[0] == '3') p.ldc = 0;
  }
#endif
  hipStream_t st = as_stream(stream);
  switch (epilogue) {
    case LLA_EPI_F16: return launch_gemm<EPI_F16, A_PLAIN>(p, st);
    case LLA_EPI_QUICKGELU_F16: return launch_gemm<EPI_QGELU, A_PLAIN>(p, st);
    case LLA_EPI_RESID_F32: return launch_gemm<EPI_RESID, A_PLAIN>(p, st);
    default: return LLA_EINVAL;
  }
}

int lla_gemm_f16_ex(const void *A, int lda, const void *W, const float *bias, void *C, int ldc,
                    const void *resid, int ldr, int M, int N, int K, int epilogue, void *stream) {
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(A);
  p.W = reinterpret_cast<const f16 *>(W);
  p.bias = bias;
  p.C = C;
  p.resid = resid;
  p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc;
  if (lda < K || (lda & 7) || (ldc & 3)) return LLA_EINVAL;
  if (ldc < N) {   // narrow output: only the first ldc columns (a multiple of 32) are stored
    if ((ldc & 31) || ldc <= 0 || (epilogue != LLA_EPI_RELU_F16 && epilogue != LLA_EPI_ADD_RELU_F16))
      return LLA_EINVAL;   // (the ReLU kinds run on the kernels whose epilogue knows about n_store)
    p.n_store = ldc;
  }
  hipStream_t st = as_stream(stream);
  switch (epilogue) {
    case LLA_EPI_F16: return launch_gemm<EPI_F16, A_PLAIN>(p, st);
    case LLA_EPI_QUICKGELU_F16: return launch_gemm<EPI_QGELU, A_PLAIN>(p, st);
    case LLA_EPI_RESID_F32: return launch_gemm<EPI_RESID, A_PLAIN>(p, st);
    case LLA_EPI_RELU_F16: return launch_gemm<EPI_RELU, A_PLAIN>(p, st);
    case LLA_EPI_ADD_RELU_F16:
      if (!resid || ldr < (ldc < N ? ldc : N) || (ldr & 3)) return LLA_EINVAL;
      return launch_gemm<EPI_ADDRELU, A_PLAIN>(p, st);
    default: return LLA_EINVAL;
  }
}

int lla_conv3x3_relu_f16(const void *in, int n, int H, int W, int pitch, int cin, const void *weights,
                         const float *bias, void *out, int ldc, int cout, void *stream) {
  if (n < 0 || H <= 0 || W <= 0 || cin <= 0 || ((cin % BK) && cin != 32) || pitch < cin || (pitch & 7) ||
      cout <= 0 || (cout % BN2) || ldc <= 0 || (ldc & 3) || (ldc < cout && (ldc & 31)))
    return LLA_EINVAL;
  if (n == 0) return LLA_OK;
  if (!in || !weights || !out) return LLA_EINVAL;
  if ((size_t)n * H * W >= (1ull << 31)) return LLA_EINVAL;
  GemmParams p{};
  p.A = reinterpret_cast<const f16 *>(in);
  p.W = reinterpret_cast<const f16 *>(weights);
  p.bias = bias;
  p.C = out;
  p.M = n * H * W; p.N = cout; p.K = (9 * cin + BK - 1) / BK * BK; p.lda = pitch; p.ldc = ldc;
  p.conv_h = H; p.conv_w = W; p.conv_cin = cin;
  if (ldc < cout) p.n_store = ldc;
  return launch_gemm<EPI_RELU, A_CONV3>(p, as_stream(stream));
}

static int layernorm_impl(const float *x, size_t row_stride, const float *w, const float *b,
                          void *y16, int rows, hipStream_t st, Profiler *prof, int rev = 0) {
  if (rows < 0 || row_stride < (size_t)kWidth || (row_stride & 3u)) return LLA_EINVAL;
  if (rows == 0) return LLA_OK;
  if (!x || !w || !b || !y16) return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_LAYERNORM, (double)rows * kWidth * 6.0);
  layernorm768_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, row_stride, w, b,
                                                      reinterpret_cast<f16 *>(y16), rows, rev);
  return check_launch();
}

static int attention_impl(const void *qkv, void *o, int B, hipStream_t st, Profiler *prof, int rev = 0) {
  if (B < 0) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!qkv || !o) return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_ATTENTION, (double)B * 12 * 4.0 * kTokens * kTokens * kHeadDim);
  attention50_kernel<<<B * 3, 256, 0, st>>>(reinterpret_cast<const f16 *>(qkv),
                                            reinterpret_cast<f16 *>(o), B, rev);
  return check_launch();
}

int lla_layernorm768(const float *x, size_t row_stride, const float *w, const float *b, void *y16,
                     int rows, void *stream) {
  return layernorm_impl(x, row_stride, w, b, y16, rows, as_stream(stream), nullptr);
}

int lla_attention50(const void *qkv, void *o, int B, void *stream) {
  return attention_impl(qkv, o, B, as_stream(stream), nullptr);
}

int lla_profiler_create(void **profiler, int max_launches) {
  if (!profiler || max_launches <= 0) return LLA_EINVAL;
  Profiler *p = new Profiler();
  p->pool.resize((size_t)max_launches);
  for (auto &r : p->pool) {
    hipError_t e = hipEventCreate(&r.a);
    if (e == hipSuccess) e = hipEventCreate(&r.b);
    if (e != hipSuccess) { delete p; return hip_fail(e); }
  }
  *profiler = p;
  return LLA_OK;
}

int lla_profiler_destroy(void *profiler) {
  Profiler *p = reinterpret_cast<Profiler *>(profiler);
  if (!p) return LLA_EINVAL;
  for (auto &r : p->pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  delete p;
  return LLA_OK;
}

int lla_profiler_collect(void *profiler, double *ms, double *work, long long *launches) {
  Profiler *p = reinterpret_cast<Profiler *>(profiler);
  if (!p || !ms || !work || !launches) return LLA_EINVAL;
  for (size_t i = 0; i < p->used; ++i) {
    auto &r = p->pool[i];
    hipError_t e = hipEventSynchronize(r.b);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e);
    ms[r.cls] += t; work[r.cls] += r.work; launches[r.cls] += 1;
  }
  p->used = 0;
  return LLA_OK;
}

static int vit_forward_impl(const void *images, int layout, int B, const void *weights, void *workspace,
                            size_t ws_bytes, int chunk, void *z_out, void *stream, void *profiler,
                            Lanes *tower, bool deferred, const void *const *pieces = nullptr, int n_pieces = 0,
                            int piece_images = 0);

int lla_tower_create(void **tower) {
  if (!tower) return LLA_EINVAL;
  Lanes *l = nullptr;
  const int rc = lanes_create(&l);
  if (rc == LLA_OK) *tower = l;
  return rc;
}

int lla_tower_destroy(void *tower) {
  if (!tower) return LLA_EINVAL;
  lanes_destroy(reinterpret_cast<Lanes *>(tower));
  return LLA_OK;
}

// Epoch of an EPI_RESID_LNX launch: what its workgroups tag their exchange words with.  Unique per launch within the
// process and never 0 (a counter, not state any result depends on: the words are compared for equality only).
static unsigned next_lnx_epoch() {
  static std::atomic<unsigned> counter{0x5EED0000u};
  unsigned e = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  return e ? e : counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
}

int lla_tower_set_option(void *tower, int option, int value) {
  Lanes *l = reinterpret_cast<Lanes *>(tower);
  if (!l) return LLA_EINVAL;
  switch (option) {
    case LLA_TOWER_OPT_LNX: l->lnx = value != 0; return LLA_OK;
    case LLA_TOWER_OPT_LNX_WAIT: l->lnx_wait = value; return LLA_OK;
    default: return LLA_EINVAL;
  }
}

int lla_tower_join(void *tower, void *stream) {
  if (!tower) return LLA_EINVAL;
  return lanes_join(reinterpret_cast<Lanes *>(tower), as_stream(stream));
}

int lla_vit_b32_forward(const void *images, int layout, int B, const void *weights,
                        void *workspace, size_t ws_bytes, int chunk, void *z_out, void *stream) {
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, nullptr, nullptr,
                          false);
}

int lla_vit_b32_forward_profiled(const void *images, int layout, int B, const void *weights,
                                 void *workspace, size_t ws_bytes, int chunk, void *z_out,
                                 void *stream, void *profiler) {
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, profiler, nullptr,
                          false);
}

int lla_vit_b32_forward_lanes(void *tower, const void *images, int layout, int B, const void *weights,
                              void *workspace, size_t ws_bytes, int chunk, void *z_out, void *stream,
                              int deferred) {
  if (!tower) return LLA_EINVAL;
  return vit_forward_impl(images, layout, B, weights, workspace, ws_bytes, chunk, z_out, stream, nullptr,
                          reinterpret_cast<Lanes *>(tower), deferred != 0);
}

int lla_vit_b32_forward_gather(void *tower, const void *const *pieces, int n_pieces, int piece_images, int layout, int B,
                               const void *weights, void *workspace, size_t ws_bytes, void *z_out, void *stream) {
  if (!tower || !pieces) return LLA_EINVAL;
  return vit_forward_impl(nullptr, layout, B, weights, workspace, ws_bytes, 0, z_out, stream, nullptr,
                          reinterpret_cast<Lanes *>(tower), false, pieces, n_pieces, piece_images);
}

static int vit_forward_impl(const void *images, int layout, int B, const void *weights, void *workspace,
                            size_t ws_bytes, int chunk, void *z_out, void *stream, void *profiler,
                            Lanes *tower, bool deferred, const void *const *pieces, int n_pieces, int piece_images) {
  Profiler *prof = reinterpret_cast<Profiler *>(profiler);
  if (pieces) {
    // the batch in pieces of piece_images images (the last one may be shorter): one slice, whole 256-row tiles
    if (n_pieces < 1 || n_pieces > 64 || piece_images <= 0 || (piece_images & 255) || B <= (n_pieces - 1) * piece_images ||
        B > n_pieces * piece_images || (B & 127) || B < 256)
      return LLA_EINVAL;
    for (int i = 0; i < n_pieces; ++i)
      if (!pieces[i]) return LLA_EINVAL;
    images = pieces[0];
    if (B > (chunk > 0 ? chunk : default_chunk())) return LLA_EINVAL;
  }
  if (!images || !weights || !workspace || !z_out || B < 0) return LLA_EINVAL;
  if (layout != LLA_LAYOUT_NHWC && layout != LLA_LAYOUT_NCHW) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (chunk <= 0) chunk = default_chunk();
  if (chunk > kMaxChunk) chunk = kMaxChunk;
  if (chunk > B) chunk = B;
  hipStream_t st_caller = as_stream(stream);
  // two lanes when the batch is large enough, the caller's buffer holds two slices and nobody is timing launches
  // Lane i's slice buffers live in the i-th half of the caller's workspace (fixed offsets: a deferred pass
  // may still be running on the other lane when the next call arrives with a different slice size).
  const size_t lane_bytes = (ws_bytes / 2) & ~(size_t)255;
  int lanes = 1;
  if (tower) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != tower->device) return LLA_EINVAL;   // a tower belongs to one device
  }
  if (!prof && tower && tower_lanes() == 2) {
    if (deferred) {
      // whole slices alternate between the lanes ACROSS calls; nothing is joined until lla_tower_join
      if (workspace_bytes(chunk) <= lane_bytes) lanes = 2;
    } else if (B >= lane_split_min()) {
      const int half = (B + 1) / 2;
      const int sub = chunk < half ? chunk : half;
      if (workspace_bytes(sub) <= lane_bytes) { lanes = 2; chunk = sub; }
    }
  }
  if (ws_bytes < workspace_bytes(chunk)) return LLA_ECAP;
  Lanes *ln = nullptr;
  int slice = 0;
  if (lanes == 1 && tower && tower->dirty) {
    // a pass on the caller's stream uses lane 0's slice buffers: deferred passes still in flight must finish first
    const int jrc = lanes_join(tower, st_caller);
    if (jrc != LLA_OK) return jrc;
  }
  if (lanes == 2) {
    ln = tower;
    const int frc = lanes_fork(ln, st_caller);
    if (frc != LLA_OK) return frc;
    if (deferred) slice = ln->next;
  }

  const uint8_t *wb = reinterpret_cast<const uint8_t *>(weights);
  auto P16 = [&](int id, int l) { return reinterpret_cast<const f16 *>(wb + param_offset(id, l)); };
  auto P32 = [&](int id, int l) { return reinterpret_cast<const float *>(wb + param_offset(id, l)); };

  Workspace wss[2];
  const size_t rows_cap = (size_t)chunk * kTokens;
  for (int i = 0; i < lanes; ++i) {
    uint8_t *w8 = reinterpret_cast<uint8_t *>(workspace) + (size_t)i * lane_bytes;
    wss[i].x = reinterpret_cast<float *>(w8);
    w8 += align_up(rows_cap * kWidth * 4);
    wss[i].h = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kWidth * 2);
    wss[i].big = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kMlp * 2);
    wss[i].xh = reinterpret_cast<f16 *>(w8);
    w8 += align_up(rows_cap * kWidth * 2);
    wss[i].part = reinterpret_cast<float *>(w8);
    w8 += align_up(rows_cap * kLnSlots * 2 * 4);
    wss[i].stats = reinterpret_cast<float *>(w8);
  }

  int rc = LLA_OK;
#define LLA_TRY(expr) do { rc = (expr); if (rc != LLA_OK) return rc; } while (0)
#ifdef LLA_PROBES
#define LLA_TRY_FUSED(expr) LLA_TRY(expr)
#else
#define LLA_TRY_FUSED(expr) return LLA_EINVAL   /* unreachable: ln_fused() is false in the product build */
#endif
#ifdef LLA_ABLATION
  // tools/snapshot_probe.py: LLA_VIT_SNAPSHOT = device address of fp32 [2 lanes][26][LLA_VIT_SNAPSHOT_ROWS][768];
  // the residual stream is copied there after ln_pre (slot 0) and after every residual GEMM (1 + 2 l, 2 + 2 l),
  // the fp16 qkv / attention output of layer LLA_VIT_SNAPSHOT_LAYER into slot 25 (as raw bytes)
  float *snap = nullptr;
  size_t snap_rows = 0;
  int snap_layer = -1;
  if (const char *e = lla_getenv("LLA_VIT_SNAPSHOT")) snap = reinterpret_cast<float *>(std::strtoull(e, nullptr, 0));
  if (const char *e = lla_getenv("LLA_VIT_SNAPSHOT_ROWS")) snap_rows = std::strtoull(e, nullptr, 0);
  if (const char *e = lla_getenv("LLA_VIT_SNAPSHOT_LAYER")) snap_layer = std::atoi(e);
  (void)snap_layer;
  // LLA_VIT_SNAPSHOT16 = device address of bytes [2 lanes][12 layers]{h after ln_2 [rows][768] f16, big after c_fc
  // [rows][3072] f16}
  unsigned char *snap16 = nullptr;
  if (const char *e = lla_getenv("LLA_VIT_SNAPSHOT16")) snap16 = reinterpret_cast<unsigned char *>(std::strtoull(e, nullptr, 0));
#define LLA_SNAP16(layer, kind, src, bytes)                                                                     \
  do {                                                                                                          \
    if (snap16 && (size_t)M <= snap_rows) {                                                                     \
      const size_t per_layer = snap_rows * (kWidth + kMlp) * 2;                                                 \
      (void)hipMemcpyAsync(snap16 + ((size_t)(lanes == 2 ? (slice & 1) : 0) * kLayers + (layer)) * per_layer +  \
                               ((kind) ? snap_rows * kWidth * 2 : 0),                                           \
                           (src), (bytes), hipMemcpyDeviceToDevice, st);                                        \
    }                                                                                                           \
  } while (0)
#define LLA_SNAP(slot, src, bytes)                                                                          \
  do {                                                                                                      \
    if (snap && (size_t)M <= snap_rows)                                                                     \
      (void)hipMemcpyAsync(snap + ((size_t)(lanes == 2 ? (slice & 1) : 0) * 26 + (slot)) * snap_rows * kWidth, \
                           (src), (bytes), hipMemcpyDeviceToDevice, st);                                    \
  } while (0)
  // LLA_VIT_SHADOW = device address of spare buffers [2 lanes]{x fp32 [rows][768], h f16 [rows][768], big f16
  // [rows][3072]} (rows = LLA_VIT_SNAPSHOT_ROWS), LLA_VIT_SHADOW_LOG = device address of u64 [1 + 4 * 255]
  unsigned char *shadow = nullptr;
  unsigned long long *shadow_log = nullptr;
  if (const char *e = lla_getenv("LLA_VIT_SHADOW")) shadow = reinterpret_cast<unsigned char *>(std::strtoull(e, nullptr, 0));
  if (const char *e = lla_getenv("LLA_VIT_SHADOW_LOG")) shadow_log = reinterpret_cast<unsigned long long *>(std::strtoull(e, nullptr, 0));
#else
#define LLA_SNAP(slot, src, bytes) do {} while (0)
#define LLA_SNAP16(layer, kind, src, bytes) do {} while (0)
  unsigned char *shadow = nullptr;
  unsigned long long *shadow_log = nullptr;
  const size_t snap_rows = 0;
#endif

  // Slices of `chunk` images; a ragged last slice of >= 256 images is cut once more so that its main part is a multiple of
  // 128 images = a whole number of 256-row tiles (50 x 128 = 25 x 256): that part runs on the four-wave GEMM, the
  // < 128 images left over on the small-M kernels.  Images are independent: same embeddings for every cut.
  for (int c0 = 0, bc = 0; c0 < B; c0 += bc, ++slice) {
    bc = (B - c0) < chunk ? (B - c0) : chunk;
    if (bc >= 256 && (bc & 127)) bc -= bc & 127;
    const int M = bc * kTokens;
    const Workspace &ws = wss[lanes == 2 ? (slice & 1) : 0];
    hipStream_t st = lanes == 2 ? ln->st[slice & 1] : st_caller;
    // (debug) shadow buffers of this lane and the compare launcher
    const bool sh = shadow && shadow_log && (size_t)M <= snap_rows;
    unsigned char *sh0 = shadow + (size_t)(lanes == 2 ? (slice & 1) : 0) * snap_rows * (kWidth * 4 + kWidth * 2 + kMlp * 2);
    float *sx = reinterpret_cast<float *>(sh0);
    f16 *shh = reinterpret_cast<f16 *>(sh0 + snap_rows * kWidth * 4);
    f16 *sbig = reinterpret_cast<f16 *>(sh0 + snap_rows * kWidth * 4 + snap_rows * kWidth * 2);
    auto shadow_cmp = [&](int tag, const void *a, const void *b, size_t bytes) {
#ifdef LLA_ABLATION
      shadow_compare_kernel<<<2048, 256, 0, st>>>(reinterpret_cast<const uint4 *>(a), reinterpret_cast<const uint4 *>(b),
                                                   bytes / 16, tag, shadow_log);
#endif
    };

    // patch embedding: conv1 as a GEMM that reads patches in place, + pos, into token rows
    GemmParams pe{};
    pe.A = reinterpret_cast<const f16 *>(images) + (size_t)c0 * kImgElems;
    pe.W = P16(layout == LLA_LAYOUT_NHWC ? LLA_VIT_CONV1_NHWC : LLA_VIT_CONV1_NCHW, 0);
    pe.bias = nullptr;
    pe.C = ws.x;
    pe.pos = P32(LLA_VIT_POS_EMB, 0);
    pe.M = bc * kPatches; pe.N = kWidth; pe.K = kPatchK; pe.lda = 0; pe.ldc = kWidth;
    if (pieces) {
      if (lanes != 1 || c0 != 0 || bc != B) return LLA_EINVAL;   // (one slice on the caller's stream)
      pe.a_chunk_images = piece_images;
      for (int i = 0; i < n_pieces; ++i) pe.a_chunk[i] = reinterpret_cast<const f16 *>(pieces[i]);
    }
    if (layout == LLA_LAYOUT_NHWC) LLA_TRY((launch_gemm<EPI_PATCH, A_PATCH_NHWC>(pe, st, prof)));
    else LLA_TRY((launch_gemm<EPI_PATCH, A_PATCH_NCHW>(pe, st, prof)));

    {
    ProfScope scope(prof, st, LLA_PROF_LAYERNORM, (double)M * kWidth * 10.0);
    ln_pre_ln1_kernel<<<(M + 3) / 4, 256, 0, st>>>(
        ws.x, P32(LLA_VIT_CLASS_EMB, 0), P32(LLA_VIT_POS_EMB, 0), P32(LLA_VIT_LN_PRE_W, 0),
        P32(LLA_VIT_LN_PRE_B, 0), P32(LLA_VIT_LN1_W, 0), P32(LLA_VIT_LN1_B, 0), ws.h, M);
    }
    LLA_TRY(check_launch());
    LLA_SNAP(0, ws.x, (size_t)M * kWidth * 4);

    // LayerNorm fused into the GEMMs around it: a residual GEMM that writes ALL rows also leaves
    // xhat = fp16(x) and per-row partial sums; ln_stats_kernel turns those into (mean, rstd); the GEMM that follows
    // the LayerNorm reads xhat with gamma folded into its weights and applies mean / rstd / beta in its epilogue.
    // Not fused: ln_1 of block 0 (made by ln_pre_ln1_kernel) and the class-token-only rows of the last block.
    const bool fuse = ln_fused();
    bool stats_ready = false;   // x of the current point in the block has xhat + stats
    auto finish_stats = [&]() -> int {
#ifdef LLA_PROBES
      ProfScope scope(prof, st, LLA_PROF_LAYERNORM, (double)M * (kLnSlots * 2 + 2) * 4.0);
      ln_stats_kernel<<<(M + 255) / 256, 256, 0, st>>>(ws.part, ws.stats, M);
      return check_launch();
#else
      return LLA_EINVAL;
#endif
    };
    int dir = 0;                       // direction of the kernel being launched (0: first rows first)
    const int zig = zigzag();
    // LayerNorm in the residual GEMMs' epilogues (EPI_RESID_LNX, gemm_q4.hip) for slices of whole 256-row tiles that the
    // four-wave kernel takes: ln_2 of every block in out-proj's epilogue, ln_1 of the next block in c_proj's; behind each
    // such GEMM lnx_cleanup_kernel redoes the row tiles whose column tiles missed each other.  Output in ws.xh (the
    // consumers' A operand then).  The exchange words of the slice's 22 launches are zeroed here, once.
    // (... and inside the four-wave kernel's 32-bit panel offsets for BOTH residual GEMMs -- c_proj's A operand has the
    // longest rows, lda = 3072: slices of 14 080+ images, ADVICE r5 -- so that launch_q4(EPI_RESID_LNX) cannot answer
    // LLA_EINVAL here; such slices take the LayerNorm kernels, with the residual GEMMs on the ping-pong kernel)
    const bool lnx = !fuse && (M & 255) == 0 && M >= 9000 && (size_t)M * kMlp * 2 < (1ull << 32) && (!tower || tower->lnx);
    const int tiles_m = M / 256;
    float *const lnx_part = ws.part;                                                       // [tiles_m][3][256] granules of 16 bytes
    unsigned *const lnx_words = reinterpret_cast<unsigned *>(ws.part + (size_t)tiles_m * 3 * 256 * 4);   // [22]{flag [tiles_m][3], done [tiles_m][3]}
    int lnx_launch = 0;
    if (lnx && hipMemsetAsync(lnx_words, 0, (size_t)(2 * kLayers - 1) * tiles_m * 6 * sizeof(unsigned), st) != hipSuccess)
      return hip_fail(hipGetLastError());
    auto lnx_gemm = [&](GemmParams g, const float *gamma, const float *beta, int &d) -> int {
      g.lnx_g = gamma; g.lnx_b = beta; g.lnx_h = ws.xh; g.lnx_part = lnx_part;
      g.lnx_flag = lnx_words + (size_t)lnx_launch * tiles_m * 6;
      g.lnx_done = g.lnx_flag + (size_t)tiles_m * 3;
      g.lnx_wait = tower ? tower->lnx_wait : kLnxWaitDefault;
      g.lnx_epoch = next_lnx_epoch();
      ++lnx_launch;
      d ^= zig; g.rev = d;
      {
        ProfScope scope(prof, st, LLA_PROF_GEMM, 2.0 * g.M * g.N * g.K);
        const int rc2 = launch_q4(EPI_RESID_LNX, g, st);
        if (rc2 != LLA_OK) return rc2;
      }
#if LLA_LNX_SYNC
      if (hipStreamSynchronize(st) != hipSuccess) return hip_fail(hipGetLastError());   // (A/B only: common.h)
#endif
      d ^= zig;
      ProfScope scope(prof, st, LLA_PROF_LAYERNORM, 0.0);
      const int split = LLA_LNX_CLEANUP_SPLIT;   // (one workgroup per row tile measured: 74 us per launch against 31 -- the few row tiles that DO need it decide)
      lnx_cleanup_kernel<<<tiles_m * split, 256, 0, st>>>(ws.x, g.lnx_done, gamma, beta, ws.xh, d, g.lnx_epoch, split);
      return check_launch();
    };
    bool ln1_by_gemm = false;          // ln_1 of this block was written to ws.xh by the c_proj GEMM of the block before
    for (int l = 0; l < kLayers; ++l) {
      const bool ln1_fused = fuse && l > 0 && stats_ready;
      if (l > 0 && !ln1_fused && !ln1_by_gemm) {
        dir ^= zig;
        LLA_TRY(layernorm_impl(ws.x, kWidth, P32(LLA_VIT_LN1_W, l), P32(LLA_VIT_LN1_B, l), ws.h,
                                 M, st, prof, dir));
        if (sh) {
          LLA_TRY(layernorm_impl(ws.x, kWidth, P32(LLA_VIT_LN1_W, l), P32(LLA_VIT_LN1_B, l), shh, M, st, prof));
          shadow_cmp(l * 8 + 0, ws.h, shh, (size_t)M * kWidth * 2);
        }
      }
      GemmParams g{};
      g.M = M;
      // Only the class token leaves the tower (ln_post(x[:, 0]) @ proj), so after the last
      // block's attention every remaining per-row op runs on the B class rows alone: row
      // stride 50*768 selects them in place, results are bit-identical to the full pass.
      const bool cls_only = (l == kLayers - 1) && prune_last_block();
      // qkv = h @ in_proj^T + b
      g.A = ln1_by_gemm ? ws.xh : ws.h; g.W = P16(LLA_VIT_QKV_W, l); g.bias = P32(LLA_VIT_QKV_B, l); g.C = ws.big;
      g.N = 3 * kWidth; g.K = kWidth; g.lda = kWidth; g.ldc = 3 * kWidth;
      if (ln1_fused) {
        g.A = ws.xh; g.W = P16(LLA_VIT_QKV_WG, l); g.bias = P32(LLA_VIT_QKV_D, l);
        g.ln_c = P32(LLA_VIT_QKV_C, l); g.ln_stats = ws.stats;
      }
      if (cls_only) {
        // ... and of the last block's queries only the class token's is used: K and V for every token
        // (columns 768 .. 2303), Q for the B class rows.  The other query rows keep stale (finite) bytes;
        // attention rows are independent, and only row 0 of every image is read afterwards.
        GemmParams kv = g;
        kv.W = g.W + (size_t)kWidth * kWidth; kv.bias = g.bias + kWidth;
        if (kv.ln_c) kv.ln_c = g.ln_c + kWidth;
        kv.C = ws.big + kWidth; kv.N = 2 * kWidth;
        GemmParams q = g;
        q.M = bc; q.N = kWidth; q.lda = kTokens * kWidth; q.ldc = kTokens * 3 * kWidth;
        if (ln1_fused) {
          q.ln_stats_stride = kTokens;   // class rows: stats of row 50 b
          LLA_TRY_FUSED((launch_gemm<EPI_F16_LN, A_PLAIN>(kv, st, prof)));
          LLA_TRY_FUSED((launch_gemm<EPI_F16_LN, A_PLAIN>(q, st, prof)));
        } else {
          LLA_TRY((launch_gemm<EPI_F16, A_PLAIN>(kv, st, prof)));
          LLA_TRY((launch_gemm<EPI_F16, A_PLAIN>(q, st, prof)));
        }
      } else {
        dir ^= zig; g.rev = dir;
        if (ln1_fused) LLA_TRY_FUSED((launch_gemm<EPI_F16_LN, A_PLAIN>(g, st, prof)));
        else LLA_TRY((launch_gemm<EPI_F16, A_PLAIN>(g, st, prof)));
        g.rev = 0;
        if (sh && !ln1_fused) {
          GemmParams g2 = g;
          g2.C = sbig;
          LLA_TRY((launch_gemm<EPI_F16, A_PLAIN>(g2, st, prof)));
          shadow_cmp(l * 8 + 1, ws.big, sbig, (size_t)M * 3 * kWidth * 2);
        }
      }
      g.ln_c = nullptr; g.ln_stats = nullptr;
      // o = softmax(q k^T / 8) v   (h is dead, reuse it)
      dir ^= zig;
      LLA_TRY(attention_impl(ws.big, ws.h, bc, st, prof, dir));
      if (sh && !cls_only) {
        LLA_TRY(attention_impl(ws.big, shh, bc, st, prof));
        shadow_cmp(l * 8 + 2, ws.h, shh, (size_t)M * kWidth * 2);
      }
      const int rows = cls_only ? bc : M;
      const int xs = cls_only ? kTokens * kWidth : kWidth;  // row stride of x / o for this pass
      // x += o @ out_proj^T + b
      g.M = rows;
      g.A = ws.h; g.W = P16(LLA_VIT_OUT_W, l); g.bias = P32(LLA_VIT_OUT_B, l); g.C = ws.x;
      g.N = kWidth; g.K = kWidth; g.lda = xs; g.ldc = xs;
      const bool ln2_fused = fuse && !cls_only;
      if (ln2_fused) { g.xhat = ws.xh; g.ln_part = ws.part; }
      if (sh && !cls_only) (void)hipMemcpyAsync(sx, ws.x, (size_t)M * kWidth * 4, hipMemcpyDeviceToDevice, st);
      const bool ln2_by_gemm = lnx && !cls_only;
      if (ln2_by_gemm) {
        LLA_TRY(lnx_gemm(g, P32(LLA_VIT_LN2_W, l), P32(LLA_VIT_LN2_B, l), dir));
      } else {
      dir ^= zig; g.rev = dir;
      if (ln2_fused) LLA_TRY_FUSED((launch_gemm<EPI_RESID_LN, A_PLAIN>(g, st, prof)));
      else LLA_TRY((launch_gemm<EPI_RESID, A_PLAIN>(g, st, prof)));
      g.rev = 0;
      }
      if (sh && !cls_only) {
        GemmParams g2 = g;
        g2.C = sx; g2.xhat = nullptr; g2.ln_part = nullptr;
        LLA_TRY((launch_gemm<EPI_RESID, A_PLAIN>(g2, st, prof)));
        shadow_cmp(l * 8 + 3, ws.x, sx, (size_t)M * kWidth * 4);
      }
      g.xhat = nullptr; g.ln_part = nullptr;
      LLA_SNAP(1 + 2 * l, ws.x, (size_t)M * kWidth * 4);
      if (ln2_fused) {
        LLA_TRY(finish_stats());
      } else if (!ln2_by_gemm) {
        dir ^= zig;
        LLA_TRY(layernorm_impl(ws.x, (size_t)xs, P32(LLA_VIT_LN2_W, l), P32(LLA_VIT_LN2_B, l), ws.h,
                               rows, st, prof, dir));
        if (sh && !cls_only) {
          LLA_TRY(layernorm_impl(ws.x, (size_t)xs, P32(LLA_VIT_LN2_W, l), P32(LLA_VIT_LN2_B, l), shh, rows, st, prof));
          shadow_cmp(l * 8 + 4, ws.h, shh, (size_t)M * kWidth * 2);
        }
      }
      LLA_SNAP16(l, 0, ws.h, (size_t)M * kWidth * 2);
      // g = quickgelu(h @ c_fc^T + b)
      g.A = ln2_by_gemm ? ws.xh : ws.h; g.W = P16(LLA_VIT_FC_W, l); g.bias = P32(LLA_VIT_FC_B, l); g.C = ws.big;
      g.N = kMlp; g.K = kWidth; g.lda = kWidth; g.ldc = kMlp;
      if (ln2_fused) {
        g.A = ws.xh; g.W = P16(LLA_VIT_FC_WG, l); g.bias = P32(LLA_VIT_FC_D, l);
        g.ln_c = P32(LLA_VIT_FC_C, l); g.ln_stats = ws.stats;
      }
      dir ^= zig; g.rev = dir;
      if (ln2_fused) LLA_TRY_FUSED((launch_gemm<EPI_QGELU_LN, A_PLAIN>(g, st, prof)));
      else LLA_TRY((launch_gemm<EPI_QGELU, A_PLAIN>(g, st, prof)));
      g.rev = 0;
      if (sh && !cls_only && !ln2_fused) {
        GemmParams g2 = g;
        g2.C = sbig;
        LLA_TRY((launch_gemm<EPI_QGELU, A_PLAIN>(g2, st, prof)));
        shadow_cmp(l * 8 + 5, ws.big, sbig, (size_t)M * kMlp * 2);
      }
      g.ln_c = nullptr; g.ln_stats = nullptr;
      LLA_SNAP16(l, 1, ws.big, (size_t)M * kMlp * 2);
      // x += g @ c_proj^T + b
      g.A = ws.big; g.W = P16(LLA_VIT_CPROJ_W, l); g.bias = P32(LLA_VIT_CPROJ_B, l); g.C = ws.x;
      g.N = kWidth; g.K = kMlp; g.lda = kMlp; g.ldc = xs;
      const bool next_fused = fuse && !cls_only && l + 1 < kLayers;   // ln_1 of the next block
      if (next_fused) { g.xhat = ws.xh; g.ln_part = ws.part; }
      if (sh && !cls_only) (void)hipMemcpyAsync(sx, ws.x, (size_t)M * kWidth * 4, hipMemcpyDeviceToDevice, st);
      ln1_by_gemm = lnx && !cls_only && l + 1 < kLayers;      // ln_1 of the next block rides in this GEMM's epilogue
      if (ln1_by_gemm) {
        LLA_TRY(lnx_gemm(g, P32(LLA_VIT_LN1_W, l + 1), P32(LLA_VIT_LN1_B, l + 1), dir));
      } else {
      dir ^= zig; g.rev = dir;
      if (next_fused) LLA_TRY_FUSED((launch_gemm<EPI_RESID_LN, A_PLAIN>(g, st, prof)));
      else LLA_TRY((launch_gemm<EPI_RESID, A_PLAIN>(g, st, prof)));
      g.rev = 0;
      }
      if (sh && !cls_only) {
        GemmParams g2 = g;
        g2.C = sx; g2.xhat = nullptr; g2.ln_part = nullptr;
        LLA_TRY((launch_gemm<EPI_RESID, A_PLAIN>(g2, st, prof)));
        shadow_cmp(l * 8 + 6, ws.x, sx, (size_t)M * kWidth * 4);
      }
      g.xhat = nullptr; g.ln_part = nullptr;
      stats_ready = false;
      if (next_fused) { LLA_TRY(finish_stats()); stats_ready = true; }
      LLA_SNAP(2 + 2 * l, ws.x, (size_t)M * kWidth * 4);
    }

    // ln_post on class tokens only, then @ proj
    LLA_TRY(layernorm_impl(ws.x, (size_t)kTokens * kWidth, P32(LLA_VIT_LN_POST_W, 0),
                           P32(LLA_VIT_LN_POST_B, 0), ws.h, bc, st, prof));
    GemmParams g{};
    g.A = ws.h; g.W = P16(LLA_VIT_PROJ_T, 0); g.bias = nullptr;
    g.C = reinterpret_cast<f16 *>(z_out) + (size_t)c0 * kOut;
    g.M = bc; g.N = kOut; g.K = kWidth; g.lda = kWidth; g.ldc = kOut;
    LLA_TRY((launch_gemm<EPI_F16, A_PLAIN>(g, st, prof)));
  }
#undef LLA_TRY
#undef LLA_TRY_FUSED
#undef LLA_SNAP
#undef LLA_SNAP16
  if (lanes == 2 && deferred) { ln->next = slice & 1; ln->dirty = true; return LLA_OK; }
  if (lanes == 2) return lanes_join(ln, st_caller);   // the caller's stream continues when both lanes are done
  return LLA_OK;
}

}  // extern "C"
