// The tower's GEMM kernels of rounds 1-2 (gfx950, v_mfma_f32_32x32x16_f16): templates shared by the product's launcher
// (gemm_pp.hip) and the tools/ builds' switchable one (ablation/gemm_select.hip).  Moved out of vit.hip verbatim in round 6
// (vit.hip split by translation unit); the product instantiates only what gemm_pp.hip launches.
//
// Stand in for the Linear / conv1 layers inside `z = self.clip(X)` (hub/compressor.py:93; clip==1.0 VisionTransformer):
//   gemm_f16_kernel         128 x 128 x 64 tiles, 4 waves (2 x 2) of 64 x 64, LDS double buffer with an XOR swizzle that makes
//                           every ds_read_b128 conflict free, operands by LDS-DMA, fused epilogues (bias / QuickGELU / residual /
//                           patch scatter + pos), A-operand modes that read 32 x 32 patches straight out of NHWC / NCHW image
//                           batches (no im2col pass).  M <= 128.
//   gemm256_f16_kernel      256 x 128 tiles, 8 waves, three-stage LDS ring, one tile per workgroup.  128 < M < 9000, and the
//                           RN50 tower's narrow / implicit 3x3 convolutions.
//   gemm_persistent_kernel  persistent 256- / 320-row tiles, all waves in lock-step (the 128-wide persistent path).
//   gemm_pp_kernel          persistent 256 / 320 x 256 tiles, the two wave rows half a K-tile out of phase (docs/history/DESIGN_rounds_1-5.md 5.1): the
//                           large GEMMs the four-wave / eight-wave kernels do not take (patch embedding, ragged M, K < 128).
#pragma once
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
constexpr int kStages = 3, kGroupM = LLA_GROUP_M;   // (BM2 x BN2 = 256 x 128: gemm_common.h)
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

// Rounds a persistent grid needs for `tiles` work items (the slowest workgroup's tile count).
inline int rounds_for(int tiles, int cus) { return (tiles + cus - 1) / cus; }

}  // namespace
}  // namespace lla
