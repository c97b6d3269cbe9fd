// Retired GEMM kernels, -DLLA_PROBES build only (make probes): the two-workgroups-per-CU kernel ("duo", round 2) and the first
// four-wave kernel ("quad", round 3) with their launchers -- measured alternatives that lost (docs/history, docs/history/DESIGN_rounds_1-5.md 5.1 / 5.5).
// Moved out of vit.hip verbatim in round 6; included by ablation/gemm_select.hip when LLA_PROBES is defined.
#pragma once
#include "../gemm_kernels.h"
#include "ablation.h"

namespace lla {
namespace {

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
// address arithmetic / waits / barriers per SIMD instead of two (docs/history/DESIGN_rounds_1-5.md 5.5).  Same persistent tile walk, LDS
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


}  // namespace
}  // namespace lla
