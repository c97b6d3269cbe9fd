// Four-wave GEMM for the tower's large layers (gfx950): C[M][N] (+)= A[M][K] W[N][K]^T with the fused epilogues.
//
// Stands in for the Linear layers inside `z = self.clip(X)` (hub/compressor.py:93; clip==1.0
// VisionTransformer: in_proj / out_proj / c_fc / c_proj), as gemm_pp_kernel does, on another layout:
//
//   * 256 x 256 x 64 tiles on FOUR waves (2 x 2), one wave per SIMD, each a 128 x 128 output tile = 16 accumulator
//     tiles of 32x32 in the AccVGPRs.  8 fragment reads per 16 MFMAs (the 160 x 64 wave tile of the ping-pong kernel:
//     7 per 10, i.e. 40 % more LDS traffic per flop), one wave's address arithmetic / waits / barriers per SIMD.
//   * FRAGMENT-MAJOR K-tiles.  A K-tile is walked in four phases, one 32-row A fragment x the wave's whole
//     128-column B operand each (16 MFMAs = 512 matrix-pipe cycles).  B stays in 64 VGPRs for the K-tile; the A
//     fragment of the NEXT phase arrives in a four-register-quad ring, every quad refilled one k-step after the
//     MFMAs that consumed it (a VGPR write behind an MFMA that still reads it stalls the wave).  In the last phase
//     the B registers are refilled the same way with the next K-tile's operand.  So a fragment read is issued
//     >= 2 k-steps (256+ cycles) before its first use and the only wave of the SIMD never waits for the LDS.
//   * The walk frees the LDS progressively: A fragment a of K-tile t is dead after phase a, the B operand after
//     phase 0 (it was read into registers during the previous K-tile) -- each is refilled by LDS-DMA
//     (global_load_lds_dwordx4, swizzle on the source address) with K-tile t+2 right away, so with the same two
//     64-KiB stages the operand stream runs 1-1.75 K-tiles (2-4k cycles) ahead of its first read.  The first cut of
//     this layout (round 3, gemm_quad_kernel) refilled a whole stage at the K-tile boundary: < 1 K-tile of lead,
//     all 16 DMA instructions in a burst, and lost a third of its time to that.
//   * DMA issue is spread: one instruction behind every fourth MFMA or so (kSched), never in a burst (the
//     vector-memory path accepts one 1-KiB request per ~16 cycles per CU; a burst blocks the issuing wave,
//     and with one wave per SIMD nothing else feeds the matrix pipe meanwhile).
//   * One s_barrier per phase.  It orders (WAR) the refill of the fragment freed by the previous phase behind
//     every wave's reads of it, and (RAW) the first reads of the pieces each wave confirmed (counted vmcnt)
//     at the end of the previous phase.  The counts come from the schedule table at compile time.
//   * Persistent XCD-contiguous tile walk, operand stream running across tile boundaries, epilogues as in the
//     other kernels (gemm_common.h): same K order per output element, hence bit-identical results.
//
// Scope: A_PLAIN operands, M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 256; everything else stays on
// gemm_pp_kernel / gemm_persistent_kernel (launch_gemm in gemm_pp.hip asks launch_q4 first).
// (round 6: the kernel template lives in this header; gemm_q4.hip holds the product's launcher, ablation/gemm_q4_select.hip the
// tools/ builds' switchable one -- VERDICT r5 #6)
#pragma once
#include "gemm_common.h"

namespace lla {
namespace {

constexpr int kQStage = 65536, kQARegion = 32768, kQPiece = 4096;   // bytes: stage = A 256 x 128 B + B 256 x 128 B
#ifndef LLA_Q4_GROUP_M
#define LLA_Q4_GROUP_M 4
#endif
constexpr int kQGroupM = LLA_Q4_GROUP_M;

// ---------------------------------------------------------------------------
// DMA schedule.  Slot (t, p) = phase p of K-tile t.  Items: A fragment i (both wave rows: pieces i and 4 + i,
// two instructions per wave) of K-tile t + d, or one B piece (one instruction) of K-tile t + 2.
// Legality (WAR): A fragment i with d = 2 only in slots p > i; with d = 1 anywhere; B pieces in slots p >= 1.
// Deadline (RAW): A fragment i of K-tile u is first read in global slot 4u + i - 1 (one k-step into the phase
// before its own), the B operand of K-tile u in global slot 4u - 1: confirmed at the end of the slot before.
// ---------------------------------------------------------------------------
struct QItem { int kind, idx, d, step; };   // kind 0 = A fragment, 1 = B piece; step = k-step of the slot it follows
struct QSched { int n[4]; QItem it[4][8]; };

constexpr QSched q_sched(int var) {
  QSched s{};
  auto put = [&s](int p, int kind, int idx, int d) { s.it[p][s.n[p]] = QItem{kind, idx, d, 0}; ++s.n[p]; };
  if (var == 0) {          // refill as soon as freed: 1 / 5 / 5 / 1 items = 2 / 6 / 6 / 2 instructions
    put(0, 0, 3, 1);
    put(1, 0, 0, 2); for (int q = 0; q < 4; ++q) put(1, 1, q, 2);
    put(2, 0, 1, 2); for (int q = 4; q < 8; ++q) put(2, 1, q, 2);
    put(3, 0, 2, 2);
  } else if (var == 1) {   // four instructions per slot
    put(0, 0, 2, 1); put(0, 0, 3, 1);
    for (int q = 0; q < 4; ++q) put(1, 1, q, 2);
    for (int q = 4; q < 8; ++q) put(2, 1, q, 2);
    put(3, 0, 0, 2); put(3, 0, 1, 2);
  } else {                 // B early (its deadline is the tightest), A fragments behind: 2 / 5 / 5 / 4
    put(0, 0, 3, 1);
    put(1, 0, 0, 2); for (int q = 0; q < 3; ++q) put(1, 1, q, 2);
    put(2, 0, 1, 2); for (int q = 3; q < 6; ++q) put(2, 1, q, 2);
    put(3, 0, 2, 2); for (int q = 6; q < 8; ++q) put(3, 1, q, 2);
  }
  // spread a slot's items over its four k-steps
  for (int p = 0; p < 4; ++p)
    for (int k = 0; k < s.n[p]; ++k) s.it[p][k].step = s.n[p] <= 4 ? k : (k * 4) / s.n[p];
  return s;
}
constexpr int q_instrs(const QItem &it) { return it.kind == 0 ? 2 : 1; }

// One phase's DMA instructions flattened (an A fragment = pieces idx and idx + 4), each with the MFMA of the phase
// (0..15 = 4 k-steps x 4 B fragments) it is issued behind.  One wave per SIMD: an MFMA covers ~32 cycles of issue
// (5-8 single-issue instructions), so fillers are placed one DMA instruction (5-6 instructions with its address
// arithmetic) per MFMA shadow, spread over the phase, and never behind the first MFMA of a k-step, where the
// operand refills go.
struct QInstr { int kind, piece, d, pos; };
struct QPhase { int n; QInstr in[12]; };
constexpr QPhase q_phase(int var, int p) {
  const QSched s = q_sched(var);
  QPhase ph{};
  for (int k = 0; k < s.n[p]; ++k) {
    const QItem &it = s.it[p][k];
    if (it.kind == 0) {
      ph.in[ph.n++] = QInstr{0, it.idx, it.d, 0};
      ph.in[ph.n++] = QInstr{0, it.idx + 4, it.d, 0};
    } else {
      ph.in[ph.n++] = QInstr{1, it.idx, it.d, 0};
    }
  }
  for (int k = 0; k < ph.n; ++k) {
    int pos = (k * 16 + 8) / ph.n;
    if ((pos & 3) == 0) ++pos;
    ph.in[k].pos = pos;
  }
  return ph;
}

// vmcnt argument at the end of slot p: instructions issued after the youngest piece that the NEXT slot reads first.
constexpr int q_confirm(int var, int p) {
  const QSched s = q_sched(var);
  // global sequence number of the last instruction of every (kind, idx, K-tile) over K-tiles 0..7 of a steady stream
  int seq = 0, last_a[12][4] = {}, last_b[12] = {}, upto[12][4] = {};
  for (int t = -2; t < 8; ++t)
    for (int q = 0; q < 4; ++q) {
      for (int k = 0; k < s.n[q]; ++k) {
        const QItem &it = s.it[q][k];
        const int u = t + it.d;
        seq += q_instrs(it);
        if (u >= 0 && u < 12) { if (it.kind == 0) last_a[u][it.idx] = seq; else last_b[u] = seq; }
      }
      if (t >= 0) upto[t][q] = seq;
    }
  const int t = 4;                      // a steady-state K-tile
  int need = 0;                         // youngest required sequence number
  auto req = [&need](int v) { if (v > need) need = v; };
  if (p == 0) req(last_a[t][2]);        // slot (t, 1) first reads fragment 2 of K-tile t
  if (p == 1) req(last_a[t][3]);
  if (p == 2) { req(last_a[t + 1][0]); req(last_b[t + 1]); }
  if (p == 3) req(last_a[t + 1][1]);
  return upto[t][p] - need;
}

// The same count when the fp16 epilogue is PIPELINED into the K loop (kPipe below): the stores of an output tile's
// four row fragments ride in the MFMA shadows of the slots (LAST, 1..3) and (FIRST of the next tile, 0), one
// 16-byte store behind every odd MFMA of the slot (8 per slot), and the wave's vector-memory queue retires in
// order, loads and stores alike (one counter): a counted wait must also let the younger STORES stay in flight, or
// it would wait for operand pieces issued a phase ago instead of a K-tile ago.  The instruction stream of a window
// of K-tiles is replayed at compile time.  kind: 0 = a K-tile with no store in its look-back (also used, one-sidedly
// safe, for the K-tile after FIRST: at most 5 old stores are waited for on top), 1 = LAST, 2 = FIRST behind a LAST.
constexpr int q_confirm_pipe(int var, int kind, int p) {
  const int TL = 5;                     // the LAST K-tile of the window; FIRST = TL + 1
  int seq = 0, last_a[14][4] = {}, last_b[14] = {}, upto[14][4] = {};
  for (int t = 0; t < 10; ++t)
    for (int q = 0; q < 4; ++q) {
      const QPhase ph = q_phase(var, q);
      const bool stores = (t == TL && q >= 1) || (t == TL + 1 && q == 0);
      for (int n = 0; n < 16; ++n) {
        for (int k = 0; k < ph.n; ++k)
          if (ph.in[k].pos == n) {
            ++seq;
            const int u = t + ph.in[k].d;
            if (ph.in[k].kind == 0) { if (last_a[u][ph.in[k].piece & 3] < seq) last_a[u][ph.in[k].piece & 3] = seq; }
            else if (last_b[u] < seq) last_b[u] = seq;
          }
        if (stores && (n & 1)) ++seq;
      }
      upto[t][q] = seq;
    }
  const int t = kind == 1 ? TL : kind == 2 ? TL + 1 : 3;
  int need = 0;
  auto req = [&need](int v) { if (v > need) need = v; };
  if (p == 0) req(last_a[t][2]);
  if (p == 1) req(last_a[t][3]);
  if (p == 2) { req(last_a[t + 1][0]); req(last_b[t + 1]); }
  if (p == 3) req(last_a[t + 1][1]);
  return upto[t][p] - need;
}
static_assert(q_confirm_pipe(1, 0, 0) == q_confirm(1, 0) && q_confirm_pipe(1, 0, 1) == q_confirm(1, 1) &&
              q_confirm_pipe(1, 0, 2) == q_confirm(1, 2) && q_confirm_pipe(1, 0, 3) == q_confirm(1, 3),
              "the replayed stream and the schedule table must agree where there are no stores");
static_assert(q_confirm_pipe(1, 1, 3) <= 63 && q_confirm_pipe(1, 2, 1) <= 63, "vmcnt is a 6-bit field");

template <int... I, class F>
__device__ __forceinline__ void q_static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void q_static_for(F &&f) { q_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// One LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS m0 + 16 lane.  M0 is declared
// clobbered instead of saved / restored around every instruction (two SALU instructions less per piece).
__device__ __forceinline__ void q_dma(unsigned voff, const unsigned char *sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1" LLA_DMA_SC
               :
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory", "m0");
}

// The same through a buffer descriptor (round 5): address = descriptor base + soff (SGPR, the K-tile panel's byte offset
// from the operand's base) + voff (VGPR: this lane's row / chunk offset INCLUDING the piece's row offset, one register
// per piece).  No 64-bit scalar address arithmetic per piece (s_add_u32 + s_addc_u32 + a 64-bit SGPR pair each), and the
// LDS destination goes to M0 by the add that forms it: 3 instructions per piece instead of 6-7 -- what hipBLASLt's kernel
// does (docs/history/DESIGN_rounds_1-5.md 5.6).  Same bytes from the same addresses into the same LDS words: bit-identical by construction.
#ifndef LLA_Q4_BUFDMA
#define LLA_Q4_BUFDMA 1
#endif
#ifndef LLA_LNX_OCKL_VOTE
#define LLA_LNX_OCKL_VOTE 0
#endif
#ifndef LLA_LNX_NO_SKIP
#define LLA_LNX_NO_SKIP 0
#endif
#ifndef LLA_LNX_TRIPLES
#define LLA_LNX_TRIPLES 1
#endif
typedef unsigned q_rsrc_t __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ void q_dma_buf(unsigned voff, q_rsrc_t rsrc, unsigned soff, unsigned lds_base) {
  asm volatile("s_add_u32 m0, %3, %4\n\t"
               "s_nop 0\n\t"
               "buffer_load_dwordx4 %0, %1, %2 offen" LLA_DMA_SC " lds"
               :
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "n"(IMM)
               : "memory", "m0", "scc");
}

__device__ __forceinline__ const unsigned char *q_uniform(const unsigned char *ptr) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const unsigned char *>(((unsigned long long)hi << 32) | lo);
}

// fp16 epilogue of a whole 128 x 128 wave tile: gemm_epilogue_swap (gemm_common.h) with the bias read from LDS
// (the whole [N] vector is put there once per launch by LDS-DMA) instead of global memory.  A bias load the
// compiler can see makes it wait `vmcnt(0)` before the first use: called per 64-column half as in the ping-pong
// kernel, the second wait sat behind the first half's stores -- an HBM write round trip per tile (cycle trace:
// 11.4k cycles per QKV tile epilogue, 27 % of the tile) -- and every such wait also drains the LDS-DMA ring.
// Same arithmetic, same roundings, same store addresses: bit-identical.
template <int EPI>
__device__ __forceinline__ void q4_epilogue_f16(const GemmParams &p, f32x16 (&acc)[2][4][2], int mw, int nw, int lane,
                                                const unsigned char *bias_lds) {
  static_assert(epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU, "fp16 outputs only");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int r32 = lane & 31, hk = lane >> 5;
  const int ncol = nw + 4 * hk;
  unsigned char *crow = reinterpret_cast<unsigned char *>(reinterpret_cast<f16 *>(p.C) + (size_t)(mw + r32) * p.ldc + nw) + 16 * hk;
  const size_t row_step = (size_t)32 * p.ldc * 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // one 32-column fragment's bias quads at a time: 16 registers instead of 64
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4 *>(bias_lds + (ncol + 32 * j + 8 * g) * 4);
    auto pack4 = [&](int i, int g, unsigned &lo, unsigned &hi) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[j >> 1][i][j & 1][4 * g + e];
      v += bias4[g];
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        unsigned ax, ay, bx, by;
        pack4(i, k, ax, ay);
        pack4(i, k + 1, bx, by);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        const u32x4 out = {rx[0], ry[0], rx[1], ry[1]};
        store16(crow + i * row_step + (32 * j + 8 * k) * 2, out);
      }
  }
}


// ---------------------------------------------------------------------------
// EPI_RESID_LNX: x += A W^T + b, AND the LayerNorm that follows in the tower (ln_2 after out-proj, ln_1 of the next
// block after c_proj; hub/compressor.py:93 -> clip VisionTransformer) applied to this workgroup's 256 x 256 chunk of
// the updated rows and written as fp16 -- so that layernorm768_kernel, which re-read all of x (3 KB per row) a moment
// later, is not launched (rounds 3-4 tried two other fusions: docs/history/DESIGN_rounds_1-5.md 5.4, 5.7; this is the one 7a.2 left open,
// done on the PRODUCER side: the consumer GEMMs are untouched).
//
// A row's statistics need all 768 columns = the three column tiles of its row tile, which three workgroups compute at
// about the same time (consecutive logical tiles of one XCD's range).  Each keeps its updated values in registers
// (the 256 accumulator registers a tile frees as it is stored), publishes exact per-row (sum, sum of squares) of its
// 256 columns -- 2 KiB, write-through (`sc1`) stores, then a flag: MI355X_MICROARCH.md "handoff-flag" -- waits a
// BOUNDED time for its two siblings' flags, adds the three partial sums in the canonical order of gemm_common.h
// (ln_finish: the same bits layernorm768_kernel computes from the stored row), normalises its chunk and stores it.
// A workgroup whose siblings are late (their tile sits in another round of the persistent grid, or another process
// holds their CU) does not wait for them: it leaves its `done` word at 0 and lnx_cleanup_kernel (layernorm_attention.hip), launched
// behind every such GEMM, redoes the few row tiles that are not complete from x.  Nothing ever spins unbounded,
// no atomics, no assumption about dispatch order or placement; both paths give the same bits by construction.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void q4_epilogue_resid_lnx(const GemmParams &p, f32x16 (&acc)[2][4][2], int m0, int n0,
                                                      int wr, int wc, int lane, int tid, unsigned char *scr,
                                                      unsigned char *xch, bool siblings_in_this_round) {
  const int mw = m0 + wr * 128, nw = n0 + wc * 128;
  const int r32 = lane & 31, hk = lane >> 5;
  const int r16 = r32 & 15, rhalf = r32 >> 4;
  unsigned char *wrow = scr + r16 * 128;
  const int wswz = r16 >> 1;
  const int rrow = lane >> 3, rch = lane & 7;   // read side: 8 rows x 8 column quads, twice
  const unsigned char *rd[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 8 * u + rrow;
    rd[u] = scr + row * 128 + ((rch ^ (row >> 1)) << 4);
  }
  auto wave_fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // per-column vectors of this lane's 4 consecutive columns in each of the wave tile's four 32-column slots
  // (UNCONDITIONAL asm loads: see gemm_epilogue_staged)
  f32x4 bias_t[4], g_t[4], b_t[4];   // [2 h + j]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = nw + 32 * q + 4 * rch;
    const float *bp = p.bias ? p.bias + col : reinterpret_cast<const float *>(g_zero_line) + 4 * rch;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias_t[q]) : "v"(bp) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g_t[q]) : "v"(p.lnx_g + col) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b_t[q]) : "v"(p.lnx_b + col) : "memory");
  }
  float *const cbase = reinterpret_cast<float *>(p.C);
  // The updated values are KEPT for the normalisation: 256 per lane, in the AccVGPRs the accumulators vacate (named
  // through the "a" constraint: left to itself the allocator wants them in arch VGPRs next to everything else and
  // spills 640 bytes per lane).  The accumulator tiles (h, i, 0..1) are dead once units k = 2 i and 2 i + 1 are staged,
  // so the two units' 32 values go to the AccVGPRs together, after the second one.
  float ka[2][4][32];           // [h][i][16 (k & 1) + 4 (2 j + u) + e], AccVGPRs
  float2 *xw = reinterpret_cast<float2 *>(xch);              // [2 wc][256 rows] (sum, sum of squares)
  unsigned coff[2][2];
  f32x4 old[2][4];
  auto request = [&](int n) {   // unit n = 8 h + k: 16 rows (k) of 64-column half h
    const int h = n >> 3, k = n & 7;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = mw + 32 * (k >> 1) + 16 * (k & 1) + 8 * u + rrow;
      coff[n & 1][u] = (unsigned)m * (unsigned)p.ldc + (unsigned)(nw + 64 * h + 4 * rch);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        asm volatile("global_load_dwordx4 %0, %1, off" LLA_RMW_SC : "=v"(old[n & 1][2 * j + u]) : "v"(cbase + coff[n & 1][u] + 32 * j) : "memory");
    }
  };
  request(0);
  f32x4 first[4];               // the even unit's values wait here for the odd one
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int h = n >> 3, k = n & 7;
    if (n + 1 < 16) request(n + 1);
    // the rows of unit n have landed once only the younger operations are outstanding: 4 loads of unit n + 1 and the 4
    // stores of unit n - 1 (the 12 parameter loads are older still)
#define LLA_WAIT_OLD(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(old[n & 1][0]), "+v"(old[n & 1][1]), "+v"(old[n & 1][2]), "+v"(old[n & 1][3]), \
                                     "+v"(bias_t[0]), "+v"(bias_t[1]), "+v"(bias_t[2]), "+v"(bias_t[3]), "+v"(g_t[0]), "+v"(g_t[1]), "+v"(g_t[2]), "+v"(g_t[3]), \
                                     "+v"(b_t[0]), "+v"(b_t[1]), "+v"(b_t[2]), "+v"(b_t[3])::"memory")
    if (n == 0) LLA_WAIT_OLD(4);
    else if (n == 15) LLA_WAIT_OLD(4);
    else LLA_WAIT_OLD(8);
#undef LLA_WAIT_OLD
    __builtin_amdgcn_sched_barrier(0);
    float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};   // [u]: this half's two slots, s0 + s1
    f32x4 cur[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // this half's 16 rows x 32 columns of raw accumulators -> scratch -> read-side layout
      if (rhalf == (k & 1)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[h][k >> 1][j][4 * g + e];
          *reinterpret_cast<f32x4 *>(wrow + (((2 * g + hk) ^ wswz) << 4)) = v;
        }
      }
      wave_fence();
      f32x4 v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) v[u] = *reinterpret_cast<const f32x4 *>(rd[u]);
      wave_fence();
      float s[2], q[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        v[u] += bias_t[2 * h + j];
        const f32x4 o = old[n & 1][2 * j + u] + v[u];       // (same operations, same order as EPI_RESID: same x)
        store16(cbase + coff[n & 1][u] + 32 * j, o);
        cur[2 * j + u] = o;
        float a = (o[0] + o[1]) + (o[2] + o[3]);
        float b = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a);   // the 8 lanes of a row: one 32-column slot
        b += dpp_f32<0xB1>(b); b += dpp_f32<0x4E>(b); b += dpp_f32<0x141>(b);
        s[u] = a; q[u] = b;
      }
      if (j == 0) { ps[0] = s[0]; ps[1] = s[1]; pq[0] = q[0]; pq[1] = q[1]; }
      else { ps[0] = ps[0] + s[0]; ps[1] = ps[1] + s[1]; pq[0] = pq[0] + q[0]; pq[1] = pq[1] + q[1]; }
    }
    // the half's sums wait in LDS (this lane's own words: program order suffices), the wave tile's replace them
    if (rch == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float2 *w = xw + wc * 256 + wr * 128 + 32 * (k >> 1) + 16 * (k & 1) + 8 * u + rrow;
        if (h == 0) *w = make_float2(ps[u], pq[u]);
        else { const float2 f = *w; *w = make_float2(f.x + ps[u], f.y + pq[u]); }
      }
    }
    if ((k & 1) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) first[q] = cur[q];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                     : "=a"(ka[h][k >> 1][4 * q]), "=a"(ka[h][k >> 1][4 * q + 1]), "=a"(ka[h][k >> 1][4 * q + 2]), "=a"(ka[h][k >> 1][4 * q + 3])
                     : "v"(first[q][0]), "v"(first[q][1]), "v"(first[q][2]), "v"(first[q][3]));
        asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                     : "=a"(ka[h][k >> 1][16 + 4 * q]), "=a"(ka[h][k >> 1][16 + 4 * q + 1]), "=a"(ka[h][k >> 1][16 + 4 * q + 2]), "=a"(ka[h][k >> 1][16 + 4 * q + 3])
                     : "v"(cur[q][0]), "v"(cur[q][1]), "v"(cur[q][2]), "v"(cur[q][3]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- the two wave columns' partial sums meet in LDS; thread t then owns row t of the tile
  float2 *xst = reinterpret_cast<float2 *>(xch + 4096);      // [256 rows] (mean, rstd)
  volatile unsigned *xready = reinterpret_cast<volatile unsigned *>(xch + 6144);
  volatile unsigned *xfresh = reinterpret_cast<volatile unsigned *>(xch + 6148);
  __syncthreads();
  const int rt = m0 >> 8, ct = n0 >> 8;
  const float2 w0 = xw[tid], w1 = xw[256 + tid];
  const float2 mine = make_float2(w0.x + w1.x, w0.y + w1.y);
  // one 16-byte granule per row: {sum, sum of squares, epoch, epoch}.  The epoch is unique per launch (host counter),
  // so a reader can tell THIS launch's sums from anything older that a cache may still hold for the address -- the flag
  // below only says when to look (MI355X_MICROARCH.md: data-tagged granules need no ordering; round 5 found that with a
  // second process on the GPU `sc1` loads do return stale lines now and then)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  {
    const float mx = mine.x, my = mine.y;      // (scalars first: see the bit casts of the loaded granules below)
    const u32x4 granule = {__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, my), p.lnx_epoch, p.lnx_epoch};
    u32x4 *dst = reinterpret_cast<u32x4 *>(p.lnx_part) + ((size_t)rt * 3 + ct) * 256 + tid;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(granule) : "memory");
  }
  __syncthreads();   // every thread's partial sums (and, the queue being in order, its stores of x) have left
  const int c1 = ct == 2 ? 0 : ct + 1, c2 = c1 == 2 ? 0 : c1 + 1;
  if (tid == 0) {
    const unsigned epoch = p.lnx_epoch;
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p.lnx_flag + rt * 3 + ct), "v"(epoch) : "memory");
    unsigned ok = 0;
    if (p.lnx_wait >= 0) {
      const unsigned *f1 = p.lnx_flag + rt * 3 + c1, *f2 = p.lnx_flag + rt * 3 + c2;
      // a sibling tile that sits in another round of the persistent grid (or in another XCD's range) is a whole tile
      // time away: look once -- it is there if its round came before this one -- and do not wait
      const unsigned long long budget = (siblings_in_this_round || p.lnx_wait >= (1 << 20)) ? (unsigned long long)p.lnx_wait : 0ull;
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      for (;;) {
        unsigned a, b;
        asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(f1), "v"(f2) : "memory");
        if (a == epoch && b == epoch) { ok = 1; break; }
        if (__builtin_amdgcn_s_memtime() - t0 > budget) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    *xready = ok;
    *xfresh = 1u;
  }
  __syncthreads();
  if (*xready == 0u) return;       // (uniform) lnx_cleanup_kernel normalises this row tile from x
  {
    u32x4 g1, g2;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(p.lnx_part) + (size_t)rt * 3 * 256 + tid;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g1), "=&v"(g2) : "v"(src + c1 * 256), "v"(src + c2 * 256) : "memory");
    // every row's two granules must carry this launch's epoch; one that does not (a stale line) sends the whole row
    // tile to the clean-up kernel
    const bool fresh = g1[2] == p.lnx_epoch && g1[3] == p.lnx_epoch && g2[2] == p.lnx_epoch && g2[3] == p.lnx_epoch;
#if LLA_LNX_OCKL_VOTE
    if (!__syncthreads_and(fresh)) return;
#else
    // vote through a second LDS word (tid 0 set it to 1 next to `xready`, a barrier ago): a wave with a stale lane clears it
    if (__builtin_amdgcn_ballot_w64(!fresh) != 0ull && lane == 0) *xfresh = 0u;
    __syncthreads();
    if (*xfresh == 0u) return;
#endif
    // (scalars first: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 with this hipcc -- common.h; that
    // turned every sibling's sum of squares into its sum for one GPU call of round 5)
    const unsigned g1s = g1[0], g1q = g1[1], g2s = g2[0], g2q = g2[1];
    const float2 t1 = make_float2(__builtin_bit_cast(float, g1s), __builtin_bit_cast(float, g1q));
    const float2 t2 = make_float2(__builtin_bit_cast(float, g2s), __builtin_bit_cast(float, g2q));
    // (t_0 + t_1) + t_2 in COLUMN-TILE order, whichever of the three this workgroup is
    const float2 a = ct == 0 ? mine : (c1 == 0 ? t1 : t2);
    const float2 b = ct == 1 ? mine : (c1 == 1 ? t1 : t2);
    const float2 c = ct == 2 ? mine : (c1 == 2 ? t1 : t2);
    float mean, rstd;
    ln_finish((a.x + b.x) + c.x, (a.y + b.y) + c.y, mean, rstd);
    xst[tid] = make_float2(mean, rstd);
  }
  __syncthreads();
  // ---- normalise the kept values, 8 bytes (4 columns) per lane, 64 contiguous bytes per row and instruction
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float2 st[2];
      f16 *hrow[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rl = wr * 128 + 32 * (k >> 1) + 16 * (k & 1) + 8 * u + rrow;
        st[u] = xst[rl];
        hrow[u] = p.lnx_h + (size_t)(m0 + rl) * kWidth + nw + 64 * h + 4 * rch;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // q = 2 j + u
        const int j = q >> 1, u = q & 1;
        float o0, o1, o2, o3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
                     : "a"(ka[h][k >> 1][16 * (k & 1) + 4 * q]), "a"(ka[h][k >> 1][16 * (k & 1) + 4 * q + 1]),
                       "a"(ka[h][k >> 1][16 * (k & 1) + 4 * q + 2]), "a"(ka[h][k >> 1][16 * (k & 1) + 4 * q + 3]));
        const f32x4 g = g_t[2 * h + j], b = b_t[2 * h + j];
        f16x4 y;
        y[0] = (f16)ln_affine(o0, st[u].x, st[u].y, g[0], b[0]);
        y[1] = (f16)ln_affine(o1, st[u].x, st[u].y, g[1], b[1]);
        y[2] = (f16)ln_affine(o2, st[u].x, st[u].y, g[2], b[2]);
        y[3] = (f16)ln_affine(o3, st[u].x, st[u].y, g[3], b[3]);
        store8(hrow[u] + 32 * j, y);
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned epoch = p.lnx_epoch;
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p.lnx_done + rt * 3 + ct), "v"(epoch) : "memory");
  }
}

#define LLA_Q4_WAIT_VM(C) __builtin_amdgcn_s_waitcnt(0x0F70 | ((C) & 15) | (((C) >> 4) << 14))   // vmcnt(C); expcnt / lgkmcnt open

// instructions of K-tile 1 the prologue issues (the d = 2 items of one K-tile's slots)
constexpr int q_prologue(int var) {
  const QSched s = q_sched(var);
  int n = 0;
  for (int p = 0; p < 4; ++p)
    for (int k = 0; k < s.n[p]; ++k)
      if (s.it[p][k].d == 2) n += q_instrs(s.it[p][k]);
  return n;
}

// DBG: timing ablations / traces of the probe build (make XFLAGS=-DLLA_Q4_PROBE, tools/q4_probe.py, tools/q4_trace.py;
// all but 20 and 30 give WRONG results): 1 = no LDS-DMA after the prologue, 2 = no s_barrier, 3 = no epilogue,
// 13 = 1 + 3, 4 = no counted vmcnt waits, 5 = every piece re-reads the first K-tile (operands cache-hot), 8 = two
// 16x16x32 MFMAs per 32x32x16 one without DMA and epilogue, 9 = the same with DMA, 20 = s_memtime stamps per K-tile
// and around the epilogue, 30 = LDS-staged fp16 epilogue (whole 128-byte lines), 31 = residual rows not read, 40..49 = the K loop on
// v_mfma_f32_16x16x32_f16 without epilogue (see M16 below; profiles/r04_q4_mfma16_probe.txt), 50 = the four waves issue a piece's DMA
// instruction behind four different MFMAs (correct results; 10 % slower).  What they showed: docs/history/DESIGN_rounds_1-5.md 5.6.
//
// PIPE (fp16 epilogues): the epilogue is software-pipelined into the K loop instead of running between two output tiles
// with the matrix pipe idle (a fifth of the tile time at K = 768).  Fragment-major K-tiles finish the accumulators of
// row fragment a at the end of phase a of the LAST K-tile, and nothing writes them again before phase a of the next
// tile's FIRST K-tile: fragment a is biased / activated / converted / stored in the MFMA shadows of the phase after
// its last one (fragment 3 in phase 0 of the next tile; after the last tile of the workgroup by itself), eight
// 32 x 16-column units of one 16-byte store each behind the odd MFMAs of the phase.  The tile's bias quads are read
// from LDS into 64 VGPRs in phase 0 of the LAST K-tile (the K loop needs ~100 of the 256).  Same arithmetic on the
// same values, same store addresses: bit-identical to the serial epilogue (LLA_Q4_PIPE=0).
template <int EPI, int VAR, int DBG = 0, int PIPE = 0>
__global__ __launch_bounds__(256, 1) void gemm_q4_kernel(GemmParams p) {
  kernel_acquire();
  constexpr QSched kSched = q_sched(VAR);
  constexpr bool kPipe = PIPE != 0 && (EPI == EPI_F16 || EPI == EPI_QGELU) && (DBG == 0 || DBG == 20);
  constexpr int kBiasOff = 2 * kQStage + 4 * 2048, kBiasBytes = 3072 * 4;   // (launch_q4 takes N <= 3072)
  __shared__ __attribute__((aligned(16))) unsigned char smem[kBiasOff + kBiasBytes];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int r32 = lane & 31, hk = lane >> 5;

  // ---- my tiles: XCD-contiguous logical range, kQGroupM row tiles per group swept over all column tiles
  const int tiles_n = p.N / 256, tiles_m = p.M / 256;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int tq = total >> 3, tr = total & 7;
  const int start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int count = tq + (xcd < tr ? 1 : 0);
  // EPI_RESID_LNX on a full grid (256 workgroups = 8 XCDs x 32 slots, 3 column tiles): the row tiles are walked in
  // TRIPLES that never straddle two rounds.  30 slots of an XCD are 10 workgroup triples (column tile = slot % 3); the two
  // left-over slots of every XCD -- 32 = 3 x 10 + 2: with the plain walk one row tile per XCD and round had its column tiles
  // in different rounds, 9.4 % of the row tiles went through lnx_cleanup_kernel and their workgroups waited in vain --
  // form 5 more triples across XCDs (the exchange is through memory: any placement works; workgroup 15 of them idles).
  // Round j holds row tiles 85 j .. 85 j + 84: unit u = 10 xcd + slot / 3 for the regular triples, 80 + e / 3 for the extra
  // ones.  1700 row tiles = 20 rounds exactly, as many as before.
#if LLA_LNX_TRIPLES
  const bool lnx_triples = EPI == EPI_RESID_LNX && nblk == 256 && tiles_n == 3;
#else
  const bool lnx_triples = false;
#endif
  const int lnx_e = 2 * xcd + (slot - 30);                           // extra workgroups 0..15 (slot >= 30)
  const int lnx_unit = slot < 30 ? 10 * xcd + slot / 3 : 80 + lnx_e / 3;
  const int lnx_ct = slot < 30 ? slot % 3 : lnx_e % 3;
  const int n_my = lnx_triples ? ((slot >= 30 && lnx_e == 15) || lnx_unit >= tiles_m ? 0 : (tiles_m - lnx_unit + 84) / 85)
                               : (slot < count ? (count - slot + nslots - 1) / nslots : 0);
  if (n_my == 0) return;
  const int group_m = p.conv_h > 0 ? p.conv_h : kQGroupM;   // (conv_h is unused by A_PLAIN GEMMs: the launcher's LLA_Q4_GROUP_M probe rides there)
  auto tile_origin = [&](int j, int &m0, int &n0) {
    if (lnx_triples) {
      const int t = 85 * j + lnx_unit;
      m0 = (p.rev ? tiles_m - 1 - t : t) * 256;
      n0 = lnx_ct * 256;
      return;
    }
    int logical = start + slot + j * nslots;
    if (p.rev) logical = total - 1 - logical;
    const int per_group = group_m * tiles_n;
    const int grp = logical / per_group;
    const int in_grp = logical - grp * per_group;
    const int gh = (tiles_m - grp * group_m) < group_m ? (tiles_m - grp * group_m) : group_m;
    const int tn = in_grp / gh;
    m0 = (grp * group_m + (in_grp - tn * gh)) * 256;
    n0 = tn * 256;
  };
  const int nk = p.K / 64;
  // ---- operand stream.  Thread -> row tid / 8 of a 32-row piece, 16-byte position tid % 8 holding source chunk
  // (tid % 8) ^ swizzle(row) (the DMA destination is lane-linear).  Sources of the K-tiles one and two ahead of the
  // one being multiplied are wave-uniform byte pointers (tile origin + K offset folded in).
  const int srow = tid >> 3, pc = tid & 7, lc = pc ^ ((srow >> 1) & 7);
  const unsigned voffA = (unsigned)(srow * p.lda + lc * 8) * 2u;
  const unsigned voffB = (unsigned)(srow * p.K + lc * 8) * 2u;
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned wave_dst = lds_base + (unsigned)wid * 1024u;
#if LLA_Q4_BUFDMA
  unsigned srcA[2], srcB[2];                 // [d - 1]: K-tile t + d: byte offset of its panel from p.A / p.W (wave-uniform)
  auto make_rsrc = [](const void *base) {    // raw buffer over the whole address range (no bounds: the panels are inside)
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    q_rsrc_t r = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a),
                  (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu, 0xFFFFFFFFu, 0x00020000u};
    return r;
  };
  const q_rsrc_t rsrcA = make_rsrc(p.A), rsrcB = make_rsrc(p.W);
#else
  const unsigned char *srcA[2], *srcB[2];    // [d - 1]: K-tile t + d
#endif
  int cur_j = 0, cur_kt = 0;                 // position of the K-tile t + 2 cursor
  auto src_of = [&](int j, int kt, auto &a, auto &b) {
    int m0, n0;
    tile_origin(j < n_my ? j : n_my - 1, m0, n0);
    if (DBG == 46 || DBG == 47 || DBG == 48) {   // (timing ablations: 46 = both operands from this workgroup's first tile, 47 = A only, 48 = B only)
      int m00, n00;
      tile_origin(0, m00, n00);
      if (DBG != 48) m0 = m00;
      if (DBG != 47) n0 = n00;
    }
    if (DBG == 49) { m0 = 0; n0 = 0; }   // (every workgroup walks K over the same first tile: 2 x 384 KiB, L2-resident, larger than the L1)
#if LLA_Q4_BUFDMA
    a = __builtin_amdgcn_readfirstlane((unsigned)(((size_t)m0 * p.lda + (size_t)kt * 64) * 2));   // (< 2^32: launch_q4 checks)
    b = __builtin_amdgcn_readfirstlane((unsigned)(((size_t)n0 * p.K + (size_t)kt * 64) * 2));
#else
    a = q_uniform(reinterpret_cast<const unsigned char *>(p.A) + ((size_t)m0 * p.lda + (size_t)kt * 64) * 2);
    b = q_uniform(reinterpret_cast<const unsigned char *>(p.W) + ((size_t)n0 * p.K + (size_t)kt * 64) * 2);
#endif
  };
  auto advance_cursor = [&] {   // K-tile t + 2 becomes t + 1; the cursor moves one K-tile on
    if (DBG == 5 || DBG == 43) return;       // (timing ablation: every piece re-reads the first K-tile: operands always cache-hot)
    srcA[0] = srcA[1]; srcB[0] = srcB[1];
    if (++cur_kt < nk) { srcA[1] += 128; srcB[1] += 128; }
    else { cur_kt = 0; ++cur_j; src_of(cur_j, 0, srcA[1], srcB[1]); }
  };
  // byte offsets of the 32-row pieces from a panel's origin (wave-uniform: SGPRs)
  unsigned offA[8], offB[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    offA[q] = __builtin_amdgcn_readfirstlane((unsigned)q * 32u * (unsigned)p.lda * 2u);
    offB[q] = __builtin_amdgcn_readfirstlane((unsigned)q * 32u * (unsigned)p.K * 2u);
  }
  // one DMA instruction into LDS stage `st` (byte offset st x 64 KiB)
#if LLA_Q4_BUFDMA
  unsigned voffA8[8], voffB8[8];             // this lane's offset inside a panel, per 32-row piece
#pragma unroll
  for (int q = 0; q < 8; ++q) { voffA8[q] = voffA + offA[q]; voffB8[q] = voffB + offB[q]; }
  auto issue1 = [&](int kind, int q, int d, unsigned st_off) {
    const unsigned base = wave_dst + st_off;
#define LLA_Q4_PIECE(Q)                                                                                  \
    case Q: if (kind == 0) q_dma_buf<Q * kQPiece>(voffA8[Q], rsrcA, srcA[d - 1], base);                  \
            else q_dma_buf<kQARegion + Q * kQPiece>(voffB8[Q], rsrcB, srcB[d - 1], base); break;
    switch (q) { LLA_Q4_PIECE(0) LLA_Q4_PIECE(1) LLA_Q4_PIECE(2) LLA_Q4_PIECE(3) LLA_Q4_PIECE(4) LLA_Q4_PIECE(5) LLA_Q4_PIECE(6) LLA_Q4_PIECE(7) }
#undef LLA_Q4_PIECE
  };
#else
  auto issue1 = [&](int kind, int q, int d, unsigned st_off) {
    if (kind == 0) q_dma(voffA, srcA[d - 1] + offA[q], wave_dst + st_off + (unsigned)q * kQPiece);
    else q_dma(voffB, srcB[d - 1] + offB[q], wave_dst + st_off + kQARegion + (unsigned)q * kQPiece);
  };
#endif
  auto issue = [&](const QItem it, unsigned st) {
    if (it.kind == 0) { issue1(0, it.idx, it.d, st * kQStage); issue1(0, it.idx + 4, it.d, st * kQStage); }
    else issue1(1, it.idx, it.d, st * kQStage);
  };

  // ---- fragment reads: byte offsets of this lane's row, k-step s (chunk XOR-swizzled by row pair)
  const int swz = (r32 >> 1) & 7;
  unsigned a_off[4], b_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned c = (unsigned)(((2 * s + hk) ^ swz) * 16);
    a_off[s] = (unsigned)((wr * 128 + r32) * 128) + c;
    b_off[s] = (unsigned)kQARegion + (unsigned)((wc * 128 + r32) * 128) + c;
  }
  f16x8 fb[4][4], fa[4];   // fb[j][s]: B fragment j, k-step s (whole K-tile); fa[s]: ring slot of k-step s
  auto read_a = [&](unsigned so, int frag, int s) {
    fa[s] = *reinterpret_cast<const f16x8 *>(smem + so + a_off[s] + frag * kQPiece);
  };
  auto read_b1 = [&](unsigned so, int j, int s) {
    fb[j][s] = *reinterpret_cast<const f16x8 *>(smem + so + b_off[s] + j * kQPiece);
  };
  auto read_b = [&](unsigned so, int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j][s] = *reinterpret_cast<const f16x8 *>(smem + so + b_off[s] + j * kQPiece);
  };

  // ---- (probe, DBG 40 / 41: timing only, no epilogue) the same K loop on v_mfma_f32_16x16x32_f16: 8 x 8 fragments of
  // 16 rows per wave tile, two k-steps of 32 per K-tile.  Lane (r16 = lane % 16, kq = lane / 16) holds halves
  // 8 kq .. 8 kq + 7 of the k-step: chunk 4 s + kq of the row.
  constexpr bool M16 = DBG >= 40 && DBG <= 49;   // 42 = no counted waits, 43 = cache-hot operands, 44 = DMA in one burst per phase, 45 = every other DMA instruction
  const int r16 = lane & 15, kq = lane >> 4;
  unsigned a16_off[2], b16_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const unsigned c = (unsigned)(((4 * s + kq) ^ ((r16 >> 1) & 7)) * 16);
    a16_off[s] = (unsigned)((wr * 128 + r16) * 128) + c;
    b16_off[s] = (unsigned)kQARegion + (unsigned)((wc * 128 + r16) * 128) + c;
  }
  f16x8 gb[8][2], ga[2][2];   // gb[b][s]: B fragment b, k-step s (whole K-tile); ga[af][s]: ring of the phase's two A fragments
  f32x4 acc16[8][8];
  auto read_a16 = [&](unsigned so, int frag, int af, int s) {
    ga[af][s] = *reinterpret_cast<const f16x8 *>(smem + so + a16_off[s] + frag * 2048);
  };
  auto read_b16 = [&](unsigned so, int b, int s) {
    gb[b][s] = *reinterpret_cast<const f16x8 *>(smem + so + b16_off[s] + b * 2048);
  };

  // ---- the bias vector [N] goes to LDS once (fp16 epilogues read it from there): 1-KiB pieces, round robin over the
  // waves; they are the oldest DMA instructions of every wave, so the prologue's counted wait covers them
  if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
    if (p.bias) {
      for (int q = wid; q * 256 < p.N; q += 4)
        q_dma((unsigned)lane * 16u, reinterpret_cast<const unsigned char *>(p.bias) + (size_t)q * 1024,
              __builtin_amdgcn_readfirstlane(lds_base + (unsigned)kBiasOff + (unsigned)q * 1024u));
    } else {
      for (int i = tid; i * 16 < p.N * 4; i += 256) *reinterpret_cast<f32x4 *>(smem + kBiasOff + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- prologue: K-tile 0 completely; of K-tile 1 what the slots of "K-tile -1" would have issued (d = 2 items,
  // in slot order: the counted waits below assume that order); then the cursors are where slot (0, 0) expects them
  src_of(0, 0, srcA[1], srcB[1]);
  srcA[0] = srcA[1]; srcB[0] = srcB[1];
#pragma unroll
  for (int q = 0; q < 4; ++q) issue(QItem{0, q, 2, 0}, 0u);
#pragma unroll
  for (int q = 0; q < 8; ++q) issue(QItem{1, q, 2, 0}, 0u);
  advance_cursor();            // srcX[1] = K-tile 1
#pragma unroll
  for (int pp = 0; pp < 4; ++pp)
#pragma unroll
    for (int k = 0; k < kSched.n[pp]; ++k)
      if (kSched.it[pp][k].d == 2) issue(kSched.it[pp][k], 1u);
  advance_cursor();            // srcX[0] = K-tile 1 (its d = 1 items are still to come), srcX[1] = K-tile 2
  {
    constexpr int kPro = q_prologue(VAR);
    LLA_Q4_WAIT_VM(kPro);      // K-tile 0 has landed (only K-tile 1's pieces may be in flight)
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (M16) {
#pragma unroll
    for (int b = 0; b < 8; ++b) read_b16(0u, b, 0);
    read_a16(0u, 0, 0, 0); read_a16(0u, 1, 1, 0); read_a16(0u, 0, 0, 1);
  } else {
#pragma unroll
    for (int s = 0; s < 3; ++s) { read_b(0u, s); read_a(0u, 0, s); }
  }

  f32x16 acc[2][4][2];   // [column half][A fragment][B fragment in the half]: the epilogues take a 64-column half
  int it = 0;            // global K-tile counter: selects the LDS stage

  // ---- pipelined fp16 epilogue (kPipe): state of the output tile whose fragments are being stored
  f32x4 ebias[4][4];                    // [B fragment j][quad g]: bias of columns nw + 32 j + 8 g + 4 hk .. + 3
  unsigned char *ecrow = nullptr;       // this lane's row r32 of the wave tile, byte address of column nw + 8 hk
  unsigned ebias_off = 0;               // (nw + 4 hk) * 4
  const size_t e_row_step = (size_t)32 * p.ldc * 2;
  auto epi_bias = [&](int q) {
    ebias[q >> 2][q & 3] = *reinterpret_cast<const f32x4 *>(smem + kBiasOff + ebias_off + (32 * (q >> 2) + 8 * (q & 3)) * 4);
  };
  auto epi_unit = [&](int fr, int u, bool asm_read = true) {  // row fragment fr, columns 32 j + 16 h .. + 15 of the wave tile (u = 2 j + h)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int j = u >> 1, k = 2 * (u & 1);
    auto pack4 = [&](int g, unsigned &lo, unsigned &hi) {
      const f32x16 &c = acc[j >> 1][fr][j & 1];
      const f32x4 b = ebias[j][g];
      // (the accumulators pass through a volatile asm: pure arithmetic may float anywhere between its operands and
      // its use at IR level -- sched_barrier only binds the machine scheduler -- and a phase's worth of QuickGELUs
      // hoisted to the top of the phase spills)
      float v0 = c[4 * g], v1 = c[4 * g + 1], v2 = c[4 * g + 2], v3 = c[4 * g + 3];
      if (asm_read) {
        // ... and the AccVGPR reads themselves are written out: left to the register allocator, all 64 copies of a
        // fragment are made at the top of the phase, in front of the first MFMA (the MFMAs that wrote these registers
        // are >= 5 MFMAs back: no hazard the assembler would have had to pad)
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)
                     : "a"(c[4 * g]), "a"(c[4 * g + 1]), "a"(c[4 * g + 2]), "a"(c[4 * g + 3]));
      } else {
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
      v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
      if constexpr (epi_base(EPI) == EPI_QGELU) { v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3); }
      typedef f16 f16x2 __attribute__((ext_vector_type(2)));
      const f16x2 a2 = {(f16)v0, (f16)v1}, b2 = {(f16)v2, (f16)v3};
      lo = __builtin_bit_cast(unsigned, a2);
      hi = __builtin_bit_cast(unsigned, b2);
    };
    unsigned ax, ay, bx, by;
    pack4(k, ax, ay);
    pack4(k + 1, bx, by);
    const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
    const u32x4 out = {rx[0], ry[0], rx[1], ry[1]};
    store16(ecrow + fr * e_row_step + (32 * j + 8 * k) * 2, out);
  };

  // FIRST: first K-tile of an output tile (C = 0 as an inline MFMA operand; its B operand and first A fragment were
  // read after the previous epilogue).  LAST: the next K-tile's operand reads of phase 3 are left to the code behind
  // the epilogue, so that no fragment register is live across it (the fp16 epilogues take ~250 VGPRs for a moment;
  // a spilled address is reloaded with `s_waitcnt vmcnt(0)`: the whole DMA ring drained).
  int n_ev = 0;
  auto stamp = [&](int tag) {
    if constexpr (DBG == 20) {
      if (p.trace && wid == 0 && bid < 8 && n_ev < 510) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) { p.trace[bid * 1024 + 2 * n_ev] = t; p.trace[bid * 1024 + 2 * n_ev + 1] = (unsigned long long)tag; }
        ++n_ev;
      }
    }
  };
  auto ktile = [&](auto first_c, auto last_c, auto pend_c) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, PEND = decltype(pend_c)::value;
    constexpr bool SKIP = LAST && !kPipe;   // (serial epilogue) the next K-tile's operand reads wait until after it
    stamp(FIRST ? 1 : LAST ? 3 : 2);
    unsigned so = (unsigned)(it & 1) * kQStage, sn = (unsigned)((it + 1) & 1) * kQStage;
    // (the four phases are instantiated, not `#pragma unroll`ed: with an epilogue in its shadows a phase is too large
    // for the pragma's size limit, and a phase loop left rolled indexes the accumulators dynamically, i.e. in scratch)
    q_static_for<4>([&](auto a_c) {
      constexpr int a = decltype(a_c)::value;
      __builtin_amdgcn_sched_barrier(0);
      if (DBG != 2) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+s"(so), "+s"(sn));
      constexpr QPhase kPh0 = q_phase(VAR, 0), kPh1 = q_phase(VAR, 1), kPh2 = q_phase(VAR, 2), kPh3 = q_phase(VAR, 3);
      const QPhase &ph = a == 0 ? kPh0 : a == 1 ? kPh1 : a == 2 ? kPh2 : kPh3;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (FIRST && s == 0) {
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[j >> 1][a][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][0], fa[0], zero16, 0, 0, 0);
          } else if (DBG == 8 || DBG == 9) {   // (timing ablation, wrong results: the same flops as two 16x16x32 MFMAs)
            f32x16 &c = acc[j >> 1][a][j & 1];
            f32x4 c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][s], fa[s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][s], fa[s], c1, 0, 0, 0);
            c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
          } else {
            acc[j >> 1][a][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][s], fa[s], acc[j >> 1][a][j & 1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- fillers in this MFMA's shadow
          // operand refills, one k-step behind the MFMAs that read the registers: A behind the first MFMA of the
          // k-step; the B operand (last phase) one fragment behind each MFMA
          const int rs = s == 0 ? 3 : s - 1;                       // k-step whose registers are refilled now
          if (j == 0) {
            if (s == 0) read_a(so, a, 3);
            else if (a < 3) read_a(so, a + 1, rs);
            else if (!SKIP) read_a(sn, 0, rs);
          }
          if (s == 0 && a == 0) read_b1(so, j, 3);
          else if (s > 0 && a == 3 && !SKIP) read_b1(sn, j, rs);
          if (DBG == 50 && ph.n == 4) {
            // (probe, correct results) the four waves issue a piece's instruction behind FOUR different MFMAs (wave w behind
            // MFMA 4 k + w) instead of all behind the same one: do they queue behind each other in the texture unit?
            const int k = s;   // 4 * s + j in [4 k, 4 k + 4)
            if (wid == j) issue1(ph.in[k].kind, ph.in[k].piece, ph.in[k].d, (unsigned)((it + ph.in[k].d) & 1) * kQStage);
          } else if (DBG != 1 && DBG != 13 && DBG != 8) {
#pragma unroll
            for (int k = 0; k < ph.n; ++k)
              if (ph.in[k].pos == 4 * s + j)
                issue1(ph.in[k].kind, ph.in[k].piece, ph.in[k].d, (unsigned)((it + ph.in[k].d) & 1) * kQStage);
          }
          if constexpr (kPipe && (LAST || PEND)) {
            // pipelined epilogue: behind the odd MFMAs.  LAST phase 0: the tile's bias quads; LAST phases 1..3: row
            // fragments 0..2; FIRST phase 0 (PEND): row fragment 3 of the tile before
            const int n = 4 * s + j;
            if (n & 1) {
              if (LAST && a == 0) { epi_bias(n - 1); epi_bias(n); }
              else if (LAST || a == 0) epi_unit((a + 3) & 3, n >> 1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // pieces the next slot reads first have landed for this wave (counted: younger ones stay in flight)
      if (DBG != 4) {
        constexpr int kind = !kPipe ? 0 : LAST ? 1 : (FIRST && PEND) ? 2 : 0;
        constexpr int c0 = q_confirm_pipe(VAR, kind, 0), c1 = q_confirm_pipe(VAR, kind, 1), c2 = q_confirm_pipe(VAR, kind, 2),
                      c3 = q_confirm_pipe(VAR, kind, 3);
        if (a == 0) LLA_Q4_WAIT_VM(c0);
        else if (a == 1) LLA_Q4_WAIT_VM(c1);
        else if (a == 2) LLA_Q4_WAIT_VM(c2);
        else LLA_Q4_WAIT_VM(c3);
      }
      asm volatile("" ::: "memory");
    });
    ++it;
    advance_cursor();
  };
  // two-part DMA issue for the 16-cycle shadows of the small MFMA: address arithmetic behind one MFMA, the
  // instruction behind the next
  unsigned dma_dst[8];
  auto ktile16 = [&](auto first_c, auto last_c) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    unsigned so = (unsigned)(it & 1) * kQStage, sn = (unsigned)((it + 1) & 1) * kQStage;
    auto phase = [&](auto a_c) {
      constexpr int a = decltype(a_c)::value;
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+s"(so), "+s"(sn));
      q_static_for<32>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr QPhase ph = q_phase(VAR, a);
        constexpr int g = n >> 3, b = n & 7, s = g >> 1, af = g & 1;
        if constexpr (FIRST && s == 0) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          acc16[2 * a + af][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gb[b][0], ga[af][0], zero4, 0, 0, 0);
        } else {
          acc16[2 * a + af][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gb[b][s], ga[af][s], acc16[2 * a + af][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // A ring: the registers of group g - 1 are refilled behind the first MFMA of group g
        if constexpr (b == 0) {
          constexpr int pg = (g + 3) & 3, ps = pg >> 1, paf = pg & 1;
          if constexpr (g == 0) read_a16(so, 2 * a + paf, paf, ps);                 // this phase's last group
          else if constexpr (a < 3) read_a16(so, 2 * (a + 1) + paf, paf, ps);
          else if constexpr (!LAST) read_a16(sn, paf, paf, ps);
        }
        // B operand: k-step 1 of this K-tile behind the odd MFMAs of groups 0, 1 of phase 0; k-step 0 of the next
        // K-tile behind those of groups 2, 3 of phase 3
        if constexpr ((n & 1) && a == 0 && n < 16) read_b16(so, n >> 1, 1);
        if constexpr ((n & 1) && a == 3 && n >= 16 && !LAST) read_b16(sn, (n - 16) >> 1, 0);
        if constexpr (DBG != 41) {
          // DMA instruction k of the phase: address arithmetic behind MFMA `at`, the instruction behind `at + 2`
          q_static_for<ph.n>([&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            constexpr int at = DBG == 44 ? 2 + 2 * k : 8 * (k % 4) + 2 + 2 * (k / 4);
            if constexpr (DBG == 45 && (k & 1)) return;
            constexpr QInstr in = ph.in[k];
            if constexpr (n == at) {
              const unsigned st_off = (unsigned)((it + in.d) & 1) * kQStage;
              dma_dst[k] = __builtin_amdgcn_readfirstlane(st_off);
            }
            if constexpr (n == at + 2) issue1(in.kind, in.piece, in.d, dma_dst[k]);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      constexpr int cf = q_confirm(VAR, a);
      if constexpr (DBG != 42 && DBG != 45) LLA_Q4_WAIT_VM(cf);
      asm volatile("" ::: "memory");
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    ++it;
    advance_cursor();
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  for (int cj = 0; cj < n_my; ++cj) {
    if constexpr (M16) {
      ktile16(T_{}, F_{});
      for (int kt = 1; kt < nk - 1; ++kt) ktile16(F_{}, F_{});
      ktile16(F_{}, T_{});
    } else if constexpr (kPipe) {
      if (cj == 0) ktile(T_{}, F_{}, F_{});
      else ktile(T_{}, F_{}, T_{});                 // ... with row fragment 3 of the tile before in its first phase
      for (int kt = 1; kt < nk - 1; ++kt) ktile(F_{}, F_{}, F_{});
      {
        int m0e, n0e;
        tile_origin(cj, m0e, n0e);
        unsigned ones = ~0u;
        asm volatile("" : "+s"(ones));
        const int el = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const int mw = m0e + wr * 128, nw = n0e + wc * 128;
        ecrow = reinterpret_cast<unsigned char *>(reinterpret_cast<f16 *>(p.C) + (size_t)(mw + (el & 31)) * p.ldc + nw) + 16 * (el >> 5);
        ebias_off = (unsigned)(nw + 4 * (el >> 5)) * 4u;
      }
      ktile(F_{}, T_{}, F_{});
      continue;
    } else {
      ktile(T_{}, F_{}, F_{});
      for (int kt = 1; kt < nk - 1; ++kt) ktile(F_{}, F_{}, F_{});
      ktile(F_{}, T_{}, F_{});
    }
    asm volatile("" ::: "memory");
    stamp(4);
    int m0c, n0c;
    tile_origin(cj, m0c, n0c);
    // the lane id is re-derived here (v_mbcnt on a mask the compiler cannot fold) rather than kept live across the
    // K loop: a spilled copy would be reloaded with `s_waitcnt vmcnt(0)`, i.e. by draining the LDS-DMA ring
    unsigned ones = ~0u;
    asm volatile("" : "+s"(ones));
    const int el = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    const int mw = m0c + wr * 128, nw = n0c + wc * 128;
    if constexpr (M16) {
      float t = 0.f;
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int b = 0; b < 8; ++b) t += acc16[f][b][0] + acc16[f][b][1] + acc16[f][b][2] + acc16[f][b][3];
      if (t == 1.2345e30f) reinterpret_cast<f16 *>(p.C)[el] = (f16)t;
    } else if (DBG == 3 || DBG == 13 || DBG == 8 || DBG == 9) {
      float t = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) t += acc[h][i][0][e] + acc[h][i][1][e];
      if (t == 1.2345e30f) reinterpret_cast<f16 *>(p.C)[el] = (f16)t;
    } else if constexpr ((epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) && DBG == 30) {
      // (probe) LDS-staged fp16 epilogue: whole 128-byte lines per row and store instruction
      gemm_epilogue_staged<EPI, 4>(p, acc[0], mw, nw, el, smem + 2 * kQStage + wid * 2048);
      gemm_epilogue_staged<EPI, 4>(p, acc[1], mw, nw + 64, el, smem + 2 * kQStage + wid * 2048);
    } else if constexpr (epi_base(EPI) == EPI_F16 || epi_base(EPI) == EPI_QGELU) {
      q4_epilogue_f16<EPI>(p, acc, mw, nw, el, smem + kBiasOff);
    } else if constexpr (EPI == EPI_RESID_LNX) {
      // (row-major tile order, group_m == 1: the row tile's three column tiles are three consecutive logical tiles;
      // this workgroup's round cj covers the logical tiles start + cj nslots .. + nslots - 1 of its XCD's range)
      const int mine = start + slot + cj * nslots;
      const int ct_l = (p.rev ? total - 1 - mine : mine) % 3;            // column tile = position in the triple
      const int first = p.rev ? mine - (2 - ct_l) : mine - ct_l;         // the triple's first logical tile in walk order
      const int r_lo = start + cj * nslots, r_hi = r_lo + nslots < start + count ? r_lo + nslots : start + count;
      q4_epilogue_resid_lnx(p, acc, m0c, n0c, wr, wc, el, (int)threadIdx.x, smem + 2 * kQStage + wid * 2048, smem + kBiasOff,
                            LLA_LNX_NO_SKIP || lnx_triples || (first >= r_lo && first + 2 < r_hi));
    } else if constexpr (DBG == 31) {   // (probe, wrong results: residual rows not read)
      gemm_epilogue_staged<EPI, 4, true>(p, acc[0], mw, nw, el, smem + 2 * kQStage + wid * 2048);
      gemm_epilogue_staged<EPI, 4, true>(p, acc[1], mw, nw + 64, el, smem + 2 * kQStage + wid * 2048);
    } else {
      // (tried in round 4: the same epilogue with the residual rows requested EIGHT 16-row units ahead -- a ring of 128
      // VGPRs, 32 KiB per wave in flight instead of 4-8 -- on the theory that 4.3 TB/s in the residual GEMMs is a
      // latency bound: out-proj 395 vs 382 us, FC2 997 vs 991: it is not; removed)
      gemm_epilogue_staged<EPI, 4>(p, acc[0], mw, nw, el, smem + 2 * kQStage + wid * 2048);
      gemm_epilogue_staged<EPI, 4>(p, acc[1], mw, nw + 64, el, smem + 2 * kQStage + wid * 2048);
    }
    asm volatile("" ::: "memory");
    stamp(5);
    {
      // first K-tile of the next output tile (confirmed before the last barrier): B operand and first A fragment,
      // k-steps 0..2.  Unconditional: after the last tile it reads bytes nobody uses.
      const unsigned so = (unsigned)(it & 1) * kQStage;
      if constexpr (M16) {
#pragma unroll
        for (int b = 0; b < 8; ++b) read_b16(so, b, 0);
        read_a16(so, 0, 0, 0); read_a16(so, 1, 1, 0); read_a16(so, 0, 0, 1);
      } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) { read_b(so, s); read_a(so, 0, s); }
      }
    }
  }
  if constexpr (kPipe) {   // row fragment 3 of the workgroup's last tile
#pragma unroll
    for (int u = 0; u < 8; ++u) epi_unit(3, u, false);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): trailing (unused) DMA pieces must land before the LDS is released
#if LLA_LNX_FENCE & 1
  if constexpr (EPI == EPI_RESID_LNX) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#endif
  kernel_release();
}


}  // namespace
}  // namespace lla
