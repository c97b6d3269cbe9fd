// Eight-wave GEMM for the tower's fp16-output layers (gfx950): C[M][N] = epi(A[M][K] W[N][K]^T + b), round 6.
//
// Stands in for in_proj (QKV) and c_fc (+ QuickGELU) inside `z = self.clip(X)` (hub/compressor.py:93; clip==1.0
// VisionTransformer / ResidualAttentionBlock), like gemm_q4_kernel, on the layout VERDICT r5 #1 asked for:
//
//   * 256 x 256 x 64 tiles on EIGHT waves (2 x 4), TWO waves per SIMD, each a 128 x 64 output tile = 8 x 4 accumulator
//     tiles of v_mfma_f32_16x16x32_f16 (128 AccVGPRs; the whole wave fits 256 registers).  The four-wave kernel is one
//     wave per SIMD: whenever its only wave sits in the issue of an LDS-DMA instruction (23-85 cycles, docs/history/DESIGN_rounds_1-5.md 5.6) or
//     a barrier, the SIMD's matrix pipe idles -- a 32-cycle MFMA hides part of that, the 16-cycle MFMA of the small
//     shape (which holds 14 % more clock at equal utilisation: profiles/r04_q4_mfma16_probe.txt) hides half as much,
//     which is why the small shape lost on four waves.  Here the partner wave of the SIMD issues meanwhile.
//   * The same operand stream as gemm_q4_kernel: two 64-KiB LDS stages (A 256 x 128 B | B 256 x 128 B, 16-byte chunks
//     XOR-swizzled by row pair on the SOURCE address, destination lane-linear), whole 128-byte lines per row and
//     LDS-DMA instruction (buffer_load_dwordx4 ... lds through a raw buffer descriptor), FRAGMENT-MAJOR K-tiles: four
//     phases per K-tile, phase a = A row fragments 2a, 2a+1 of the wave x its 4 B fragments x 2 k-steps of 32 = 16
//     MFMAs; the B operand of a K-tile lives in 32 VGPRs, the A fragments in a ring of four register quads refilled
//     one group (4 MFMAs) behind the MFMAs that read them.  The LDS frees progressively (A pair a after phase a, the B
//     region after phase 0) and is refilled at once with K-tile t + 2 (t + 1 for pairs 2, 3): every piece is issued
//     >= 4 phases (one K-tile time) before its first read.  One s_barrier per phase; the pieces the next phase reads
//     first are confirmed by a counted vmcnt before it (w_confirm: computed from the schedule at compile time).
//   * 8 LDS-DMA instructions per wave and K-tile (1 KiB each: 8 rows x 128 B), two per phase.
//   * fp16 epilogue in registers, between two output tiles: bias (from an LDS copy) / QuickGELU / conversion in the MFMA
//     layout (lane = row l % 16, 4 consecutive columns per accumulator tile), one v_permlane16_swap per dword pairs two
//     neighbouring column fragments so that every lane owns 8 consecutive columns: 16-byte stores, 64 contiguous bytes
//     per row and instruction.
//
// Measured (tools/w8_probe.py, M = 217 600, profiles/r06_w8_probe.txt): QKV 770 us against the four-wave kernel's 815
// (+5.5 .. +7.5 %), c_fc 1015-1030 against 1070-1080 (+4 .. +5.5 %).  What bounds it is the CU's vector-memory path, not the
// matrix pipe: the K loop takes the same time with every MFMA deleted (620 vs 632 us), 502 us without the operand stream,
// 554 with cache-hot operands; the counted waits never block (without them: same time, same bits); the epilogue's 16 stores
// per wave cost 138 us per launch whatever their place (between tiles, in the K loop's MFMA shadows, 64- or 128-byte
// row segments, `nt`, or all aimed at one L2-resident tile) while its arithmetic costs 8 us.
//
// Numerics: v_mfma_f32_16x16x32_f16 adds 32 products per instruction where the 32x32x16 shape of the other kernels adds
// 16.  docs/history/DESIGN_rounds_1-5.md 5.6 assumed the two round differently; measured, they do not: the WHOLE outputs of this kernel and of
// gemm_q4_kernel are equal bit for bit at M = 217 600 (QKV 501 M values, c_fc + QuickGELU 668 M; tools/w8_probe.py
// checksums, tests/test_gpu_variants.py) -- both shapes evidently add a k-step's products in the same order -- so this
// kernel mixes freely with the other GEMM paths and an image's embedding still does not depend on the batch it travels in.
// launch_w8 takes every M (a ragged last row tile is computed from the panel that ends at row M and stored masked).
//
// Scope: A_PLAIN operands, EPI_F16 / EPI_QGELU, N % 256 == 0, N <= 3072, K % 64 == 0, K >= 128.
// (the kernel template lives in this header; gemm_w8.hip holds the product's launcher, ablation/gemm_w8_select.hip the tools/
// builds' switchable one)
#pragma once
#include "gemm_common.h"

namespace lla {
namespace {

constexpr int kWStage = 65536, kWARegion = 32768;
constexpr int kWBiasOff = 2 * kWStage, kWBiasBytes = 3072 * 4;
#ifndef LLA_W8_GROUP_M
#define LLA_W8_GROUP_M 4
#endif
// MFMAs of a phase behind which its two LDS-DMA instructions are issued (A/B: make w8variant NAME=x DEFS="-DLLA_W8_DMA_N0=..";
// profiles/r06_w8_probe.txt: 2 / 10 level, 7 / 15 1.5 % slower)
#ifndef LLA_W8_DMA_N0
#define LLA_W8_DMA_N0 5
#endif
#ifndef LLA_W8_DMA_N1
#define LLA_W8_DMA_N1 13
#endif

// DMA schedule: two items per phase and wave.  kind 0 = A row-fragment pair idx (the 32 rows of phase idx in both wave
// rows: 64 rows x 128 B = 8 waves x 1 KiB), kind 1 = B slot idx (64 rows); d = K-tiles ahead of the one being multiplied.
// Legality (WAR): pair i of K-tile t + 2 only in phases p > i (its stage still holds K-tile t), of K-tile t + 1 anywhere;
// B slots of K-tile t + 2 in phases >= 1.  Deadlines (RAW): pair i of K-tile u is first read in phase i - 1 of K-tile u
// (pair 0: phase 3 of K-tile u - 1), the B operand of K-tile u in phase 3 of K-tile u - 1.
struct WItem { int kind, idx, d; };
constexpr WItem w_item(int p, int k) {
  return p == 0 ? WItem{0, 2 + k, 1} : p == 1 ? WItem{1, k, 2} : p == 2 ? WItem{1, 2 + k, 2} : WItem{0, k, 2};
}
// vmcnt argument at the end of phase p: instructions issued after the youngest piece the NEXT phase reads first
constexpr int w_confirm(int p) {
  int seq = 0, last_a[14][4] = {}, last_b[14] = {}, upto[14][4] = {};
  for (int t = -2; t < 10; ++t)
    for (int q = 0; q < 4; ++q) {
      for (int k = 0; k < 2; ++k) {
        const WItem it = w_item(q, k);
        const int u = t + it.d;
        ++seq;
        if (u >= 0 && u < 14) { if (it.kind == 0) last_a[u][it.idx] = seq; else last_b[u] = seq; }
      }
      if (t >= 0) upto[t][q] = seq;
    }
  const int t = 4;
  int need = 0;
  if (p == 0) need = last_a[t][2];
  if (p == 1) need = last_a[t][3];
  if (p == 2) need = last_a[t + 1][0] > last_b[t + 1] ? last_a[t + 1][0] : last_b[t + 1];
  if (p == 3) need = last_a[t + 1][1];
  return upto[t][p] - need;
}
static_assert(w_confirm(0) == 9 && w_confirm(1) == 10 && w_confirm(2) == 7 && w_confirm(3) == 8, "schedule / wait counts out of step");
// The same count when the epilogue is PIPELINED into the K loop (kPipe): the wave's vector-memory queue retires loads and
// stores in order through one counter, so a counted wait must also let the YOUNGER STORES stay in flight, or it would
// wait for operand pieces issued a phase ago instead of a K-tile ago.  The instruction stream of a window of K-tiles is
// replayed: per phase the two LDS-DMA instructions behind MFMAs n0 / n1 and, in the phases that carry an epilogue
// (1..3 of a LAST K-tile, 0 of the FIRST behind it), one store behind MFMAs 3, 7, 11, 15.  kind: 0 = no store in the
// look-back (also used for every other K-tile: fewer stores assumed than there are only makes a wait longer, never
// shorter), 1 = LAST, 2 = FIRST behind a LAST.
constexpr int w_confirm_pipe(int kind, int p, int n0, int n1) {
  const int TL = 6;                     // the LAST K-tile of the window; FIRST = TL + 1
  int seq = 0, last_a[16][4] = {}, last_b[16] = {}, upto[16][4] = {};
  for (int t = 0; t < 12; ++t)
    for (int q = 0; q < 4; ++q) {
      const bool stores = (t == TL && q >= 1) || (t == TL + 1 && q == 0);
      for (int n = 0; n < 16; ++n) {
        if (n == n0 || n == n1) {
          const WItem it = w_item(q, n == n1 ? 1 : 0);
          const int u = t + it.d;
          ++seq;
          if (it.kind == 0) last_a[u][it.idx] = seq; else last_b[u] = seq;
        }
        if (stores && (n & 3) == 3) ++seq;
      }
      upto[t][q] = seq;
    }
  const int t = kind == 1 ? TL : kind == 2 ? TL + 1 : 3;
  int need = 0;
  if (p == 0) need = last_a[t][2];
  if (p == 1) need = last_a[t][3];
  if (p == 2) need = last_a[t + 1][0] > last_b[t + 1] ? last_a[t + 1][0] : last_b[t + 1];
  if (p == 3) need = last_a[t + 1][1];
  return upto[t][p] - need;
}
static_assert(w_confirm_pipe(0, 0, 5, 13) == 9 && w_confirm_pipe(0, 1, 5, 13) == 10 && w_confirm_pipe(0, 2, 5, 13) == 7 &&
              w_confirm_pipe(0, 3, 5, 13) == 8, "the replayed stream and the schedule table must agree where there are no stores");
static_assert(w_confirm_pipe(1, 0, 5, 13) == 9 && w_confirm_pipe(1, 1, 5, 13) == 14 && w_confirm_pipe(1, 2, 5, 13) == 15 &&
              w_confirm_pipe(1, 3, 5, 13) == 20 && w_confirm_pipe(2, 0, 5, 13) == 25 && w_confirm_pipe(2, 1, 5, 13) == 26 &&
              w_confirm_pipe(2, 2, 5, 13) == 14 && w_confirm_pipe(2, 3, 5, 13) == 13, "hand count of the default placement");

template <int... I, class F>
__device__ __forceinline__ void w_static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void w_static_for(F &&f) { w_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

typedef unsigned w_rsrc_t __attribute__((ext_vector_type(4)));
// One LDS-DMA instruction: 64 lanes x 16 bytes from (descriptor base + soff + voff) to LDS (lds_base + IMM) + 16 lane
template <int IMM>
__device__ __forceinline__ void w_dma(unsigned voff, w_rsrc_t rsrc, unsigned soff, unsigned lds_base) {
  asm volatile("s_add_u32 m0, %3, %4\n\t"
               "s_nop 0\n\t"
               "buffer_load_dwordx4 %0, %1, %2 offen" LLA_DMA_SC " lds"
               :
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "n"(IMM)
               : "memory", "m0", "scc");
}
__device__ __forceinline__ void w_dma_flat(unsigned voff, const unsigned char *sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory", "m0");
}

#define LLA_W8_WAIT_VM(C) __builtin_amdgcn_s_waitcnt(0x0F70 | ((C) & 15) | (((C) >> 4) << 14))   // vmcnt(C); expcnt / lgkmcnt open

// DBG (tools/ build, timing only, WRONG results; tools/w8_probe.py, profiles/r06_w8_probe.txt): 1 = no LDS-DMA after the
// prologue, 2 = no s_barrier, 3 = no epilogue, 4 = no counted waits (still the right bits: the data is always there in time),
// 5 = the epilogue's arithmetic without its stores, 13 = 1 + 3, 15 = 3 with every piece re-reading the workgroup's first
// K-tile (operands cache-hot).
//
// PIPE (tools/ build, LLA_W8_PIPE=1; same bits; NOT the product: 0.5-1.5 % slower than the serial epilogue, see the file
// header of profiles/r06_w8_probe.txt -- the stores, not the arithmetic, are the epilogue's cost, and they cost the same
// wherever they are issued): the epilogue is software-pipelined into the K loop (as gemm_q4's): fragment-major K-tiles finish the accumulators
// of row-fragment pair a at the end of phase a of an output tile's LAST K-tile and nothing writes them again before phase
// a of the next tile's FIRST K-tile, so pair a is biased / activated / converted / stored in the MFMA shadows of the phase
// after its last one (pair 3 in phase 0 of the next tile -- PEND --, after the workgroup's last tile by itself): four
// units (row fragment x column-fragment pair) of 8 adds (+ QuickGELU) + 4 v_cvt_pk_f16_f32 + 2 v_permlane16_swap + one
// 16-byte store behind MFMAs 3, 7, 11, 15 of the phase; the tile's 16 bias registers are read from LDS in phase 0 of the
// LAST K-tile.  Same arithmetic on the same values, same store addresses as the serial epilogue: bit-identical.
template <int EPI, int DBG = 0, int PIPE = 0>
__global__ __launch_bounds__(512) void gemm_w8_kernel(GemmParams p) {
  static_assert(EPI == EPI_F16 || EPI == EPI_QGELU, "fp16 outputs only");
  constexpr bool kPipe = PIPE != 0 && DBG == 0;
  kernel_acquire();
  __shared__ __attribute__((aligned(16))) unsigned char smem[kWBiasOff + kWBiasBytes];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- my tiles: XCD-contiguous logical range, group_m row tiles per group swept over all column tiles (as gemm_q4)
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  const int total = tiles_m * tiles_n;
  const int bid = blockIdx.x, nblk = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslots = (nblk - xcd + 7) >> 3;
  const int tq = total >> 3, tr = total & 7;
  const int start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int count = tq + (xcd < tr ? 1 : 0);
  const int n_my = slot < count ? (count - slot + nslots - 1) / nslots : 0;
  if (n_my == 0) return;
  constexpr int group_m = LLA_W8_GROUP_M;
  auto tile_origin = [&](int j, int &m0, int &n0) {
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

  // ---- operand stream.  Wave w, lane l -> row l / 8 of its 8-row group, 16-byte position l % 8 holding source chunk
  // (l % 8) ^ swizzle(row): A pair a = rows (w / 4) 128 + 32 a + (w % 4) 8 + l / 8, B slot q = rows 64 q + 8 w + l / 8;
  // swizzle(row) = (row / 2) % 8 = (w % 2) 4 + l / 16 for both.
  const int pc = lane & 7, lc = pc ^ (((wid & 1) << 2) | (lane >> 4));
  const int rowA = (wid >> 2) * 128 + (wid & 3) * 8 + (lane >> 3);
  const int rowB = wid * 8 + (lane >> 3);
  unsigned voffA[4], voffB[4];               // this lane's byte offset inside a tile's K-tile panel, per pair / slot
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    voffA[a] = (unsigned)((rowA + 32 * a) * p.lda + lc * 8) * 2u;
    voffB[a] = (unsigned)((rowB + 64 * a) * p.K + lc * 8) * 2u;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned dstA = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((wid >> 2) * 16384 + (wid & 3) * 1024));
  const unsigned dstB = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(kWARegion + wid * 1024));
  auto make_rsrc = [](const void *base) {    // raw buffer over the whole address range (no bounds: the panels are inside)
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    w_rsrc_t r = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a),
                  (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu, 0xFFFFFFFFu, 0x00020000u};
    return r;
  };
  const w_rsrc_t rsrcA = make_rsrc(p.A), rsrcB = make_rsrc(p.W);
  unsigned srcA[2], srcB[2];                 // [d - 1]: K-tile t + d: byte offset of its panel from p.A / p.W (wave-uniform)
  int cur_j = 0, cur_kt = 0;                 // position of the K-tile t + 2 cursor
  // rows beyond M (the last row tile of a ragged M): the panel origin is pulled back so that its 256 rows end at row M
  // -- the tile then recomputes rows of the tile before it, which the store side masks (M >= 256), or, for M < 256,
  // the lanes' row offsets are clamped to row M - 1 (mclamp below)
  auto src_of = [&](int j, int kt, unsigned &a, unsigned &b) {
    int m0, n0;
    tile_origin(j < n_my ? j : n_my - 1, m0, n0);
    if (m0 + 256 > p.M) m0 = p.M >= 256 ? p.M - 256 : 0;
    a = __builtin_amdgcn_readfirstlane((unsigned)(((size_t)m0 * p.lda + (size_t)kt * 64) * 2));   // (< 2^32: launch_w8 checks)
    b = __builtin_amdgcn_readfirstlane((unsigned)(((size_t)n0 * p.K + (size_t)kt * 64) * 2));
  };
  auto advance_cursor = [&] {   // K-tile t + 2 becomes t + 1; the cursor moves one K-tile on
    if (DBG == 15) return;      // (timing probe: every piece re-reads the workgroup's first K-tile: operands cache-hot)
    srcA[0] = srcA[1]; srcB[0] = srcB[1];
    if (++cur_kt < nk) { srcA[1] += 128; srcB[1] += 128; }
    else { cur_kt = 0; ++cur_j; src_of(cur_j, 0, srcA[1], srcB[1]); }
  };
  if (p.M < 256) {   // (tiny problems: rows >= M read row M - 1)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = rowA + 32 * a;
      voffA[a] = (unsigned)((r < p.M ? r : p.M - 1) * p.lda + lc * 8) * 2u;
    }
  }
  auto issue = [&](auto kind_c, auto idx_c, int d, unsigned st_off) {
    constexpr int kind = decltype(kind_c)::value, idx = decltype(idx_c)::value;
    if constexpr (kind == 0) w_dma<idx * 4096>(voffA[idx], rsrcA, srcA[d - 1], dstA + st_off);
    else w_dma<idx * 8192>(voffB[idx], rsrcB, srcB[d - 1], dstB + st_off);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  // ---- fragment reads: lane (r16, kq) holds halves 8 kq .. 8 kq + 7 of k-step s: logical chunk 4 s + kq of its row
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const unsigned c = (unsigned)(((4 * s + kq) ^ ((r16 >> 1) & 7)) * 16);
    a_off[s] = (unsigned)((wr * 128 + r16) * 128) + c;
    b_off[s] = (unsigned)kWARegion + (unsigned)((wc * 64 + r16) * 128) + c;
  }
  f16x8 gb[4][2], ga[2][2];   // gb[b][s]: B fragment b, k-step s (whole K-tile); ga[af][s]: ring, group g = 2 s + af
  f32x4 acc[8][4];
  auto read_a = [&](unsigned so, int frag, int af, int s) {
    ga[af][s] = *reinterpret_cast<const f16x8 *>(smem + so + a_off[s] + frag * 2048);
  };
  auto read_b = [&](unsigned so, int b, int s) {
    gb[b][s] = *reinterpret_cast<const f16x8 *>(smem + so + b_off[s] + b * 2048);
  };

  // ---- the bias vector [N] goes to LDS once: 1-KiB pieces round robin over the waves (the oldest DMA instructions
  // of every wave: the prologue's counted wait covers them)
  if (p.bias) {
    for (int q = wid; q * 256 < p.N; q += 8)
      w_dma_flat((unsigned)lane * 16u, reinterpret_cast<const unsigned char *>(p.bias) + (size_t)q * 1024,
                 __builtin_amdgcn_readfirstlane(lds_base + (unsigned)kWBiasOff + (unsigned)q * 1024u));
  } else {
    for (int i = tid; i * 16 < p.N * 4; i += 512) *reinterpret_cast<f32x4 *>(smem + kWBiasOff + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- prologue: K-tile 0 completely; of K-tile 1 what the phases of "K-tile -1" would have issued (the d = 2 items, in
  // phase order: the counted waits assume that order); then the cursors are where phase (0, 0) expects them
  src_of(0, 0, srcA[1], srcB[1]);
  srcA[0] = srcA[1]; srcB[0] = srcB[1];
  issue(I0{}, I0{}, 2, 0u); issue(I0{}, I1{}, 2, 0u); issue(I0{}, I2{}, 2, 0u); issue(I0{}, I3{}, 2, 0u);
  issue(I1{}, I0{}, 2, 0u); issue(I1{}, I1{}, 2, 0u); issue(I1{}, I2{}, 2, 0u); issue(I1{}, I3{}, 2, 0u);
  advance_cursor();            // srcX[1] = K-tile 1
  issue(I1{}, I0{}, 2, (unsigned)kWStage); issue(I1{}, I1{}, 2, (unsigned)kWStage);
  issue(I1{}, I2{}, 2, (unsigned)kWStage); issue(I1{}, I3{}, 2, (unsigned)kWStage);
  issue(I0{}, I0{}, 2, (unsigned)kWStage); issue(I0{}, I1{}, 2, (unsigned)kWStage);
  advance_cursor();            // srcX[0] = K-tile 1 (pairs 2, 3 still to come), srcX[1] = K-tile 2
  LLA_W8_WAIT_VM(6);           // K-tile 0 has landed (only K-tile 1's six pieces may be in flight)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int b = 0; b < 4; ++b) read_b(0u, b, 0);
  read_a(0u, 0, 0, 0); read_a(0u, 1, 1, 0); read_a(0u, 0, 0, 1);

  int it = 0;            // global K-tile counter: selects the LDS stage

  // ---- epilogue of one output tile.  A ragged last row tile was computed from the panel that ENDS at row M (src_of):
  // its rows are mbase .. M - 1 with mbase = M - 256 <= m0; the rows below m0 belong to the tile before and are not
  // stored (same values anyway).  State of the tile whose row fragments are being stored:
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 ebias[4];                       // [B fragment b]: bias of columns nw + 16 b + 4 (lane / 16) .. + 3
  unsigned char *ecrow = nullptr;       // this lane's row of the wave tile (+ 16 f rows), byte address of the 8 columns it
                                        // owns after the swap: 16 (bp + (q & 1)) + 8 (q >> 1) of the pair (bp, bp + 1), q = lane / 16
  unsigned elive = 0;                   // bit f: this lane's row of row fragment f is stored
  unsigned ebias_off = 0;
  const size_t e_f_step = (size_t)16 * p.ldc * 2;
  auto epi_setup = [&](int cj) {
    int m0c, n0c;
    tile_origin(cj, m0c, n0c);
    // the lane id is re-derived here (v_mbcnt on a mask the compiler cannot fold) rather than kept live across the K loop
    unsigned ones = ~0u;
    asm volatile("" : "+s"(ones));
    const int el = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    const int er = el & 15, eq = el >> 4;
    int mbase = m0c, mskip = 0;
    if (m0c + 256 > p.M) { mbase = p.M >= 256 ? p.M - 256 : 0; mskip = m0c - mbase; }
    const int nw = n0c + wc * 64;
    // store side: after the swap lane (er, eq) owns columns 16 (bp + (eq & 1)) + 8 (eq >> 1) .. + 7 of the column-fragment
    // pair (bp, bp + 1) in row er: 64 contiguous bytes per row and store instruction (whole 128-byte lines per
    // instruction -- rows 8..15 handed to the lanes of rows 0..7 through row_ror:8 -- measured level to 2 % slower)
    const int rl0 = wr * 128 + er;                       // row inside the tile (+ 16 f)
    const int col = 16 * (eq & 1) + 8 * (eq >> 1);
    elive = 0;
#pragma unroll
    for (int f = 0; f < 8; ++f) elive |= (rl0 + 16 * f >= mskip && mbase + rl0 + 16 * f < p.M) ? (1u << f) : 0u;
    ecrow = reinterpret_cast<unsigned char *>(reinterpret_cast<f16 *>(p.C) + (size_t)(mbase + rl0) * p.ldc + nw + col);
    ebias_off = (unsigned)(nw + 4 * eq) * 4u;
  };
  auto epi_bias = [&](int b) {
    ebias[b] = *reinterpret_cast<const f32x4 *>(smem + kWBiasOff + ebias_off + 64 * b);
  };
  auto epi_frag = [&](int f) {   // row fragment f: 16 rows x the wave tile's 64 columns -> two 16-byte stores per lane
    u32x4 o[2];
#pragma unroll
    for (int bp = 0; bp < 4; bp += 2) {
      unsigned lo[2], hi[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 v = acc[f][bp + h];
        // (through a volatile asm: pure arithmetic may float anywhere between its operands and its use at IR level --
        // sched_barrier only binds the machine scheduler -- and a phase's worth of QuickGELUs hoisted to its top spills)
        asm volatile("" : "+v"(v));
        v += ebias[bp + h];
        if constexpr (EPI == EPI_QGELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        typedef f16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 a2 = {(f16)v[0], (f16)v[1]}, b2 = {(f16)v[2], (f16)v[3]};
        lo[h] = __builtin_bit_cast(unsigned, a2);
        hi[h] = __builtin_bit_cast(unsigned, b2);
      }
      const auto rx = __builtin_amdgcn_permlane16_swap(lo[0], lo[1], false, false);
      const auto ry = __builtin_amdgcn_permlane16_swap(hi[0], hi[1], false, false);
      o[bp >> 1] = u32x4{rx[0], ry[0], rx[1], ry[1]};
    }
    if constexpr (DBG == 5) asm volatile("" ::"v"(o[0]), "v"(o[1]));      // (probe: the arithmetic without the stores)
    else if ((elive >> f) & 1u) { store16(ecrow + f * e_f_step, o[0]); store16(ecrow + f * e_f_step + 64, o[1]); }
  };

  // FIRST: first K-tile of an output tile (C = 0 as an inline MFMA operand).  LAST / PEND: pipelined epilogue (kPipe) --
  // LAST phase 0: the tile's bias registers; LAST phases 1..3: pairs 0..2; PEND = FIRST behind a LAST: pair 3 in phase 0.
  auto ktile = [&](auto first_c, auto last_c, auto pend_c) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, PEND = decltype(pend_c)::value;
    unsigned so = (unsigned)(it & 1) * kWStage, sn = (unsigned)((it + 1) & 1) * kWStage;
    w_static_for<4>([&](auto a_c) {
      constexpr int a = decltype(a_c)::value;
      __builtin_amdgcn_sched_barrier(0);
      if (DBG != 2) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+s"(so), "+s"(sn));
      w_static_for<16>([&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr int g = n >> 2, b = n & 3, s = g >> 1, af = g & 1;
        if constexpr (FIRST && s == 0) {
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          acc[2 * a + af][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gb[b][0], ga[af][0], zero4, 0, 0, 0);
        } else {
          acc[2 * a + af][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gb[b][s], ga[af][s], acc[2 * a + af][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // A ring: the register quad of group g - 1 is refilled behind the first MFMA of group g
        if constexpr (b == 0) {
          constexpr int pg = (g + 3) & 3, ps = pg >> 1, paf = pg & 1;
          if constexpr (g == 0) read_a(so, 2 * a + paf, paf, ps);                 // this phase's last group
          else if constexpr (a < 3) read_a(so, 2 * (a + 1) + paf, paf, ps);
          else read_a(sn, paf, paf, ps);
        }
        // B operand: k-step 1 of this K-tile behind the first four MFMAs of phase 0 (its registers were last read by
        // the last two groups of the K-tile before); k-step 0 of the next K-tile behind MFMAs 8..11 of phase 3
        if constexpr (a == 0 && n < 4) read_b(so, n, 1);
        if constexpr (a == 3 && n >= 8 && n < 12) read_b(sn, n - 8, 0);
        // LDS-DMA: the phase's two instructions behind MFMAs 5 and 13 (LLA_W8_DMA_N0 / N1)
        if constexpr (DBG != 1 && DBG != 13 && (n == LLA_W8_DMA_N0 || n == LLA_W8_DMA_N1)) {
          constexpr WItem item = w_item(a, n == LLA_W8_DMA_N1 ? 1 : 0);
          issue(std::integral_constant<int, item.kind>{}, std::integral_constant<int, item.idx>{}, item.d,
                (unsigned)((it + item.d) & 1) * kWStage);
        }
        if constexpr (kPipe && (LAST || PEND)) {
          if constexpr (LAST && a == 0) { if constexpr ((n & 3) == 3) epi_bias(n >> 2); }
          else if constexpr ((n & 7) == 7) {
            if constexpr (LAST) epi_frag(2 * (a - 1) + (n >> 3));
            else if constexpr (a == 0) epi_frag(6 + (n >> 3));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // pieces the next phase reads first have landed for this wave (counted: younger ones -- pieces and, with the
      // pipelined epilogue, its stores -- stay in flight)
      if (DBG != 4) {
        constexpr int kind = !kPipe ? 0 : LAST ? 1 : PEND ? 2 : 0;
        constexpr int cf = w_confirm_pipe(kind, a, LLA_W8_DMA_N0, LLA_W8_DMA_N1);
        static_assert(cf <= 63, "vmcnt is a 6-bit field");
        LLA_W8_WAIT_VM(cf);
      }
      asm volatile("" ::: "memory");
    });
    ++it;
    advance_cursor();
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  for (int cj = 0; cj < n_my; ++cj) {
    if constexpr (kPipe) {
      if (cj == 0) ktile(T_{}, F_{}, F_{});
      else ktile(T_{}, F_{}, T_{});                 // ... with row-fragment pair 3 of the tile before in its first phase
      for (int kt = 1; kt < nk - 1; ++kt) ktile(F_{}, F_{}, F_{});
      epi_setup(cj);
      ktile(F_{}, T_{}, F_{});
      continue;
    }
    ktile(T_{}, F_{}, F_{});
    for (int kt = 1; kt < nk; ++kt) ktile(F_{}, F_{}, F_{});
    asm volatile("" ::: "memory");
    if constexpr (DBG == 3 || DBG == 13 || DBG == 15) {
      unsigned ones = ~0u;
      asm volatile("" : "+s"(ones));
      const int el = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
      float t = 0.f;
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int b = 0; b < 4; ++b) t += acc[f][b][0] + acc[f][b][1] + acc[f][b][2] + acc[f][b][3];
      if (t == 1.2345e30f) reinterpret_cast<f16 *>(p.C)[el] = (f16)t;
    } else {
      epi_setup(cj);
#pragma unroll
      for (int b = 0; b < 4; ++b) epi_bias(b);
#pragma unroll
      for (int f = 0; f < 8; ++f) epi_frag(f);
    }
    asm volatile("" ::: "memory");
  }
  if constexpr (kPipe) {   // row-fragment pair 3 of the workgroup's last tile
    epi_frag(6); epi_frag(7);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): trailing (unused) DMA pieces must land before the LDS is released
  kernel_release();
}


}  // namespace
}  // namespace lla
