// The tower's GEMM launcher of the tools/ builds (make ablation / make probes): the product's selection (../gemm_pp.hip)
// plus every A/B switch of rounds 1-5 (LLA_GEMM_TILE, LLA_GEMM_PP, LLA_GEMM_Q4, LLA_GEMM_W8, LLA_GEMM_EPILOGUE, LLA_GEMM_TALL,
// LLA_GEMM_BALANCED, LLA_GEMM_PERSIST, LLA_GEMM_KB, LLA_RN_PERSIST ...) and, under -DLLA_PROBES, the timing ablations and the
// retired kernels.  Compiled INSTEAD of ../gemm_pp.hip (same lla::launch_gemm symbol); never part of liblossyless_amd.so.
// With no switch set it must select what the product selects: tests/test_gpu_variants.py::ablation_build_defaults.
// Moved out of vit.hip verbatim in round 6.
#include "../gemm_kernels.h"
#include "../gemm_launch.h"
#include "ablation.h"
#ifdef LLA_PROBES
#include "gemm_retired.h"
#endif

namespace lla {
namespace {

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
    if (w8 && (big_enough || w8 == 2) && p.n_store == p.N) {
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
    if (q4 && big_enough && p.ldc == p.N) {
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

}  // namespace

LLA_DEFINE_LAUNCH_GEMM

}  // namespace lla
