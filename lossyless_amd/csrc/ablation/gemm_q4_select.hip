// launch_q4 of the tools/ builds (make ablation / make probes): the product's launcher (../gemm_q4.hip) plus the A/B switches
// (LLA_Q4_SCHED, LLA_Q4_PIPE, LLA_Q4_GROUP_M, the fp16-output instantiations behind LLA_GEMM_W8=0) and, under -DLLA_PROBES, the
// timing ablations (LLA_Q4_DBG, LLA_Q4_TRACE) and the LayerNorm-folding epilogues of round 3.  Compiled INSTEAD of
// ../gemm_q4.hip; verbatim from it as of round 5.
#include "../gemm_q4_kernel.h"
#include "ablation.h"

namespace lla {
namespace {

template <int EPI>
int launch_q4_epi(const GemmParams &p_in, hipStream_t st) {
  GemmParams p = p_in;
  static const int gm = [] { const char *e = lla_getenv("LLA_Q4_GROUP_M"); return e ? std::atoi(e) : 0; }();
  p.conv_h = gm;
  // (EPI_RESID_LNX: the three column tiles of a row tile are consecutive logical tiles, so that they run in the same
  // round of the persistent grid on three neighbouring workgroups of one XCD and find each other's partial sums in time)
  if (EPI == EPI_RESID_LNX) p.conv_h = 1;
  const int cus = num_cus();
  const int total = (p.M / 256) * (p.N / 256);
  int grid = total < cus ? total : cus;
  // balanced persistent grid (as launch_pp): only as many workgroups as the round count needs, a multiple of the 8 XCDs
  if (total > cus) {
    const int rounds = (total + cus - 1) / cus;
    const int need = ((total + rounds - 1) / rounds + 7) & ~7;
    if (need < grid) grid = need;
  }
#if LLA_LNX_TRIPLES
  // (the triple walk of EPI_RESID_LNX is laid out for 8 x 32 workgroups: same number of rounds as the balanced grid --
  // ceil(row tiles / 85) against ceil(3 row tiles / 256) --, no row tile split over two rounds)
  if (EPI == EPI_RESID_LNX && cus == 256 && total > cus) grid = 256;
#endif
  // LLA_Q4_SCHED: DMA schedule (q_sched): 1 = four instructions per phase (default; 905-909 TFLOP/s per layer at M = 217 600
  // against 903-906 for 0 and 2, same box)
  static const int var = [] { const char *e = lla_getenv("LLA_Q4_SCHED"); return e ? std::atoi(e) : 1; }();
#if defined(LLA_PROBES) || defined(LLA_Q4_PROBE)
  static const int dbg = [] { const char *e = lla_getenv("LLA_Q4_DBG"); return e ? std::atoi(e) : 0; }();
  if (dbg == 1) { gemm_q4_kernel<EPI, 1, 1><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 2) { gemm_q4_kernel<EPI, 1, 2><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 3) { gemm_q4_kernel<EPI, 1, 3><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 13) { gemm_q4_kernel<EPI, 1, 13><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 4) { gemm_q4_kernel<EPI, 1, 4><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 5) { gemm_q4_kernel<EPI, 1, 5><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 9) { gemm_q4_kernel<EPI, 1, 9><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 30) { gemm_q4_kernel<EPI, 1, 30><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if constexpr (EPI == EPI_RESID) { if (dbg == 31) { gemm_q4_kernel<EPI, 1, 31><<<grid, 256, 0, st>>>(p); return check_launch(); } }
  if (dbg == 20) {
    static unsigned long long *const tr = [] { const char *e = lla_getenv("LLA_Q4_TRACE"); return e ? reinterpret_cast<unsigned long long *>(std::strtoull(e, nullptr, 0)) : nullptr; }();
    p.trace = tr;
    gemm_q4_kernel<EPI, 1, 20><<<grid, 256, 0, st>>>(p); return check_launch();
  }
  if (dbg == 8) { gemm_q4_kernel<EPI, 1, 8><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 40) { gemm_q4_kernel<EPI, 1, 40><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 41) { gemm_q4_kernel<EPI, 1, 41><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 42) { gemm_q4_kernel<EPI, 1, 42><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 43) { gemm_q4_kernel<EPI, 1, 43><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 44) { gemm_q4_kernel<EPI, 1, 44><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 45) { gemm_q4_kernel<EPI, 1, 45><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 46) { gemm_q4_kernel<EPI, 1, 46><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 47) { gemm_q4_kernel<EPI, 1, 47><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 48) { gemm_q4_kernel<EPI, 1, 48><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 49) { gemm_q4_kernel<EPI, 1, 49><<<grid, 256, 0, st>>>(p); return check_launch(); }
  if (dbg == 50) { gemm_q4_kernel<EPI, 1, 50><<<grid, 256, 0, st>>>(p); return check_launch(); }
#endif
  // The product library holds ONE instantiation per epilogue: DMA schedule 1, the fp16 epilogues pipelined into the K
  // loop.  LLA_Q4_PIPE=0 / LLA_Q4_SCHED=0|2 (A/B; same bits: tests/test_gpu_variants.py) exist in the tools/ build only.
#ifdef LLA_ABLATION
  static const int pipe = [] { const char *e = lla_getenv("LLA_Q4_PIPE"); return e ? std::atoi(e) : 1; }();
  if constexpr (EPI == EPI_F16 || EPI == EPI_QGELU) {
    if (pipe && var == 1) { gemm_q4_kernel<EPI, 1, 0, 1><<<grid, 256, 0, st>>>(p); return check_launch(); }
  }
  if (var == 0) gemm_q4_kernel<EPI, 0><<<grid, 256, 0, st>>>(p);
  else if (var == 2) gemm_q4_kernel<EPI, 2><<<grid, 256, 0, st>>>(p);
  else gemm_q4_kernel<EPI, 1><<<grid, 256, 0, st>>>(p);
#else
  (void)var;
  if constexpr (EPI == EPI_F16 || EPI == EPI_QGELU) gemm_q4_kernel<EPI, 1, 0, 1><<<grid, 256, 0, st>>>(p);
  else gemm_q4_kernel<EPI, 1><<<grid, 256, 0, st>>>(p);
#endif
  return check_launch();
}

}  // namespace

int launch_q4(int epi, const GemmParams &p, hipStream_t st) {
  if (p.M <= 0 || (p.M & 255) || (p.N & 255) || p.N > 3072 || (p.K & 63) || p.K < 256 || p.lda < p.K || (p.lda & 7)) return LLA_EINVAL;
  // 32-bit byte offsets inside a tile's operand panel
  if ((size_t)256 * (size_t)p.lda * 2 >= (1ull << 31) || (size_t)256 * (size_t)p.K * 2 >= (1ull << 31)) return LLA_EINVAL;
#if LLA_Q4_BUFDMA
  // ... and 32-bit byte offsets of the panels from the operands' bases (the buffer descriptor's scalar offset)
  if ((size_t)p.M * (size_t)p.lda * 2 >= (1ull << 32) || (size_t)p.N * (size_t)p.K * 2 >= (1ull << 32)) return LLA_EINVAL;
#endif
  switch (epi) {
#if defined(LLA_ABLATION) || !LLA_W8_DEFAULT
    // (round 6: in the product the large fp16-output GEMMs run on gemm_w8.hip -- launch_gemm asks it first and it takes every
    // shape this kernel takes -- so the four-wave kernel's fp16 instantiations exist in the tools/ build only:
    // LLA_GEMM_W8=0, tests/test_gpu_variants.py)
    case EPI_F16: return launch_q4_epi<EPI_F16>(p, st);
    case EPI_QGELU: return launch_q4_epi<EPI_QGELU>(p, st);
#endif
    case EPI_RESID: return launch_q4_epi<EPI_RESID>(p, st);
    case EPI_RESID_LNX:
      if (p.N != kWidth || p.ldc != kWidth || !p.lnx_g || !p.lnx_b || !p.lnx_h || !p.lnx_part || !p.lnx_flag || !p.lnx_done)
        return LLA_EINVAL;
      return launch_q4_epi<EPI_RESID_LNX>(p, st);
    default: return LLA_EINVAL;
  }
}

}  // namespace lla

