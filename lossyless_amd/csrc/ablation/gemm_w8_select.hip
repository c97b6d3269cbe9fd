// launch_w8 of the tools/ builds (make ablation / make probes / make w8variant): the product's launcher (../gemm_w8.hip) plus
// the timing ablations (LLA_W8_DBG) and the pipelined epilogue (LLA_W8_PIPE=1) of tools/w8_probe.py.  Compiled INSTEAD of
// ../gemm_w8.hip.
#include "../gemm_w8_kernel.h"
#include "ablation.h"

namespace lla {
namespace {

template <int EPI>
int launch_w8_epi(const GemmParams &p, hipStream_t st) {
  const int cus = num_cus();
  const int total = ((p.M + 255) / 256) * (p.N / 256);
  int grid = total < cus ? total : cus;
  if (total > cus) {   // balanced persistent grid: only as many workgroups as the round count needs, a multiple of the 8 XCDs
    const int rounds = (total + cus - 1) / cus;
    const int need = ((total + rounds - 1) / rounds + 7) & ~7;
    if (need < grid) grid = need;
  }
#ifdef LLA_ABLATION   // (tools/w8_probe.py)
  static const int dbg = [] { const char *e = lla_getenv("LLA_W8_DBG"); return e ? std::atoi(e) : 0; }();
  if (dbg == 1) { gemm_w8_kernel<EPI, 1><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 2) { gemm_w8_kernel<EPI, 2><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 3) { gemm_w8_kernel<EPI, 3><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 4) { gemm_w8_kernel<EPI, 4><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 5) { gemm_w8_kernel<EPI, 5><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 13) { gemm_w8_kernel<EPI, 13><<<grid, 512, 0, st>>>(p); return check_launch(); }
  if (dbg == 15) { gemm_w8_kernel<EPI, 15><<<grid, 512, 0, st>>>(p); return check_launch(); }
  static const int pipe = [] { const char *e = lla_getenv("LLA_W8_PIPE"); return e ? std::atoi(e) : 0; }();
  if (pipe) { gemm_w8_kernel<EPI, 0, 1><<<grid, 512, 0, st>>>(p); return check_launch(); }
#endif
  gemm_w8_kernel<EPI><<<grid, 512, 0, st>>>(p);
  return check_launch();
}

}  // namespace

int launch_w8(int epi, const GemmParams &p, hipStream_t st) {
  if (p.M <= 0 || (p.N & 255) || p.N > 3072 || (p.K & 63) || p.K < 128 || p.lda < p.K || (p.lda & 7) || p.ldc < p.N || (p.ldc & 7)) return LLA_EINVAL;
  // 32-bit byte offsets: inside a tile's operand panel, and of the panels from the operands' bases (the descriptor's scalar offset)
  if ((size_t)256 * (size_t)p.lda * 2 >= (1ull << 31) || (size_t)256 * (size_t)p.K * 2 >= (1ull << 31)) return LLA_EINVAL;
  if ((size_t)p.M * (size_t)p.lda * 2 >= (1ull << 32) || (size_t)p.N * (size_t)p.K * 2 >= (1ull << 32)) return LLA_EINVAL;
  switch (epi) {
    case EPI_F16: return launch_w8_epi<EPI_F16>(p, st);
    case EPI_QGELU: return launch_w8_epi<EPI_QGELU>(p, st);
    default: return LLA_EINVAL;
  }
}

}  // namespace lla

