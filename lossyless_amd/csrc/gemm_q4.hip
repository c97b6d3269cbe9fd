// Four-wave GEMM (gemm_q4_kernel.h): the product's launcher.  In the product the kernel takes the tower's RESIDUAL layers
// (out-proj, c_proj; M % 256 == 0): x += A W^T + b (EPI_RESID) and the same with the LayerNorm that follows applied in the
// epilogue (EPI_RESID_LNX); the fp16-output layers run on the eight-wave kernel (gemm_w8.hip).  One instantiation per epilogue:
// DMA schedule 1, no switch.  (The tools/ builds compile ablation/gemm_q4_select.hip instead.)
#include "gemm_q4_kernel.h"

namespace lla {
namespace {

template <int EPI>
int launch_q4_epi(const GemmParams &p_in, hipStream_t st) {
  GemmParams p = p_in;
  // (conv_h is unused by A_PLAIN GEMMs: the tile-group height rides there; 0 = the kernel's default.  EPI_RESID_LNX: the three
  // column tiles of a row tile are consecutive logical tiles, so that they run in the same round of the persistent grid on
  // three neighbouring workgroups of one XCD and find each other's partial sums in time)
  p.conv_h = EPI == EPI_RESID_LNX ? 1 : 0;
  const int cus = num_cus();
  const int total = (p.M / 256) * (p.N / 256);
  int grid = total < cus ? total : cus;
  // balanced persistent grid: only as many workgroups as the round count needs, a multiple of the 8 XCDs
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
  gemm_q4_kernel<EPI, 1><<<grid, 256, 0, st>>>(p);
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
    case EPI_RESID: return launch_q4_epi<EPI_RESID>(p, st);
    case EPI_RESID_LNX:
      if (p.N != kWidth || p.ldc != kWidth || !p.lnx_g || !p.lnx_b || !p.lnx_h || !p.lnx_part || !p.lnx_flag || !p.lnx_done)
        return LLA_EINVAL;
      return launch_q4_epi<EPI_RESID_LNX>(p, st);
    default: return LLA_EINVAL;
  }
}

}  // namespace lla
