// The tower's GEMM launcher, product build: which kernel takes which shape.  No switches here -- the A/B selection of rounds
// 1-5 lives in ablation/gemm_select.hip, compiled instead of this file by `make ablation` / `make probes`; with no switch
// set it selects exactly this (tests/test_gpu_variants.py::ablation_build_defaults, bit for bit).
//
// Stands in for the Linear / conv1 layers inside `z = self.clip(X)` (hub/compressor.py:93) and the 1x1 / 3x3 convolutions of
// the RN50-CLIP tower (lossyless/architectures.py:367-371):
//   M >= 9000 rows (batches of ~190+ images):
//     fp16-output layers (QKV, c_fc), N % 256 == 0, K >= 128 ........ gemm_w8_kernel   (gemm_w8.hip: eight waves, 16x16x32 MFMA)
//     residual layers (out-proj, c_proj), M % 256 == 0, K >= 256 .... gemm_q4_kernel   (gemm_q4.hip: four waves; the tower calls
//                                                                     launch_q4(EPI_RESID_LNX) itself where LayerNorm rides along)
//     everything else 256 columns wide (patch embedding, ragged M) .. gemm_pp_kernel   (two wave rows out of phase)
//     narrower outputs ............................................... gemm_persistent_kernel
//     ReLU / add + ReLU convolutions, N % 256 == 0 ................... gemm_persistent_kernel (line-assembling epilogue)
//   128 < M < 9000, and every implicit 3x3 convolution ............... gemm256_f16_kernel (one 256 x 128 tile per workgroup)
//   M <= 128 .......................................................... gemm_f16_kernel    (one 128 x 128 tile per workgroup)
// All of them accumulate K in the same order: an output does not depend on the kernel that computed it.
#include "gemm_kernels.h"
#include "gemm_launch.h"

namespace lla {
namespace {

// Persistent 256- / 320-row tiles, lock-step waves (N / (128 NJ) column tiles).  Tile height: 320 rows when that shortens the
// critical path (cost ~ rounds x rows; on a tie the taller tile wins: 10 % fewer operand bytes per flop).
template <int EPI, int AMODE, int NJ>
int launch_persistent(const GemmParams &p, hipStream_t st) {
  const int cus = num_cus();
  const int tiles_n = p.N / (128 * NJ);
  const int t256 = ((p.M + 255) / 256) * tiles_n, t320 = ((p.M + 319) / 320) * tiles_n;
  const bool tall = NJ == 2 && rounds_for(t320, cus) * 320 <= rounds_for(t256, cus) * 256;
  const int total = tall ? t320 : t256;
  const int grid = total < cus ? total : cus;
  if constexpr (NJ == 2) {
    if (tall) gemm_persistent_kernel<EPI, AMODE, 2, 64, 2, 0, 5><<<grid, 512, 0, st>>>(p);
    else gemm_persistent_kernel<EPI, AMODE, 2, 64, 2, 0, 4><<<grid, 512, 0, st>>>(p);
  } else {
    gemm_persistent_kernel<EPI, AMODE, 1, 64, 3, 0, 4><<<grid, 512, 0, st>>>(p);
  }
  return check_launch();
}

// The ping-pong kernel: persistent 256 / 320 x 256 tiles on a BALANCED grid -- the launch lasts rounds_for(total, cus) tiles
// per workgroup whatever happens, so only as many workgroups as that round count needs are started (rounded up to a
// multiple of the 8 XCDs): 51 200 rows -> 1440 / 1920 / 480 tiles = exactly 6 / 8 / 2 rounds on 240 workgroups.
template <int EPI, int AMODE>
int launch_pp(const GemmParams &p, hipStream_t st) {
  const int cus = num_cus();
  const int tiles_n = p.N / 256;
  const int t256 = ((p.M + 255) / 256) * tiles_n, t320 = ((p.M + 319) / 320) * tiles_n;
  const bool tall = !p.a_chunk_images && rounds_for(t320, cus) * 320 <= rounds_for(t256, cus) * 256;
  const int total = tall ? t320 : t256;
  int grid = total < cus ? total : cus;
  if (total > cus) {
    const int rounds = rounds_for(total, cus);
    const int need = ((total + rounds - 1) / rounds + 7) & ~7;
    if (need < grid) grid = need;
  }
  if (tall) gemm_pp_kernel<EPI, AMODE, 5><<<grid, 512, 0, st>>>(p);
  else gemm_pp_kernel<EPI, AMODE, 4><<<grid, 512, 0, st>>>(p);
  return check_launch();
}

template <int EPI, int AMODE>
int launch_gemm(const GemmParams &p_in, hipStream_t st, Profiler *prof) {
  GemmParams p = p_in;
  if (p.M <= 0) return LLA_OK;
  if (p.N % BN || p.K % BK || !p.A || !p.W || !p.C) return LLA_EINVAL;
  if (p.a_chunk_images) {   // (the image batch in pieces: patch embedding of a chip-filling pass, on the ping-pong kernel only)
    if (AMODE == A_PLAIN || AMODE == A_CONV3 || EPI != EPI_PATCH || (p.a_chunk_images & 255) || p.M < 9000 ||
        p.N % 256 || p.N < 768 || p.K < 256)
      return LLA_EINVAL;
  }
  if (p.n_store <= 0 || p.n_store > p.N) p.n_store = p.N;
  // the fp32 epilogues address C with 32-bit element offsets (registers are scarce there)
  if ((EPI == EPI_RESID || EPI == EPI_PATCH) && ((size_t)p.M + (size_t)p.M / kPatches + 2) * (size_t)p.ldc >= (1ull << 32))
    return LLA_EINVAL;
  ProfScope scope(prof, st, LLA_PROF_GEMM, 2.0 * p.M * p.N * p.K);
  // Small problems (< ~9k rows: batches under ~190 images) do not fill 256 persistent workgroups with 256-wide tiles;
  // measured at batch 128: 40.6k img/s persistent vs 48.4k with the one-tile-per-workgroup 256 x 128 kernel.
  const bool big = p.M >= 9000;
  if constexpr (EPI == EPI_RELU || EPI == EPI_ADDRELU) {
    // ResNet-tower GEMMs.  1x1 convolutions whose output is a multiple of 256 channels wide run on
    // the persistent 256-wide kernel with the line-assembling epilogue (whole 128-byte lines instead of 16-byte pieces per
    // row took the add + ReLU convolution of layer1 from 3.2 to 5.3 TB/s); narrow outputs and the implicit 3x3
    // convolutions stay on the one-tile-per-workgroup kernel.
    if constexpr (AMODE == A_PLAIN) {
      if (big && p.N % 256 == 0 && p.n_store == p.N) return launch_persistent<EPI, AMODE, 2>(p, st);
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
    if (!big && p.M > 128) {
      const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2);
      gemm256_f16_kernel<EPI, AMODE><<<tiles2, 512, 0, st>>>(p);
      return check_launch();
    }
    if constexpr (AMODE == A_PLAIN && (EPI == EPI_F16 || EPI == EPI_QGELU)) {
      if (big) {   // (takes ragged M; LLA_EINVAL for shapes outside its scope: N % 256, K < 128, ...)
        const int rc = launch_w8(EPI, p, st);
        if (rc != LLA_EINVAL) return rc;
      }
    }
    if constexpr (AMODE == A_PLAIN && EPI == EPI_RESID) {
      if (big && p.ldc == p.N) {   // (whole 256-row tiles only: LLA_EINVAL otherwise)
        const int rc = launch_q4(EPI, p, st);
        if (rc != LLA_EINVAL) return rc;
      }
    }
    if (p.M > 128) {
      if (p.N % 256 == 0 && p.N >= 768 && p.K >= 256) return launch_pp<EPI, AMODE>(p, st);
      if (p.N % 256 == 0 && p.N >= 768) return launch_persistent<EPI, AMODE, 2>(p, st);
      return launch_persistent<EPI, AMODE, 1>(p, st);
    }
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    gemm_f16_kernel<EPI, AMODE, true><<<tiles, kGemmThreads, 0, st>>>(p);
    return check_launch();
  }
}

}  // namespace

LLA_DEFINE_LAUNCH_GEMM

}  // namespace lla
