// fp32 Linear layers on the matrix cores (gfx950): C[M][N] = act(A[M][K] W[N][K]^T + bias).
//
// Stands in for the `nn.Linear` layers of the reference's hyperprior MLPs, which it evaluates in fp32 under
// `autocast(False)` (lossyless/rates.py:104 "precision here is important", :631-639, :687-699 -- side_encoder,
// z_encoder; lossyless/architectures.py:94-168).  The scale indexes, and with them the bitstreams, depend on
// these values, so the arithmetic is the reference's: fp32 operands, fp32 accumulation, one rounding per
// product -- `v_mfma_f32_32x32x2_f32` is bitwise an fp32 fma chain.
//
// Shape of the work: 512-wide layers over a few thousand rows (0.3-0.5 GFLOP per layer at 1024 rows) -- small
// against the 157 TFLOP/s fp32 matrix peak, so the kernel is the simplest one that is correct and
// batch-invariant: one 32 x 32 output tile per wave, operands straight from global memory (16-byte loads along
// K; the 64 x 64 workgroup tile's rows are shared by its waves through L1), no LDS.  The K order of every output
// element is fixed by the kernel alone (it does not depend on M or on which tile the row falls in), so a decoder
// that evaluates the network at another batch size reproduces the encoder's values bit for bit.
#include "common.h"

namespace lla {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Swapped operands (D^T = W X^T), as in the fp16 GEMMs: lane l holds output row m = m0 + (l & 31) and the
// columns n0 + 8 g + 4 (l >> 5) + e of register 4 g + e -> four 16-byte stores per lane.
// 32x32x2 operand layout: "A" lane l = W[n0 + (l & 31)][k + (l >> 5)], "B" lane l = X[m0 + (l & 31)][k + (l >> 5)].
// A lane loads 4 consecutive k (16 bytes) at k0 + 4 (l >> 5); MFMA j of the group multiplies element j of both
// lanes' vectors, i.e. logical k = k0 + 4 h + j for h = 0, 1: a permutation of the 8 k of the group, the same
// for both operands.
template <bool RELU>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, int lda,
                                                       const float *__restrict__ W, int ldw,
                                                       const float *__restrict__ bias, float *__restrict__ C,
                                                       int ldc, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r32 = lane & 31, hk = lane >> 5;
  const int tiles_n = (N + 63) / 64;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 64 + (wid >> 1) * 32, n0 = tn * 64 + (wid & 1) * 32;
  if (m0 >= M || n0 >= N) return;
  int m = m0 + r32, n = n0 + r32;
  const bool n_ok = n < N;
  if (m >= M) m = M - 1;       // clamped rows are computed and not stored
  if (!n_ok) n = N - 1;
  const float *ap = A + (size_t)m * lda + 4 * hk;
  const float *wp = W + (size_t)n * ldw + 4 * hk;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + k);
    f32x4 w = *reinterpret_cast<const f32x4 *>(wp + k);
    if (!n_ok) w = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], a[j], acc, 0, 0, 0);
  }
  const int mo = m0 + r32;
  if (mo >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nc = n0 + 8 * g + 4 * hk;
    if (nc >= N) continue;     // N % 4 == 0: a quad is inside or outside
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (bias) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + nc);
      v += b;
    }
    if (RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4 *>(C + (size_t)mo * ldc + nc) = v;
  }
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" int lla_gemm_f32(const float *A, int lda, const float *W, int ldw, const float *bias, float *C,
                            int ldc, int M, int N, int K, int relu, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || (K & 7) || (N & 3) || lda < K || ldw < K || ldc < N || (lda & 3) ||
      (ldw & 3) || (ldc & 3))
    return LLA_EINVAL;
  if (M == 0) return LLA_OK;
  if (!A || !W || !C) return LLA_EINVAL;
  const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
  if (tiles > 0x7fffffffLL) return LLA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (relu) gemm_f32_kernel<true><<<(int)tiles, 256, 0, st>>>(A, lda, W, ldw, bias, C, ldc, M, N, K);
  else gemm_f32_kernel<false><<<(int)tiles, 256, 0, st>>>(A, lda, W, ldw, bias, C, ldc, M, N, K);
  return check_launch();
}
