// CLIP preprocessing on the GPU: Resize(224, bicubic) -> CenterCrop(224) -> ToTensor ->
// Normalize -> fp16 NHWC, bit-exact against Pillow + the reference's transform chain.
//
// Stands in for `compressor.preprocess` = clip._transform (clip==1.0: torchvision
// Resize(224, BICUBIC), CenterCrop(224), ToTensor, Normalize(CLIP mean/std)), which the
// reference runs per image with PIL inside DataLoader worker processes
// (hub/compressor.py:155,186; same chain in utils/data/images.py:383-411).  SURVEY.md 8(f)
// rank 2.  Pillow's 8-bit resampler is two 1-D passes in 22-bit fixed point with a uint8
// intermediate image (horizontal first, then vertical); the per-output-pixel tap windows and
// integer coefficients are computed on the host exactly as Pillow does (float64) and handed
// over as tables, so the device side is pure integer work and reproduces Pillow's bytes.
#include "common.h"

namespace lla {
namespace {

typedef _Float16 f16;
constexpr int kPrecisionBits = 22;  // Pillow: 32 - 8 - 2
constexpr int kOut = 224;

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1: rows [row0, row0 + nrows) of every image, 224 cropped output columns
__global__ void resample_h_kernel(const uint8_t *__restrict__ img, int B, int H, int W, int row0,
                                  int nrows, const int *__restrict__ bounds,
                                  const int *__restrict__ coef, int ksize,
                                  uint8_t *__restrict__ tmp) {
  const size_t n = (size_t)B * nrows * kOut * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const int x = (int)((i / 3) % kOut);
    const int r = (int)((i / (3 * kOut)) % nrows);
    const int b = (int)(i / ((size_t)3 * kOut * nrows));
    const int xmin = bounds[2 * x], xmax = bounds[2 * x + 1];
    const uint8_t *src = img + (((size_t)b * H + row0 + r) * W + xmin) * 3 + c;
    const int *k = coef + x * ksize;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < xmax; ++t) ss += (int)src[3 * t] * k[t];
    tmp[i] = (uint8_t)clip8(ss);
  }
}

// pass 2 + ToTensor + Normalize + half: (u8 / 255 - mean) / std, each op rounded to fp32
__global__ void resample_v_norm_kernel(const uint8_t *__restrict__ tmp, int B, int row0, int nrows,
                                       const int *__restrict__ bounds, const int *__restrict__ coef,
                                       int ksize, float m0, float m1, float m2, float s0, float s1,
                                       float s2, f16 *__restrict__ out) {
  const size_t n = (size_t)B * kOut * kOut * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const int x = (int)((i / 3) % kOut);
    const int y = (int)((i / (3 * kOut)) % kOut);
    const int b = (int)(i / ((size_t)3 * kOut * kOut));
    const int ymin = bounds[2 * y] - row0, ymax = bounds[2 * y + 1];
    const uint8_t *src = tmp + (((size_t)b * nrows + ymin) * kOut + x) * 3 + c;
    const int *k = coef + y * ksize;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < ymax; ++t) ss += (int)src[(size_t)t * kOut * 3] * k[t];
    const float u = __fdiv_rn((float)clip8(ss), 255.f);
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = (f16)__fdiv_rn(__fsub_rn(u, mean), stdv);
  }
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" {

size_t lla_preprocess_workspace_bytes(int B, int nrows) {
  return (size_t)(B > 0 ? B : 0) * (size_t)(nrows > 0 ? nrows : 0) * kOut * 3;
}

int lla_preprocess_clip(const uint8_t *images, int B, int H, int W, int row0, int nrows,
                        const int32_t *h_bounds, const int32_t *h_coef, int h_ksize,
                        const int32_t *v_bounds, const int32_t *v_coef, int v_ksize,
                        const float *mean3, const float *std3, void *workspace,
                        size_t workspace_bytes, void *out_nhwc_f16, void *stream) {
  if (B < 0 || H <= 0 || W <= 0 || row0 < 0 || nrows <= 0 || row0 + nrows > H || h_ksize <= 0 ||
      v_ksize <= 0 || !mean3 || !std3)
    return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!images || !h_bounds || !h_coef || !v_bounds || !v_coef || !workspace || !out_nhwc_f16)
    return LLA_EINVAL;
  if (workspace_bytes < lla_preprocess_workspace_bytes(B, nrows)) return LLA_ECAP;
  hipStream_t st = as_stream(stream);
  const size_t n1 = (size_t)B * nrows * kOut * 3, n2 = (size_t)B * kOut * kOut * 3;
  auto grid = [](size_t n) { size_t g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : g); };
  resample_h_kernel<<<grid(n1), 256, 0, st>>>(images, B, H, W, row0, nrows, h_bounds, h_coef,
                                              h_ksize, reinterpret_cast<uint8_t *>(workspace));
  resample_v_norm_kernel<<<grid(n2), 256, 0, st>>>(
      reinterpret_cast<const uint8_t *>(workspace), B, row0, nrows, v_bounds, v_coef, v_ksize,
      mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], reinterpret_cast<f16 *>(out_nhwc_f16));
  return check_launch();
}

}  // extern "C"
