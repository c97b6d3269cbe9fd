// CLIP RN50 visual tower (ModifiedResNet + attention pool) for gfx950: fp16 NHWC activations, fp32
// accumulation, BatchNorm folded into the convolutions.
//
// Stands in for `clip.load("RN50")[0].visual` as the reference's pretrained featuriser uses it
// (lossyless/architectures.py:367-371: `arch = "ViT-B/32" if "vit" in self.model else "RN50"`;
// clip==1.0 ModifiedResNet: 3-conv stem + avgpool, bottlenecks [3, 4, 6, 3] with anti-aliasing
// average pools in place of strided convolutions, AttentionPool2d(7, 2048, 32 heads, 1024)).
// SURVEY.md 8(f) rank 4.
//
// Every convolution is a GEMM on the tower's MFMA kernels: 1x1 convolutions read the NHWC activation
// matrix [B*H*W][channel pitch] in place (lla_gemm_f16_ex); stride-1 3x3 convolutions are implicit
// GEMMs whose loader gathers the nine taps (lla_conv3x3_relu_f16; K ordered (kh, kw, c), zero-padded to
// a multiple of 64); only the first stem convolution (3 channels, stride 2) goes through an im2col
// matrix.  ReLU and the bottleneck's "+ identity, ReLU" are GEMM epilogues.  Weight rows are padded to
// a multiple of 128 output columns (zero weights), but 32- / 64-channel outputs are stored with a 32- /
// 64-element pitch (the padding columns are computed and dropped).
#include "common.h"
#include "switches.h"
#include <cstdlib>

#include <vector>

namespace lla {
namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvDesc {
  int cin, cout, ksize, stride;   // stride only for the first stem convolution (2)
  int kpad, npad;                 // GEMM K (multiple of 64) and N (multiple of 128)
  size_t w_off, b_off;            // byte offsets in the blob: fp16 [npad][kpad], fp32 [npad]
};

constexpr int kStages = 4;
constexpr int kBlocks[kStages] = {3, 4, 6, 3};
constexpr int kPlanes[kStages] = {64, 128, 256, 512};
constexpr int kEmbed = 2048, kHeads = 32, kTokens = 50, kOutDim = 1024;
constexpr size_t kAlign = 256;
constexpr size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Layout {
  std::vector<ConvDesc> convs;
  // first block of every stage: conv3 and the downsample convolution as ONE 1x1 convolution over the concatenated inputs
  // [main path (planes) | block input (inplanes)], weights [4 planes][planes + inplanes] = [W3 | Wds], bias b3 + bds
  ConvDesc fused[kStages];
  size_t pos_off, q_w, q_b, kv_w, kv_b, c_w, c_b, total;
};

const Layout &layout() {
  static const Layout L = [] {
    Layout l;
    size_t off = 0;
    auto add = [&](int cin, int cout, int k, int stride) {
      ConvDesc d{cin, cout, k, stride, round_up(cin * k * k, 64), round_up(cout, 128), 0, 0};
      d.w_off = off; off += align_up((size_t)d.npad * d.kpad * 2);
      d.b_off = off; off += align_up((size_t)d.npad * 4);
      l.convs.push_back(d);
    };
    add(3, 32, 3, 2); add(32, 32, 3, 1); add(32, 64, 3, 1);
    int inplanes = 64;
    for (int s = 0; s < kStages; ++s)
      for (int b = 0; b < kBlocks[s]; ++b) {
        const int p = kPlanes[s];
        add(inplanes, p, 1, 1); add(p, p, 3, 1); add(p, 4 * p, 1, 1);
        if (b == 0) add(inplanes, 4 * p, 1, 1);   // downsample
        inplanes = 4 * p;
      }
    inplanes = 64;
    for (int s = 0; s < kStages; ++s) {
      const int p = kPlanes[s];
      ConvDesc d{p + inplanes, 4 * p, 1, 1, round_up(p + inplanes, 64), round_up(4 * p, 128), 0, 0};
      d.w_off = off; off += align_up((size_t)d.npad * d.kpad * 2);
      d.b_off = off; off += align_up((size_t)d.npad * 4);
      l.fused[s] = d;
      inplanes = 4 * p;
    }
    l.pos_off = off; off += align_up((size_t)kTokens * kEmbed * 4);
    l.q_w = off; off += align_up((size_t)kEmbed * kEmbed * 2);
    l.q_b = off; off += align_up((size_t)kEmbed * 4);
    l.kv_w = off; off += align_up((size_t)2 * kEmbed * kEmbed * 2);
    l.kv_b = off; off += align_up((size_t)2 * kEmbed * 4);
    l.c_w = off; off += align_up((size_t)kOutDim * kEmbed * 2);
    l.c_b = off; off += align_up((size_t)kOutDim * 4);
    l.total = off;
    return l;
  }();
  return L;
}

// ---------------------------------------------------------------------------
// im2col for 3x3 / pad 1 convolutions over NHWC fp16 [B][H][W][pitch] (first cin channels used):
// col[(b, oy, ox)][(kh*3 + kw) * cin + c], rows of kpad halfs (tail zero).  One thread per 8 halfs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const f16 *__restrict__ in, int H, int W, int pitch,
                                                        int cin, int stride, int Ho, int Wo, int kpad,
                                                        f16 *__restrict__ col, size_t n_vec) {
  const int vec_per_row = kpad >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / vec_per_row;
    const int k0 = (int)(i - row * vec_per_row) * 8;
    const int ox = (int)(row % Wo);
    const int oy = (int)((row / Wo) % Ho);
    const size_t b = row / ((size_t)Wo * Ho);
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((cin & 7) == 0) {
      if (k0 < 9 * cin) {
        const int tap = k0 / cin, c = k0 - tap * cin;
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
          v = *reinterpret_cast<const f16x8 *>(in + (((size_t)b * H + iy) * W + ix) * pitch + c);
      }
    } else {   // the RGB stem convolution: element-wise
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        if (k < 9 * cin) {
          const int tap = k / cin, c = k - tap * cin;
          const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = in[(((size_t)b * H + iy) * W + ix) * pitch + c];
        }
      }
    }
    *reinterpret_cast<f16x8 *>(col + row * kpad + k0) = v;
  }
}

// 2x2 average pool over NHWC (fp32 sum, x 0.25, one rounding), 8 channels per thread
__global__ __launch_bounds__(256) void avgpool2_kernel(const f16 *__restrict__ in, int H, int W, int pitch,
                                                       int C, f16 *__restrict__ out, int out_pitch, size_t n_vec) {
  const int Ho = H >> 1, Wo = W >> 1, vec_per_px = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / vec_per_px;
    const int c = (int)(i - px * vec_per_px) * 8;
    const int ox = (int)(px % Wo);
    const int oy = (int)((px / Wo) % Ho);
    const size_t b = px / ((size_t)Wo * Ho);
    const f16 *p00 = in + (((size_t)b * H + 2 * oy) * W + 2 * ox) * pitch + c;
    const f16x8 a = *reinterpret_cast<const f16x8 *>(p00), bb = *reinterpret_cast<const f16x8 *>(p00 + pitch),
                cc = *reinterpret_cast<const f16x8 *>(p00 + (size_t)W * pitch),
                dd = *reinterpret_cast<const f16x8 *>(p00 + (size_t)W * pitch + pitch);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((((float)a[e] + (float)bb[e]) + ((float)cc[e] + (float)dd[e])) * 0.25f);
    *reinterpret_cast<f16x8 *>(out + px * out_pitch + c) = o;
  }
}

// attention-pool tokens: t[b][0] = mean_j x[b][j] + pos[0]; t[b][1 + j] = x[b][j] + pos[1 + j]  (fp32, one rounding)
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const f16 *__restrict__ x, const float *__restrict__ pos,
                                                              f16 *__restrict__ t, int B) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < kEmbed; c += blockDim.x) {
    float sum = 0.f;
    const f16 *xp = x + (size_t)b * 49 * kEmbed + c;
    f16 *tp = t + (size_t)b * kTokens * kEmbed + c;
    for (int j = 0; j < 49; ++j) {
      const float v = (float)xp[(size_t)j * kEmbed];
      sum += v;
      tp[(size_t)(1 + j) * kEmbed] = (f16)(v + pos[(size_t)(1 + j) * kEmbed + c]);
    }
    tp[0] = (f16)(sum * (1.f / 49.f) + pos[c]);
  }
}

// single-query attention: one wave per (image, head); q [B][2048], kv [B*50][4096] (k | v), o [B][2048]
__global__ __launch_bounds__(256) void attnpool_attend_kernel(const f16 *__restrict__ q, const f16 *__restrict__ kv,
                                                              f16 *__restrict__ o, int B) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = wave / kHeads, h = wave - b * kHeads;
  if (b >= B) return;
  const float qd = (float)q[(size_t)b * kEmbed + h * 64 + lane] * 0.125f;   // head_dim^-0.5 after the projection
  const f16 *kp = kv + (size_t)b * kTokens * (2 * kEmbed) + h * 64 + lane;
  float s[kTokens];
  float mx = -3.0e38f;
  for (int j = 0; j < kTokens; ++j) {
    const float p = wave_sum_f32(qd * (float)kp[(size_t)j * (2 * kEmbed)]);   // (DPP sum: common.h)
    s[j] = p;
    mx = fmaxf(mx, p);
  }
  float den = 0.f;
  for (int j = 0; j < kTokens; ++j) { s[j] = __expf(s[j] - mx); den += s[j]; }
  const float inv = 1.f / den;
  float acc = 0.f;
  const f16 *vp = kp + kEmbed;
  for (int j = 0; j < kTokens; ++j) acc += s[j] * inv * (float)vp[(size_t)j * (2 * kEmbed)];
  o[(size_t)b * kEmbed + h * 64 + lane] = (f16)acc;
}

// conv3 + downsample of a stage's first block as one GEMM over [main | block input] (no identity tensor written and
// read back: 3.3 GB per 1024 images in layer1 alone); LLA_RN50_FUSE_DS=0: two GEMMs, the identity rounded to fp16 in between
inline bool fuse_downsample() { return sw::rn50_fuse_downsample(); }

inline bool direct_conv() { return sw::rn50_direct_conv(); }

inline int grid_for(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 65535 * 16 ? 65535 * 16 : (g ? g : 1)); }

// per-image workspace elements (halfs): three activation buffers + identity + im2col
constexpr size_t kActElems = (size_t)112 * 112 * 128;   // largest activation (stem, pitch 128)
constexpr size_t kColElems = (size_t)112 * 112 * 320;   // largest im2col matrix (stem conv2 / conv3)
size_t workspace_bytes(int chunk) {
  return (size_t)chunk * (4 * kActElems + kColElems) * 2 + (size_t)chunk * (kTokens * kEmbed + kTokens * 2 * kEmbed + 2 * kEmbed) * 2 + 4096;
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" {

size_t lla_rn50_weights_bytes(void) { return layout().total; }
int lla_rn50_conv_count(void) { return (int)layout().convs.size(); }
int lla_rn50_conv_desc(int i, int64_t *out8) {
  const auto &L = layout();
  if (i < 0 || i >= (int)L.convs.size() || !out8) return LLA_EINVAL;
  const ConvDesc &d = L.convs[(size_t)i];
  out8[0] = d.cin; out8[1] = d.cout; out8[2] = d.ksize; out8[3] = d.stride; out8[4] = d.kpad; out8[5] = d.npad;
  out8[6] = (int64_t)d.w_off; out8[7] = (int64_t)d.b_off;
  return LLA_OK;
}
int lla_rn50_fused_desc(int stage, int64_t *out8) {
  if (stage < 0 || stage >= kStages || !out8) return LLA_EINVAL;
  const ConvDesc &d = layout().fused[stage];
  out8[0] = d.cin; out8[1] = d.cout; out8[2] = kPlanes[stage]; out8[3] = d.cin - kPlanes[stage]; out8[4] = d.kpad; out8[5] = d.npad;
  out8[6] = (int64_t)d.w_off; out8[7] = (int64_t)d.b_off;
  return LLA_OK;
}
int lla_rn50_attnpool_offsets(int64_t *out7) {
  if (!out7) return LLA_EINVAL;
  const auto &L = layout();
  out7[0] = (int64_t)L.pos_off; out7[1] = (int64_t)L.q_w; out7[2] = (int64_t)L.q_b; out7[3] = (int64_t)L.kv_w;
  out7[4] = (int64_t)L.kv_b; out7[5] = (int64_t)L.c_w; out7[6] = (int64_t)L.c_b;
  return LLA_OK;
}
size_t lla_rn50_workspace_bytes(int chunk) {   // slice buffers for both tower lanes
  return (size_t)tower_lanes() * workspace_bytes(chunk > 0 ? chunk : 32);
}

static int rn50_slices(const void *images_nhwc_f16, int c_begin, int c_end, int chunk, const void *weights,
                       void *workspace, void *z_out, void *stream);

int lla_rn50_forward(const void *images_nhwc_f16, int B, const void *weights, void *workspace,
                     size_t workspace_bytes_given, int chunk, void *z_out, void *stream, void *tower) {
  if (B < 0) return LLA_EINVAL;
  if (B == 0) return LLA_OK;
  if (!images_nhwc_f16 || !weights || !workspace || !z_out) return LLA_EINVAL;
  if (chunk <= 0) chunk = 32;
  if (chunk > B) chunk = B;
  // Two tower lanes (tower.hip): from 32 images on the batch is cut into (at least) two slices that alternate
  // between the two HIP streams of the caller's tower handle, each with its own half of the workspace: the tail of one slice's
  // one-tile-per-workgroup GEMMs and its pooling kernels run beside the other slice's GEMMs (31.0k -> 34.5k
  // img/s at batch 256, tools/rn50_two_stream_probe.py).  Same embeddings: images are independent.
  const size_t lane_bytes = (workspace_bytes_given / 2) & ~(size_t)255;
  if (tower && tower_lanes() == 2 && B >= 32) {
    const int half = (B + 1) / 2, sub = chunk < half ? chunk : half;
    if (workspace_bytes(sub) <= lane_bytes) {
      Lanes *ln = reinterpret_cast<Lanes *>(tower);
      int rc = ln->dirty ? lanes_join(ln, as_stream(stream)) : LLA_OK;   // (deferred ViT passes share the lanes)
      if (rc == LLA_OK) rc = lanes_fork(ln, as_stream(stream));
      if (rc != LLA_OK) return rc;
      int lane = 0;
      for (int c0 = 0; c0 < B; c0 += sub, lane ^= 1) {
        const int c1 = c0 + sub < B ? c0 + sub : B;
        rc = rn50_slices(images_nhwc_f16, c0, c1, sub, weights,
                         reinterpret_cast<uint8_t *>(workspace) + (size_t)lane * lane_bytes, z_out, ln->st[lane]);
        if (rc != LLA_OK) return rc;
      }
      return lanes_join(ln, as_stream(stream));
    }
  }
  if (workspace_bytes_given < workspace_bytes(chunk)) return LLA_ECAP;
  return rn50_slices(images_nhwc_f16, 0, B, chunk, weights, workspace, z_out, stream);
}

// images [c_begin, c_end) in slices of `chunk`, all on `stream`, through the slice buffers at `workspace`
static int rn50_slices(const void *images_nhwc_f16, int c_begin, int c_end, int chunk, const void *weights,
                       void *workspace, void *z_out, void *stream) {
  const int B = c_end;
  hipStream_t st = as_stream(stream);
  const Layout &L = layout();
  const uint8_t *wb = reinterpret_cast<const uint8_t *>(weights);
  f16 *ws = reinterpret_cast<f16 *>(workspace);
  f16 *bufA = ws, *bufB = bufA + (size_t)chunk * kActElems, *bufC = bufB + (size_t)chunk * kActElems,
      *bufD = bufC + (size_t)chunk * kActElems, *col = bufD + (size_t)chunk * kActElems;
  f16 *tok = col + (size_t)chunk * kColElems, *kvb = tok + (size_t)chunk * kTokens * kEmbed,
      *qb = kvb + (size_t)chunk * kTokens * 2 * kEmbed, *ob = qb + (size_t)chunk * kEmbed;
  int rc = LLA_OK;
#define LLA_TRY(expr) do { rc = (expr); if (rc != LLA_OK) return rc; } while (0)
  auto W16 = [&](const ConvDesc &d) { return wb + d.w_off; };
  auto B32 = [&](const ConvDesc &d) { return reinterpret_cast<const float *>(wb + d.b_off); };
  // conv (+ folded BN) as GEMM; in: [n][H][W][pitch]; out: [n][Ho][Wo][d.npad]
  // channel pitch of a convolution's output: its padded width, except that narrow ReLU outputs (32 / 64
  // channels: stem and layer 1, the largest activations) keep a narrow pitch -- the GEMM still computes 128
  // columns but stores only the real ones, halving / quartering those layers' activation traffic
  auto opitch = [](const ConvDesc &d) { return d.cout % 128 == 0 ? d.npad : (d.cout + 31) / 32 * 32; };
  auto conv = [&](const ConvDesc &d, const f16 *in, int n, int H, int Wd, int pitch, f16 *out, int epi,
                  const f16 *resid, int ldr) -> int {
    const int ldo = (epi == LLA_EPI_RELU_F16 || epi == LLA_EPI_ADD_RELU_F16) ? opitch(d) : d.npad;
    if (d.ksize == 1) {
      return lla_gemm_f16_ex(in, pitch, W16(d), B32(d), out, ldo, resid, ldr, n * H * Wd, d.npad, d.kpad, epi, stream);
    }
    const int Ho = (H - 1) / d.stride + 1, Wo = (Wd - 1) / d.stride + 1;   // k 3, pad 1
    // stride-1 convolutions over >= 64-channel inputs followed by ReLU (conv2 of every bottleneck): implicit
    // GEMM, the loader gathers the taps itself (LLA_RN50_IM2COL=1 keeps the im2col path for A/B)
    const bool use_im2col = sw::rn50_im2col();
    // the narrow ones (stem 32 -> 32 / 64, layer1 64 -> 64): direct convolution, one 8 x 8 tile per wave (conv_direct.hip;
    // bit-identical; LLA_RN50_DIRECT=0 keeps them on the implicit GEMM for A/B)
    if (direct_conv() && !use_im2col && d.stride == 1 && epi == LLA_EPI_RELU_F16 && !resid && H % 8 == 0 && Wd % 8 == 0 &&
        ((d.cin == 32 && (d.cout == 32 || d.cout == 64)) || (d.cin == 64 && d.cout == 64)))
      return lla_conv3x3_direct_relu_f16(in, n, H, Wd, pitch, d.cin, W16(d), d.kpad, B32(d), out, ldo, d.cout, 0, stream);
    if (!use_im2col && d.stride == 1 && (d.cin % 64 == 0 || d.cin == 32) && epi == LLA_EPI_RELU_F16 && !resid)
      return lla_conv3x3_relu_f16(in, n, H, Wd, pitch, d.cin, W16(d), B32(d), out, ldo, d.npad, stream);
    if (direct_conv() && !use_im2col && d.stride == 2 && d.cin == 3 && d.cout == 32 && pitch == 3 && epi == LLA_EPI_RELU_F16 &&
        !resid && H % 16 == 0 && Wd % 16 == 0)   // the RGB stem convolution: direct as well (no im2col matrix)
      return lla_conv3x3_rgb_s2_relu_f16(in, n, H, Wd, W16(d), d.kpad, B32(d), out, ldo, stream);
    const size_t rows = (size_t)n * Ho * Wo, n_vec = rows * (d.kpad >> 3);
    im2col3x3_kernel<<<grid_for(n_vec), 256, 0, st>>>(in, H, Wd, pitch, d.cin, d.stride, Ho, Wo, d.kpad, col, n_vec);
    if (int e = check_launch()) return e;
    return lla_gemm_f16_ex(col, d.kpad, W16(d), B32(d), out, ldo, resid, ldr, (int)rows, d.npad, d.kpad, epi, stream);
  };
  auto pool = [&](const f16 *in, int n, int H, int Wd, int pitch, int C, f16 *out, int out_pitch) -> int {
    const size_t n_vec = (size_t)n * (H / 2) * (Wd / 2) * (C >> 3);
    avgpool2_kernel<<<grid_for(n_vec), 256, 0, st>>>(in, H, Wd, pitch, C, out, out_pitch, n_vec);
    return check_launch();
  };

  for (int c0 = c_begin; c0 < B; c0 += chunk) {
    const int n = (B - c0) < chunk ? (B - c0) : chunk;
    const f16 *img = reinterpret_cast<const f16 *>(images_nhwc_f16) + (size_t)c0 * 224 * 224 * 3;
    size_t ci = 0;
    // stem
    const int p0 = opitch(L.convs[0]), p1 = opitch(L.convs[1]), p2 = opitch(L.convs[2]);      // 32, 32, 64
    LLA_TRY(conv(L.convs[ci++], img, n, 224, 224, 3, bufA, LLA_EPI_RELU_F16, nullptr, 0));    // 112x112x32
    LLA_TRY(conv(L.convs[ci++], bufA, n, 112, 112, p0, bufB, LLA_EPI_RELU_F16, nullptr, 0));
    f16 *x = bufB, *t1 = bufA, *t2 = bufC, *idb = bufD;
    int H = 56, pitch = p2;
    // layer1's first block multiplies [conv2 output | block input] in one GEMM (fuse_downsample): the stem then writes its
    // output straight into the right half of that 128-channel-pitch buffer
    // (with the whole first block as one kernel -- bottleneck_fused.hip, CIN = 64 -- the stem's output is a plain 64-channel tensor)
    // (the fused kernel addresses its input with 32-bit byte offsets: slices beyond that keep the three kernels)
    auto fits32 = [&](int channels) { return (size_t)n * 56 * 56 * channels * 2 < (1ull << 31); };
    const bool block0_fused = sw::rn50_fused_bottleneck() && fuse_downsample() && direct_conv() && fits32(64);
    const bool fuse0 = fuse_downsample() && direct_conv() && !block0_fused;
    if (direct_conv()) {   // third stem convolution and the stem's average pool in one kernel: 56x56x64 straight away
      const ConvDesc &d3 = L.convs[ci++];
      if (fuse0) {
        LLA_TRY(lla_conv3x3_direct_relu_f16(bufB, n, 112, 112, p1, d3.cin, W16(d3), d3.kpad, B32(d3), bufD + 64, 128, d3.cout, 1, stream));
        x = bufD + 64; pitch = 128; t1 = bufA; t2 = bufC; idb = bufB;
      } else {
        LLA_TRY(lla_conv3x3_direct_relu_f16(bufB, n, 112, 112, p1, d3.cin, W16(d3), d3.kpad, B32(d3), bufA, p2, d3.cout, 1, stream));
        x = bufA; t1 = bufB;
      }
    } else {
      LLA_TRY(conv(L.convs[ci++], bufB, n, 112, 112, p1, bufA, LLA_EPI_RELU_F16, nullptr, 0));  // 64 channels
      LLA_TRY(pool(bufA, n, 112, 112, p2, 64, bufB, p2));                                       // 56x56x64
    }
    for (int s = 0; s < kStages; ++s)
      for (int b = 0; b < kBlocks[s]; ++b) {
        const int stride = (s > 0 && b == 0) ? 2 : 1;
        const ConvDesc &c1 = L.convs[ci], &c2 = L.convs[ci + 1], &c3 = L.convs[ci + 2];
        ci += 3;
        // layer1's blocks (56 x 56; block 0: 64 -> 64 -> 64 -> 256 with the downsample convolution folded into conv3; blocks 1, 2:
        // 256 -> 64 -> 64 -> 256, identity = the input): one kernel each, the 64-channel
        // intermediates never leave the CU (bottleneck_fused.hip; LLA_RN50_FUSED_BLOCK=0 keeps the three kernels for A/B)
        if (s == 0 && b == 0 && block0_fused && H % 14 == 0 && c1.cin == 64) {   // conv3 | downsample over [t2 | x]: Layout::fused
          const ConvDesc &fd = L.fused[0];
          ++ci;                                                                  // (the downsample convolution's own descriptor)
          LLA_TRY(lla_rn50_bottleneck_f16(x, n, H, H, pitch, c1.cin, W16(c1), c1.kpad, B32(c1), W16(c2), c2.kpad, B32(c2), W16(fd),
                                          fd.kpad, B32(fd), t2, c3.npad, stream));
          { f16 *o = x; x = t2; t2 = o; }
          pitch = c3.npad;
          continue;
        }
        if (s == 0 && b > 0 && sw::rn50_fused_bottleneck() && H % 14 == 0 && c1.cin == 256 && pitch == c3.npad && fits32(pitch)) {
          f16 *out = t2;
          LLA_TRY(lla_rn50_bottleneck_f16(x, n, H, H, pitch, c1.cin, W16(c1), c1.kpad, B32(c1), W16(c2), c2.kpad, B32(c2), W16(c3),
                                          c3.kpad, B32(c3), out, c3.npad, stream));
          { f16 *o = x; x = t2; t2 = o; }
          continue;
        }
        LLA_TRY(conv(c1, x, n, H, H, pitch, t1, LLA_EPI_RELU_F16, nullptr, 0));
        if (b == 0 && fuse_downsample() && (s > 0 || fuse0)) {
          const ConvDesc &ds = L.convs[ci++], &fd = L.fused[s];
          (void)ds;
          const int P = fd.cin, Ho = H / stride;
          f16 *cat, *out;
          if (s == 0) {   // x already sits in columns 64..127 of its buffer; conv2 (direct) fills columns 0..63
            cat = x - 64;
            LLA_TRY(lla_conv3x3_direct_relu_f16(t1, n, H, H, opitch(c1), c2.cin, W16(c2), c2.kpad, B32(c2), cat, P, c2.cout, 0, stream));
            out = t2;
          } else {        // both inputs pass their anti-aliasing average pool on the way into the concatenated buffer
            LLA_TRY(conv(c2, t1, n, H, H, opitch(c1), t2, LLA_EPI_RELU_F16, nullptr, 0));
            cat = idb;
            LLA_TRY(pool(t2, n, H, H, opitch(c2), c2.cout, cat, P));
            LLA_TRY(pool(x, n, H, H, pitch, P - c2.cout, cat + c2.cout, P));
            out = x;
          }
          LLA_TRY(lla_gemm_f16_ex(cat, P, W16(fd), B32(fd), out, c3.npad, nullptr, 0, n * Ho * Ho, fd.npad, fd.kpad,
                                  LLA_EPI_RELU_F16, stream));
          if (s == 0) { f16 *o = t2; t2 = idb; idb = cat; x = o; }   // (four distinct buffers again: x, t1, t2, idb)
          H = Ho;
          pitch = c3.npad;
          continue;
        }
        LLA_TRY(conv(c2, t1, n, H, H, opitch(c1), t2, LLA_EPI_RELU_F16, nullptr, 0));
        const f16 *main_in = t2;
        int Ho = H;
        if (stride == 2) {
          LLA_TRY(pool(t2, n, H, H, opitch(c2), opitch(c2), t1, opitch(c2)));
          main_in = t1;
          Ho = H / 2;
        }
        const f16 *ident = x;
        int ld_ident = pitch;
        if (b == 0) {
          const ConvDesc &ds = L.convs[ci++];
          const f16 *ds_in = x;
          if (stride == 2) {
            f16 *pooled = stride == 2 && main_in == t1 ? t2 : t1;   // t2 is free once pooled into t1
            LLA_TRY(pool(x, n, H, H, pitch, ds.cin, pooled, pitch));
            ds_in = pooled;
          }
          LLA_TRY(conv(ds, ds_in, n, Ho, Ho, pitch, idb, LLA_EPI_F16, nullptr, 0));
          ident = idb;
          ld_ident = ds.npad;
        }
        // out = relu(conv3(main) + identity): written over the buffer that is dead now
        f16 *out = (main_in == t1) ? t2 : t1;
        if (b == 0 && stride == 2) out = x;   // x (and its pooled copy) were consumed by the downsample branch
        LLA_TRY(conv(c3, main_in, n, Ho, Ho, opitch(c2), out, LLA_EPI_ADD_RELU_F16, ident, ld_ident));
        // rotate buffers: the new x must not alias t1 / t2 / idb of the next block
        if (out == x) { /* in place */ }
        else if (out == t1) { f16 *o = x; x = t1; t1 = o; }
        else { f16 *o = x; x = t2; t2 = o; }
        H = Ho;
        pitch = c3.npad;
      }
    // attention pool over the 7x7 map (x: [n][49][2048])
    attnpool_tokens_kernel<<<n, 256, 0, st>>>(x, reinterpret_cast<const float *>(wb + L.pos_off), tok, n);
    LLA_TRY(check_launch());
    LLA_TRY(lla_gemm_f16_ex(tok, kEmbed, wb + L.kv_w, reinterpret_cast<const float *>(wb + L.kv_b), kvb, 2 * kEmbed,
                            nullptr, 0, n * kTokens, 2 * kEmbed, kEmbed, LLA_EPI_F16, stream));
    LLA_TRY(lla_gemm_f16_ex(tok, kTokens * kEmbed, wb + L.q_w, reinterpret_cast<const float *>(wb + L.q_b), qb, kEmbed,
                            nullptr, 0, n, kEmbed, kEmbed, LLA_EPI_F16, stream));
    attnpool_attend_kernel<<<(n * kHeads + 3) / 4, 256, 0, st>>>(qb, kvb, ob, n);
    LLA_TRY(check_launch());
    LLA_TRY(lla_gemm_f16_ex(ob, kEmbed, wb + L.c_w, reinterpret_cast<const float *>(wb + L.c_b),
                            reinterpret_cast<f16 *>(z_out) + (size_t)c0 * kOutDim, kOutDim, nullptr, 0, n, kOutDim, kEmbed,
                            LLA_EPI_F16, stream));
  }
#undef LLA_TRY
  return LLA_OK;
}

}  // extern "C"
