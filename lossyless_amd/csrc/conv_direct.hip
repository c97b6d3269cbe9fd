// Direct 3x3 / stride 1 / pad 1 convolution + bias + ReLU (+ 2x2 average pool) for the NARROW layers of the RN50 tower
// (stem conv2 32 -> 32, stem conv3 32 -> 64 with the stem's pool, layer1 conv2 64 -> 64): NHWC fp16 in and out.
//
// Why not the implicit GEMM (gemm_kernels.h gemm256_f16_kernel<EPI_RELU, A_CONV3>): its 256 x 128 tile computes 128 output
// columns for 32 / 64 real ones, and every one of the nine taps re-stages its operand rows from L2 into LDS -- at 112 x 112
// that is 1.5-1.8 ms per stem convolution against an HBM floor of 0.3-0.45 ms (profiles/r04_rn50_*).  Here a WAVE owns an
// 8 x 8 output tile: the 10 x 10 x cin halo goes into LDS once and the nine taps are nine LDS addresses of the same
// pixels; the weights stay in registers (32 -> 32) or in LDS (64 outputs) for the whole kernel; waves are persistent and
// independent (no workgroup barrier after start-up), the next tile's halo is in flight while this one is multiplied.
//
// Bit-identical to the implicit-GEMM path by construction: same v_mfma_f32_32x32x16_f16, K walked in the same order
// (tap, then channel, 16 per step; the GEMM's zero padding of K adds exact zeros), same fp32 bias + ReLU + one
// rounding; the pooled variant rounds to fp16 first and then averages exactly as avgpool2_kernel (rn50.hip) does
// -- ((a + b) + (c + d)) * 0.25f -- so conv3 + pool in one kernel writes the bytes the two kernels wrote.
// tests/test_gpu_rn50.py compares the tower with LLA_RN50_DIRECT=0 / 1.
#include "common.h"

#include <cstdlib>

namespace lla {
namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// LDS staging written as dwords and read as halfs: without may_alias the type-based alias analysis lets hipcc hoist the reads
// over the writes (it did: every tile multiplied the first tile's pixels)
typedef unsigned __attribute__((may_alias)) u32_lds;
typedef f16 __attribute__((may_alias)) f16_lds;

__device__ __attribute__((aligned(16))) f16 g_conv_zero[8];   // (zero-initialised)

template <int CIN, int COUT, bool POOL>
__global__ __launch_bounds__(256, CIN == 64 ? 1 : 2) void conv3x3_direct_kernel(const f16 *__restrict__ in, int H, int W, int pitch,
                                                             const f16 *__restrict__ wt, int kpad,
                                                             const float *__restrict__ bias, f16 *__restrict__ out,
                                                             int out_pitch, int n_images) {
  constexpr int KS = CIN / 16, NF = COUT / 32;
  constexpr bool W_LDS = COUT > 32;                // weights in LDS when 9 x KS x NF fragments (144 / 288 registers) would cost the second wave per SIMD
  constexpr int PSTR = CIN * 2 + 16;               // bytes per halo pixel: + 16 so that 16 consecutive pixels cover all banks
  constexpr int HALO = 100 * PSTR + 16;            // one wave's 10 x 10 halo (+ a slot the idle lanes of the last sweep write)
  constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel
  constexpr int NCH = 100 * CPP, NLD = (NCH + 63) / 64;
  // weights in LDS, fragment-major: [(tap, k-step)][k half (lane / 32)][COUT rows][16 bytes]: a fragment read is 32
  // consecutive 16-byte slots per half-wave
  constexpr int WBYTES = W_LDS ? 9 * KS * 2 * COUT * 16 : 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * HALO + WBYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hk = lane >> 5;
  unsigned char *my = smem + wid * HALO;
  unsigned char *wl = smem + 4 * HALO;

  f16x8 fb[W_LDS ? 1 : 9][W_LDS ? 1 : KS][W_LDS ? 1 : NF];
  if constexpr (W_LDS) {
    for (int q = tid; q < 9 * KS * 2 * COUT; q += 256) {
      const int n = q % COUT, h = (q / COUT) & 1, ts = q / (2 * COUT);   // ts = tap * KS + ks
      *reinterpret_cast<f16x8 *>(wl + q * 16) =
          *reinterpret_cast<const f16x8 *>(wt + (size_t)n * kpad + (ts / KS) * CIN + 16 * (ts % KS) + 8 * h);
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          fb[tap][ks][nf] = *reinterpret_cast<const f16x8 *>(wt + (size_t)(32 * nf + r32) * kpad + tap * CIN + 16 * ks + 8 * hk);
  }
  f32x4 bias4[NF][4];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[nf][g] = *reinterpret_cast<const f32x4 *>(bias + 32 * nf + 8 * g + 4 * hk);

  // this lane's halo chunks: chunk q = lane + 64 i -> halo pixel q / CPP, 16-byte piece q % CPP
  int goff[NLD], loff[NLD], hyx[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int q = lane + 64 * i;
    const int hp = q / CPP, cc = q - hp * CPP, hy = hp / 10, hx = hp - hy * 10;
    goff[i] = (hy * W + hx) * pitch + cc * 8;
    loff[i] = q < NCH ? hp * PSTR + cc * 16 : 100 * PSTR;
    hyx[i] = hy * 16 + hx;
  }
  const int tiles_x = W >> 3, tiles_y = H >> 3, per_image = tiles_x * tiles_y;
  const int total = n_images * per_image;
  const int stride = gridDim.x * 4;
  auto tile_of = [&](int t, int &b, int &ty, int &tx) {
    b = t / per_image;
    const int r = t - b * per_image;
    ty = r / tiles_x;
    tx = r - ty * tiles_x;
  };
  auto load_halo = [&](int t, f16x8 (&regs)[NLD]) {
    int b, ty, tx;
    tile_of(t, b, ty, tx);
    const int y0 = ty * 8 - 1, x0 = tx * 8 - 1;
    const f16 *base = in + ((ptrdiff_t)((size_t)b * H) + y0) * (ptrdiff_t)W * pitch + (ptrdiff_t)x0 * pitch;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int iy = y0 + (hyx[i] >> 4), ix = x0 + (hyx[i] & 15);
      const bool ok = lane + 64 * i < NCH && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      // out-of-image pixels read a zero line: an UNCONDITIONAL load from a selected address (a load behind a branch is
      // waited for inside the branch: seven serial round trips per tile, 6.5 us where the tile's MFMAs take 0.6)
      const f16 *src = ok ? base + goff[i] : g_conv_zero;
      regs[i] = *reinterpret_cast<const f16x8 *>(src);
    }
  };

  const int a_base = ((r32 >> 3) * 10 + (r32 & 7)) * PSTR + 16 * hk;   // this lane's pixel of M-fragment 0, tap (0, 0)
  int t = blockIdx.x * 4 + wid;
  f16x8 pre[NLD];
  if (t < total) load_halo(t, pre);
  for (; t < total; t += stride) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<f16x8 *>(my + loff[i]) = pre[i];
    if (t + stride < total) load_halo(t + stride, pre);   // in flight under this tile's MFMAs

    f32x16 acc[2][NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = ((tap / 3) * 10 + tap % 3) * PSTR + 32 * ks;
        const f16x8 a0 = *reinterpret_cast<const f16x8 *>(my + a_base + off);
        const f16x8 a1 = *reinterpret_cast<const f16x8 *>(my + a_base + off + 40 * PSTR);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          f16x8 b;
          if constexpr (W_LDS) b = *reinterpret_cast<const f16x8 *>(wl + (((tap * KS + ks) * 2 + hk) * COUT + 32 * nf + r32) * 16);
          else b = fb[tap][ks][nf];
          acc[0][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a0, acc[0][nf], 0, 0, 0);
          acc[1][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a1, acc[1][nf], 0, 0, 0);
        }
      }

    // ---- epilogue.  Swapped-operand C/D layout: this lane holds pixel 32 mf + r32 of the tile and channels
    // 32 nf + 8 g + 4 hk + e (register 4 g + e); v_permlane32_swap pairs the half-waves' quads into 8 consecutive
    // channels per lane (gemm_common.h gemm_epilogue_swap)
    int b, ty, tx;
    tile_of(t, b, ty, tx);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      unsigned char *dst;
      bool writer = true;
      if constexpr (POOL) {
        const int Ho = H >> 1, Wo = W >> 1;
        const int oy = ty * 4 + 2 * mf + (r32 >> 4), ox = tx * 4 + ((r32 & 7) >> 1);
        dst = reinterpret_cast<unsigned char *>(out + (((size_t)b * Ho + oy) * Wo + ox) * out_pitch) + 16 * hk;
        writer = (r32 & 9) == 0;     // even column, even row of the 2 x 2 block
      } else {
        const int gy = ty * 8 + 4 * mf + (r32 >> 3), gx = tx * 8 + (r32 & 7);
        dst = reinterpret_cast<unsigned char *>(out + (((size_t)b * H + gy) * W + gx) * out_pitch) + 16 * hk;
      }
      auto pack4 = [&](int nf, int g, unsigned &lo, unsigned &hi) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mf][nf][4 * g + e];
        v += bias4[nf][g];
        f16 h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (f16)fmaxf(v[e], 0.f);
        if constexpr (POOL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = (float)h[e];
            const float s = x + dpp_f32<0xB1>(x);       // + the pixel beside it (lane ^ 1)
            const float q = s + dpp_f32<0x128>(s);      // + the row below / above (lane ^ 8: row_ror:8)
            h[e] = (f16)(q * 0.25f);
          }
        }
        const f16x2 p0 = {h[0], h[1]}, p1 = {h[2], h[3]};
        lo = __builtin_bit_cast(unsigned, p0);
        hi = __builtin_bit_cast(unsigned, p1);
      };
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          unsigned ax, ay, bx, by;
          pack4(nf, k, ax, ay);
          pack4(nf, k + 1, bx, by);
          const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          const u32x4 o = {rx[0], ry[0], rx[1], ry[1]};
          if (writer) *reinterpret_cast<u32x4 *>(dst + (32 * nf + 8 * k) * 2) = o;
        }
    }
  }
}

// The tower's first convolution: 3 -> 32 channels, 3 x 3, stride 2, pad 1, + bias + ReLU, on NHWC fp16 [n][H][W][3] (pitch
// 3: pixels are 6 bytes).  K = 27 in the order (kh, kw, c), padded to 32 here and to 64 in the GEMM it replaces (im2col
// matrix of 64 halfs per output pixel -- 1.6 GB per 1024 images -- then a 128-column GEMM for 32 real ones: 2.2 ms; this
// kernel reads the image once and writes the activation once).  A wave owns an 8 x 8 output tile = 17 x 17 input pixels,
// staged in LDS as 17 rows of 18 pixels (one more on the left: rows then start on a dword); every lane assembles its two
// A fragments per M-fragment from 16 two-byte LDS reads; four MFMAs per tile.  Same MFMA, same K order, zeros where the GEMM
// has zeros: bit-identical to im2col + GEMM.
__global__ __launch_bounds__(256, 2) void conv3x3_rgb_s2_kernel(const f16 *__restrict__ in, int H, int W,
                                                                const f16 *__restrict__ wt, int kpad,
                                                                const float *__restrict__ bias, f16 *__restrict__ out,
                                                                int out_pitch, int n_images) {
  constexpr int RSTR = 54 * 2 + 4;        // bytes per staged row: 18 pixels x 3 halfs (+ 4: rows fall on different banks)
  constexpr int HALO = 17 * RSTR + 4;     // (+ the slot the idle lanes of the last sweep write)
  constexpr int NDW = 17 * 27, NLD = (NDW + 63) / 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * ((HALO + 15) & ~15)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hk = lane >> 5;
  unsigned char *my = smem + wid * ((HALO + 15) & ~15);

  f16x8 fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fb[ks] = *reinterpret_cast<const f16x8 *>(wt + (size_t)r32 * kpad + 16 * ks + 8 * hk);
  f32x4 bias4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4 *>(bias + 8 * g + 4 * hk);

  // LDS byte offset of element e of k-step ks of this lane's A fragment, from the lane's pixel: k = 16 ks + 8 hk + e =
  // 3 tap + c -> staged row kh, staged pixel 2 px + kw + 1 (the staged row starts two pixels left of the tile)
  int aoff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 8 * hk + e, tap = k / 3, c = k - 3 * tap, kh = tap / 3, kw = tap - 3 * kh;
      aoff[ks][e] = k < 27 ? kh * RSTR + ((kw + 1) * 3 + c) * 2 : -1;
    }
  const int a_base = 2 * (r32 >> 3) * RSTR + 2 * (r32 & 7) * 6;   // output pixel (py, px) of M-fragment 0 -> input (2 py, 2 px)

  int goff[NLD], loff[NLD], rj[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int q = lane + 64 * i, r = q / 27, j = q - r * 27;
    goff[i] = r * W * 3 + 2 * j;                      // halfs from the staged region's first element
    loff[i] = q < NDW ? r * RSTR + 4 * j : 17 * RSTR;
    rj[i] = r * 32 + j;
  }
  const int Ho = H >> 1, Wo = W >> 1;
  const int tiles_x = Wo >> 3, tiles_y = Ho >> 3, per_image = tiles_x * tiles_y;
  const int total = n_images * per_image;
  const int stride = gridDim.x * 4;
  auto tile_of = [&](int t, int &b, int &ty, int &tx) {
    b = t / per_image;
    const int r = t - b * per_image;
    ty = r / tiles_x;
    tx = r - ty * tiles_x;
  };
  auto load_halo = [&](int t, unsigned (&regs)[NLD]) {
    int b, ty, tx;
    tile_of(t, b, ty, tx);
    const int y0 = ty * 16 - 1, x0 = tx * 16 - 2;
    const f16 *base = in + (((ptrdiff_t)((size_t)b * H) + y0) * (ptrdiff_t)W + x0) * 3;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int iy = y0 + (rj[i] >> 5), j = rj[i] & 31;
      // (the region never leaves the image on the right or at the bottom: 16 tx + 15 <= W - 1)
      const bool ok = lane + 64 * i < NDW && iy >= 0 && (x0 >= 0 || j >= 3);
      const f16 *src = ok ? base + goff[i] : g_conv_zero;
      regs[i] = *reinterpret_cast<const unsigned *>(src);
    }
  };

  int t = blockIdx.x * 4 + wid;
  unsigned pre[NLD];
  if (t < total) load_halo(t, pre);
  for (; t < total; t += stride) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32_lds *>(my + loff[i]) = pre[i];
    if (t + stride < total) load_halo(t + stride, pre);

    f32x16 acc[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f16 v = *reinterpret_cast<const f16_lds *>(my + a_base + mf * 8 * RSTR + (aoff[ks][e] < 0 ? 0 : aoff[ks][e]));
          a[e] = aoff[ks][e] < 0 ? (f16)0 : v;
        }
        acc[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks], a, acc[mf], 0, 0, 0);
      }
    }

    int b, ty, tx;
    tile_of(t, b, ty, tx);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int gy = ty * 8 + 4 * mf + (r32 >> 3), gx = tx * 8 + (r32 & 7);
      unsigned char *dst = reinterpret_cast<unsigned char *>(out + (((size_t)b * Ho + gy) * Wo + gx) * out_pitch) + 16 * hk;
      auto pack4 = [&](int g, unsigned &lo, unsigned &hi) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mf][4 * g + e];
        v += bias4[g];
        const f16x2 p0 = {(f16)fmaxf(v[0], 0.f), (f16)fmaxf(v[1], 0.f)}, p1 = {(f16)fmaxf(v[2], 0.f), (f16)fmaxf(v[3], 0.f)};
        lo = __builtin_bit_cast(unsigned, p0);
        hi = __builtin_bit_cast(unsigned, p1);
      };
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        unsigned ax, ay, bx, by;
        pack4(k, ax, ay);
        pack4(k + 1, bx, by);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        const u32x4 o = {rx[0], ry[0], rx[1], ry[1]};
        *reinterpret_cast<u32x4 *>(dst + 8 * k * 2) = o;
      }
    }
  }
}

inline int cu_count() {
  static const int v = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return v;
}

template <int CIN, int COUT, bool POOL>
int launch_direct(const f16 *in, int n, int H, int W, int pitch, const f16 *wt, int kpad, const float *bias, f16 *out,
                  int ldo, hipStream_t st) {
  const int tiles = n * (H / 8) * (W / 8);
  // waves are persistent: as many workgroups as fit a CU at once (LDS: 4 halos [+ the weights]: 32 / 69 / 131 KiB; two waves per SIMD)
  const int per_cu = CIN == 64 ? 1 : 2;
  int grid = cu_count() * per_cu;
  if (grid > (tiles + 3) / 4) grid = (tiles + 3) / 4;
  conv3x3_direct_kernel<CIN, COUT, POOL><<<grid, 256, 0, st>>>(in, H, W, pitch, wt, kpad, bias, out, ldo, n);
  return check_launch();
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" int lla_conv3x3_direct_relu_f16(const void *in, int n, int H, int W, int pitch, int cin, const void *weights,
                                           int kpad, const void *bias, void *out, int ldo, int cout, int pool,
                                           void *stream) {
  if (n < 0 || !in || !weights || !bias || !out) return LLA_EINVAL;
  if (n == 0) return LLA_OK;
  if (H <= 0 || W <= 0 || (H & 7) || (W & 7) || pitch < cin || (pitch & 7) || ldo < cout || (ldo & 7) || kpad < 9 * cin || (kpad & 7))
    return LLA_EINVAL;
  if ((size_t)12 * W * pitch >= (1ull << 31)) return LLA_EINVAL;   // 32-bit element offsets inside a tile's halo
  const f16 *x = reinterpret_cast<const f16 *>(in), *w = reinterpret_cast<const f16 *>(weights);
  const float *b = reinterpret_cast<const float *>(bias);
  f16 *y = reinterpret_cast<f16 *>(out);
  hipStream_t st = as_stream(stream);
  if (cin == 32 && cout == 32 && !pool) return launch_direct<32, 32, false>(x, n, H, W, pitch, w, kpad, b, y, ldo, st);
  if (cin == 32 && cout == 64 && !pool) return launch_direct<32, 64, false>(x, n, H, W, pitch, w, kpad, b, y, ldo, st);
  if (cin == 32 && cout == 64 && pool) return launch_direct<32, 64, true>(x, n, H, W, pitch, w, kpad, b, y, ldo, st);
  if (cin == 64 && cout == 64 && !pool) return launch_direct<64, 64, false>(x, n, H, W, pitch, w, kpad, b, y, ldo, st);
  return LLA_EINVAL;
}

extern "C" int lla_conv3x3_rgb_s2_relu_f16(const void *in, int n, int H, int W, const void *weights, int kpad, const void *bias,
                                           void *out, int ldo, void *stream) {
  if (n < 0 || !in || !weights || !bias || !out) return LLA_EINVAL;
  if (n == 0) return LLA_OK;
  if (H <= 0 || W <= 0 || (H & 15) || (W & 15) || ldo < 32 || (ldo & 7) || kpad < 32 || (kpad & 7)) return LLA_EINVAL;
  if ((size_t)20 * W * 3 >= (1ull << 31)) return LLA_EINVAL;
  const int tiles = n * (H / 16) * (W / 16);
  int grid = cu_count() * 2;
  if (grid > (tiles + 3) / 4) grid = (tiles + 3) / 4;
  conv3x3_rgb_s2_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f16 *>(in), H, W,
                                                             reinterpret_cast<const f16 *>(weights), kpad,
                                                             reinterpret_cast<const float *>(bias),
                                                             reinterpret_cast<f16 *>(out), ldo, n);
  return check_launch();
}
