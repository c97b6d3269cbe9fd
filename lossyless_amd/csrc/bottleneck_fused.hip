// One bottleneck of the RN50 tower's layer1 (56 x 56, planes = 64) as ONE kernel: conv1 (1x1, 256 -> 64) + ReLU, conv2
// (3x3, 64 -> 64) + ReLU, conv3 (1x1, 64 -> 256) + identity + ReLU -- clip/model.py Bottleneck as loaded at
// lossyless/architectures.py:367-371, BatchNorm folded at pack time, fp16 NHWC in and out, fp32 accumulation, the two
// 64-channel intermediates rounded to fp16 exactly where the three-kernel path stores them -- but they never leave the CU.
// (CIN = 64: the stage's first block, conv3 and the downsample convolution as one product over [t2 | x]; see the kernel.)
//
// Why: at 56 x 56 the three kernels are HBM-bound (profiles/r06_rn50_layer_table_three_kernels.txt: 2.06 + 0.82 + 3.70 GB per
// 1024 images at 4.3-5.4 TB/s = 1.45 ms per block); fused, a block reads its input once and writes its output once (3.3 GB +
// halo).  What bounds the fused kernel is the CU's vector-memory path: ~64-68 clocks per 64-lane x 16-byte instruction and CU,
// global loads, LDS-DMAs and stores alike, whether 32 or 256 workgroups run (profiles/r06_rn50_fused_bottleneck.txt) -- so every
// byte goes through it ONCE, and the epilogues' arithmetic went to the matrix pipe where it could.
//
// Shape of the kernel (256 threads = one wave per SIMD, one workgroup per CU, persistent over a contiguous range of tiles):
//   * a tile is 14 x 14 output pixels; its 16 x 16 halo is exactly the 256-row M of conv1, recomputed per tile (conv1 is
//     24 % of the block's FLOPs: +7 % work for no intermediate in HBM);
//   * x streams through LDS in chunks of 32 channels (global_load_lds_dwordx4, XOR-swizzled on the source side so that
//     fragment reads are conflict-free): chunks 0-4 of the NEXT tile go out between the MFMAs of conv2 (chunk 3 under conv1's
//     tail) into five 16-KiB buffers, chunks 5 and 6 into the idle t1 region at the top of their tile, chunk 7 into buffer 0
//     once chunk 0 is consumed;
//   * conv1: the waves split the halo pixels (64 each), W1 is LDS-resident in fragment order; t1 = ReLU(.) goes to LDS as
//     [pixel][64 + 8 halfs], zero where the halo leaves the image (conv2's padding);
//   * conv2: M = 14 rows x 16 columns (two garbage columns per row keep a tap a constant row shift); a wave owns one half
//     of the output channels for every second 32-pixel block, and its half of W2 (36 KiB: 144 registers of the 512 a lone
//     wave per SIMD owns) stays in REGISTERS for the whole kernel; t2 overwrites t1;
//   * conv3: wave w owns the output chunks (32 channels) w and w + 4 (W3 slice: 32 registers), so that the identity rows it
//     adds are the INPUT chunks w and w + 4: every wave copies them out of LDS into registers (112, as MFMA B fragments) with
//     the same instructions while the chunks are there -- no second read of x -- and conv3 adds them on the MATRIX pipe (two
//     MFMAs per block against a one-hot fragment: exact), not with 64 conversions and adds per lane and block;
//   * MFMA rows are PERMUTED output channels so that a lane's 16 accumulators are 16 CONSECUTIVE channels of one pixel:
//     16-byte LDS writes and global stores throughout; the bias is the C operand of a block's first MFMA (conv2, conv3) or the
//     accumulators' start (conv1); ReLU on packed halfs after the rounding; fragment reads are register-double-buffered by
//     hand (sched_barrier).
#include "common.h"

namespace lla {
namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct BottleneckParams {
  const f16 *x; int pitch;            // [n][H][W][pitch], first CIN channels
  const f16 *w1; int k1pad; const float *b1;
  const f16 *w2; int k2pad; const float *b2;
  const f16 *w3; int k3pad; const float *b3;
  f16 *out; int ldo;                  // [n][H][W][ldo], first 256 channels; must not overlap x
  int n, H, W;
  int x_last_pixel;                   // byte offset of the tensor's last pixel (source clamp of the halo loads)
};

constexpr int kTile = 14;    // (halo: 16 x 16)
constexpr int kT1Stride = 144;                    // bytes per t1 / t2 pixel: 64 halfs + 16 (16 consecutive pixels cover all banks)
constexpr int kT1Rows = 264;                      // conv2's garbage columns read up to row 223 + 34
constexpr int kBufBytes = 256 * 64;               // one chunk: 256 halo pixels x 32 channels
constexpr int kBufs = 5;
constexpr int kW1Off = 0;                         // W1: [k-step][half][64 rows][16 B]
constexpr int kBiasOff = 32768;                   // b1[64] b2[64] b3[256] fp32
constexpr int kT1Off = kBiasOff + 384 * 4;
constexpr int kBufOff = kT1Off + kT1Rows * kT1Stride;
constexpr int kLdsBytes = kBufOff + kBufs * kBufBytes;
static_assert(kLdsBytes <= 160 * 1024 && kT1Off % 16 == 0 && kBufOff % 16 == 0, "LDS map");
// Where chunk c of a 256-channel tile lives: chunks 0-4 in the five buffers (prefetched under the previous tile); chunks 5 and 6
// in the t1 / t2 region, which is idle from the end of the previous tile's conv3 to this tile's conv1 epilogue -- they are
// requested at the very top of the tile; chunk 7 in buffer 0 once chunk 0 is consumed.
constexpr int chunk_off(int c) { return c < kBufs ? kBufOff + c * kBufBytes : c < 7 ? kT1Off + (c - 5) * kBufBytes : kBufOff; }
static_assert(2 * kBufBytes <= kT1Rows * kT1Stride, "chunks 5 and 6 in the t1 region");

__device__ __forceinline__ int perm_row(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }

// 64 lanes x 16 bytes from (sbase + voff) to LDS lds_dst + 16 lane.  M0 is saved / restored: it belongs to the compiler.
__device__ __forceinline__ void bn_dma(unsigned voff, const void *sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %3\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void bn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the same without the compiler fence, for DMAs issued between the MFMAs of a phase that does not touch their buffers (the
// fragment reads of that phase may then be scheduled across them; volatile keeps them in order with barriers and waits)
__device__ __forceinline__ void bn_dma_nofence(unsigned voff, const void *sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %3\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst));
}

// accumulators START at the bias: register t of a lane <-> channel (block base) + 16 hk + t
__device__ __forceinline__ f32x16 bias16(const unsigned char *bias_lds) {
  f32x16 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[4 * q + e] = b[e];
  }
  return a;
}
// (ReLU AFTER the rounding, on packed halfs: max(round(v), 0) == round(max(v, 0)); two VALU operations per value instead of three)
__device__ __forceinline__ f16x8 relu_pack8(const f32x16 &acc, int half, bool keep) {
  typedef f16 f16x2 __attribute__((ext_vector_type(2)));
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f16x2 v = {(f16)acc[8 * half + e], (f16)acc[8 * half + e + 1]};
    const f16x2 z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    o[e] = v[0];
    o[e + 1] = v[1];
  }
  const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  return keep ? o : zero;
}

// DBG (tools/ builds only: make tuvariant TU=bottleneck_fused NAME=.. DEFS=-DLLA_BN_DBG=..; WRONG results): 9 = shader-clock
// stamps between the phases, summed over the workgroup's tiles and written over the first output bytes (64 B per wave).
//
// CIN = 64 is layer1's FIRST block (its input is the stem's 64 channels; conv3 and the downsample convolution are one 1x1
// convolution over [t2 | x], K = 128, no identity: rn50.hip Layout::fused): both chunks of a tile are prefetched, stay in LDS
// until conv3 has multiplied them, and the tiles alternate between buffers {0, 1} and {2, 3}.
template <int CIN, int DBG>
__global__ __launch_bounds__(256, 1) void bottleneck14_kernel(BottleneckParams p) {
  constexpr bool CAT = CIN == 64;
  constexpr int KS1 = CIN / 16, NCH = CIN / 32;     // conv1 k-steps, 32-channel chunks per tile
  constexpr int KS3 = CAT ? 8 : 4;                  // conv3 k-steps
  constexpr int NPRE = NCH < kBufs ? NCH : kBufs;   // chunks prefetched under the previous tile
  static_assert((NCH == 8 || NCH == 2) && kBufs == 5, "chunk schedule");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), r32 = lane & 31, hk = lane >> 5;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

  // ---- resident weights
  const int mu = wid & 1, nu = wid >> 1;     // conv2: this wave's pixel-block parity and its half of the 64 output channels
  // conv3: wave w owns the 32-channel chunks w and w + 4 of the 256 outputs (block i of its accumulators = chunk w + 4 i, a lane's
  // 16 registers = 16 consecutive channels of it), so that the identity rows it adds (CIN = 256) are the INPUT chunks w (among
  // 0-3: copied at the top of the tile) and w + 4 (among 4-7: copied once chunk 7 has landed): EVERY wave copies them out of LDS
  // with the same instructions while those chunks are there.
  auto out_channel = [&](int i, int r) { return 32 * (wid + 4 * i) + perm_row(r); };
  f16x8 w2f[9][4], w3f[KS3][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      w2f[tap][ks] = *reinterpret_cast<const f16x8 *>(p.w2 + (size_t)(32 * nu + perm_row(r32)) * p.k2pad + tap * 64 + 16 * ks + 8 * hk);
#pragma unroll
  for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      w3f[ks][i] = *reinterpret_cast<const f16x8 *>(p.w3 + (size_t)out_channel(i, r32) * p.k3pad + 16 * ks + 8 * hk);
  f16x8 eye[2];                                  // one-hot rows: MFMA row r32 of a block is channel perm_row(r32) of its chunk
#pragma unroll
  for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
    for (int j = 0; j < 8; ++j) eye[sidx][j] = perm_row(r32) == 16 * sidx + 8 * hk + j ? (f16)1 : (f16)0;
  for (int q = tid; q < KS1 * 2 * 64; q += 256) {
    const int row = q & 63, h = (q >> 6) & 1, ks = q >> 7;
    *reinterpret_cast<f16x8 *>(smem + kW1Off + q * 16) =
        *reinterpret_cast<const f16x8 *>(p.w1 + (size_t)((row & 32) + perm_row(row & 31)) * p.k1pad + 16 * ks + 8 * h);
  }
  for (int q = tid; q < 384; q += 256)
    reinterpret_cast<float *>(smem + kBiasOff)[q] = q < 64 ? p.b1[q] : q < 128 ? p.b2[q - 64] : p.b3[q - 128];
  for (int q = tid; q < (kT1Rows - 256) * kT1Stride / 4; q += 256) reinterpret_cast<unsigned *>(smem + kT1Off + 256 * kT1Stride)[q] = 0u;

  // ---- tiles of this workgroup: a contiguous range (an image's 16 tiles run back to back on one CU: halo re-reads hit its L2)
  const int tiles_x = p.W / kTile, per_image = tiles_x * (p.H / kTile), total = p.n * per_image;
  const int t_begin = (int)((long long)total * blockIdx.x / gridDim.x), t_end = (int)((long long)total * (blockIdx.x + 1) / gridDim.x);
  if (t_begin >= t_end) return;
  const int pix_bytes = p.pitch * 2, row_bytes = p.W * pix_bytes;

  // halo loads: instruction i of this lane fetches 16 bytes of halo pixel (4 i + wid, (tid >> 2) & 15), LDS slot tid & 3
  const int hx_l = (tid >> 2) & 15;
  const int seg_off = ((tid & 3) ^ ((tid >> 3) & 3)) * 16;
  int pix[4];
  auto tile_origin = [&](int t, int &img, int &y0, int &x0) {
    img = t / per_image;
    const int r = t - img * per_image, ty = r / tiles_x;
    y0 = ty * kTile - 1;
    x0 = (r - ty * tiles_x) * kTile - 1;
  };
  auto set_pix = [&](int t) {
    int img, y0, x0;
    tile_origin(t, img, y0, x0);
    const int base = (img * p.H + y0) * row_bytes + (x0 + hx_l) * pix_bytes;   // (may be negative: clamped below)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int o = base + (4 * i + wid) * row_bytes;
      o = o < 0 ? 0 : (o > p.x_last_pixel ? p.x_last_pixel : o);
      pix[i] = o + seg_off;
    }
  };
  auto issue_chunk = [&](int c) {
    const unsigned dst = lds_base + chunk_off(c) + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) bn_dma((unsigned)(pix[i] + c * 64), p.x, dst + i * 4096);
  };

  set_pix(t_begin);
#pragma unroll
  for (int c = 0; c < NPRE; ++c) issue_chunk(c);
  bn_wait_vm<0>();

  const int swz = (r32 >> 1) & 3, swz_id = ((r32 + 17) >> 1) & 3;
  const unsigned char *bias = smem + kBiasOff;
  unsigned long long stamp = 0, phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto mark = [&](int k) {
    if constexpr (DBG == 9) {
      const unsigned long long now = __builtin_readcyclecounter();
      phase[k] += now - stamp;
      stamp = now;
    }
  };
  for (int t = t_begin; t < t_end; ++t) {
    int img, y0, x0;
    tile_origin(t, img, y0, x0);
    if constexpr (DBG == 9) stamp = __builtin_readcyclecounter();
    const int bofs = CAT ? ((t - t_begin) & 1) * 2 * kBufBytes : 0;   // this tile's buffer set (CIN = 64)
    // every load older than the previous tile's 28 output stores has landed (this tile's chunks 0-4 among them)
    bn_wait_vm<28>();
    __syncthreads();
    mark(0);

    // conv3's output pixel of block blk for this lane: (2 blk + (r32 >> 4), r32 & 15); the two garbage columns are not stored
    const int ox = r32 & 15;
    const bool store = ox < kTile;
    const long long opix = (long long)(img * p.H + y0 + 1 + (r32 >> 4)) * p.W + (x0 + 1 + ox);
    unsigned char *outp = reinterpret_cast<unsigned char *>(p.out) + opix * p.ldo * 2 + (32 * wid + 16 * hk) * 2;   // + 256 i + 16 h
    // the identity of conv3 (CIN = 256) comes out of the SAME chunks, as MFMA B fragments: interior pixel m = halo pixel m + 17,
    // k-step s of chunk wid + 4 i = its piece 2 s + hk (channels 16 s + 8 hk .. + 7); conv3 ADDS it on the matrix pipe -- two more
    // MFMAs per block against a one-hot weight fragment, exact (1.0 x value into the fp32 accumulator: one rounding, as the fp32
    // add of the three-kernel epilogue) -- instead of 64 conversions and adds per lane and block on the VALU, which conv3's
    // epilogue was bound by.  Chunks 0-3 (i = 0) are copied at the top of the tile, 4-7 (i = 1) once chunk 7 has landed.
    f16x8 idr[7][2][2];
    auto grab_identity = [&](int i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int off = i == 0 ? chunk_off(0) + wid * kBufBytes
                               : wid == 0 ? chunk_off(4) : wid == 1 ? chunk_off(5) : wid == 2 ? chunk_off(6) : chunk_off(7);
        const unsigned char *src = smem + off + (r32 + 17) * 64 + (((2 * h + hk) ^ swz_id) * 16);
#pragma unroll
        for (int blk = 0; blk < 7; ++blk) idr[blk][i][h] = *reinterpret_cast<const f16x8 *>(src + 32 * blk * 64);
      }
    };
    if constexpr (!CAT) {
      issue_chunk(5);                                              // (into the idle t1 region: requested before anything else)
      issue_chunk(6);
      grab_identity(0);
    }

    // ---------------- conv1: halo pixels 64 wid + 32 j + r32, all 64 channels
    f32x16 acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc1[i][0] = bias16(bias + (32 * i + 16 * hk) * 4);
      acc1[i][1] = acc1[i][0];
    }
    // k-step ks = 2 c + s of chunk c: fragments of step ks + 1 are read BEFORE the MFMAs of step ks (register double buffer,
    // pinned with sched_barrier: hipcc otherwise hoists reads until the register file spills); the pipeline restarts where a
    // barrier separates chunks (after chunk 0: its buffer takes chunk 7; before chunks 5 and 7: they have landed)
    f16x8 fa[2][2], fb[2][2];
    auto read1 = [&](int ks, int bufsel) {
      const unsigned char *buf = smem + (CAT ? kBufOff + bofs + (ks >> 1) * kBufBytes : chunk_off(ks >> 1));
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[bufsel][i] = *reinterpret_cast<const f16x8 *>(smem + kW1Off + ((ks * 2 + hk) * 64 + 32 * i + r32) * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[bufsel][j] = *reinterpret_cast<const f16x8 *>(buf + (64 * wid + 32 * j + r32) * 64 + (((2 * (ks & 1) + hk) ^ swz) * 16));
    };
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      if (!CAT && ks == 2) {
        __syncthreads();                                           // chunk 0 is consumed (and copied): its buffer takes chunk 7
        issue_chunk(7);
      }
      if (!CAT && ks == 10) {
        bn_wait_vm<4>();                                           // chunks 5 and 6 have landed (7 may still fly)
        __syncthreads();
      }
      if (!CAT && ks == 14) {
        bn_wait_vm<0>();
        __syncthreads();
        grab_identity(1);
        // chunk 3's buffer has been free since the barrier before chunk 5: the next tile's chunk 3 goes out here, under this
        // tile's last MFMAs and its epilogue (the other four chunks between the MFMAs of conv2)
        set_pix(t + 1 < t_end ? t + 1 : t);                        // (the last tile re-fetches itself: no branch around the DMAs)
        issue_chunk(3);
      }
      if (ks == 0 || (!CAT && (ks == 2 || ks == 10 || ks == 14))) read1(ks, ks & 1);
      if (ks != KS1 - 1 && (CAT || (ks != 1 && ks != 9 && ks != 13))) read1(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][i], fb[ks & 1][j], acc1[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!CAT) __syncthreads();                           // chunks 5 and 6 are consumed and copied: t1 may overwrite them
    mark(1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hp = 64 * wid + 32 * j + r32, iy = y0 + (hp >> 4), ix = x0 + (hp & 15);
      const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<f16x8 *>(smem + kT1Off + hp * kT1Stride + (32 * i + 16 * hk + 8 * h) * 2) = relu_pack8(acc1[i][j], h, inside);
    }
    __syncthreads();                                               // t1 complete; (CIN = 256) every chunk buffer is free
    mark(2);

    // the next tile's chunks 0-4 go out BETWEEN the MFMAs of conv2, one instruction per window of four: a burst would hold
    // the wave at the vector-memory queue (~64 clocks per instruction and CU) with the matrix pipe idle
    if constexpr (CAT) set_pix(t + 1 < t_end ? t + 1 : t);         // (the last tile re-fetches itself: no branch around the DMAs)
    const unsigned dma_dst = lds_base + kBufOff + wid * 1024 + (CAT ? 2 * kBufBytes - bofs : 0);   // (CIN = 64: the OTHER buffer set)
    mark(3);

    // ---------------- conv2: output channels 32 nu .. 32 nu + 31 of the 32-pixel blocks mu, mu + 2, mu + 4 (, 6) of the 14 x 16
    // output rows, one block at a time (a dependent MFMA chain on one accumulator issues back to back; the register file is full)
    f16x8 h2[4][2];
    {
      const unsigned char *t1p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t1p[u] = smem + kT1Off + (32 * (mu + 2 * u < 7 ? mu + 2 * u : 6) + r32) * kT1Stride + 8 * hk * 2;   // (wave mu = 1 has three blocks: the fourth repeats block 6)
      // window w = (block u, tap, half of the tap's four k-steps): the fragments of window w + 1 are read before the MFMAs of
      // window w
      f16x8 bf[2][2];
      auto read2 = [&](int w, int bufsel) {
        const int u = w / 18, tap = (w % 18) >> 1, k0 = 2 * (w & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bf[bufsel][ks] = *reinterpret_cast<const f16x8 *>(t1p[u] + (16 * (tap / 3) + tap % 3) * kT1Stride + 16 * (k0 + ks) * 2);
      };
      f32x16 acc2;
      const f32x16 bias2 = bias16(bias + (64 + 32 * nu + 16 * hk) * 4);
      read2(0, 0);
#pragma unroll
      for (int w = 0; w < 72; ++w) {
        const int u = w / 18, tap = (w % 18) >> 1, k0 = 2 * (w & 1);
        if (w + 1 < 72) read2(w + 1, (w + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)   // (a block's first MFMA takes the bias as its C operand)
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[tap][k0 + ks], bf[w & 1][ks], w % 18 == 0 && ks == 0 ? bias2 : acc2, 0, 0, 0);
        if (CAT ? (w < 8 * NPRE && !(w & 1)) : (w & 1) && (4 * (w >> 1)) % 9 < 4) {   // CIN = 256: 16 of the 36 taps (chunks 0, 1, 2, 4)
          const int k = CAT ? w >> 1 : (4 * (w >> 1)) / 9, c = CAT ? k >> 2 : (k >> 2) + (k >> 2 == 3), i = k & 3;
          bn_dma_nofence((unsigned)(pix[i] + c * 64), p.x, dma_dst + c * kBufBytes + i * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (w % 18 == 17) {
#pragma unroll
          for (int h = 0; h < 2; ++h) h2[u][h] = relu_pack8(acc2, h, true);
        }
      }
    }
    mark(4);
    __syncthreads();                                               // every wave has read its taps: t2 may overwrite t1
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (mu + 2 * u < 7)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<f16x8 *>(smem + kT1Off + (32 * (mu + 2 * u) + r32) * kT1Stride + (32 * nu + 16 * hk + 8 * h) * 2) = h2[u][h];
    __syncthreads();
    mark(5);

    // ---------------- conv3: output channels 64 wid .. 64 wid + 63 of all seven blocks, + identity (CIN = 256), ReLU.
    // CIN = 64: k-steps 4-7 multiply the tile's own input pixels out of the chunk buffers (output pixel m = halo pixel m + 17)
    f16x8 cf[2][KS3];
    f32x16 bias3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bias3[i] = bias16(bias + (128 + 32 * (wid + 4 * i) + 16 * hk) * 4);
    auto read3 = [&](int blk, int bufsel) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cf[bufsel][ks] = *reinterpret_cast<const f16x8 *>(smem + kT1Off + (32 * blk + r32) * kT1Stride + (16 * ks + 8 * hk) * 2);
      if constexpr (CAT) {
#pragma unroll
        for (int ks = 4; ks < 8; ++ks)
          cf[bufsel][ks] = *reinterpret_cast<const f16x8 *>(smem + kBufOff + bofs + ((ks - 4) >> 1) * kBufBytes + (32 * blk + r32 + 17) * 64 +
                                                            (((2 * (ks & 1) + hk) ^ swz_id) * 16));
      }
    };
    read3(0, 0);
#pragma unroll
    for (int blk = 0; blk < 7; ++blk) {
      f32x16 acc3[2];
      if (blk + 1 < 7) read3(blk + 1, (blk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)      // (the bias is the first MFMA's C operand: resident registers, no copy into the accumulators)
        acc3[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[0][i], cf[blk & 1][0], bias3[i], 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < KS3; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc3[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[ks][i], cf[blk & 1][ks], acc3[i], 0, 0, 0);
      if constexpr (!CAT) {
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc3[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(eye[sidx], idr[blk][i][sidx], acc3[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (store) *reinterpret_cast<f16x8 *>(outp + (size_t)blk * 2 * p.W * p.ldo * 2 + 256 * i + 16 * h) = relu_pack8(acc3[i], h, true);
    }
    mark(6);
  }
  bn_wait_vm<0>();                                                 // (the last tile's self-prefetch is still writing LDS)
  if constexpr (DBG == 9) {
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) reinterpret_cast<unsigned long long *>(p.out)[(blockIdx.x * 4 + wid) * 8 + k] = phase[k];
  }
}

#ifndef LLA_BN_DBG
#define LLA_BN_DBG 0
#endif

inline int bn_cu_count() {
  static const int v = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return v;
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" int lla_rn50_bottleneck_f16(const void *x, int n, int H, int W, int pitch, int cin, const void *w1, int k1pad,
                                       const void *b1, const void *w2, int k2pad, const void *b2, const void *w3, int k3pad,
                                       const void *b3, void *out, int ldo, void *stream) {
  if (n < 0 || !x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) return LLA_EINVAL;
  if (n == 0) return LLA_OK;
  if ((cin != 256 && cin != 64) || H <= 0 || W <= 0 || H % kTile || W % kTile || pitch < cin || (pitch & 7) || ldo < 256 || (ldo & 7) ||
      k1pad < cin || (k1pad & 7) || k2pad < 576 || (k2pad & 7) || k3pad < (cin == 64 ? 128 : 64) || (k3pad & 7))
    return LLA_EINVAL;
  const size_t pixels = (size_t)n * H * W;
  if (pixels * pitch * 2 >= (1ull << 31) || pixels * ldo * 2 >= (1ull << 40)) return LLA_EINVAL;   // 32-bit source offsets of the halo loads
  const char *xb = reinterpret_cast<const char *>(x), *ob = reinterpret_cast<const char *>(out);
  if (xb < ob + pixels * ldo * 2 && ob < xb + pixels * pitch * 2) return LLA_EINVAL;                 // halos are read after neighbours are written
  BottleneckParams p;
  p.x = reinterpret_cast<const f16 *>(x); p.pitch = pitch;
  p.w1 = reinterpret_cast<const f16 *>(w1); p.k1pad = k1pad; p.b1 = reinterpret_cast<const float *>(b1);
  p.w2 = reinterpret_cast<const f16 *>(w2); p.k2pad = k2pad; p.b2 = reinterpret_cast<const float *>(b2);
  p.w3 = reinterpret_cast<const f16 *>(w3); p.k3pad = k3pad; p.b3 = reinterpret_cast<const float *>(b3);
  p.out = reinterpret_cast<f16 *>(out); p.ldo = ldo;
  p.n = n; p.H = H; p.W = W;
  p.x_last_pixel = (int)((pixels - 1) * pitch * 2);
  const int tiles = n * (H / kTile) * (W / kTile);
  int grid = bn_cu_count();
  if (grid > tiles) grid = tiles;
  if (cin == 256) bottleneck14_kernel<256, LLA_BN_DBG><<<grid, 256, 0, as_stream(stream)>>>(p);
  else bottleneck14_kernel<64, LLA_BN_DBG><<<grid, 256, 0, as_stream(stream)>>>(p);
  return check_launch();
}
