// Entropy stage of the compress_dataset hot path on gfx950: quantise, rANS encode,
// stream compaction, rANS decode, dequantise.  Integer / bit-manipulation work, no
// MFMA.  One image per lane: a wave64 carries 64 independent rANS states through
// the 512-symbol dependency chain in lock-step over the channel index, so every
// lane of a wave reads the SAME table row at the same time (LDS broadcast / no
// bank conflicts), and the table (u16 [C][W], <= 34 KB) lives in LDS.
//
// Reference behaviour being replaced (compressai==1.1.5, not vendored):
//   RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes
//   (cpp_exts/rans/rans_interface.cpp, third_party/ryg_rans/rans64.h), called per
//   image from hub/compressor.py:98,124.  Algorithm: SURVEY.md 8(a) A13 / A14.
#include "common.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

namespace lla {

thread_local int g_last_hip_error = 0;

namespace {

constexpr uint32_t kProbBits = 16;
constexpr uint64_t kStateLow = 1ull << 31;
constexpr int kEncThreads = 256;  // 4 waves = 256 images per workgroup

// ---------------------------------------------------------------------------
// table staging: int32 cdf[C][W] (global) -> u16 [C][W] (LDS).  65536 wraps to 0,
// which is harmless: it is only ever used as the upper edge of the last bin and
// (hi - lo) & 0xffff recovers the frequency (< 65536 by construction).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void stage_table(uint16_t *lds, const int32_t *__restrict__ cdf,
                                            int n_entries) {
  // 16-byte loads, four in flight per thread: written as `lds[i] = cdf[i]` the loop waits for every
  // 4-byte load before its ds_write (64 dependent L2 round trips per thread for a 512 x 32 table =
  // ~15 % of the encoder's time at batch 1024)
  const int quads = ((reinterpret_cast<uintptr_t>(cdf) & 15u) == 0) ? n_entries >> 2 : 0;
  const int4 *src = reinterpret_cast<const int4 *>(cdf);
  for (int i0 = threadIdx.x; i0 < quads; i0 += 4 * blockDim.x) {
    int4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      q[u] = i < quads ? src[i] : int4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i < quads) {
        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u16x4 *>(lds + 4 * i) =
            u16x4{(unsigned short)q[u].x, (unsigned short)q[u].y, (unsigned short)q[u].z, (unsigned short)q[u].w};
      }
    }
  }
  for (int i = 4 * quads + threadIdx.x; i < n_entries; i += blockDim.x) lds[i] = (uint16_t)cdf[i];
  __syncthreads();
}

// t / f and t % f for t < f * 2^16 (so the quotient is < 2^16), 1 <= f < 2^16.
// hipcc expands a u32 division into ~40 dependent instructions; here the quotient is
// estimated in fp32 and corrected, ~12 instructions.  Exactness: float(t) and v_rcp_f32 are
// each within 2^-23 relative, so the estimate is within q * 2^-21 < 1/32 of t / f and its
// floor is q - 1, q or q + 1; the remainder test below moves it to q in either case.
__device__ __forceinline__ void divmod16(uint32_t t, uint32_t f, uint32_t &q, uint32_t &r) {
  uint32_t qe = (uint32_t)((float)t * __builtin_amdgcn_rcpf((float)f));
  int32_t rem = (int32_t)(t - qe * f);  // |rem| < 2 f: fits, wrap-around is harmless
  if (rem < 0) { --qe; rem += (int32_t)f; }
  if (rem >= (int32_t)f) { ++qe; rem -= (int32_t)f; }
  q = qe;
  r = (uint32_t)rem;
}

// x / f and x % f for x < 2^47 * f, 1 <= f < 2^16, as three long-division steps in base 2^16
// (each partial dividend is < f * 2^16, which is what divmod16 needs).
// Returns (q << 16) + r, i.e. the rANS state before `start` is added.
__device__ __forceinline__ uint64_t div_step(uint64_t x, uint32_t f) {
  const uint32_t hi = (uint32_t)(x >> 32);
  const uint32_t lo = (uint32_t)x;
  uint32_t q1, r1, q2, r2, q3, r3;
  divmod16(hi, f, q1, r1);
  divmod16((r1 << 16) | (lo >> 16), f, q2, r2);
  divmod16((r2 << 16) | (lo & 0xffffu), f, q3, r3);
  // q = q1 * 2^32 + q2 * 2^16 + q3 ; result = q * 2^16 + r3
  return ((uint64_t)q1 << 48) | ((uint64_t)q2 << 32) | ((uint64_t)q3 << 16) | (uint64_t)r3;
}

struct __attribute__((aligned(16))) ChanParams {
  float bias, es, med;
  int32_t off;
  int32_t len;
  int32_t pad[3];
};
__host__ __device__ inline size_t enc_table_bytes(int C, int W) {
  return ((size_t)C * W * sizeof(uint16_t) + 15u) & ~(size_t)15u;
}
__host__ __device__ inline size_t enc_lds_bytes(int C, int W) {
  return enc_table_bytes(C, W) + (size_t)C * sizeof(ChanParams);
}

struct EncState {
  uint64_t x;
  uint32_t *wp;  // next free word is wp[-1]; stream grows towards lower addresses
};

__device__ __forceinline__ void put_symbol(EncState &s, uint32_t start, uint32_t freq) {
  // x >= ((2^31 >> 16) << 32) * freq = freq << 47; the bound's low word is 0: compare high words
  if ((uint32_t)(s.x >> 32) >= (freq << 15)) {
    *--s.wp = (uint32_t)s.x;
    s.x >>= 32;
  }
  s.x = div_step(s.x, freq) + start;
}

__device__ __forceinline__ void put_digit(EncState &s, uint32_t d) {
  if ((uint32_t)(s.x >> 32) >= (1u << 27)) {  // x >= ((2^31 >> 16) << 32) * 2^12 = 2^59
    *--s.wp = (uint32_t)s.x;
    s.x >>= 32;
  }
  s.x = (s.x << 4) | d;
}

// One channel of one image, in two halves so that the caller can run the state-independent half
// (table look-ups) for a whole group of channels before the serial state updates of that group:
// the LDS round trips of a group then overlap instead of sitting in the 512-step dependency chain.
struct Prepared {
  uint32_t start, freq, raw;
  int nd;  // -1: regular symbol; 0..8: escape with that many 4-bit payload digits
};

template <typename RowT>
__device__ __forceinline__ Prepared prepare_channel(const RowT *row, int len, int off, int32_t sym) {
  const int esc = len - 2;
  const int v0 = sym - off;
  const bool neg = v0 < 0, over = v0 >= esc, escaped = neg || over;
  Prepared p;
  p.raw = neg ? (uint32_t)(-2 * v0 - 1) : (over ? (uint32_t)(2 * (v0 - esc)) : 0u);
  // raw fits 32 bits -> at most 8 digits -> the count is a single digit (< 15)
  const int nd = p.raw ? (35 - __clz((int)p.raw)) >> 2 : 0;
  p.nd = escaped ? nd : -1;
  const int v = escaped ? esc : v0;
  p.start = (uint32_t)row[v];
  p.freq = ((uint32_t)row[v + 1] - p.start) & 0xffffu;  // 65536 (or its u16 wrap 0) minus start
  return p;
}

// Symbols are consumed last-to-first, so for an escaped value the payload digits go in
// most-significant first, then the digit count, then the escape symbol itself -- the mirror
// image of the decoder's read order.
__device__ __forceinline__ void emit_channel(EncState &s, const Prepared &p) {
  if (p.nd >= 0) {
    for (int d = p.nd - 1; d >= 0; --d) put_digit(s, (p.raw >> (4 * d)) & 15u);
    put_digit(s, (uint32_t)p.nd);
  }
  put_symbol(s, p.start, p.freq);
}

__device__ __forceinline__ int32_t quantise_one(float z, float bias, float es, float med) {
  // three separately rounded fp32 operations, then round-half-even
  const float t = __fadd_rn(z, bias);
  const float u = __fmul_rn(t, es);
  const float d = __fsub_rn(u, med);
  return (int32_t)rintf(d);
}

// ZMODE 0: int32 symbols, 1: fp16 z, 2: fp32 z.  G = channels fetched per 16-byte load.
template <int ZMODE>
struct Fetch;
template <>
struct Fetch<0> {
  static constexpr int G = 4;
  using elem = int32_t;
};
template <>
struct Fetch<1> {
  static constexpr int G = 8;
  using elem = __half;
};
template <>
struct Fetch<2> {
  static constexpr int G = 4;
  using elem = float;
};

template <int ZMODE>
__global__ __launch_bounds__(kEncThreads) void rans_encode_kernel(
    const void *__restrict__ in, int B, int C, const float *__restrict__ bias,
    const float *__restrict__ exp_scale, const float *__restrict__ median,
    const int32_t *__restrict__ cdf, int W, const int32_t *__restrict__ cdf_len,
    const int32_t *__restrict__ offset, uint8_t *__restrict__ scratch, size_t stride,
    uint32_t *__restrict__ lengths, int32_t *__restrict__ symbols_out) {
  kernel_acquire();
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
  // per-channel scalars next to the table: as uniform global reads they become s_load_dword,
  // which shares lgkmcnt with the LDS and returns out of order, so every use forced
  // s_waitcnt lgkmcnt(0) -- two scalar-memory round trips per symbol in the serial loop
  ChanParams *par = reinterpret_cast<ChanParams *>(smem + enc_table_bytes(C, W));
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    ChanParams q;
    q.bias = ZMODE ? bias[c] : 0.f;
    q.es = ZMODE ? exp_scale[c] : 0.f;
    q.med = ZMODE ? median[c] : 0.f;
    q.off = offset[c];
    q.len = cdf_len[c];
    par[c] = q;
  }
  stage_table(tab, cdf, C * W);

  const int img = blockIdx.x * kEncThreads + threadIdx.x;
  if (img >= B) return;

  using F = Fetch<ZMODE>;
  constexpr int G = F::G;
  using elem = typename F::elem;
  const elem *src = reinterpret_cast<const elem *>(in) + (size_t)img * C;

  uint8_t *end = scratch + (size_t)img * stride + stride;
  EncState s;
  s.x = kStateLow;
  s.wp = reinterpret_cast<uint32_t *>(end);

  auto prep = [&](int c, elem raw_in) {
    const ChanParams q = par[c];
    int32_t sym;
    if constexpr (ZMODE == 0) {
      sym = raw_in;
    } else if constexpr (ZMODE == 1) {
      sym = quantise_one(__half2float(raw_in), q.bias, q.es, q.med);
    } else {
      sym = quantise_one(raw_in, q.bias, q.es, q.med);
    }
    if (symbols_out) symbols_out[(size_t)img * C + c] = sym;
    return prepare_channel(tab + c * W, q.len, q.off, sym);
  };

  const int tail = C % G;
  for (int c = C - 1; c >= C - tail; --c) emit_channel(s, prep(c, src[c]));
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && ((C - tail) > 0);
  for (int g = (C - tail) / G - 1; g >= 0; --g) {
    elem v[G];
    if (vec_ok) {
      *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(src + g * G);
    } else {
#pragma unroll
      for (int k = 0; k < G; ++k) v[k] = src[g * G + k];
    }
    Prepared pr[G];
#pragma unroll
    for (int k = 0; k < G; ++k) pr[k] = prep(g * G + k, v[k]);
#pragma unroll
    for (int k = G - 1; k >= 0; --k) emit_channel(s, pr[k]);
  }

  s.wp -= 2;
  s.wp[0] = (uint32_t)s.x;
  s.wp[1] = (uint32_t)(s.x >> 32);
  lengths[img] = (uint32_t)(end - reinterpret_cast<uint8_t *>(s.wp));
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
struct DecState {
  uint64_t x;
  const uint32_t *w;
  uint32_t pos, nwords;
};

// (A 16-byte word window reloaded every fourth renormalisation, and four symbols per store, were both
// tried: the extra selects and divergent blocks on the serial per-symbol chain cost more than the
// vector-memory round trips they save -- 2.7 vs 3.1 M img/s at batch 1024.)
__device__ __forceinline__ void refill(DecState &s) {
  if (s.x < kStateLow) {
    const uint32_t word = s.pos < s.nwords ? s.w[s.pos] : 0u;
    s.x = (s.x << 32) | word;
    ++s.pos;
  }
}

__device__ __forceinline__ uint32_t take_digit(DecState &s) {
  const uint32_t d = (uint32_t)s.x & 15u;
  s.x >>= 4;
  refill(s);
  return d;
}

// Slot search over one CDF row: largest k in [0, len-2] with row[k] <= cf (row[len-1] stands for 65536
// and is never read; entries at or beyond it count as 65536).  The row sits in LDS (u16) or in global
// memory (int32); either way a classic binary search is a chain of log2(len) DEPENDENT reads, and with
// one stream per lane nothing else hides their latency.  Here three tree levels are resolved per round
// trip: the seven candidates at lo + step * {1..7} are requested together, then picked by compares and
// selects (span / 8 per round; the last round handles span 4 or 2).  A 32-entry row (every shipped
// table) takes 2 round trips instead of 5 + 1, a 3133-entry scale-table row 4 instead of 12.
// Returns the slot and the two edges it lies between.
template <typename RowT>
__device__ __forceinline__ int search_row(const RowT *row, int len, uint32_t cf, uint32_t &start,
                                          uint32_t &next) {
  const int last = len - 1;   // first index that counts as 65536
  int span = 2;               // smallest power of two >= len - 1 (>= 2): wave-uniform
  while (span < last) span <<= 1;
  int lo = 0;
  uint32_t vlo = 0u, vhi = 65536u;
  // candidates lo + step * (j + 1), j < N: ALL reads are issued before the first value is looked at
  // (left to itself hipcc waits for each read in front of the select that patches the 65536 sentinel in)
  auto fetch = [&](auto n_c, int step, uint32_t (&v)[7]) {
    constexpr int N = decltype(n_c)::value;
    uint32_t raw[7];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int i = lo + step * (j + 1);
      raw[j] = (uint32_t)row[i < last ? i : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = lo + step * (j + 1) < last ? raw[j] : 65536u;
  };
  auto take = [&](bool le, uint32_t v) {
    vlo = le ? v : vlo;
    vhi = le ? vhi : v;
  };
  uint32_t v[7];
  while (span >= 8) {
    const int step = span >> 3;
    fetch(std::integral_constant<int, 7>{}, step, v);
    const bool c4 = v[3] <= cf;
    take(c4, v[3]);
    const uint32_t m2 = c4 ? v[5] : v[1];
    const bool c2 = m2 <= cf;
    take(c2, m2);
    const uint32_t m1 = c4 ? (c2 ? v[6] : v[4]) : (c2 ? v[2] : v[0]);
    const bool c1 = m1 <= cf;
    take(c1, m1);
    lo += step * ((c4 ? 4 : 0) + (c2 ? 2 : 0) + (c1 ? 1 : 0));
    span = step;
  }
  if (span == 4) {
    fetch(std::integral_constant<int, 3>{}, 1, v);
    const bool c2 = v[1] <= cf;
    take(c2, v[1]);
    const uint32_t m1 = c2 ? v[2] : v[0];
    const bool c1 = m1 <= cf;
    take(c1, m1);
    lo += (c2 ? 2 : 0) + (c1 ? 1 : 0);
  } else if (span == 2) {
    fetch(std::integral_constant<int, 1>{}, 1, v);
    const bool c1 = v[0] <= cf;
    take(c1, v[0]);
    lo += c1 ? 1 : 0;
  }
  start = vlo;
  next = vhi;
  return lo;
}

// The plain binary search: for the short rows of the factorized model (<= 33 entries in LDS, 5 steps)
// it measured faster than the three-levels-per-round-trip search above (3.1 vs 2.7 M img/s at batch
// 1024: fewer instructions on the serial chain matter more than two LDS round trips).
template <typename RowT>
__device__ __forceinline__ int search_row_binary(const RowT *row, int len, uint32_t cf, uint32_t &start,
                                                 uint32_t &next) {
  int lo = 0, hi = len - 1;
  uint32_t vlo = 0u, vhi = 65536u;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    const uint32_t v = (uint32_t)row[mid];   // mid <= len-2: the 65536 entry is never read
    const bool le = v <= cf;
    lo = le ? mid : lo;
    hi = le ? hi : mid;
    vlo = le ? v : vlo;
    vhi = le ? vhi : v;
  }
  start = vlo;
  next = vhi;
  return lo;
}

// One symbol off the stream: search the row, advance the state, read escape digits.
template <bool TREE, typename RowT>
__device__ __forceinline__ int32_t decode_symbol(DecState &s, const RowT *row, int len) {
  const int esc = len - 2;
  const uint32_t cf = (uint32_t)s.x & 0xffffu;
  uint32_t start, next;
  const int lo = TREE ? search_row(row, len, cf, start, next) : search_row_binary(row, len, cf, start, next);
  const uint32_t freq = next - start;
  s.x = (uint64_t)freq * (s.x >> kProbBits) + cf - start;
  refill(s);
  int32_t v = lo;
  if (lo == esc) {
    uint32_t d = take_digit(s);
    uint32_t nd = d;
    while (d == 15u && nd < 64u) {
      d = take_digit(s);
      nd += d;
    }
    uint32_t raw = 0;
    for (uint32_t j = 0; j < nd; ++j) {
      d = take_digit(s);
      if (j < 8) raw |= d << (4 * j);
    }
    const int32_t sraw = (int32_t)raw;
    v = sraw >> 1;
    v = (sraw & 1) ? -v - 1 : v + esc;
  }
  return v;
}

__device__ __forceinline__ bool open_stream(DecState &s, const uint8_t *payload, const uint64_t *off,
                                            int skip, int i) {
  const uint64_t begin = off[i] + (uint64_t)skip;
  const uint64_t endb = off[i + 1];
  s.w = reinterpret_cast<const uint32_t *>(payload + begin);
  s.nwords = endb > begin ? (uint32_t)((endb - begin) >> 2) : 0u;
  if (s.nwords < 2 || ((endb - begin) & 3u) || (reinterpret_cast<uintptr_t>(s.w) & 3u)) return false;
  s.x = (uint64_t)s.w[0] | ((uint64_t)s.w[1] << 32);
  s.pos = 2;
  return true;
}

__global__ __launch_bounds__(kEncThreads) void rans_decode_kernel(
    const uint8_t *__restrict__ payload, const uint64_t *__restrict__ off, int skip, int B, int C,
    const int32_t *__restrict__ cdf, int W, const int32_t *__restrict__ cdf_len,
    const int32_t *__restrict__ offset, int32_t *__restrict__ out, int32_t *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
  // (cdf length, offset) per channel in LDS: see the encoder for why not uniform global reads
  int2 *par = reinterpret_cast<int2 *>(smem + enc_table_bytes(C, W));
  for (int c = threadIdx.x; c < C; c += blockDim.x) par[c] = make_int2(cdf_len[c], offset[c]);
  stage_table(tab, cdf, C * W);

  const int img = blockIdx.x * kEncThreads + threadIdx.x;
  if (img >= B) return;

  DecState s;
  int32_t *dst = out + (size_t)img * C;
  if (!open_stream(s, payload, off, skip & 0xff, img)) {
    for (int c = 0; c < C; ++c) dst[c] = 0;
    if (status) status[img] = 1;
    return;
  }
  for (int c = 0; c < C; ++c) {
    const int2 q = par[c];
    dst[c] = decode_symbol<false>(s, tab + c * W, q.x) + q.y;
  }
  if (status) status[img] = s.pos > s.nwords ? 1 : 0;
}

// ---------------------------------------------------------------------------
// Arbitrary table row per symbol (compressai encode_with_indexes / decode_with_indexes as
// GaussianConditional uses them: lossyless/rates.py:694-729).  One string per lane; rows are
// read from global memory (a 64-level scale table is T x W = 64 x ~3100 int32, too large for
// LDS, and the rows differ per lane anyway).  Row indexes are clamped to [0, T).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kEncThreads) void rans_encode_indexed_kernel(
    const int32_t *__restrict__ symbols, const int32_t *__restrict__ indexes, int B, int n,
    const int32_t *__restrict__ cdf, int T, int W, const int32_t *__restrict__ cdf_len,
    const int32_t *__restrict__ offset, uint8_t *__restrict__ scratch, size_t stride,
    uint32_t *__restrict__ lengths) {
  const int i = blockIdx.x * kEncThreads + threadIdx.x;
  if (i >= B) return;
  const int32_t *sym = symbols + (size_t)i * n;
  const int32_t *idx = indexes + (size_t)i * n;
  uint8_t *end = scratch + (size_t)i * stride + stride;
  EncState s;
  s.x = kStateLow;
  s.wp = reinterpret_cast<uint32_t *>(end);
  constexpr int G = 4;
  auto prep = [&](int k) {
    const int t = min(max(idx[k], 0), T - 1);
    return prepare_channel(cdf + (size_t)t * W, cdf_len[t], offset[t], sym[k]);
  };
  const int tail = n % G;
  for (int k = n - 1; k >= n - tail; --k) emit_channel(s, prep(k));
  for (int g = (n - tail) / G - 1; g >= 0; --g) {
    Prepared pr[G];
#pragma unroll
    for (int k = 0; k < G; ++k) pr[k] = prep(g * G + k);
#pragma unroll
    for (int k = G - 1; k >= 0; --k) emit_channel(s, pr[k]);
  }
  s.wp -= 2;
  s.wp[0] = (uint32_t)s.x;
  s.wp[1] = (uint32_t)(s.x >> 32);
  lengths[i] = (uint32_t)(end - reinterpret_cast<uint8_t *>(s.wp));
}

// Decoder: the rows are searched, not just indexed, so their latency sits on every symbol's critical
// path.  All T rows are therefore packed back to back into LDS as u16 (only their cdf_len[t] valid
// entries: the 64-level scale table of lossyless/rates.py:567-569 is 64 x 3133 int32 = 800 KB padded
// but ~27 000 valid entries = 54 KB) and searched there; when they do not fit the LDS the kernel was
// given, the rows are searched in global memory (same code, int32 rows).
__global__ __launch_bounds__(kEncThreads) void rans_decode_indexed_kernel(
    const uint8_t *__restrict__ payload, const uint64_t *__restrict__ off, int skip, int B, int n,
    const int32_t *__restrict__ indexes, const int32_t *__restrict__ cdf, int T, int W,
    const int32_t *__restrict__ cdf_len, const int32_t *__restrict__ offset,
    int32_t *__restrict__ out, int32_t *__restrict__ status, unsigned lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [T+1] row starts (u32) | [T] (len, offset) | packed rows (u16)
  uint32_t *rowoff = reinterpret_cast<uint32_t *>(smem);
  int2 *par = reinterpret_cast<int2 *>(smem + (((size_t)(T + 1) * 4 + 7) & ~(size_t)7));
  uint16_t *tab = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(par) + (size_t)T * sizeof(int2));
  const size_t head = (size_t)(reinterpret_cast<uint8_t *>(tab) - smem);
  bool in_lds = head < lds_bytes;
  if (in_lds) {
    for (int t = threadIdx.x; t < T; t += blockDim.x) par[t] = make_int2(min(max(cdf_len[t], 3), W), offset[t]);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t acc = 0;
      for (int t = 0; t < T; ++t) { rowoff[t] = acc; acc += (uint32_t)par[t].x; }
      rowoff[T] = acc;
    }
    __syncthreads();
    in_lds = head + (size_t)rowoff[T] * 2 <= lds_bytes;   // uniform
    if (in_lds) {
      for (int t = 0; t < T; ++t) {
        const int len = par[t].x;
        const int32_t *src = cdf + (size_t)t * W;
        uint16_t *dst = tab + rowoff[t];
        for (int i = threadIdx.x; i < len; i += blockDim.x) dst[i] = (uint16_t)src[i];
      }
      __syncthreads();
    }
  }
  const int i = blockIdx.x * kEncThreads + threadIdx.x;
  if (i >= B) return;
  DecState s;
  int32_t *dst = out + (size_t)i * n;
  if (!open_stream(s, payload, off, skip, i)) {
    for (int k = 0; k < n; ++k) dst[k] = 0;
    if (status) status[i] = 1;
    return;
  }
  const int32_t *idx = indexes + (size_t)i * n;
  if (in_lds) {
    int t_next = min(max(idx[0], 0), T - 1);
    for (int k = 0; k < n; ++k) {
      const int t = t_next;
      if (k + 1 < n) t_next = min(max(idx[k + 1], 0), T - 1);   // (requested one symbol ahead)
      const int2 q = par[t];
      dst[k] = decode_symbol<true>(s, tab + rowoff[t], q.x) + q.y;
    }
  } else {
    for (int k = 0; k < n; ++k) {
      const int t = min(max(idx[k], 0), T - 1);
      dst[k] = decode_symbol<true>(s, cdf + (size_t)t * W, min(max(cdf_len[t], 3), W)) + offset[t];   // (same clamp as the LDS path)
    }
  }
  if (status) status[i] = s.pos > s.nwords ? 1 : 0;
}

// ---------------------------------------------------------------------------
// elementwise: quantise / dequantise / represent
// ---------------------------------------------------------------------------
template <int ZMODE>
__global__ void quantise_kernel(const void *__restrict__ z, size_t n, int C,
                                const float *__restrict__ bias, const float *__restrict__ es,
                                const float *__restrict__ med, int32_t *__restrict__ sym) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    float v;
    if constexpr (ZMODE == 1) v = __half2float(reinterpret_cast<const __half *>(z)[i]);
    else v = reinterpret_cast<const float *>(z)[i];
    sym[i] = quantise_one(v, bias[c], es[c], med[c]);
  }
}

__device__ __forceinline__ float dequantise_one(float q, float bias, float es, float med) {
  const float zh = __fadd_rn(q, med);
  return __fsub_rn(__fdiv_rn(zh, es), bias);
}

__global__ void dequantise_kernel(const int32_t *__restrict__ sym, size_t n, int C,
                                  const float *__restrict__ bias, const float *__restrict__ es,
                                  const float *__restrict__ med, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    out[i] = dequantise_one((float)sym[i], bias[c], es[c], med[c]);
  }
}

template <int ZMODE>
__global__ void represent_kernel(const void *__restrict__ z, size_t n, int C,
                                 const float *__restrict__ bias, const float *__restrict__ es,
                                 const float *__restrict__ med, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    float v;
    if constexpr (ZMODE == 1) v = __half2float(reinterpret_cast<const __half *>(z)[i]);
    else v = reinterpret_cast<const float *>(z)[i];
    const float t = __fadd_rn(v, bias[c]);
    const float u = __fmul_rn(t, es[c]);
    const float q = rintf(__fsub_rn(u, med[c]));  // EntropyBottleneck.forward, eval mode
    out[i] = dequantise_one(q, bias[c], es[c], med[c]);
  }
}

// ---------------------------------------------------------------------------
// compaction: lengths -> exclusive offsets (wave scans) -> wave-per-image copy
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

// Inclusive scan across the workgroup (Hillis-Steele through LDS memory, double-buffered); returns this
// thread's inclusive value and the workgroup total.  buf must hold 2 * blockDim.x entries of LDS.  At 16 Ki
// lengths per launch the scan's latency does not matter; ten barrier-separated steps are the simplest
// thing that is obviously right.
__device__ __forceinline__ uint64_t block_inclusive_scan(uint64_t v, uint64_t *buf, uint64_t *total) {
  const int t = threadIdx.x, n = blockDim.x;
  int cur = 0;
  buf[t] = v;
  __syncthreads();
  for (int d = 1; d < n; d <<= 1) {
    uint64_t x = buf[cur * n + t];
    if (t >= d) x += buf[cur * n + t - d];
    cur ^= 1;
    buf[cur * n + t] = x;
    __syncthreads();
  }
  *total = buf[cur * n + n - 1];
  return buf[cur * n + t];
}

__global__ __launch_bounds__(kScanThreads) void block_sums_kernel(
    const uint32_t *__restrict__ lengths, int B, uint32_t extra, uint64_t *__restrict__ sums) {
  __shared__ uint64_t wave_tot[2 * kScanThreads];
  const int i = blockIdx.x * kScanThreads + threadIdx.x;
  const uint64_t v = i < B ? (uint64_t)lengths[i] + extra : 0;
  uint64_t total;
  block_inclusive_scan(v, wave_tot, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of sums[0..n) in place, total -> *grand
__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(uint64_t *__restrict__ sums,
                                                                 int n,
                                                                 uint64_t *__restrict__ grand) {
  __shared__ uint64_t wave_tot[2 * kScanThreads];
  uint64_t carry = 0;
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const uint64_t v = i < n ? sums[i] : 0;
    uint64_t total;
    const uint64_t inc = block_inclusive_scan(v, wave_tot, &total);
    if (i < n) sums[i] = carry + inc - v;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *grand = carry;
}

__global__ __launch_bounds__(kScanThreads) void final_offsets_kernel(
    const uint32_t *__restrict__ lengths, int B, uint32_t extra,
    const uint64_t *__restrict__ sums_ex, uint64_t *__restrict__ out_off) {
  __shared__ uint64_t wave_tot[2 * kScanThreads];
  const int i = blockIdx.x * kScanThreads + threadIdx.x;
  const uint64_t v = i < B ? (uint64_t)lengths[i] + extra : 0;
  uint64_t total;
  const uint64_t inc = block_inclusive_scan(v, wave_tot, &total);
  if (i < B) out_off[i] = sums_ex[blockIdx.x] + inc - v;
}

__global__ __launch_bounds__(256) void copy_streams_kernel(
    const uint8_t *__restrict__ scratch, size_t stride, const uint32_t *__restrict__ lengths,
    int B, int record_prefix, uint8_t *__restrict__ out, size_t cap,
    const uint64_t *__restrict__ out_off) {
  const int img = blockIdx.x * 4 + (threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  if (img >= B) return;
  const uint32_t len = lengths[img];
  const uint64_t o = out_off[img];
  const uint64_t need = o + len + (record_prefix ? 4u : 0u);
  if (need > cap) return;  // caller learns the required size from out_off[B]
  const uint32_t *src =
      reinterpret_cast<const uint32_t *>(scratch + (size_t)img * stride + stride - len);
  uint32_t *dst = reinterpret_cast<uint32_t *>(out + o);
  if (record_prefix) {
    if (lane == 0) dst[0] = __builtin_bswap32(len);  // big-endian u32, hub/compressor.py:258
    ++dst;
  }
  const uint32_t nw = len >> 2;
  for (uint32_t w = lane; w < nw; w += kWave) dst[w] = src[w];
}

inline int grid_for(size_t n, int threads, int cap = 2048) {
  size_t g = (n + threads - 1) / threads;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

bool table_args_ok(int B, int C, int W, const void *cdf, const void *cdf_len, const void *off) {
  return B >= 0 && C > 0 && W >= 3 && (size_t)C * W * 2 <= 64 * 1024 && cdf && cdf_len && off;
}

}  // namespace
}  // namespace lla

using namespace lla;

extern "C" {

int lla_abi_version(void) { return LLA_ABI_VERSION; }
int lla_last_hip_error(void) { return g_last_hip_error; }

size_t lla_rans_max_encoded_bytes(int C) {
  // per symbol at most 16 bits for the escape symbol + 1 count digit + 8 payload
  // digits = 52 bits; +2 flush words; +1 word of slack; rounded up to 16 bytes
  const size_t bits = (size_t)C * 52u;
  const size_t bytes = 4u * ((bits + 31u) / 32u) + 8u + 4u;
  return (bytes + 15u) & ~(size_t)15u;
}

int lla_quantise(const void *z, int z_dtype, int B, int C, const float *bias,
                 const float *exp_scale, const float *median, int32_t *symbols, void *stream) {
  if (!z || !bias || !exp_scale || !median || !symbols || B < 0 || C <= 0) return LLA_EINVAL;
  if (z_dtype != LLA_Z_F16 && z_dtype != LLA_Z_F32) return LLA_EINVAL;
  const size_t n = (size_t)B * C;
  if (n == 0) return LLA_OK;
  const int g = grid_for(n, 256);
  if (z_dtype == LLA_Z_F16)
    quantise_kernel<1><<<g, 256, 0, as_stream(stream)>>>(z, n, C, bias, exp_scale, median, symbols);
  else
    quantise_kernel<2><<<g, 256, 0, as_stream(stream)>>>(z, n, C, bias, exp_scale, median, symbols);
  return check_launch();
}

int lla_rans_encode_batch(const int32_t *symbols, int B, int C, const int32_t *cdf, int W,
                          const int32_t *cdf_len, const int32_t *offset, uint8_t *scratch,
                          size_t stride, uint32_t *lengths, void *stream) {
  if (B == 0) return LLA_OK;
  if (!symbols || !scratch || !lengths || !table_args_ok(B, C, W, cdf, cdf_len, offset))
    return LLA_EINVAL;
  if (stride < lla_rans_max_encoded_bytes(C) || (stride & 3u)) return LLA_ECAP;
  const int grid = (B + kEncThreads - 1) / kEncThreads;
  const size_t lds = enc_lds_bytes(C, W);
  if (lds > 64 * 1024) return LLA_EINVAL;
  rans_encode_kernel<0><<<grid, kEncThreads, lds, as_stream(stream)>>>(
      symbols, B, C, nullptr, nullptr, nullptr, cdf, W, cdf_len, offset, scratch, stride, lengths,
      nullptr);
  return check_launch();
}

int lla_quantise_encode(const void *z, int z_dtype, int B, int C, const float *bias,
                        const float *exp_scale, const float *median, const int32_t *cdf, int W,
                        const int32_t *cdf_len, const int32_t *offset, uint8_t *scratch,
                        size_t stride, uint32_t *lengths, int32_t *symbols_out, void *stream) {
  if (B == 0) return LLA_OK;
  if (!z || !bias || !exp_scale || !median || !scratch || !lengths ||
      !table_args_ok(B, C, W, cdf, cdf_len, offset))
    return LLA_EINVAL;
  if (z_dtype != LLA_Z_F16 && z_dtype != LLA_Z_F32) return LLA_EINVAL;
  if (stride < lla_rans_max_encoded_bytes(C) || (stride & 3u)) return LLA_ECAP;
  const int grid = (B + kEncThreads - 1) / kEncThreads;
  const size_t lds = enc_lds_bytes(C, W);
  if (lds > 64 * 1024) return LLA_EINVAL;
  if (z_dtype == LLA_Z_F16)
    rans_encode_kernel<1><<<grid, kEncThreads, lds, as_stream(stream)>>>(
        z, B, C, bias, exp_scale, median, cdf, W, cdf_len, offset, scratch, stride, lengths,
        symbols_out);
  else
    rans_encode_kernel<2><<<grid, kEncThreads, lds, as_stream(stream)>>>(
        z, B, C, bias, exp_scale, median, cdf, W, cdf_len, offset, scratch, stride, lengths,
        symbols_out);
  return check_launch();
}

size_t lla_rans_compact_workspace_bytes(int B) {
  const size_t nblk = ((size_t)(B > 0 ? B : 1) + kScanThreads - 1) / kScanThreads;
  return (nblk + 1) * sizeof(uint64_t);
}

int lla_rans_compact(const uint8_t *scratch, size_t stride, const uint32_t *lengths, int B,
                     int record_prefix, uint8_t *out, size_t cap, uint64_t *out_off,
                     void *workspace, size_t workspace_bytes, void *stream) {
  hipStream_t st = as_stream(stream);
  if (B == 0) {
    if (!out_off) return LLA_EINVAL;
    hipError_t e = hipMemsetAsync(out_off, 0, sizeof(uint64_t), st);
    return e == hipSuccess ? LLA_OK : hip_fail(e);
  }
  if (!scratch || !lengths || !out || !out_off || !workspace || B < 0) return LLA_EINVAL;
  if (workspace_bytes < lla_rans_compact_workspace_bytes(B)) return LLA_ECAP;
  if ((reinterpret_cast<uintptr_t>(out) & 3u) || (stride & 3u)) return LLA_EINVAL;
  uint64_t *sums = reinterpret_cast<uint64_t *>(workspace);
  const int nblk = (B + kScanThreads - 1) / kScanThreads;
  const uint32_t extra = record_prefix ? 4u : 0u;
  block_sums_kernel<<<nblk, kScanThreads, 0, st>>>(lengths, B, extra, sums);
  scan_sums_kernel<<<1, kScanThreads, 0, st>>>(sums, nblk, out_off + B);
  final_offsets_kernel<<<nblk, kScanThreads, 0, st>>>(lengths, B, extra, sums, out_off);
  copy_streams_kernel<<<(B + 3) / 4, 256, 0, st>>>(scratch, stride, lengths, B, record_prefix, out,
                                                   cap, out_off);
  return check_launch();
}

int lla_rans_decode_batch(const uint8_t *payload, const uint64_t *off, int record_prefix, int B,
                          int C, const int32_t *cdf, int W, const int32_t *cdf_len,
                          const int32_t *offset, int32_t *symbols_out, int32_t *status,
                          void *stream) {
  if (B == 0) return LLA_OK;
  if (!payload || !off || !symbols_out || !table_args_ok(B, C, W, cdf, cdf_len, offset))
    return LLA_EINVAL;
  const int grid = (B + kEncThreads - 1) / kEncThreads;
  const size_t lds = enc_table_bytes(C, W) + (size_t)C * sizeof(int2);
  if (lds > 64 * 1024) return LLA_EINVAL;
  int skip = record_prefix ? 4 : 0;
  rans_decode_kernel<<<grid, kEncThreads, lds, as_stream(stream)>>>(
      payload, off, skip, B, C, cdf, W, cdf_len, offset, symbols_out, status);
  return check_launch();
}

int lla_rans_encode_indexed(const int32_t *symbols, const int32_t *indexes, int B, int n,
                            const int32_t *cdf, int T, int W, const int32_t *cdf_len,
                            const int32_t *offset, uint8_t *scratch, size_t stride,
                            uint32_t *lengths, void *stream) {
  if (B == 0) return LLA_OK;
  if (B < 0 || n <= 0 || T <= 0 || W < 3 || !symbols || !indexes || !cdf || !cdf_len || !offset ||
      !scratch || !lengths)
    return LLA_EINVAL;
  if (stride < lla_rans_max_encoded_bytes(n) || (stride & 3u)) return LLA_ECAP;
  const int grid = (B + kEncThreads - 1) / kEncThreads;
  rans_encode_indexed_kernel<<<grid, kEncThreads, 0, as_stream(stream)>>>(
      symbols, indexes, B, n, cdf, T, W, cdf_len, offset, scratch, stride, lengths);
  return check_launch();
}

int lla_rans_decode_indexed(const uint8_t *payload, const uint64_t *off, int record_prefix, int B,
                            int n, const int32_t *indexes, const int32_t *cdf, int T, int W,
                            const int32_t *cdf_len, const int32_t *offset, int32_t *symbols_out,
                            int32_t *status, void *stream) {
  if (B == 0) return LLA_OK;
  if (B < 0 || n <= 0 || T <= 0 || W < 3 || !payload || !off || !indexes || !cdf || !cdf_len ||
      !offset || !symbols_out)
    return LLA_EINVAL;
  const int grid = (B + kEncThreads - 1) / kEncThreads;
  // LDS for the packed rows: their valid entries number at most T*W (known exactly only on the device), so the
  // launch asks for min(device limit, header + 2*T*W) -- small tables leave room for several workgroups per CU.
  // The limit and the kernel's opt-in to > 48 KiB are per DEVICE (a process may drive several GPUs).
  const unsigned lds_max = dynamic_lds_limit(reinterpret_cast<const void *>(rans_decode_indexed_kernel));
  const size_t head = (((size_t)(T + 1) * 4 + 7) & ~(size_t)7) + (size_t)T * sizeof(int2);
  const size_t want = head + 2 * (size_t)T * (size_t)W + 64;
  const unsigned lds = head + 64 > lds_max ? 0u   // (absurdly many rows: global-memory search)
                                           : (unsigned)(want < lds_max ? ((want + 255) & ~(size_t)255) : lds_max);
  rans_decode_indexed_kernel<<<grid, kEncThreads, lds, as_stream(stream)>>>(
      payload, off, record_prefix ? 4 : 0, B, n, indexes, cdf, T, W, cdf_len, offset, symbols_out,
      status, lds);
  return check_launch();
}

int lla_dequantise(const int32_t *symbols, int B, int C, const float *bias,
                   const float *exp_scale, const float *median, float *z_hat, void *stream) {
  if (!symbols || !bias || !exp_scale || !median || !z_hat || B < 0 || C <= 0) return LLA_EINVAL;
  const size_t n = (size_t)B * C;
  if (n == 0) return LLA_OK;
  dequantise_kernel<<<grid_for(n, 256), 256, 0, as_stream(stream)>>>(symbols, n, C, bias, exp_scale,
                                                                     median, z_hat);
  return check_launch();
}

int lla_represent(const void *z, int z_dtype, int B, int C, const float *bias,
                  const float *exp_scale, const float *median, float *z_hat, void *stream) {
  if (!z || !bias || !exp_scale || !median || !z_hat || B < 0 || C <= 0) return LLA_EINVAL;
  if (z_dtype != LLA_Z_F16 && z_dtype != LLA_Z_F32) return LLA_EINVAL;
  const size_t n = (size_t)B * C;
  if (n == 0) return LLA_OK;
  const int g = grid_for(n, 256);
  if (z_dtype == LLA_Z_F16)
    represent_kernel<1><<<g, 256, 0, as_stream(stream)>>>(z, n, C, bias, exp_scale, median, z_hat);
  else
    represent_kernel<2><<<g, 256, 0, as_stream(stream)>>>(z, n, C, bias, exp_scale, median, z_hat);
  return check_launch();
}

}  // extern "C"
