// Host-side entry points of liblossyless_amd.so that have no device work.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "../../include/lossyless_amd.h"

extern "C" int lla_pmf_to_quantized_cdf(const float *pmf, int n, int precision,
                                        uint32_t *cdf_out) {
  // Stands in for compressai._CXX.pmf_to_quantized_cdf (cpp_exts/ops/ops.cpp), reached
  // from EntropyBottleneck.update() at hub/compressor.py:63.  Algorithm: SURVEY.md 8(a)
  // row A12 -- scale to 2^precision, renormalise by the integer total, prefix-sum, pin the
  // last edge, then widen every empty bin by one count taken from the narrowest bin that
  // can spare it (first such bin on ties), shifting the edges in between.
  if (!pmf || !cdf_out || n < 1 || precision < 1 || precision > 16) return LLA_EINVAL;
  const uint64_t full = uint64_t(1) << precision;
  std::vector<uint64_t> edge(size_t(n) + 1, 0);
  for (int i = 0; i < n; ++i) {
    const float scaled = pmf[i] * float(full);  // float multiply, like the source
    if (!(scaled >= 0.0f)) return LLA_EDATA;    // negative or NaN mass
    edge[size_t(i) + 1] = uint64_t(std::round(scaled));
  }
  const uint64_t total = std::accumulate(edge.begin(), edge.end(), uint64_t(0));
  if (total == 0) return LLA_EDATA;
  for (auto &e : edge) e = (full * e) / total;
  std::partial_sum(edge.begin(), edge.end(), edge.begin());
  edge.back() = full;

  for (int i = 0; i < n; ++i) {
    if (edge[size_t(i)] != edge[size_t(i) + 1]) continue;
    int donor = -1;
    uint64_t donor_width = ~uint64_t(0);
    for (int j = 0; j < n; ++j) {
      const uint64_t width = edge[size_t(j) + 1] - edge[size_t(j)];
      if (width > 1 && width < donor_width) {
        donor_width = width;
        donor = j;
      }
    }
    if (donor < 0) return LLA_EDATA;
    if (donor < i) {
      for (int j = donor + 1; j <= i; ++j) edge[size_t(j)] -= 1;
    } else {
      for (int j = i + 1; j <= donor; ++j) edge[size_t(j)] += 1;
    }
  }
  for (int i = 0; i <= n; ++i) cdf_out[i] = uint32_t(edge[size_t(i)]);
  return LLA_OK;
}

static inline uint32_t load_be32(const uint8_t *p) {
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

extern "C" int lla_container_index(const uint8_t *blob, size_t nbytes, uint64_t *off,
                                   size_t off_cap, uint32_t *n_out) {
  // Record walk of the reference's reader loop, hub/compressor.py:233-237.
  if (!blob || !n_out || nbytes < 4) return LLA_EINVAL;
  const uint32_t n = load_be32(blob);
  *n_out = n;
  // every record carries at least its 4-byte length: a count the file cannot hold is corrupt
  // (checked before the caller sizes anything by it)
  if (uint64_t(n) * 4 + 4 > uint64_t(nbytes)) return LLA_EDATA;
  if (!off) return LLA_OK;
  if (off_cap < size_t(n) + 1) return LLA_ECAP;
  size_t pos = 4;
  for (uint32_t i = 0; i < n; ++i) {
    if (pos + 4 > nbytes) return LLA_EDATA;
    off[i] = pos - 4;
    const uint32_t len = load_be32(blob + pos);
    pos += 4 + size_t(len);
    if (pos > nbytes) return LLA_EDATA;
  }
  off[n] = pos - 4;
  return pos == nbytes ? LLA_OK : LLA_EDATA;
}

// ---------------------------------------------------------------------------
// Host rANS coder (SURVEY.md 8(b) "host" entry points).  The reference decodes on the CPU by
// default (decompress_dataset(is_cpu=True), hub/compressor.py:227-229: the module is moved to
// the host and RansDecoder.decode_with_indexes runs per image); these entry points give that
// mode without a GPU -- B images per call, one std::thread per slice of images.  Same streams
// as the device kernels and as compressai's rans_interface.cpp: 64-bit state, lower bound 2^31,
// 32-bit renormalisation words, 16-bit probabilities, 4-bit bypass digits.
// ---------------------------------------------------------------------------
#include <thread>

namespace {

constexpr int kPrecision = 16;
constexpr uint64_t kLower = uint64_t(1) << 31;

// Stream builder for one image.  Coding runs over the symbols back to front and every
// renormalisation word is appended; the finished stream is the two state words followed by the
// renormalisation words in REVERSE order of emission (the decoder consumes front to back).
struct StreamBuilder {
  uint64_t state = kLower;
  std::vector<uint32_t> spill;  // renormalisation words, oldest first

  inline void shrink_for(uint64_t freq, int bits) {
    // largest state that stays below 2^63 after absorbing a symbol of `freq` / 2^bits
    const uint64_t limit = ((kLower >> bits) << 32) * freq;
    if (state >= limit) {
      spill.push_back(uint32_t(state));
      state >>= 32;
    }
  }
  inline void absorb(uint32_t start, uint32_t freq) {
    shrink_for(freq, kPrecision);
    state = ((state / freq) << kPrecision) + (state % freq) + start;
  }
  inline void absorb_digit(uint32_t digit) {  // uniform over 16 values: freq 2^12 of 2^16
    shrink_for(uint64_t(1) << (kPrecision - 4), kPrecision);
    state = (state << 4) | digit;
  }
  size_t bytes() const { return 4 * (spill.size() + 2); }
  void write(uint8_t *dst) const {  // little-endian words, as the reference's native u32 buffer
    auto put = [&](uint32_t w) { dst[0] = uint8_t(w); dst[1] = uint8_t(w >> 8); dst[2] = uint8_t(w >> 16); dst[3] = uint8_t(w >> 24); dst += 4; };
    put(uint32_t(state));
    put(uint32_t(state >> 32));
    for (size_t i = spill.size(); i-- > 0;) put(spill[i]);
  }
};

// What the forward pass of encode_with_indexes pushes for one value: the table slot, and for an
// out-of-window value the bypass digits (count in base 15 with 15 as "more", then the payload,
// least significant digit first).
struct Coded {
  int slot;
  uint32_t raw;
  bool escaped;
};
inline Coded classify(int32_t sym, int32_t off, int esc_slot) {
  const int64_t v = int64_t(sym) - off;
  if (v < 0) return {esc_slot, uint32_t(-2 * v - 1), true};
  if (v >= esc_slot) return {esc_slot, uint32_t(2 * (v - esc_slot)), true};
  return {int(v), 0u, false};
}
inline int digits_of(uint32_t raw) {
  int n = 0;
  while (raw) { ++n; raw >>= 4; }
  return n;
}

void encode_image(const int32_t *sym, int C, const int32_t *cdf, int W, const int32_t *cdf_len,
                  const int32_t *offset, StreamBuilder &sb) {
  sb.state = kLower;
  sb.spill.clear();
  for (int c = C - 1; c >= 0; --c) {  // last pushed symbol is coded first
    const int32_t *row = cdf + size_t(c) * W;
    const int esc = cdf_len[c] - 2;
    const Coded k = classify(sym[c], offset[c], esc);
    if (k.escaped) {
      // forward order was: escape slot, count digits, payload digits (LSB first) -> reverse it
      const int nd = digits_of(k.raw);
      for (int d = nd - 1; d >= 0; --d) sb.absorb_digit((k.raw >> (4 * d)) & 15u);
      int full = nd / 15, last = nd % 15;  // count = 15 + 15 + ... + last
      sb.absorb_digit(uint32_t(last));
      for (int i = 0; i < full; ++i) sb.absorb_digit(15u);
    }
    const uint32_t start = uint32_t(row[k.slot]);
    sb.absorb(start, uint32_t(row[k.slot + 1]) - start);
  }
}

struct StreamReader {
  uint64_t state;
  const uint8_t *p, *end;
  bool overrun = false;
  inline uint32_t word() {
    if (p + 4 > end) { overrun = true; return 0; }
    const uint32_t w = uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
    p += 4;
    return w;
  }
  inline void refill() { if (state < kLower) state = (state << 32) | word(); }
  inline uint32_t digit() {
    const uint32_t d = uint32_t(state & 15u);
    state >>= 4;
    refill();
    return d;
  }
};

int decode_image(const uint8_t *s, size_t n, int C, const int32_t *cdf, int W,
                 const int32_t *cdf_len, const int32_t *offset, int32_t *out) {
  StreamReader r{0, s, s + n};
  const uint32_t lo = r.word(), hi = r.word();
  r.state = uint64_t(lo) | (uint64_t(hi) << 32);
  for (int c = 0; c < C; ++c) {
    const int32_t *row = cdf + size_t(c) * W;
    const int len = cdf_len[c], esc = len - 2;
    const uint32_t target = uint32_t(r.state & 0xffffu);
    int slot = int(std::upper_bound(row, row + len, int32_t(target)) - row) - 1;
    if (slot < 0) slot = 0;
    if (slot > esc) slot = esc;
    const uint32_t start = uint32_t(row[slot]), freq = uint32_t(row[slot + 1]) - start;
    r.state = uint64_t(freq) * (r.state >> kPrecision) + target - start;
    r.refill();
    int64_t v = slot;
    if (slot == esc) {
      uint32_t d = r.digit();
      int nd = int(d);
      while (d == 15u && !r.overrun) { d = r.digit(); nd += int(d); }
      uint64_t raw = 0;
      for (int j = 0; j < nd && !r.overrun; ++j) raw |= uint64_t(r.digit()) << (4 * (j & 15));
      v = int64_t(raw >> 1);
      v = (raw & 1u) ? -v - 1 : v + esc;
    }
    out[c] = int32_t(v + offset[c]);
  }
  return r.overrun ? 1 : 0;
}

template <typename F>
void for_images(int B, F body) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = int(std::min<unsigned>(hw ? hw : 1u, 32u));
  nt = std::max(1, std::min(nt, B / 64));  // a thread is not worth fewer than 64 images
  if (nt == 1) { body(0, B); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    const int lo = int(int64_t(B) * t / nt), hi = int(int64_t(B) * (t + 1) / nt);
    pool.emplace_back([=] { body(lo, hi); });
  }
  for (auto &th : pool) th.join();
}

bool tables_ok(int C, const int32_t *cdf, int W, const int32_t *cdf_len, const int32_t *offset) {
  if (C < 0 || W < 3 || (C && (!cdf || !cdf_len || !offset))) return false;
  for (int c = 0; c < C; ++c)
    if (cdf_len[c] < 3 || cdf_len[c] > W) return false;
  return true;
}

}  // namespace

extern "C" int lla_rans_encode_batch_host(const int32_t *symbols, int B, int C, const int32_t *cdf,
                                          int W, const int32_t *cdf_len, const int32_t *offset,
                                          int record_prefix, uint8_t *out, size_t cap,
                                          uint64_t *out_off) {
  if (B < 0 || !out_off || (B && C && !symbols) || !tables_ok(C, cdf, W, cdf_len, offset))
    return LLA_EINVAL;
  std::vector<std::vector<uint8_t>> streams((size_t)B);
  for_images(B, [&](int lo, int hi) {
    StreamBuilder sb;
    for (int b = lo; b < hi; ++b) {
      encode_image(symbols + size_t(b) * C, C, cdf, W, cdf_len, offset, sb);
      streams[size_t(b)].resize(sb.bytes());
      sb.write(streams[size_t(b)].data());
    }
  });
  const size_t pre = record_prefix ? 4 : 0;
  out_off[0] = 0;
  for (int b = 0; b < B; ++b) out_off[b + 1] = out_off[b] + pre + streams[size_t(b)].size();
  if (out_off[B] > cap) return LLA_ECAP;  // out_off still reports the size needed
  if (out_off[B] && !out) return LLA_EINVAL;
  for_images(B, [&](int lo, int hi) {
    for (int b = lo; b < hi; ++b) {
      uint8_t *dst = out + out_off[b];
      const auto &s = streams[size_t(b)];
      if (pre) {
        const uint32_t n = uint32_t(s.size());
        dst[0] = uint8_t(n >> 24); dst[1] = uint8_t(n >> 16); dst[2] = uint8_t(n >> 8); dst[3] = uint8_t(n);
        dst += 4;
      }
      std::copy(s.begin(), s.end(), dst);
    }
  });
  return LLA_OK;
}

extern "C" int lla_rans_decode_batch_host(const uint8_t *payload, const uint64_t *off,
                                          int record_prefix, int B, int C, const int32_t *cdf, int W,
                                          const int32_t *cdf_len, const int32_t *offset,
                                          int32_t *symbols_out, int32_t *status) {
  if (B < 0 || (B && (!payload || !off || !status || (C && !symbols_out))) ||
      !tables_ok(C, cdf, W, cdf_len, offset))
    return LLA_EINVAL;
  const size_t pre = record_prefix ? 4 : 0;
  for_images(B, [&](int lo, int hi) {
    for (int b = lo; b < hi; ++b) {
      const uint64_t a = off[b] + pre, e = off[b + 1];
      status[b] = (e < a || e - a < 8)
                      ? 1
                      : decode_image(payload + a, size_t(e - a), C, cdf, W, cdf_len, offset,
                                     symbols_out + size_t(b) * C);
    }
  });
  return LLA_OK;
}

extern "C" int lla_dequantise_host(const int32_t *symbols, int B, int C, const float *bias,
                                   const float *exp_scale, const float *median, float *z_hat) {
  // EntropyModel.dequantize + process_z_out (hub/compressor.py:111-115), each fp32 operation
  // rounded on its own (this file is built with -ffp-contract=off): same values as lla_dequantise.
  if (B < 0 || C < 0 || (B && C && (!symbols || !bias || !exp_scale || !median || !z_hat)))
    return LLA_EINVAL;
  for_images(B, [&](int lo, int hi) {
    for (int b = lo; b < hi; ++b)
      for (int c = 0; c < C; ++c) {
        const float v = float(symbols[size_t(b) * C + c]) + median[c];
        const float q = v / exp_scale[c];
        z_hat[size_t(b) * C + c] = q - bias[c];
      }
  });
  return LLA_OK;
}

// Pillow's tap tables (src/libImaging/Resample.c: bicubic_filter, precompute_coeffs, normalize_coeffs_8bpc),
// restated in the same double arithmetic: the window of output position xx is
// [int(center - support + 0.5), int(center + support + 0.5)) clipped to the axis, weights are
// bicubic((x - center + 0.5) / filterscale) normalised by their running sum, then rounded half away from zero
// to 22-bit fixed point.  The device kernels (preprocess.hip) consume these integers.
static inline double pillow_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int lla_pillow_bicubic_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return LLA_EINVAL;
  double filterscale = double(float(in_size) - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return int(std::ceil(2.0 * filterscale)) * 2 + 1;
}

extern "C" int lla_pillow_bicubic_taps(int in_size, int out_size, int first, int count, int ksize,
                                       int32_t *bounds, int32_t *coef) {
  if (in_size <= 0 || out_size <= 0 || first < 0 || count < 0 || first + count > out_size || !bounds || !coef ||
      ksize != lla_pillow_bicubic_ksize(in_size, out_size))
    return LLA_EINVAL;
  const double scale = double(float(in_size) - 0.0f) / out_size;   // box = (0, in_size) held as floats
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  std::vector<double> k(size_t(ksize), 0.0);
  for (int i = 0; i < count; ++i) {
    const int xx = first + i;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = int(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = int(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      const double w = pillow_bicubic((x + xmin - center + 0.5) * ss);
      k[size_t(x)] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[size_t(x)] /= ww;
    int32_t *row = coef + size_t(i) * ksize;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[size_t(x)] : 0.0;
      row[x] = v < 0 ? int32_t(-0.5 + v * double(1 << 22)) : int32_t(0.5 + v * double(1 << 22));
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xmax;
  }
  return LLA_OK;
}

// (the Makefile passes -DLLA_SOURCE_SHA to this file only and rebuilds it whenever any source changes)
#ifndef LLA_SOURCE_SHA
#define LLA_SOURCE_SHA "unknown"
#endif
// (the marker lets lossyless_amd/_lib.py read the sha from the FILE, before -- and without -- loading the library)
static const char kSourceSha[] = "LLA_SOURCE_SHA=" LLA_SOURCE_SHA;
extern "C" const char *lla_source_sha(void) { return kSourceSha + 15; }
