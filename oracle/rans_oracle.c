/*
 * oracle/rans_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar CPU restatement of the integer entropy-coder stage that the
 * reference's hot path calls.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the shipped package
 * (lossyless_amd/) never does.
 *
 * The arithmetic lives in a third-party dependency that is NOT vendored in
 * /root/reference: compressai==1.1.5 (requirements/environment.yaml:105),
 *   cpp_exts/ops/ops.cpp               pmf_to_quantized_cdf
 *   cpp_exts/rans/rans_interface.cpp   RansEncoder / RansDecoder
 *   third_party/ryg_rans/rans64.h      64-bit rANS, 32-bit renormalisation
 * It is restated here from its published algorithm (SURVEY.md section 8(a) rows
 * A12, A13, A14).  Parity anchoring: the reference's call sites are
 *   hub/compressor.py:63   entropy_bottleneck.update()   -> A12
 *   hub/compressor.py:98   entropy_bottleneck.compress() -> A13
 *   hub/compressor.py:124  entropy_bottleneck.decompress() -> A14
 *   lossyless/rates.py:299,559,563 (training-side twin of the same calls)
 * The reference holds no unit tests or golden vectors for this path
 * (SURVEY.md section 4), and compressai is not importable here, so:
 *   PARITY UNPINNED against a live compressai build; pinned against hand-derived
 *   known-answer streams (tests/golden/rans_kat.json), the rANS identities
 *   (decode(encode(s)) == s, state bounds), and the reference's recorded
 *   aggregate rate (notebooks/Hub.ipynb:253) as a plausibility band.
 *
 * Constants: state is u64 with lower bound 2^31, initial
 * state 2^31; probabilities have 16 bits; renormalisation moves 32-bit words;
 * escape ("bypass") payloads travel as 4-bit digits with an implied
 * frequency of 2^12.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PROB_BITS 16u
#define ORC_STATE_LOW (1ull << 31)
#define ORC_DIGIT_BITS 4u
#define ORC_DIGIT_MAX 15u

/* ------------------------------------------------------------------ A12 */
/* pmf -> integer CDF with exactly `precision` bits of total mass.
 * Follows compressai/cpp_exts/ops/ops.cpp:pmf_to_quantized_cdf as reached from
 * EntropyBottleneck.update() (hub/compressor.py:63).  cdf_out has n+1 slots.
 * Returns 0, or -1 when no donor frequency exists for a zero-width bin. */
int orc_pmf_to_quantized_cdf(const float *pmf, int n, int precision,
                             uint32_t *cdf_out) {
  const uint64_t one = 1ull << precision;
  cdf_out[0] = 0;
  for (int i = 0; i < n; ++i) {
    /* round-half-away of p * 2^precision, evaluated in float like the source */
    float scaled = pmf[i] * (float)one;
    cdf_out[i + 1] = (uint32_t)(int64_t)__builtin_roundf(scaled);
  }
  uint64_t total = 0;
  for (int i = 0; i <= n; ++i) total += cdf_out[i];
  if (total == 0) return -1;
  for (int i = 0; i <= n; ++i)
    cdf_out[i] = (uint32_t)((one * (uint64_t)cdf_out[i]) / total);
  for (int i = 1; i <= n; ++i) cdf_out[i] += cdf_out[i - 1];
  cdf_out[n] = (uint32_t)one;

  for (int i = 0; i < n; ++i) {
    if (cdf_out[i] != cdf_out[i + 1]) continue;
    /* bin i has zero width: take one count from the narrowest bin wider than 1 */
    uint32_t best_freq = ~0u;
    int best = -1;
    for (int j = 0; j < n; ++j) {
      uint32_t freq = cdf_out[j + 1] - cdf_out[j];
      if (freq > 1 && freq < best_freq) {
        best_freq = freq;
        best = j;
      }
    }
    if (best < 0) return -1;
    if (best < i) {
      for (int j = best + 1; j <= i; ++j) cdf_out[j] -= 1;
    } else {
      for (int j = i + 1; j <= best; ++j) cdf_out[j] += 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ A13 */
typedef struct {
  uint16_t start;
  uint16_t range;
  uint8_t raw; /* 1: 4-bit digit stored in `start` */
} orc_item;

/* Worst case items per symbol: 1 coded + 1 count digit + 8 payload digits
 * (raw value fits 32 bits -> at most 8 digits, count < 15). */
#define ORC_ITEMS_PER_SYMBOL 10

/* Encode one image's symbol vector into one rANS stream.
 * Mirrors RansEncoder.encode_with_indexes (hub/compressor.py:98 ->
 * EntropyModel.compress) with indexes[i] = channel i.
 *   symbols[n], cdf[C*W] row-major, cdf_len[C], offset[C], index[n] (or NULL = i)
 * Writes the byte string to out (capacity cap) and returns its length, or -1. */
long orc_rans_encode(const int32_t *symbols, const int32_t *index, int n,
                     const int32_t *cdf, int W, const int32_t *cdf_len,
                     const int32_t *offset, uint8_t *out, size_t cap) {
  orc_item *items = (orc_item *)malloc(sizeof(orc_item) * (size_t)n *
                                       ORC_ITEMS_PER_SYMBOL + sizeof(orc_item));
  if (!items) return -1;
  size_t n_items = 0;

  for (int i = 0; i < n; ++i) {
    const int c = index ? index[i] : i;
    const int32_t *row = cdf + (size_t)c * W;
    const int32_t escape = cdf_len[c] - 2;
    int32_t v = symbols[i] - offset[c];
    uint32_t raw = 0;
    if (v < 0) {
      raw = (uint32_t)(-2 * v - 1);
      v = escape;
    } else if (v >= escape) {
      raw = (uint32_t)(2 * (v - escape));
      v = escape;
    }
    items[n_items].start = (uint16_t)row[v];
    items[n_items].range = (uint16_t)(row[v + 1] - row[v]);
    items[n_items].raw = 0;
    ++n_items;
    if (v == escape) {
      int digits = 0;
      while (digits < 8 && (raw >> (digits * ORC_DIGIT_BITS)) != 0) ++digits;
      int left = digits;
      while (left >= (int)ORC_DIGIT_MAX) {
        items[n_items].start = ORC_DIGIT_MAX;
        items[n_items].raw = 1;
        ++n_items;
        left -= ORC_DIGIT_MAX;
      }
      items[n_items].start = (uint16_t)left;
      items[n_items].raw = 1;
      ++n_items;
      for (int d = 0; d < digits; ++d) {
        items[n_items].start = (uint16_t)((raw >> (d * ORC_DIGIT_BITS)) & ORC_DIGIT_MAX);
        items[n_items].raw = 1;
        ++n_items;
      }
    }
  }

  /* words are produced last-to-first; collect them in a scratch that is filled
   * from its end */
  size_t max_words = n_items + 2;
  uint32_t *words = (uint32_t *)malloc(sizeof(uint32_t) * max_words);
  if (!words) {
    free(items);
    return -1;
  }
  uint32_t *wp = words + max_words;
  uint64_t x = ORC_STATE_LOW;
  for (size_t k = n_items; k-- > 0;) {
    const orc_item it = items[k];
    if (!it.raw) {
      const uint64_t freq = it.range;
      const uint64_t limit = ((ORC_STATE_LOW >> ORC_PROB_BITS) << 32) * freq;
      if (x >= limit) {
        *--wp = (uint32_t)x;
        x >>= 32;
      }
      x = ((x / freq) << ORC_PROB_BITS) + (x % freq) + it.start;
    } else {
      const uint64_t freq = 1ull << (ORC_PROB_BITS - ORC_DIGIT_BITS);
      const uint64_t limit = ((ORC_STATE_LOW >> ORC_PROB_BITS) << 32) * freq;
      if (x >= limit) {
        *--wp = (uint32_t)x;
        x >>= 32;
      }
      x = (x << ORC_DIGIT_BITS) | it.start;
    }
  }
  wp -= 2;
  wp[0] = (uint32_t)x;
  wp[1] = (uint32_t)(x >> 32);

  size_t nbytes = (size_t)((words + max_words) - wp) * sizeof(uint32_t);
  long ret = -1;
  if (nbytes <= cap) {
    memcpy(out, wp, nbytes); /* native little-endian words */
    ret = (long)nbytes;
  }
  free(words);
  free(items);
  return ret;
}

/* ------------------------------------------------------------------ A14 */
static inline uint32_t orc_take_digit(uint64_t *x, const uint32_t **wp) {
  uint32_t d = (uint32_t)(*x & ORC_DIGIT_MAX);
  *x >>= ORC_DIGIT_BITS;
  if (*x < ORC_STATE_LOW) {
    *x = (*x << 32) | **wp;
    ++*wp;
  }
  return d;
}

/* Inverse of orc_rans_encode: RansDecoder.decode_with_indexes
 * (hub/compressor.py:124 -> EntropyModel.decompress).  Returns bytes consumed
 * or -1 on overrun. */
long orc_rans_decode(const uint8_t *in, size_t nbytes, const int32_t *index,
                     int n, const int32_t *cdf, int W, const int32_t *cdf_len,
                     const int32_t *offset, int32_t *symbols_out) {
  if (nbytes < 8 || (nbytes & 3)) return -1;
  /* one guard word so that a trailing renormalisation read stays in bounds */
  uint32_t *buf = (uint32_t *)calloc(nbytes / 4 + 2, sizeof(uint32_t));
  if (!buf) return -1;
  memcpy(buf, in, nbytes);
  const uint32_t *wp = buf;
  uint64_t x = (uint64_t)wp[0] | ((uint64_t)wp[1] << 32);
  wp += 2;
  const uint32_t mask = (1u << ORC_PROB_BITS) - 1;

  for (int i = 0; i < n; ++i) {
    const int c = index ? index[i] : i;
    const int32_t *row = cdf + (size_t)c * W;
    const int32_t len = cdf_len[c];
    const int32_t escape = len - 2;
    const uint32_t cf = (uint32_t)(x & mask);
    int s = 0;
    while (s < len && !((uint32_t)row[s] > cf)) ++s;
    s -= 1;
    const uint64_t start = (uint32_t)row[s];
    const uint64_t freq = (uint32_t)(row[s + 1] - row[s]);
    x = freq * (x >> ORC_PROB_BITS) + (x & mask) - start;
    if (x < ORC_STATE_LOW) {
      x = (x << 32) | *wp;
      ++wp;
    }
    int32_t v = s;
    if (v == escape) {
      int32_t d = (int32_t)orc_take_digit(&x, &wp);
      int32_t digits = d;
      while (d == (int32_t)ORC_DIGIT_MAX) {
        d = (int32_t)orc_take_digit(&x, &wp);
        digits += d;
      }
      int32_t raw = 0;
      for (int j = 0; j < digits; ++j) {
        d = (int32_t)orc_take_digit(&x, &wp);
        raw |= (int32_t)((uint32_t)d << (j * ORC_DIGIT_BITS));
      }
      v = raw >> 1;
      if (raw & 1)
        v = -v - 1;
      else
        v += escape;
    }
    symbols_out[i] = v + offset[c];
    if ((size_t)(wp - buf) > nbytes / 4 + 1) {
      free(buf);
      return -1;
    }
  }
  long used = (long)((wp - buf) * sizeof(uint32_t));
  free(buf);
  return used;
}

/* Batch helpers used by the cpu_baseline leg of bench.py and by tests: B
 * independent images, strings packed back to back, out_off[B+1] byte offsets. */
long orc_rans_encode_batch(const int32_t *symbols, int B, int C,
                           const int32_t *cdf, int W, const int32_t *cdf_len,
                           const int32_t *offset, uint8_t *out, size_t cap,
                           uint64_t *out_off) {
  size_t pos = 0;
  out_off[0] = 0;
  for (int b = 0; b < B; ++b) {
    long k = orc_rans_encode(symbols + (size_t)b * C, NULL, C, cdf, W, cdf_len,
                             offset, out + pos, cap - pos);
    if (k < 0) return -1;
    pos += (size_t)k;
    out_off[b + 1] = pos;
  }
  return (long)pos;
}

int orc_rans_decode_batch(const uint8_t *in, const uint64_t *off, int B, int C,
                          const int32_t *cdf, int W, const int32_t *cdf_len,
                          const int32_t *offset, int32_t *symbols_out) {
  for (int b = 0; b < B; ++b) {
    long k = orc_rans_decode(in + off[b], (size_t)(off[b + 1] - off[b]), NULL,
                             C, cdf, W, cdf_len, offset,
                             symbols_out + (size_t)b * C);
    if (k < 0) return -1;
  }
  return 0;
}

/* A4 + quantiser of A13: sym = rint((float(z) + bias) * exp_scale - median),
 * each operation rounded to fp32 separately (hub/compressor.py:105-109 then
 * EntropyModel.quantize "symbols": torch.round(x - means).int()). */
void orc_quantise(const float *z, int B, int C, const float *bias,
                  const float *exp_scale, const float *median,
                  int32_t *symbols_out) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      volatile float t = z[(size_t)b * C + c] + bias[c];
      volatile float u = t * exp_scale[c];
      volatile float d = u - median[c];
      symbols_out[(size_t)b * C + c] = (int32_t)__builtin_rintf(d);
    }
}
