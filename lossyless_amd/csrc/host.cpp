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
