// synth.cpp -- bit-reproducible synthetic workloads (host side; no GPU involved).
// The reference ships no corpus; these stand in for the inputs BASELINE.md section 3
// names.  S-ramp is the reference's own test pattern (deflate-fast_test.mbt:15-24).
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "flate_hip.h"

namespace {

inline uint64_t splitmix64(uint64_t &x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

constexpr int kVocab = 4096;

struct Vocab {
  std::vector<uint8_t> bytes;
  uint32_t off[kVocab + 1];
  uint64_t cum[kVocab];  // cumulative Zipf(1) weights floor(2^32 / (k+1))
  uint64_t total;
  Vocab() {
    static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
    uint32_t o = 0;
    uint64_t c = 0;
    for (int w = 0; w < kVocab; ++w) {
      uint64_t st = 0x5EED0000ull + (uint64_t)w * 0x100000001B3ull;
      uint64_t r = splitmix64(st);
      int lg = 0;
      while ((2u << lg) <= (unsigned)(w + 2)) ++lg;  // floor(log2(w+2))
      int span = 2 + std::min(10, lg);
      int len = 2 + (int)(r % (uint64_t)span);
      off[w] = o;
      for (int k = 0; k < len; ++k) {
        uint64_t q = splitmix64(st);
        int a = (int)(q % 26), b = (int)((q >> 20) % 26);
        bytes.push_back((uint8_t)letters[std::min(a, b)]);
      }
      o += (uint32_t)len;
      c += (1ull << 32) / (uint64_t)(w + 1);
      cum[w] = c;
    }
    off[kVocab] = o;
    total = c;
  }
};

const Vocab &vocab() {
  static const Vocab v;
  return v;
}

void fill_text(uint64_t seed, uint64_t stream, uint8_t *dst, uint64_t len) {
  const Vocab &v = vocab();
  uint64_t st = seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ 0x5EED0001ull;
  uint64_t pos = 0;
  while (pos < len) {
    uint64_t r = splitmix64(st);
    uint64_t u = r % v.total;
    int w = (int)(std::upper_bound(v.cum, v.cum + kVocab, u) - v.cum);
    if (w >= kVocab) w = kVocab - 1;
    const uint8_t *wb = v.bytes.data() + v.off[w];
    uint32_t wl = v.off[w + 1] - v.off[w];
    for (uint32_t k = 0; k < wl && pos < len; ++k) dst[pos++] = wb[k];
    uint64_t p = r >> 40;
    if (p % 61 == 0) {
      if (pos < len) dst[pos++] = '.';
      if (pos < len) dst[pos++] = '\n';
    } else if (p % 13 == 0) {
      if (pos < len) dst[pos++] = ',';
      if (pos < len) dst[pos++] = ' ';
    } else {
      if (pos < len) dst[pos++] = ' ';
    }
  }
}

void fill_rand(uint64_t seed, uint64_t stream, uint8_t *dst, uint64_t len) {
  uint64_t st = seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ 0x5EED0002ull;
  uint64_t pos = 0;
  while (pos < len) {
    uint64_t r = splitmix64(st);
    for (int k = 0; k < 8 && pos < len; ++k) dst[pos++] = (uint8_t)(r >> (8 * k));
  }
}

void fill_one(int kind, uint64_t seed, uint64_t stream, uint8_t *dst, uint64_t len) {
  switch (kind) {
    case FLATE_SYNTH_RAMP:
      for (uint64_t i = 0; i < len; ++i) dst[i] = (uint8_t)(i & 127);
      break;
    case FLATE_SYNTH_TEXT:
      fill_text(seed, stream, dst, len);
      break;
    case FLATE_SYNTH_RAND:
      fill_rand(seed, stream, dst, len);
      break;
    default:
      memset(dst, 0, len);
      break;
  }
}

}  // namespace

extern "C" int flate_hip_synth_fill(int kind, uint64_t seed, uint64_t first_stream,
                                    uint32_t n_streams, uint64_t stream_len, uint8_t *out,
                                    int nthreads) {
  if (!out || kind < 0 || kind > FLATE_SYNTH_ZERO) return FLATE_HIP_E_INVALID;
  (void)vocab();
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n_streams) nthreads = n_streams ? (int)n_streams : 1;
  auto work = [&](int t) {
    uint32_t lo = (uint32_t)((uint64_t)n_streams * t / nthreads);
    uint32_t hi = (uint32_t)((uint64_t)n_streams * (t + 1) / nthreads);
    for (uint32_t i = lo; i < hi; ++i)
      fill_one(kind, seed, first_stream + i, out + (uint64_t)i * stream_len, stream_len);
  };
  if (nthreads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  return FLATE_HIP_OK;
}
