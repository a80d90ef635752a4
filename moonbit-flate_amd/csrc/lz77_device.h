// lz77_device.h -- device helpers shared by the match-finder kernels (lz77_kernels.hip,
// lz77_team_kernels.hip): unaligned loads, lane masks, 16-byte prefix compare, match extension
// (match_len, deflate-fast.mbt:286-342) and the per-stream geometry of the LZ77 chunks.
#pragma once

#include "flate_kernels.h"

namespace flate {

constexpr int kDenseKeep = 61;  // keep using a dense batch while the next s-1 lane <= this (58..61 measured: 61 best by 0.8 %)
// An event that starts at lane 62 or 63 has no scan lane left in its batch: the general path's probe masks would
// shift by 64 (undefined; the hardware shifts by 0) and the batch ends where it began -- the build with 62 hung its
// run (profiles/r05/README.md section 9).  The parser's progress guard (kStatusNoProgress) is the second line.
static_assert(kDenseKeep >= 0 && kDenseKeep <= 61, "an event must start at a lane that leaves it a scan lane (a + 2 <= 63)");
// multi-window streams: 16-bit modular table slots with periodic sweeps (see lz77_stream).  The sweep period and the
// reach of a sparse batch share one budget (the assert below); round 6 moved it from 8192 / 16384 to 20480 / 4096:
// 2.5 sweeps fewer per 65535-byte window (config 3's match finder 17.69 -> 17.40 ms, profiles/r06/README.md section 4),
// and what the smaller reach costs is a few batches on incompressible windows only: the stride of a scan without a match
// is about (positions scanned) / 32 (deflate-fast.mbt:178-187), so beyond 2048 such positions a sparse batch holds
// 4096 / stride events instead of 64 -- about 15 more batches in a window of random bytes, which takes ~300 probes in all.
#ifndef FLATE_LZ_SWEEP_EVERY  // (A/B builds: tools/build_variant.sh ... -DFLATE_LZ_SWEEP_EVERY=.. -DFLATE_LZ_SPAN_MAX=.. -DFLATE_LZ_MARKER_BACK=..)
#define FLATE_LZ_SWEEP_EVERY 20480
#endif
#ifndef FLATE_LZ_SPAN_MAX
#define FLATE_LZ_SPAN_MAX 4096
#endif
#ifndef FLATE_LZ_MARKER_BACK
#define FLATE_LZ_MARKER_BACK 36864
#endif
constexpr uint32_t kSweepEvery = FLATE_LZ_SWEEP_EVERY, kSpanMax = FLATE_LZ_SPAN_MAX, kMarkerBack = FLATE_LZ_MARKER_BACK;
// A sweep at R leaves no slot older than 32768 and writes dead slots as "kMarkerBack behind R".  Until the next sweep a
// lookup comes from at most R + kSweepEvery + one batch (64 positions dense, kSpanMax sparse): the marker must read as
// out of range (> 32768) and every distance, the marker's included, must stay below 2^16 to be told apart.
static_assert(kMarkerBack > 32768u && kSweepEvery + kSpanMax + 64u + kMarkerBack < 65536u && kSpanMax >= 64u,
              "16-bit modular table slots: sweep period + batch span + marker distance must stay below 2^16");

FLATE_D uint32_t ld32(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);  // gfx950: one unaligned global_load_dword
  return v;
}

// 1..3 trailing bytes (never reads past p[rem-1])
FLATE_D uint32_t ld_partial(const uint8_t *p, int rem) {
  uint32_t v = p[0];
  if (rem > 1) v |= (uint32_t)p[1] << 8;
  if (rem > 2) v |= (uint32_t)p[2] << 16;
  return v;
}

FLATE_D uint32_t rdlane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

struct ChunkGeom {
  const uint8_t *stream;  // first byte of the stream
  uint64_t len;           // stream length
  uint32_t nchunks;       // LZ77 chunks of this stream (enc_speed policy)
  uint32_t chunk0;        // global index of the first chunk
  uint64_t mbase;         // first match record of the first chunk
};

FLATE_D ChunkGeom stream_geom(const LzParams &P, uint32_t sid) {
  ChunkGeom g;
  uint64_t a = P.in_off[sid], b = P.in_off[sid + 1];
  g.stream = P.in + a;
  g.len = b - a;
  g.chunk0 = P.chunk_base[sid];
  g.nchunks = P.chunk_base[sid + 1] - g.chunk0;
  g.mbase = (uint64_t)g.chunk0 * kMatchCapPerChunk;
  return g;
}

FLATE_D uint64_t lanes_below(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1ull); }
FLATE_D uint64_t lanes_upto(int l) { return l >= 63 ? ~0ull : ((1ull << (l + 1)) - 1ull); }
FLATE_D int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }

// wave-wide OR of a 32-bit value (DPP row shifts / broadcasts, no LDS traffic)
template <int CTRL, int ROW_MASK>
FLATE_D uint32_t dpp_or(uint32_t v) {
  return v | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
FLATE_D uint32_t wave_or(uint32_t v) {
  v = dpp_or<0x111, 0xf>(v);
  v = dpp_or<0x112, 0xf>(v);
  v = dpp_or<0x114, 0xf>(v);
  v = dpp_or<0x118, 0xf>(v);
  v = dpp_or<0x142, 0xa>(v);
  v = dpp_or<0x143, 0xc>(v);
  return rdlane(v, 63);
}

FLATE_D uint4 ld128(const uint8_t *p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);  // one unaligned global_load_dwordx4
  return v;
}

// common prefix (bytes, 0..16) of two 16-byte strings
FLATE_D int prefix16(uint4 a, uint4 b) {
  const uint64_t lo = (uint64_t)(a.x ^ b.x) | ((uint64_t)(a.y ^ b.y) << 32);
  const uint64_t hi = (uint64_t)(a.z ^ b.z) | ((uint64_t)(a.w ^ b.w) << 32);
  if (lo) return __builtin_ctzll(lo) >> 3;
  if (hi) return 8 + (__builtin_ctzll(hi) >> 3);
  return 16;
}

// Total match length at chunk position pf against absolute position cand, `have` bytes
// already known equal (match_len, deflate-fast.mbt:286-342), 64 lanes x 4 bytes.
FLATE_D int extend_match(const uint8_t *src, const uint8_t *stream, uint32_t W, int n, int pf,
                         uint32_t cand, int have, uint32_t compat_go, int lane) {
  if (!compat_go && cand + 4 < W) return 4;  // MoonBit: prev window is empty
  int limit = n - pf;
  if (limit > 258) limit = 258;
  const int o = have + 4 * lane;
  uint32_t x = 0;
  if (o < limit) {
    const int r = limit - o;
    const uint8_t *pa = src + pf + o, *pb = stream + cand + o;
    x = r >= 4 ? (ld32(pa) ^ ld32(pb)) : (ld_partial(pa, r) ^ ld_partial(pb, r));
  }
  const uint64_t mm = __ballot(x != 0);
  if (!mm) return limit;
  const int k = __builtin_ctzll(mm);
  return have + 4 * k + (__builtin_ctz(rdlane(x, k)) >> 3);
}

}  // namespace flate
