// chain_stats.cpp -- EXPERIMENT (round 3 design study, not product code).
// Runs the reference's sequential deflate-fast parse (deflate-fast.mbt:123-270) over synthetic
// single-window streams and reports how the "slot chain + inserted set" formulation would behave:
// for every probe the depth k of the first INSERTED predecessor on the chain of positions with
// the same hash slot, match-length distribution, short-distance predecessors, etc.
//   g++ -O2 -std=c++17 -I include tools/experiments/chain_stats.cpp moonbit-flate_amd/csrc/synth.cpp -lpthread -o /tmp/chain_stats
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "flate_hip.h"

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash4(uint32_t u) { return (u * 0x1e35a7bdu) >> 18; }

int main(int argc, char **argv) {
  int kind = argc > 1 ? atoi(argv[1]) : FLATE_SYNTH_TEXT;
  int nstreams = argc > 2 ? atoi(argv[2]) : 8;
  const int n = 65535;
  std::vector<uint8_t> buf((size_t)nstreams * 65536 + 64);
  if (argc > 3) {  // raw file of nstreams x 65536 bytes (tests/util.py kinds "low", "runs", ...)
    FILE *f = fopen(argv[3], "rb");
    if (!f || fread(buf.data(), 65536, nstreams, f) != (size_t)nstreams) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
    fclose(f);
  } else {
    flate_hip_synth_fill(kind, 0x5EED0001ull, 0, nstreams, 65536, buf.data(), 1);
  }
  uint64_t probes = 0, succ = 0, fail = 0, matches = 0, inserted_total = 0, positions = 0;
  uint64_t depth_succ[34] = {0}, depth_fail[34] = {0}, depth_all_lanes[34] = {0};
  uint64_t lenhist[260] = {0};
  uint64_t shortpred[4] = {0};  // probes whose candidate is < 64, < 128, < 256, <512 back
  uint64_t events = 0;
  uint64_t fail_kind[3] = {0};  // chain empty/out of range, other bytes, (unused)
  for (int si = 0; si < nstreams; ++si) {
    const uint8_t *src = buf.data() + (size_t)si * 65536;
    std::vector<int32_t> prev(n, -1);  // previous position with the same slot
    {
      std::vector<int32_t> last(16384, -1);
      for (int p = 0; p + 4 <= n; ++p) { uint32_t h = hash4(ld32(src + p)); prev[p] = last[h]; last[h] = p; }
    }
    std::vector<uint8_t> ins(n, 0);
    std::vector<int32_t> table(16384, -1);
    auto lookup = [&](int s, uint32_t cv, int *depth, int *kindp) -> int {
      // candidate through the chain; also cross-check with the real table
      int k = 0, q = prev[s];
      int cand = -1;
      while (q >= 0 && s - q <= 32768) { ++k; if (ins[q]) { cand = q; break; } q = prev[q]; }
      uint32_t h = hash4(cv);
      int t = table[h];
      if (t >= 0 && s - t > 32768) t = -1;
      if (t != cand) { fprintf(stderr, "MISMATCH s=%d t=%d cand=%d\n", s, t, cand); exit(1); }
      *depth = cand >= 0 ? k : -k;  // negative: walked k steps and found none
      bool ok = cand >= 0 && ld32(src + cand) == cv;
      *kindp = cand < 0 ? 0 : (ok ? 2 : 1);
      return ok ? cand : -1;
    };
    auto insert = [&](int s, uint32_t cv) { table[hash4(cv)] = s; ins[s] = 1; };
    const int s_limit = n - 15;
    int s = 0; uint32_t cv = ld32(src);
    bool done = false;
    auto account = [&](int s, int depth, int kindv) {
      ++probes;
      int d = depth < 0 ? -depth : depth;
      if (d > 33) d = 33;
      if (kindv == 2) { ++succ; depth_succ[d]++; } else { ++fail; depth_fail[d]++; fail_kind[kindv]++; }
    };
    while (!done) {
      int skip = 32, next_s = s, cand = -1;
      for (;;) {
        s = next_s; int step = skip >> 5; next_s = s + step; skip += step;
        if (next_s > s_limit) { done = true; break; }
        int depth, kv; cand = lookup(s, cv, &depth, &kv); account(s, depth, kv);
        uint32_t now = ld32(src + next_s);
        insert(s, cv);
        if (cand < 0) { cv = now; continue; }
        break;
      }
      if (done) break;
      ++events;
      for (;;) {
        int pf = s; s += 4; int t = cand + 4; int limit = std::min(n - s, 254); int l = 0;
        while (l < limit && src[s + l] == src[t + l]) ++l;
        ++matches; lenhist[l + 4]++;
        int dist = pf - cand;
        if (dist < 64) shortpred[0]++; if (dist < 128) shortpred[1]++; if (dist < 256) shortpred[2]++; if (dist < 512) shortpred[3]++;
        s += l;
        if (s >= s_limit) { done = true; break; }
        insert(s - 1, ld32(src + s - 1));
        uint32_t x1 = ld32(src + s);
        int depth, kv; cand = lookup(s, x1, &depth, &kv); account(s, depth, kv);
        insert(s, x1);
        if (cand < 0) { cv = ld32(src + s + 1); s += 1; break; }
      }
    }
    for (int p = 0; p < n; ++p) inserted_total += ins[p];
    positions += n;
    // depth for ALL positions at the end-state inserted set (what a speculative all-lane evaluation sees)
    for (int p = 0; p + 4 <= n; ++p) {
      int k = 0, q = prev[p]; bool f = false;
      while (q >= 0 && p - q <= 32768) { ++k; if (ins[q]) { f = true; break; } q = prev[q]; }
      int d = std::min(k, 33); (void)f;
      depth_all_lanes[d]++;
    }
  }
  printf("streams %d positions %llu inserted %.3f probes/window %.0f matches/window %.0f events/window %.0f\n", nstreams,
         (unsigned long long)positions, (double)inserted_total / positions, (double)probes / nstreams, (double)matches / nstreams, (double)events/nstreams);
  printf("successful lookups %.3f of probes; failed: empty-chain %.3f other-bytes %.3f\n", (double)succ / probes,
         (double)fail_kind[0] / probes, (double)fail_kind[1] / probes);
  double cs = 0, cf = 0;
  printf("depth  succ(cum)  fail(cum)  all-lanes(cum)\n");
  double ca = 0; uint64_t tot_all = 0; for (int d = 0; d < 34; ++d) tot_all += depth_all_lanes[d];
  for (int d = 0; d < 34; ++d) {
    cs += depth_succ[d]; cf += depth_fail[d]; ca += depth_all_lanes[d];
    if (d <= 16 || d == 33) printf("%3d   %.4f    %.4f   %.4f\n", d, cs / succ, cf / fail, ca / tot_all);
  }
  uint64_t ge16 = 0, ge17 = 0, ge32 = 0, ge20=0; double sumlen = 0;
  for (int l = 4; l < 260; ++l) { if (l >= 16) ge16 += lenhist[l]; if (l >= 17) ge17 += lenhist[l]; if (l>=20) ge20+=lenhist[l]; if (l >= 32) ge32 += lenhist[l]; sumlen += (double)l * lenhist[l]; }
  printf("match len: mean %.2f  >=16 %.4f  >=17 %.4f >=20 %.4f >=32 %.4f\n", sumlen / matches, (double)ge16 / matches, (double)ge17 / matches, (double)ge20/matches, (double)ge32 / matches);
  printf("match dist <64 %.4f <128 %.4f <256 %.4f <512 %.4f\n", (double)shortpred[0] / matches, (double)shortpred[1] / matches, (double)shortpred[2] / matches, (double)shortpred[3]/matches);
  return 0;
}
