// slot_cache_model.cpp -- EXPERIMENT (round 5 design study, not product code).
// would an exact write-back slot cache in LDS, in front of a guest block's
// L2-resident hash table, absorb the guests' table traffic?  This replays the WAVE algorithm's table
// accesses (the lane-accurate host model tests/host_model/lz77_wave_model.cpp: a dense batch gathers the
// slot of every lane speculatively, the commit writes the inserted lanes) through direct-mapped caches of
// 512 .. 8192 entries {slot, position, dirty} and reports
//   read hit rate   = gathers served by the cache / gathers the guest issues today (after the 2-bit tags)
//   write-backs     = dirty evictions / table writes the guest issues today
// for both allocation policies (write-allocate only; read- and write-allocate).
// Kill criterion of the verdict: read-hit < 50 % on S-text at 4 KiB of LDS per guest.
//   g++ -O2 -std=c++17 -I include tools/experiments/slot_cache_model.cpp moonbit-flate_amd/csrc/synth.cpp -lpthread -o /tmp/slot_cache_model
//   /tmp/slot_cache_model <kind 0 ramp|1 text|2 rand|3 zero> <nstreams> [windows per stream] [raw file]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "flate_hip.h"

namespace {
constexpr int kTableSize = 16384;
constexpr int kDenseKeep = 61;
inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint32_t hash4(uint32_t u) { return (u * 0x1e35a7bdu) >> 18; }
inline uint32_t tag_of(uint32_t u) { return ((u * 0x1e35a7bdu) >> 16) & 3u; }
inline int ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }
inline uint64_t below(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1); }
inline uint64_t upto(int l) { return l >= 63 ? ~0ull : ((1ull << (l + 1)) - 1); }
int scan_off(int e, int *step, const std::vector<uint32_t> &tab) {
  if (e < 32) { *step = 1; return e; }
  if (e < 48) { *step = 2; return 32 + 2 * (e - 32); }
  if (e < 59) { *step = 3; return 64 + 3 * (e - 48); }
  if (e < 67) { *step = 4; return 97 + 4 * (e - 59); }
  if (e + 1 >= (int)tab.size()) { *step = 1; return 1 << 24; }
  *step = (int)(tab[e + 1] - tab[e]);
  return (int)tab[e];
}
int common_prefix16(const uint8_t *a, const uint8_t *b) { int i = 0; while (i < 16 && a[i] == b[i]) ++i; return i; }

struct Cache {
  int n;            // entries (power of two), direct-mapped on the low bits of the slot
  bool read_alloc;  // a read miss brings the slot in (clean)
  std::vector<int32_t> slot;  // -1 = invalid
  std::vector<uint8_t> dirty;
  uint64_t reads = 0, read_hits = 0, writes = 0, write_hits = 0, writebacks = 0, fills = 0;
  Cache(int n_, bool ra) : n(n_), read_alloc(ra), slot(n_, -1), dirty(n_, 0) {}
  void reset() { std::fill(slot.begin(), slot.end(), -1); std::fill(dirty.begin(), dirty.end(), 0); }
  void read(uint32_t h) {
    ++reads;
    const int i = h & (n - 1);
    if (slot[i] == (int32_t)h) { ++read_hits; return; }
    if (read_alloc) {
      if (slot[i] >= 0 && dirty[i]) ++writebacks;
      slot[i] = (int32_t)h; dirty[i] = 0; ++fills;
    }
  }
  void write(uint32_t h) {
    ++writes;
    const int i = h & (n - 1);
    if (slot[i] == (int32_t)h) { ++write_hits; dirty[i] = 1; return; }
    if (slot[i] >= 0 && dirty[i]) ++writebacks;
    slot[i] = (int32_t)h; dirty[i] = 1;
  }
};
}  // namespace

int main(int argc, char **argv) {
  int kind = argc > 1 ? atoi(argv[1]) : FLATE_SYNTH_TEXT;
  int nstreams = argc > 2 ? atoi(argv[2]) : 16;
  const int nwin = argc > 3 ? atoi(argv[3]) : 1;  // LZ77 windows per stream (4 = config 3: the table persists, MoonBit rule for cross-window matches)
  const int n = 65535;
  const size_t slen = (size_t)nwin * 65536;
  std::vector<uint8_t> buf((size_t)nstreams * slen + 64);
  if (argc > 4) {
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(buf.data(), slen, nstreams, f) != (size_t)nstreams) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
    fclose(f);
  } else {
    flate_hip_synth_fill(kind, 0x5EED0001ull, 0, nstreams, slen, buf.data(), 1);
  }
  std::vector<uint32_t> scantab;
  { uint32_t skip = 32, pos = 0; while (pos <= 65535) { scantab.push_back(pos); uint32_t st = skip >> 5; pos += st; skip += st; } scantab.push_back(1 << 24); }

  const int sizes[] = {512, 1024, 2048, 4096, 8192};
  std::vector<Cache> caches;           // fed with the gathers that pass the 2-bit tags (tags kept beside the cache)
  std::vector<Cache> caches_notag;     // fed with every lane's gather (the cache replaces the tags)
  for (int ra = 0; ra < 2; ++ra) for (int s : sizes) { caches.emplace_back(s, ra != 0); caches_notag.emplace_back(s, ra != 0); }
  uint64_t batches = 0, sparse_batches = 0, gathers_all = 0, gathers_tag = 0, commits = 0, matches = 0;
  // a second question: how many BATCHES would need no L2 gather at all (the wave's wait disappears only then)
  std::vector<uint64_t> batch_allhit(caches.size(), 0);

  for (int si = 0; si < nstreams; ++si) {
    const uint8_t *stream = buf.data() + (size_t)si * slen;
    std::vector<uint32_t> table(kTableSize, 0);
    std::vector<uint8_t> tags(kTableSize, 0);
    for (int win = 0; win < nwin; ++win) {
    const uint32_t W = (uint32_t)win * 65535u;
    const uint8_t *src = stream + W;
    // (a window unit may run on another block: the cache starts cold, its dirty entries were written back)
    for (auto &c : caches) { for (int i = 0; i < c.n && win > 0; ++i) if (c.slot[i] >= 0 && c.dirty[i]) ++c.writebacks; c.reset(); }
    for (auto &c : caches_notag) { for (int i = 0; i < c.n && win > 0; ++i) if (c.slot[i] >= 0 && c.dirty[i]) ++c.writebacks; c.reset(); }
    const int s_limit = n - 15;
    int s = -1; bool sparse = false; int scan_base = 0, e_idx = 0; bool done = false;
    auto extend = [&](int pf, uint32_t cand, int have) -> int {
      int limit = n - pf; if (limit > 258) limit = 258;
      if (cand + 4 < W) return 4;  // MoonBit: prev is empty
      int l = have; const uint8_t *a = src + pf, *b = stream + cand;
      while (l < limit && a[l] == b[l]) ++l; return l;
    };
    while (!done) {
      if (!sparse) {
        ++batches;
        const int B = s - 1;
        int q[64]; uint32_t cv[64], h[64], old[64], A1[64]; uint8_t own[64][16]; int mlen[64];
        uint64_t LD = 0, E1 = 0, E2 = 0, OK = 0, DUP = 0;
        for (int L = 0; L < 64; ++L) {
          q[L] = B + L; cv[L] = h[L] = old[L] = 0; mlen[L] = 0; A1[L] = W + (uint32_t)q[L] + 1;
          if (q[L] >= 0 && q[L] + 1 <= s_limit) E1 |= 1ull << L;
          if (q[L] >= 0 && q[L] + 2 <= s_limit) E2 |= 1ull << L;
        }
        LD = E1;
        std::vector<uint8_t> hit_all(caches.size(), 1);
        for (int L = 0; L < 64; ++L) {
          if (!((LD >> L) & 1)) continue;
          memcpy(own[L], src + q[L], 16);
          cv[L] = ld32(src + q[L]); h[L] = hash4(cv[L]);
          const bool tag_ok = tags[h[L]] == tag_of(cv[L]);
          ++gathers_all;
          for (auto &c : caches_notag) c.read(h[L]);
          if (tag_ok) {
            ++gathers_tag;
            for (size_t k = 0; k < caches.size(); ++k) {
              const uint64_t before = caches[k].read_hits;
              caches[k].read(h[L]);
              if (caches[k].read_hits == before) hit_all[k] = 0;
            }
          }
          old[L] = table[h[L]];
          if (old[L] != 0 && A1[L] - old[L] <= 32768u) {
            mlen[L] = common_prefix16(own[L], stream + (old[L] - 1));
            if (mlen[L] >= 4) OK |= 1ull << L;
          }
        }
        for (size_t k = 0; k < caches.size(); ++k) batch_allhit[k] += hit_all[k];
        for (int L = 0; L < 64; ++L) for (int M = 0; M < 64; ++M)
          if (L != M && ((LD >> L) & 1) && ((LD >> M) & 1) && h[L] == h[M]) DUP |= 1ull << L;
        uint64_t INS = 0; int a = 0; bool batch_over = false;
        while (!batch_over) {
          const uint64_t a_ins = (LD >> a) & 1 ? (1ull << a) : 0;
          uint64_t R = 0;
          if (a + 1 <= 63 && ((LD >> (a + 1)) & 1) && q[a + 1] >= 0) R |= 1ull << (a + 1);
          bool scan_ended = false; int consumed = 0;
          { const int b = a + 2;
            for (int e = 0;; ++e) { int step; const int L = b + scan_off(e, &step, scantab); if (L > 63) break;
              const bool ex = step == 1 ? ((E1 >> L) & 1) : ((E2 >> L) & 1); if (!ex) { scan_ended = true; break; }
              R |= 1ull << L; consumed = e + 1; } }
          uint64_t T = 0, rem = R; int f = 64; uint32_t cand = 0; int have = 0;
          for (;;) {
            const int fv = ctz64(OK & rem & ~DUP), fd = ctz64(DUP & rem);
            if (fv < fd) { f = fv; cand = old[fv] - 1; have = mlen[fv]; T |= rem & upto(fv); break; }
            if (fd == 64) { T |= rem; break; }
            T |= rem & below(fd);
            uint64_t G = 0;
            for (int L = 0; L < fd; ++L) if (((INS | T | a_ins) >> L) & 1 && h[L] == h[fd]) G |= 1ull << L;
            bool v; uint32_t cnd; int ml;
            if (G) { const int i = 63 - __builtin_clzll(G); v = cv[i] == cv[fd]; cnd = W + (uint32_t)q[i]; ml = common_prefix16(own[fd], own[i]); }
            else { v = (OK >> fd) & 1; cnd = old[fd] - 1; ml = mlen[fd]; }
            T |= 1ull << fd;
            if (v) { f = fd; cand = cnd; have = ml; break; }
            rem &= ~upto(fd);
          }
          if (f == 64) {
            if (scan_ended) { INS |= T | a_ins; done = true; }
            else if (a == 0) { INS |= T | a_ins; sparse = true; scan_base = s + 1; e_idx = consumed; }
            batch_over = true;
          } else {
            INS |= T | a_ins; ++matches;
            const int pf = q[f];
            const int total = have < 16 ? ((cand + 4 < W) ? 4 : have) : extend(pf, cand, 16);
            s = pf + total;
            if (s >= s_limit) { done = true; batch_over = true; }
            else { a = s - 1 - B; if (a > kDenseKeep) batch_over = true; }
          }
        }
        // commit: the last inserted lane of every same-slot group writes (what the guest does)
        for (int L = 0; L < 64; ++L) if ((INS >> L) & 1) {
          bool later = false;
          for (int M = L + 1; M < 64; ++M) if (((INS >> M) & 1) && h[M] == h[L]) later = true;
          table[h[L]] = A1[L]; tags[h[L]] = (uint8_t)tag_of(cv[L]);
          if (!later) { ++commits; for (auto &c : caches) c.write(h[L]); for (auto &c : caches_notag) c.write(h[L]); }
        }
      } else {
        ++sparse_batches;
        int p[64], step[64]; uint64_t EX = 0;
        for (int L = 0; L < 64; ++L) { p[L] = scan_base + scan_off(e_idx + L, &step[L], scantab); if (p[L] + step[L] <= s_limit) EX |= 1ull << L; }
        const int nexist = __builtin_popcountll(EX);
        if (nexist == 0) { done = true; break; }
        int f = 64; uint32_t cand = 0;
        for (int L = 0; L < nexist; ++L) {
          const uint32_t cvL = ld32(src + p[L]), hL = hash4(cvL), A1L = W + (uint32_t)p[L] + 1;
          const uint32_t o = table[hL];
          ++gathers_all; ++gathers_tag; ++commits;
          for (auto &c : caches) { c.read(hL); c.write(hL); }
          for (auto &c : caches_notag) { c.read(hL); c.write(hL); }
          table[hL] = A1L; tags[hL] = (uint8_t)tag_of(cvL);
          if (o != 0 && A1L - o <= 32768u && ld32(stream + (o - 1)) == cvL) { f = L; cand = o - 1; break; }
        }
        if (f == 64) { if (nexist < 64) { done = true; break; } e_idx += 64; continue; }
        ++matches;
        const int pf = p[f]; const int total = extend(pf, cand, 4);
        s = pf + total; sparse = false; if (s >= s_limit) done = true;
      }
    }
    }
  }
  printf("streams %d x %d windows: dense batches/stream %.0f sparse %.1f matches %.0f | table gathers per dense batch: all lanes %.1f, after 2-bit tags %.1f (%.1f %% stopped) | table writes per batch %.1f\n",
         nstreams, nwin, (double)batches / nstreams, (double)sparse_batches / nstreams, (double)matches / nstreams,
         (double)gathers_all / batches, (double)gathers_tag / batches, 100.0 * (1.0 - (double)gathers_tag / gathers_all), (double)commits / batches);
  printf("%-34s %8s %10s %12s %14s %16s\n", "cache", "entries", "read hit", "write hit", "write-backs/wr", "batches w/o L2 rd");
  for (int pass = 0; pass < 2; ++pass) {
    auto &cs = pass == 0 ? caches : caches_notag;
    for (size_t k = 0; k < cs.size(); ++k) {
      const Cache &c = cs[k];
      char name[64];
      snprintf(name, sizeof name, "%s, %s", pass == 0 ? "beside the tags" : "instead of the tags", c.read_alloc ? "rd+wr alloc" : "wr alloc");
      printf("%-34s %8d %9.1f%% %11.1f%% %13.1f%% ", name, c.n, 100.0 * c.read_hits / c.reads, 100.0 * c.write_hits / c.writes, 100.0 * c.writebacks / c.writes);
      if (pass == 0) printf("%15.1f%%\n", 100.0 * batch_allhit[k] / batches); else printf("%16s\n", "-");
    }
  }
  return 0;
}
