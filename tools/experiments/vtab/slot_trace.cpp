// slot_trace.cpp -- EXPERIMENT (round 6): the table accesses of the WAVE algorithm on
// S-text, batch by batch, as a guest block issues them (the lane-accurate replay of tools/experiments/
// slot_cache_model.cpp / tests/host_model/lz77_wave_model.cpp), written as a trace that vtab_bench.hip replays
// through a hash table kept in VGPRs.
//   per batch: 64 x u32 word = slot h [13:0] | lookup << 14 | insert << 15 | (position + 1) << 16
//              64 x u16 expected result of the lookup (the slot's value before the batch's inserts; 0 where no lookup)
//   lookup: the lane gathers its slot (every lane that may be probed: the register-table form needs no tags)
//   insert: the lane commits its position (the last inserted lane of every same-slot group, as the guests do)
//   g++ -O2 -std=c++17 -I include tools/experiments/vtab/slot_trace.cpp moonbit-flate_amd/csrc/synth.cpp -lpthread -o build/exp/slot_trace
//   build/exp/slot_trace <nstreams> <out file>
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

}  // namespace

int main(int argc, char **argv) {
  const int nstreams = argc > 1 ? atoi(argv[1]) : 8;
  const char *path = argc > 2 ? argv[2] : "slot_trace.bin";
  const int n = 65535;
  const size_t slen = 65536;
  std::vector<uint8_t> buf((size_t)nstreams * slen + 64);
  flate_hip_synth_fill(FLATE_SYNTH_TEXT, 0x5EED0001ull, 0, nstreams, slen, buf.data(), 1);
  std::vector<uint32_t> scantab;
  { uint32_t skip = 32, pos = 0; while (pos <= 65535) { scantab.push_back(pos); uint32_t st = skip >> 5; pos += st; skip += st; } scantab.push_back(1 << 24); }
  std::vector<uint32_t> words;     // 64 per batch
  std::vector<uint16_t> expect;    // 64 per batch
  std::vector<uint32_t> first(nstreams + 1, 0);  // first batch of every stream
  uint64_t lookups = 0, inserts = 0, batches = 0, sparse_batches = 0;
  for (int si = 0; si < nstreams; ++si) {
    const uint8_t *stream = buf.data() + (size_t)si * slen;
    std::vector<uint32_t> table(kTableSize, 0);
    const uint32_t W = 0;
    const uint8_t *src = stream;
    const int s_limit = n - 15;
    int s = -1; bool sparse = false; int scan_base = 0, e_idx = 0; bool done = false;
    auto extend = [&](int pf, uint32_t cand, int have) -> int {
      int limit = n - pf; if (limit > 258) limit = 258;
      int l = have; const uint8_t *a = src + pf, *b = stream + cand;
      while (l < limit && a[l] == b[l]) ++l; return l;
    };
    auto emit = [&](const uint32_t *w, const uint16_t *e) { words.insert(words.end(), w, w + 64); expect.insert(expect.end(), e, e + 64); ++batches; };
    while (!done) {
      uint32_t tw[64] = {0}; uint16_t te[64] = {0};
      if (!sparse) {
        const int B = s - 1;
        int q[64]; uint32_t cv[64], h[64], old[64], A1[64]; uint8_t own[64][16]; int mlen[64];
        uint64_t LD = 0, E1 = 0, E2 = 0, OK = 0, DUP = 0;
        for (int L = 0; L < 64; ++L) {
          q[L] = B + L; cv[L] = h[L] = old[L] = 0; mlen[L] = 0; A1[L] = W + (uint32_t)q[L] + 1;
          if (q[L] >= 0 && q[L] + 1 <= s_limit) E1 |= 1ull << L;
          if (q[L] >= 0 && q[L] + 2 <= s_limit) E2 |= 1ull << L;
        }
        LD = E1;
        for (int L = 0; L < 64; ++L) {
          if (!((LD >> L) & 1)) continue;
          memcpy(own[L], src + q[L], 16);
          cv[L] = ld32(src + q[L]); h[L] = hash4(cv[L]);
          old[L] = table[h[L]];
          tw[L] = h[L] | (1u << 14) | (A1[L] << 16); te[L] = (uint16_t)old[L]; ++lookups;
          if (old[L] != 0 && A1[L] - old[L] <= 32768u) {
            mlen[L] = common_prefix16(own[L], stream + (old[L] - 1));
            if (mlen[L] >= 4) OK |= 1ull << L;
          }
        }
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
            INS |= T | a_ins;
            const int pf = q[f];
            const int total = have < 16 ? have : extend(pf, cand, 16);
            s = pf + total;
            if (s >= s_limit) { done = true; batch_over = true; }
            else { a = s - 1 - B; if (a > kDenseKeep) batch_over = true; }
          }
        }
        for (int L = 0; L < 64; ++L) if ((INS >> L) & 1) {
          bool later = false;
          for (int M = L + 1; M < 64; ++M) if (((INS >> M) & 1) && h[M] == h[L]) later = true;
          table[h[L]] = A1[L];
          if (!later) { tw[L] |= 1u << 15; ++inserts; }
        }
        emit(tw, te);
      } else {
        ++sparse_batches;
        int p[64], step[64]; uint64_t EX = 0;
        for (int L = 0; L < 64; ++L) { p[L] = scan_base + scan_off(e_idx + L, &step[L], scantab); if (p[L] + step[L] <= s_limit) EX |= 1ull << L; }
        const int nexist = __builtin_popcountll(EX);
        if (nexist == 0) { done = true; break; }
        int f = 64; uint32_t cand = 0;
        // (the kernel gathers all existing lanes before any insert of the batch, and commits the lanes up to the first
        // valid one; a batch with two lanes on one slot is replayed in order there -- here the trace keeps the gather
        // values of the batch's start and flags the last inserted lane of each slot)
        uint32_t hs[64], olds[64];
        for (int L = 0; L < nexist; ++L) { hs[L] = hash4(ld32(src + p[L])); olds[L] = table[hs[L]]; }
        int lim = nexist - 1;
        for (int L = 0; L < nexist; ++L) {
          const uint32_t cvL = ld32(src + p[L]), hL = hs[L], A1L = W + (uint32_t)p[L] + 1;
          const uint32_t o = table[hL];
          table[hL] = A1L;
          if (o != 0 && A1L - o <= 32768u && ld32(stream + (o - 1)) == cvL) { f = L; cand = o - 1; lim = L; break; }
        }
        for (int L = 0; L < nexist; ++L) {
          const uint32_t A1L = W + (uint32_t)p[L] + 1;
          tw[L] = hs[L] | (1u << 14) | (A1L << 16); te[L] = (uint16_t)olds[L]; ++lookups;
          if (L <= lim) {
            bool later = false;
            for (int M = L + 1; M <= lim; ++M) if (hs[M] == hs[L]) later = true;
            if (!later) { tw[L] |= 1u << 15; ++inserts; }
          }
        }
        emit(tw, te);
        if (f == 64) { if (nexist < 64) { done = true; break; } e_idx += 64; continue; }
        const int pf = p[f]; const int total = extend(pf, cand, 4);
        s = pf + total; sparse = false; if (s >= s_limit) done = true;
      }
    }
    first[si + 1] = (uint32_t)batches;
  }
  FILE *f = fopen(path, "wb");
  if (!f) { perror(path); return 2; }
  const uint32_t hdr[4] = {0x56544142u, (uint32_t)nstreams, (uint32_t)batches, 0};
  fwrite(hdr, 4, 4, f);
  fwrite(first.data(), 4, first.size(), f);
  fwrite(words.data(), 4, words.size(), f);
  fwrite(expect.data(), 2, expect.size(), f);
  fclose(f);
  printf("streams %d: batches/stream %.0f (sparse %.1f) | per batch: lookups %.1f, inserts %.1f -> %s\n", nstreams,
         (double)batches / nstreams, (double)sparse_batches / nstreams, (double)lookups / batches, (double)inserts / batches, path);
  return 0;
}
