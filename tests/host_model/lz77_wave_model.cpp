// lz77_wave_model.cpp -- TEST INFRASTRUCTURE.  Lane-accurate host model of the wave64
// match-finder kernel (moonbit-flate_amd/csrc/lz77_kernels.hip): every per-lane value is
// an array of 64, every ballot a 64-bit mask.  It lets the CPU test-suite fuzz the batch
// algorithm (dense multi-event batches + sparse scan batches + duplicate-slot handling)
// against the oracle without a GPU.  It follows the kernel, not the reference: the
// reference semantics it must reproduce are DeflateFast::encode, deflate-fast.mbt:123-270.
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int kTableSize = 16384;
constexpr int kWin = 65535;
constexpr int kSmallLzMin = 128;
constexpr int kDenseKeep = 61;  // continue in the same dense batch while the next s-1 lane <= this

inline uint32_t ld32(const uint8_t *p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
inline uint32_t hash4(uint32_t u) { return (u * 0x1e35a7bdu) >> 18; }
// the guest blocks' 2-bit slot tag (lz77_kernels.hip, tag_of): the two product bits below the slot index
inline uint32_t tag_of(uint32_t u) { return ((u * 0x1e35a7bdu) >> 16) & 3u; }
inline int ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }
inline uint64_t below(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1); }  // lanes < l
inline uint64_t upto(int l) { return l >= 63 ? ~0ull : ((1ull << (l + 1)) - 1); }  // lanes <= l

int scan_off(int e, int *step, const std::vector<uint32_t> &tab) {
  if (e < 32) { *step = 1; return e; }
  if (e < 48) { *step = 2; return 32 + 2 * (e - 32); }
  if (e < 59) { *step = 3; return 64 + 3 * (e - 48); }
  if (e < 67) { *step = 4; return 97 + 4 * (e - 59); }
  if (e + 1 >= (int)tab.size()) { *step = 1; return 1 << 24; }
  *step = (int)(tab[e + 1] - tab[e]);
  return (int)tab[e];
}

int common_prefix16(const uint8_t *a, const uint8_t *b) {
  int i = 0;
  while (i < 16 && a[i] == b[i]) ++i;
  return i;
}

struct Rec { uint32_t pos, tok; };

struct Stats { uint64_t dense_batches, sparse_batches, events, dup_evals, discarded; };

}  // namespace

// flags: bit 0 = compat_go; bit 1 = model the guest blocks' slot tags (a dense-batch lane whose own
// tag differs from its slot's skips the slot read: its candidate could not pass `cv == cand.val`)
extern "C" int model_lz77(const uint8_t *stream, uint64_t len, int flags, uint32_t *recs_out,
                          uint32_t *chunk_nmatch, uint64_t *stats_out) {
  const int compat_go = flags & 1;
  const bool use_tags = (flags & 2) != 0;
  std::vector<uint32_t> table(kTableSize, 0);
  std::vector<uint8_t> tags(kTableSize, 0);
  std::vector<uint32_t> scantab;
  {
    uint32_t skip = 32, pos = 0;
    while (pos <= 65535) { scantab.push_back(pos); uint32_t st = skip >> 5; pos += st; skip += st; }
    scantab.push_back(1 << 24);
  }
  Stats st = {0, 0, 0, 0, 0};
  const uint64_t full = len / kWin, r = len % kWin;
  const uint32_t nchunks = (uint32_t)(full + (r >= kSmallLzMin ? 1 : 0));
  uint64_t rec_base = 0;
  for (uint32_t c = 0; c < nchunks; ++c) {
    const uint32_t W = c * (uint32_t)kWin;
    const int n = (int)((len - W) < (uint64_t)kWin ? (len - W) : kWin);
    const uint8_t *src = stream + W;
    const int s_limit = n - 15;
    Rec *out = reinterpret_cast<Rec *>(recs_out) + rec_base;
    uint32_t nm = 0;

    int s = -1;          // POST state: re-insert s-1, probe s, scan from s+1 (chunk start: s = -1)
    bool sparse = false; // SPARSE state: continue the scan at (scan_base, e_idx)
    int scan_base = 0, e_idx = 0;
    bool done = false;

    auto extend = [&](int pf, uint32_t cand, int have) -> int {
      // total match length given `have` (>= 4) already verified bytes
      int limit = n - pf;
      if (limit > 258) limit = 258;
      if (!compat_go && cand + 4 < W) return 4;  // MoonBit: prev window is empty
      int l = have;
      const uint8_t *a = src + pf, *b = stream + cand;
      while (l < limit && a[l] == b[l]) ++l;
      return l;
    };

    while (!done) {
      if (!sparse) {
        // ----------------------------- dense batch --------------------------------
        st.dense_batches++;
        const int B = s - 1;
        int q[64];
        uint32_t cv[64], h[64], old[64], A1[64];
        uint8_t own[64][16];
        int mlen[64];
        uint64_t LD = 0, E1 = 0, E2 = 0, OK = 0, DUP = 0;
        for (int L = 0; L < 64; ++L) {
          q[L] = B + L;
          cv[L] = h[L] = old[L] = 0;
          mlen[L] = 0;
          A1[L] = W + (uint32_t)q[L] + 1;
          if (q[L] >= 0 && q[L] + 1 <= s_limit) E1 |= 1ull << L;
          if (q[L] >= 0 && q[L] + 2 <= s_limit) E2 |= 1ull << L;
        }
        LD = E1;
        for (int L = 0; L < 64; ++L) {
          if (!((LD >> L) & 1)) continue;
          memcpy(own[L], src + q[L], 16);
          cv[L] = ld32(src + q[L]);
          h[L] = hash4(cv[L]);
          old[L] = (use_tags && tags[h[L]] != tag_of(cv[L])) ? 0u : table[h[L]];
          if (old[L] != 0 && A1[L] - old[L] <= 32768u) {
            mlen[L] = common_prefix16(own[L], stream + (old[L] - 1));
            if (mlen[L] >= 4) OK |= 1ull << L;
          }
        }
        for (int L = 0; L < 64; ++L)
          for (int M = 0; M < 64; ++M)
            if (L != M && ((LD >> L) & 1) && ((LD >> M) & 1) && h[L] == h[M]) DUP |= 1ull << L;

        uint64_t INS = 0;
        int a = 0;
        bool batch_over = false;
        while (!batch_over) {
          // probe lanes of this event
          const uint64_t a_ins = (LD >> a) & 1 ? (1ull << a) : 0;
          uint64_t R = 0;
          if (a + 1 <= 63 && ((LD >> (a + 1)) & 1) && q[a + 1] >= 0) R |= 1ull << (a + 1);
          bool scan_ended = false, truncated = false;
          int consumed = 0;
          {
            const int b = a + 2;
            for (int e = 0;; ++e) {
              int step;
              const int L = b + scan_off(e, &step, scantab);
              if (L > 63) { truncated = true; break; }
              const bool ex = step == 1 ? ((E1 >> L) & 1) : ((E2 >> L) & 1);
              if (!ex) { scan_ended = true; break; }
              R |= 1ull << L;
              consumed = e + 1;
            }
          }
          uint64_t T = 0, rem = R;
          int f = 64;
          uint32_t cand = 0;
          int have = 0;
          for (;;) {
            const int fv = ctz64(OK & rem & ~DUP), fd = ctz64(DUP & rem);
            if (fv < fd) {
              f = fv; cand = old[fv] - 1; have = mlen[fv];
              T |= rem & upto(fv);
              break;
            }
            if (fd == 64) { T |= rem; break; }
            st.dup_evals++;
            T |= rem & below(fd);
            uint64_t G = 0;
            for (int L = 0; L < fd; ++L)
              if (((INS | T | a_ins) >> L) & 1 && h[L] == h[fd]) G |= 1ull << L;
            bool v;
            uint32_t cnd;
            int ml;
            if (G) {
              const int i = 63 - __builtin_clzll(G);
              v = cv[i] == cv[fd];
              cnd = W + (uint32_t)q[i];
              ml = common_prefix16(own[fd], own[i]);
            } else {
              v = (OK >> fd) & 1;
              cnd = old[fd] - 1;
              ml = mlen[fd];
            }
            T |= 1ull << fd;
            if (v) { f = fd; cand = cnd; have = ml; break; }
            rem &= ~upto(fd);
          }
          if (f == 64) {
            if (scan_ended) {
              INS |= T | a_ins;
              done = true;
            } else if (a == 0) {  // nothing in this whole batch: continue as a sparse scan
              INS |= T | a_ins;
              sparse = true;
              scan_base = s + 1;
              e_idx = consumed;
            } else {
              st.discarded++;  // partial event at the end of the batch: redo it in a new batch
            }
            batch_over = true;
            (void)truncated;
          } else {
            INS |= T | a_ins;
            st.events++;
            const int pf = q[f];
            const int total = have < 16 ? ((!compat_go && cand + 4 < W) ? 4 : have) : extend(pf, cand, 16);
            out[nm].pos = (uint32_t)pf;
            out[nm].tok = (1u << 30) | ((uint32_t)(total - 3) << 22) | ((W + (uint32_t)pf) - cand - 1);
            ++nm;
            s = pf + total;
            if (s >= s_limit) { done = true; batch_over = true; }
            else {
              a = s - 1 - B;
              if (a > kDenseKeep) batch_over = true;  // start a fresh dense batch at s
            }
          }
        }
        // commit the inserts in position order (later positions overwrite earlier ones)
        for (int L = 0; L < 64; ++L)
          if ((INS >> L) & 1) {
            table[h[L]] = A1[L];
            tags[h[L]] = (uint8_t)tag_of(cv[L]);
          }
      } else {
        // ----------------------------- sparse batch -------------------------------
        st.sparse_batches++;
        int p[64], step[64];
        uint64_t EX = 0;
        for (int L = 0; L < 64; ++L) {
          p[L] = scan_base + scan_off(e_idx + L, &step[L], scantab);
          if (p[L] + step[L] <= s_limit) EX |= 1ull << L;
        }
        const int nexist = __builtin_popcountll(EX);
        if (nexist == 0) { done = true; break; }
        int f = 64;
        uint32_t cand = 0;
        for (int L = 0; L < nexist; ++L) {  // in-order replay (the kernel does this in parallel)
          const uint32_t cvL = ld32(src + p[L]), hL = hash4(cvL), A1L = W + (uint32_t)p[L] + 1;
          const uint32_t o = table[hL];
          table[hL] = A1L;
          tags[hL] = (uint8_t)tag_of(cvL);
          if (o != 0 && A1L - o <= 32768u && ld32(stream + (o - 1)) == cvL) { f = L; cand = o - 1; break; }
        }
        if (f == 64) {
          if (nexist < 64) { done = true; break; }
          e_idx += 64;
          continue;
        }
        st.events++;
        const int pf = p[f];
        const int total = extend(pf, cand, 4);
        out[nm].pos = (uint32_t)pf;
        out[nm].tok = (1u << 30) | ((uint32_t)(total - 3) << 22) | ((W + (uint32_t)pf) - cand - 1);
        ++nm;
        s = pf + total;
        sparse = false;
        if (s >= s_limit) done = true;
      }
    }
    chunk_nmatch[c] = nm;
    rec_base += 16384;
  }
  if (stats_out) {
    stats_out[0] = st.dense_batches; stats_out[1] = st.sparse_batches; stats_out[2] = st.events;
    stats_out[3] = st.dup_evals; stats_out[4] = st.discarded;
  }
  return (int)nchunks;
}
