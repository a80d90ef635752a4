// lz77_pipe_kernels.hip -- deflate-fast match finder, one wavefront per stream, with the front end
// of the NEXT dense batch issued before the events of the current one ("pipelined").
//
// Same results as lz77_wave_kernel (lz77_kernels.hip), i.e. DeflateFast::encode (reference
// deflate-fast.mbt:123-270), same batch algorithm.  What changes: in the one-wave kernel every
// batch starts with a chain of three dependent memory round trips -- input load -> hash -> table
// gather -> candidate gather -- that nothing overlaps (profiles/r02: 2200 of the 4600 cycles of a
// batch), because the batch starts where the previous one ended.  Here a batch is position-based:
// lane L owns chunk position G + L whatever the parse does, and the parse enters it at lane
// a0 = (s - 1) - G (lanes below a0 are never visited).  A dense batch stops when the next event
// starts beyond lane 58, so the next batch is prepared at the guessed base G + 59 while the current
// one is still being parsed:
//
//   batch k:   issue  own' = input[G+59 ..]                         (round trip 1 of batch k+1)
//              refresh batch k's slots, evaluate its lanes
//              issue  old' = table[hash(own')], cb' = input[cand(old')]   (round trips 2, 3)
//              events (scalar chase, general path), match records, table commit
//
// What the guess cannot know is the table as batch k leaves it: batch k+1 therefore re-reads its
// slots first and, where one changed, takes the candidate bytes from batch k's registers (a changed
// slot holds a position batch k inserted).  A guess that misses (long match, sparse scan, chunk
// start: ~14 % of the batches on text) costs one unhidden front end, as before.  Lanes with equal
// hashes are found by ballots, so the table is written only by the committed inserts.
#include "lz77_device.h"

namespace flate {

#ifdef FLATE_LZ_STAMPS
#define PSTAMP(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define PADD(acc, expr) acc += (expr)
#else
#define PSTAMP(var)
#define PADD(acc, expr)
#endif

constexpr int kPipeStride = kDenseKeep + 1;  // guessed base of the next batch
constexpr int kPipeA0Max = 24;               // largest entry lane for which the guess is used

struct PipeFront {  // one prepared batch: 64 consecutive chunk positions G .. G+63, one per lane
  int G;
  bool e1, e2;      // position may be inserted / probed with step 1 resp. probed with step 2
  uint4 own;        // 16 input bytes at the position
  uint4 cb;         // 16 bytes at the candidate predicted from `old`
  uint32_t h, old;  // slot and its value when the batch was prepared
  uint64_t eq;      // lanes of the batch whose slot is mine (including me)
};

template <bool MULTI, bool GUEST>
FLATE_D void lz77_pipe_stream(const LzParams &P, const uint32_t sid, uint16_t *table, const int lane) {
  using E = uint16_t;
  constexpr uint32_t kEMask = 0xffffu;
  {
    uint4 *t4 = reinterpret_cast<uint4 *>(table);
    const uint32_t fill = MULTI ? (((0u - kMarkerBack + 1u) & 0xffffu) * 0x10001u) : 0u;
    const uint4 z = make_uint4(fill, fill, fill, fill);
    for (int i = lane; i < (int)(kTableSize * sizeof(E) / 16); i += 64) t4[i] = z;
  }
  __syncthreads();
  volatile E *vtable = table;
  uint32_t next_sweep = kSweepEvery;  // MULTI: absolute position at which the next sweep is due
  auto sweep = [&](uint32_t R) {
    uint32_t *t32 = reinterpret_cast<uint32_t *>(table);
    const uint32_t marker = (R - kMarkerBack + 1u) & 0xffffu;
    for (int i = lane; i < kTableSize / 2; i += 64) {
      const uint32_t v = t32[i];
      const uint32_t d0 = (R + 1u - v) & 0xffffu, d1 = (R + 1u - (v >> 16)) & 0xffffu;
      const uint32_t lo = (d0 == 0 || d0 > 32768u) ? marker : (v & 0xffffu);
      const uint32_t hi = (d1 == 0 || d1 > 32768u) ? marker : (v >> 16);
      t32[i] = lo | (hi << 16);
    }
    __syncthreads();
    next_sweep = R + kSweepEvery;
  };

  const ChunkGeom g = stream_geom(P, sid);
  const uint16_t *scan_tab = P.scan_off;
  uint32_t pf_val = 0, pf_sink = 0;

  for (uint32_t c = 0; c < g.nchunks; ++c) {
    const uint32_t W = c * (uint32_t)kMaxStoreBlockSize;
    const uint64_t rem_len = g.len - W;
    const int n = rem_len < (uint64_t)kMaxStoreBlockSize ? (int)rem_len : kMaxStoreBlockSize;
    const uint8_t *src = g.stream + W;
    uint2 *mout = P.matches + g.mbase + (uint64_t)c * kMatchCapPerChunk;
    uint32_t nm = 0;
    uint32_t acc_len = 0;  // per-lane partial sums of match lengths
    const int s_limit = n - kInputMargin;
#ifdef FLATE_LZ_STAMPS
    uint64_t st_front = 0, st_eval = 0, st_next = 0, st_events = 0, st_commit = 0, st_nb = 0, st_fresh = 0,
             st_changed = 0;
#endif

    auto clamp_pos = [&](int q) -> int {
      int qa = q < 0 ? 0 : q;
      return qa > n - 16 ? n - 16 : qa;  // (a chunk has at least 128 bytes)
    };
    // lanes with equal hashes, one ballot per hash bit (the table is not touched)
    auto equal_hash_lanes = [&](uint32_t h, bool e1) -> uint64_t {
      uint64_t eq = __ballot(e1);
#pragma unroll
      for (int k = 0; k < kTableBits; ++k) {
        const bool bit = (h >> k) & 1u;
        const uint64_t m = __ballot(bit);
        eq &= bit ? m : ~m;
      }
      return e1 ? eq : 0ull;
    };
    // candidate address predicted from a slot value (or the lane's own bytes when out of range)
    auto cand_of = [&](uint32_t old, int q, bool e1, bool *inr, uint32_t *dist) -> uint32_t {
      const uint32_t A1 = W + (uint32_t)q + 1u;
      const uint32_t d = MULTI ? ((A1 - old) & 0xffffu) : (A1 - old);
      *dist = d;
      *inr = e1 && (MULTI ? d != 0 : old != 0) && d <= (uint32_t)kMaxMatchOffset;
      return A1 - 1u - d;
    };
    // a whole front end with nothing to hide it behind (chunk start, missed guess, after a sparse scan)
    auto fresh_front = [&](const int G) -> PipeFront {
      PipeFront f;
      f.G = G;
      const int q = G + lane;
      f.e1 = q >= 0 && q + 1 <= s_limit;
      f.e2 = q >= 0 && q + 2 <= s_limit;
      f.own = ld128(src + clamp_pos(q));
      f.h = f.e1 ? hash4(f.own.x) : 0u;
      f.old = (uint32_t)vtable[f.h];
      bool inr;
      uint32_t dist;
      const uint32_t ca = cand_of(f.old, q, f.e1, &inr, &dist);
      f.cb = ld128(g.stream + (inr ? ca : W + (uint32_t)clamp_pos(q)));
      f.eq = equal_hash_lanes(f.h, f.e1);
      return f;
    };

    int s = -1;
    bool sparse = false;
    int scan_base = 0, e_idx = 0;
    bool done = false;
    PipeFront F;
    F.G = 0;
    F.e1 = F.e2 = false;
    F.own = F.cb = make_uint4(0, 0, 0, 0);
    F.h = F.old = 0;
    F.eq = 0;
    bool f_valid = false;
    uint4 pown = make_uint4(0, 0, 0, 0);  // input bytes of the previous dense batch (its base: pG)
    int pG = 0;
    bool p_valid = false;

    while (!done) {
      if (!sparse) {
        // =============================== dense batch ===============================
        PSTAMP(t0);
        const int a0g = s - 1 - F.G;
        const bool usable = f_valid && a0g >= 0 && a0g <= kPipeA0Max;
        if (MULTI) {
          const uint32_t first = W + (uint32_t)(s - 1 < 0 ? 0 : s - 1);
          if (first >= next_sweep) sweep(first);
        }
        if (!usable) {
          F = fresh_front(s - 1);
          p_valid = false;  // the table was read after every commit: nothing can be stale
          PADD(st_fresh, 1);
        }
        const int B = F.G;
        const int a0 = s - 1 - B;
        const int q = B + lane;
        const bool e1 = F.e1, e2 = F.e2;
        const uint4 own = F.own;
        const uint32_t h = F.h;
        const uint32_t A1 = W + (uint32_t)q + 1u;
        // ---- round trip 1 of the next batch: its input bytes
        const int Gn = B + kPipeStride;
        const int qn = Gn + lane;
        PipeFront N;
        N.G = Gn;
        N.e1 = qn >= 0 && qn + 1 <= s_limit;
        N.e2 = qn >= 0 && qn + 2 <= s_limit;
        pf_sink ^= pf_val;  // retire the previous look-ahead load
        N.own = ld128(src + clamp_pos(qn));
        {  // look-ahead: pull the next lines of this stream towards L2
          int pq = B + 768 + 4 * lane;
          if (pq > n - 4) pq = n - 4;
          pf_val = ld32(src + pq);
        }
        // ---- refresh: this batch's slots as the previous batch left them
        uint32_t old = F.old;
        uint4 cb = F.cb;
        if (usable && p_valid) {
          const uint32_t cur = e1 ? (uint32_t)vtable[h] : old;
          const bool chg = cur != old;
          if (__ballot(chg)) {
            PADD(st_changed, 1);
            old = cur;
            bool in2;
            uint32_t d2;
            const uint32_t ca2 = cand_of(old, q, e1, &in2, &d2);
            // a changed slot holds a position the previous batch inserted: its bytes are in pown
            const int li = (int)(ca2 - (W + (uint32_t)pG));
            const bool from_prev = chg && in2 && li >= 0 && li < 64;
            const uint4 pb = make_uint4(__shfl(pown.x, li & 63), __shfl(pown.y, li & 63),
                                        __shfl(pown.z, li & 63), __shfl(pown.w, li & 63));
            if (from_prev) cb = pb;
            if (__ballot(chg && in2 && !from_prev)) {  // (inserted by a sparse scan in between: rare)
              const uint4 gb = ld128(g.stream + (in2 ? ca2 : W + (uint32_t)clamp_pos(q)));
              if (chg && !from_prev) cb = gb;
            }
          }
        }
        bool inr;
        uint32_t dist;
        const uint32_t cand_abs = cand_of(old, q, e1, &inr, &dist);
        const uint64_t E1 = __ballot(e1), E2 = __ballot(e2);
        // lanes below the entry lane belong to the previous batch: they take no part
        const uint64_t eq = (lane >= a0) ? (F.eq & ~lanes_below(a0)) : 0ull;
        const uint64_t DUP = __ballot((eq & lanes_below(lane)) != 0);

        // probe lanes of the event that would start with s-1 == my lane
        const int bsh = lane + 2;  // (lanes 62, 63 have no scan lanes left: shift counts stay < 64)
        const uint64_t specR = (E1 & (2ull << lane)) |
                               (bsh < 64 ? (((0x00000000ffffffffull << bsh) & E1) |
                                            ((0x5555555500000000ull << bsh) & E2))
                                         : 0ull);
        const uint64_t dupR = DUP & specR;
        const int fd0 = dupR ? __builtin_ctzll(dupR) : 64;
        const int mlen = inr ? prefix16(own, cb) : 0;
        const uint64_t OK = __ballot(mlen >= 4);
        const bool cross = MULTI && !P.compat_go && cand_abs + 4 < W;  // MoonBit: prev window is empty
        const int tot_self = cross ? 4 : mlen;                         // 16 => needs extension (slow)
        uint32_t rec_tok = kMatchType | ((uint32_t)(tot_self - 3) << kLengthShift) | (dist - 1u);
        // ev: [6:0] match lane fv, [15:8] total length, bit16 general path needed, bit17 the match
        // ends the chunk, bit18 stop chasing, [31:24] lane of the next event start
        uint32_t ev;
        {
          const uint64_t okR = OK & specR & ~DUP;
          const int fv = okR ? __builtin_ctzll(okR) : 64;
          const int tf = __shfl(tot_self, fv & 63);
          const bool slow = !(fv < fd0) || tf >= 16;
          const bool ends = B + fv + tf >= s_limit;
          const int nxt = fv + tf - 1;
          ev = (uint32_t)(fv & 127) | ((uint32_t)tf << 8) | (slow ? 1u << 16 : 0u) |
               (ends ? 1u << 17 : 0u) | ((ends || nxt > kDenseKeep) ? 1u << 18 : 0u) |
               ((uint32_t)(nxt & 255) << 24);
        }
        PSTAMP(t1);
        // ---- round trips 2 and 3 of the next batch: its slots and the candidates they predict
        N.h = N.e1 ? hash4(N.own.x) : 0u;
        N.old = (uint32_t)vtable[N.h];
        N.eq = equal_hash_lanes(N.h, N.e1);
        {
          bool inn;
          uint32_t dn;
          const uint32_t can = cand_of(N.old, qn, N.e1, &inn, &dn);
          N.cb = ld128(g.stream + (inn ? can : W + (uint32_t)clamp_pos(qn)));
        }
        PSTAMP(t2);

        // Lanes inserted by the fast events whose start lanes are in `vis` (see lz77_kernels.hip)
        auto fast_inserts = [&](uint64_t vis) -> uint64_t {
          const uint64_t below_me = vis & lanes_upto(lane);
          const int j = below_me ? 63 - __builtin_clzll(below_me) : 0;
          const int fvj = (int)(__shfl(ev, j) & 127u);
          const int d = lane - j, o = d - 2;
          const bool sched = (d <= 1 || o < 32) ? e1 : (((o & 1) == 0) ? e2 : false);
          return __ballot(below_me != 0 && sched && lane <= fvj);
        };
        uint64_t INS = 0, M = 0, MF = 0;
        uint64_t VISall = 0;
        int a = a0;
        for (;;) {  // events inside this batch
          uint32_t x, tmp;
          uint64_t VIS = 0, MFl = 0;
          int a_s = __builtin_amdgcn_readfirstlane(a);
          asm volatile(
              "1:\n\t"
              "s_nop 1\n\t"
              "v_readlane_b32 %[x], %[ev], %[a]\n\t"
              "s_nop 3\n\t"
              "s_bitcmp1_b32 %[x], 16\n\t"
              "s_cbranch_scc1 2f\n\t"
              "s_bitset1_b64 %[vis], %[a]\n\t"
              "s_and_b32 %[t], %[x], 0x7f\n\t"
              "s_bitset1_b64 %[mf], %[t]\n\t"
              "s_lshr_b32 %[a], %[x], 24\n\t"
              "s_bitcmp1_b32 %[x], 18\n\t"
              "s_cbranch_scc0 1b\n\t"
              "2:\n\t"
              : [x] "=&s"(x), [a] "+s"(a_s), [vis] "+s"(VIS), [mf] "+s"(MFl), [t] "=&s"(tmp)
              : [ev] "v"(ev)
              : "scc");
          a = a_s;
          MF |= MFl;
          VISall |= VIS;
          s = B + a + 1;
          if (!(x & (1u << 16))) {  // stopped after a fast event
            if (x & (1u << 17)) done = true;
            break;
          }
          // ---- general event (shared slots, long matches, end of scan) ----
          const uint64_t FINS = fast_inserts(VISall);
          const uint64_t a_ins = E1 & (1ull << a);
          const int b = a + 2;
          const uint64_t full = 0x55555555ffffffffull << b;
          const uint64_t scanR = ((0x00000000ffffffffull << b) & E1) | ((0x5555555500000000ull << b) & E2);
          const uint64_t R = (E1 & (2ull << a)) | scanR;
          const bool scan_ended = scanR != full;

          uint64_t T = 0, rem = R;
          int f = 64, have = 0;
          uint32_t cand = 0;
          for (;;) {
            const int fv = ffs64(OK & rem & ~DUP), fd = ffs64(DUP & rem);
            if (fv < fd) {
              f = fv;
              cand = rdlane(cand_abs, fv);
              have = (int)rdlane((uint32_t)mlen, fv);
              T |= rem & lanes_upto(fv);
              break;
            }
            if (fd == 64) {
              T |= rem;
              break;
            }
            // lane fd shares its slot with other lanes of this batch: judge it against the latest
            // position this batch has already inserted into that slot
            T |= rem & lanes_below(fd);
            const uint32_t hfd = rdlane(h, fd);
            const uint64_t Gm = __ballot(e1 && h == hfd) & (INS | FINS | T | a_ins) & lanes_below(fd);
            bool v;
            uint32_t cnd;
            int ml;
            if (Gm) {
              const int i = 63 - __builtin_clzll(Gm);
              v = rdlane(own.x, i) == rdlane(own.x, fd);
              cnd = W + (uint32_t)(B + i);
              const uint4 oi = make_uint4(rdlane(own.x, i), rdlane(own.y, i), rdlane(own.z, i), rdlane(own.w, i));
              const uint4 of = make_uint4(rdlane(own.x, fd), rdlane(own.y, fd), rdlane(own.z, fd), rdlane(own.w, fd));
              ml = prefix16(of, oi);
            } else {
              v = (OK >> fd) & 1;
              cnd = rdlane(cand_abs, fd);
              ml = (int)rdlane((uint32_t)mlen, fd);
            }
            T |= 1ull << fd;
            if (v) {
              f = fd;
              cand = cnd;
              have = ml;
              break;
            }
            rem &= ~lanes_upto(fd);
          }

          if (f == 64) {
            if (scan_ended) {  // the scan ran into s_limit: emit_remainder (:152-159)
              INS |= T | a_ins;
              done = true;
            } else if (a == 0) {  // 47 probes without a candidate: continue as a sparse scan
              INS |= T | a_ins;
              sparse = true;
              scan_base = s + 1;
              e_idx = 47;
            }  // else: partial event at the end of the batch; redo it in a fresh batch
            break;
          }
          INS |= T | a_ins;
          const int pf = B + f;
          int total;
          if (have < 16)
            total = (!P.compat_go && cand + 4 < W) ? 4 : have;
          else
            total = extend_match(src, g.stream, W, n, pf, cand, 16, P.compat_go, lane);
          M |= 1ull << f;
          if (lane == f) {
            rec_tok = kMatchType | ((uint32_t)(total - 3) << kLengthShift) | ((W + (uint32_t)pf) - cand - 1);
            acc_len += (uint32_t)total;
          }
          s = pf + total;
          if (s >= s_limit) {
            done = true;
            break;
          }
          a = s - 1 - B;
          if (a > kDenseKeep) break;
        }
        M |= MF;
        INS |= fast_inserts(VISall);
        PSTAMP(t3);
        if ((MF >> lane) & 1) acc_len += (uint32_t)tot_self;
        // match records of this batch, in position order, one coalesced streaming store
        if ((M >> lane) & 1) {
          const unsigned long long rec = (unsigned long long)(uint32_t)q | ((unsigned long long)rec_tok << 32);
          __builtin_nontemporal_store(
              rec, reinterpret_cast<unsigned long long *>(mout + nm + (uint32_t)__popcll(M & lanes_below(lane))));
        }
        nm += (uint32_t)__popcll(M);
        // commit: the inserted lanes write, and where several share a slot only the last of them
        // (position order: the latest insert wins)
        asm volatile("" ::: "memory");
        if (e1 && ((INS >> lane) & 1) && (eq & INS & ~lanes_upto(lane)) == 0) vtable[h] = (E)A1;
        asm volatile("" ::: "memory");
        // hand the prepared batch over
        pown = own;
        pG = B;
        p_valid = true;
        F = N;
        f_valid = !sparse;
        PSTAMP(t4);
        PADD(st_front, t1 - t0);
        PADD(st_next, t2 - t1);
        PADD(st_events, t3 - t2);
        PADD(st_commit, t4 - t3);
        PADD(st_nb, 1);
      } else {
        // =============================== sparse batch ==============================
        f_valid = false;  // a sparse scan inserts far ahead: the prepared batch is dropped
        const int e = e_idx + lane;
        int p, step;
        if (e < kScanClosedForm) {
          p = scan_base + scan_off_small(e, &step);
        } else {
          const int ec = e < P.scan_len - 1 ? e : P.scan_len - 2;
          const int o0 = scan_tab[ec], o1 = scan_tab[ec + 1];
          p = scan_base + o0 + (e - ec) * 65536;  // beyond the table => never exists
          step = o1 - o0;
        }
        const bool exists_all = p + step <= s_limit;  // the `next_s > s_limit` test of :188
        const int p0 = (int)rdlane((uint32_t)p, 0);
        const bool exists = exists_all && (!MULTI || (uint32_t)(p - p0) < kSpanMax);
        const int nall = __popcll(__ballot(exists_all));
        const int nexist = __popcll(__ballot(exists));  // events are a prefix of the lanes
        if (nall == 0) break;                           // emit_remainder (:152-159)
        if (MULTI && W + (uint32_t)p0 >= next_sweep) sweep(W + (uint32_t)p0);

        uint32_t cv = 0, hs = 0, olds = 0;
        if (exists) {
          cv = ld32(src + p);
          hs = hash4(cv);
          olds = (uint32_t)vtable[hs];
        }
        const uint32_t As = W + (uint32_t)p + 1;
        const uint32_t dists = MULTI ? ((As - olds) & 0xffffu) : (As - olds);
        const uint32_t cands = As - 1u - dists;
        bool ok = false;
        if (exists && (MULTI ? dists != 0 : olds != 0) && dists <= (uint32_t)kMaxMatchOffset)
          ok = ld32(g.stream + cands) == cv;
        const uint64_t V = __ballot(ok);
        const int f0 = ffs64(V);

        // commit the inserts of lanes <= first valid lane; detect same-slot collisions
        const int lim = f0 < nexist - 1 ? f0 : nexist - 1;
        const bool ins = lane <= lim;
        uint64_t C;
        if (GUEST) {  // same-slot lanes among the inserted ones by ballots
          uint64_t eqs = __ballot(ins);
#pragma unroll
          for (int k = 0; k < kTableBits; ++k) {
            const bool bit = (hs >> k) & 1u;
            const uint64_t m = __ballot(bit);
            eqs &= bit ? m : ~m;
          }
          C = __ballot(ins && (eqs & ~(1ull << lane)) != 0);
          if (C == 0 && ins) vtable[hs] = (E)As;
        } else {  // LDS table: insert and read back
          if (ins) vtable[hs] = (E)As;
          const uint32_t rb = ins ? (uint32_t)vtable[hs] : (As & kEMask);
          C = __ballot(rb != (As & kEMask));
        }

        int f = f0;
        uint32_t cand = 0;  // absolute candidate position
        if (C == 0) {
          if (f0 < 64) cand = rdlane(cands, f0);
        } else {
          // two lanes of this batch share a slot: replay the batch in order
          if (!GUEST && ins) vtable[hs] = (E)olds;
          f = 64;
          for (int e2i = 0; e2i < nexist; ++e2i) {
            const uint32_t he = rdlane(hs, e2i);
            const uint32_t pe1 = rdlane(As, e2i);
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vtable[he]);
            vtable[he] = (E)pe1;
            bool v;
            uint32_t cnd;
            if (cur == rdlane(olds, e2i)) {
              v = (V >> e2i) & 1;
              cnd = rdlane(cands, e2i);
            } else {  // candidate was inserted by an earlier lane of this batch
              const uint64_t m = __ballot(exists && (As & kEMask) == cur);
              v = false;
              cnd = 0;
              if (m) {
                const int i = __builtin_ctzll(m);
                v = rdlane(cv, i) == rdlane(cv, e2i);
                cnd = rdlane(As, i) - 1u;
              }
            }
            if (v) {
              f = e2i;
              cand = cnd;
              break;
            }
          }
        }
        if (f == 64) {
          if (nexist == nall && nall < 64) break;  // the scan ran into s_limit
          e_idx += nexist;
          continue;
        }
        const int pf = (int)rdlane((uint32_t)p, f);
        const int total = extend_match(src, g.stream, W, n, pf, cand, 4, P.compat_go, lane);
        if (lane == 0)
          mout[nm] = make_uint2((uint32_t)pf, kMatchType | ((uint32_t)(total - 3) << kLengthShift) |
                                                  ((W + (uint32_t)pf) - cand - 1));
        ++nm;
        if (lane == 0) acc_len += (uint32_t)total;
        s = pf + total;
        sparse = false;
        if (s >= s_limit) done = true;
      }
    }
    uint32_t sumlen = acc_len;
    for (int d = 32; d >= 1; d >>= 1) sumlen += __shfl_xor(sumlen, d);
    if (pf_sink == 0x9e3779b9u && P.debug) P.debug[0] = pf_sink;  // keeps the look-ahead loads alive
    if (lane == 0) {
      P.chunk_nmatch[g.chunk0 + c] = nm;
      P.chunk_ntok[g.chunk0 + c] = (uint32_t)n - sumlen + nm;
#ifdef FLATE_LZ_STAMPS
      if (P.debug) {
        uint64_t *d = P.debug + (uint64_t)(g.chunk0 + c) * 8;
        d[0] = st_front; d[1] = st_next; d[2] = st_events; d[3] = st_commit; d[4] = st_nb; d[5] = nm;
        d[6] = st_fresh; d[7] = st_changed;
      }
#endif
    }
  }
}

// Resident kernel: table in LDS (32 KiB per stream).
template <bool MULTI>
__global__ __launch_bounds__(64) void lz77_pipe_kernel(LzParams P) {
  __shared__ uint16_t table[kTableSize];
  const int lane = threadIdx.x;
  for (bool first = true;; first = false) {
    uint32_t q;
    if (P.queue) {
      q = 0;
      if (lane == 0) q = atomicAdd(P.queue, 1u);
      q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
      if (q >= P.queue_end) break;
    } else {
      if (!first) break;
      q = blockIdx.x;
    }
    __syncthreads();
    lz77_pipe_stream<MULTI, false>(P, P.stream_ids ? P.stream_ids[q] : q, table, lane);
    __syncthreads();
  }
}

// Guest kernel: the table is a 32 KiB slice of HBM scratch that stays in the XCD's L2.
template <bool MULTI>
__global__ __launch_bounds__(64) void lz77_pipe_guest_kernel(LzParams P) {
  uint16_t *table = reinterpret_cast<uint16_t *>(P.gtables) + (size_t)blockIdx.x * kTableSize;
  const int lane = threadIdx.x;
  for (;;) {
    uint32_t q = 0;
    if (lane == 0) q = atomicAdd(P.queue, 1u);
    q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (q >= P.queue_end) break;
    __syncthreads();
    lz77_pipe_stream<MULTI, true>(P, P.stream_ids[q], table, lane);
    __syncthreads();
  }
}

template __global__ void lz77_pipe_kernel<false>(LzParams);
template __global__ void lz77_pipe_kernel<true>(LzParams);
template __global__ void lz77_pipe_guest_kernel<false>(LzParams);
template __global__ void lz77_pipe_guest_kernel<true>(LzParams);

}  // namespace flate
