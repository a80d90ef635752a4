// lz77_team_kernels.hip -- deflate-fast match finder, TWO wavefronts per stream ("team").
//
// Same results as lz77_wave_kernel (lz77_kernels.hip), i.e. DeflateFast::encode
// (reference deflate-fast.mbt:123-270); same batch algorithm (dense batches with per-lane
// speculative events and a scalar chase, sparse scan batches, exact handling of lanes that share
// a table slot).  What changes is the schedule.  A single wavefront is bound by its own
// instruction stream (profiles/r02: ~420 instructions and ~4600 cycles per 58-position batch,
// 45 % of them waiting for the input load -> table gather -> candidate gather chain), and the
// number of streams in flight is bound by table storage, not by SIMD time.  So a stream gets two
// wavefronts that alternate:
//
//   round r:   wave (r & 1)      parses batch r: refresh, events, commit, hand over `s`
//              wave (r & 1) ^ 1  meanwhile prepares batch r+1 SPECULATIVELY at the guessed base
//                                G + 59 (a dense batch stops when the next event starts beyond
//                                lane 58): input load, hash, table gather, candidate gather,
//                                equal-hash lanes
//
// One s_barrier per round.  The prepared batch is position-based, not start-based: the parse
// enters it at lane a0 = (s - 1) - base, lanes below a0 are simply never visited.  What the
// speculation cannot know is the table as the previous batch leaves it; the parser therefore
// re-reads its slots after the hand-over (LDS, or an L2-served load for the guest form) and
// re-gathers the candidates only where a slot changed.  The table itself is never written
// speculatively: lanes with equal hashes are found by ballots, and the only writes are the
// committed inserts (the latest position of a slot group wins).  A guess that misses (long
// match, sparse scan, chunk start) costs one unhidden front end, as in the one-wave kernel.
#include "lz77_device.h"

namespace flate {

// Diagnostic build only (-DFLATE_LZ_STAMPS): per-chunk s_memtime sums of the team's phases.
#ifdef FLATE_LZ_STAMPS
#define TSTAMP(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define TADD(acc, expr) acc += (expr)
#else
#define TSTAMP(var)
#define TADD(acc, expr)
#endif

constexpr int kTeamStride = kDenseKeep + 1;  // guessed base of the next batch
constexpr int kTeamA0Max = 24;               // largest entry lane for which the guess is used

struct TeamToken {   // hand-over between the two waves of a team (LDS), double-buffered by round
  int s;             // POST(s): the next event starts at chunk position s (its batch base is s-1)
  uint32_t nm;       // match records of this chunk so far
  uint32_t flags;    // bit 0: chunk finished
  uint32_t next_sweep;
};
struct TeamShared {
  TeamToken t[2];
  uint32_t sum[2];   // per-wave sums of match lengths (chunk epilogue)
  uint32_t q;        // stream index fetched from the queue
  uint32_t pad;
};

// Barrier between the two waves of a team.  Only what the partner reads must have landed: LDS
// writes (token, LDS table) -- and for the guest form the table stores to global memory.
template <bool GUEST>
FLATE_D void team_barrier() {
  if (GUEST)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A table slot as the partner wave left it: a read the compiler may not merge with the speculative
// one of the front end.
template <bool GUEST>
FLATE_D uint32_t slot_now(const uint16_t *p) {
  // (both waves of a team run on one CU and share its vector L1, which is coherent for that CU's
  // own stores once they are acknowledged -- team_barrier<true> waits for vmcnt(0) -- so a plain
  // load is enough in the guest form too: AMDGPU memory model, workgroup scope, non-tgsplit mode)
  return (uint32_t)*reinterpret_cast<const volatile uint16_t *>(p);
}

struct TeamFront {  // one prepared batch: 64 consecutive chunk positions G .. G+63, one per lane
  int G;
  bool e1, e2;      // position may be inserted / probed with step 1 resp. probed with step 2
  uint4 own;        // 16 input bytes at the position
  uint4 cb;         // 16 bytes at the candidate predicted from `old`
  uint32_t h, old;  // slot and its value when the batch was prepared
  uint64_t eq;      // lanes of the batch whose slot is mine (including me)
};

template <bool MULTI, bool GUEST>
FLATE_D void lz77_team_stream(const LzParams &P, const uint32_t sid, uint16_t *table, TeamShared *sh,
                              const int lane, const int wave) {
  using E = uint16_t;
  constexpr uint32_t kEMask = 0xffffu;
  const int tid = wave * 64 + lane;
  {
    uint4 *t4 = reinterpret_cast<uint4 *>(table);
    const uint32_t fill = MULTI ? (((0u - kMarkerBack + 1u) & 0xffffu) * 0x10001u) : 0u;
    const uint4 z = make_uint4(fill, fill, fill, fill);
    for (int i = tid; i < (int)(kTableSize * sizeof(E) / 16); i += 128) t4[i] = z;
  }
  volatile TeamShared *vs = sh;
  uint32_t sweep_at = kSweepEvery;  // kept in the token from the first round on
  __syncthreads();
  volatile E *vtable = table;

  const ChunkGeom g = stream_geom(P, sid);
  const uint16_t *scan_tab = P.scan_off;
  uint32_t pf_val = 0, pf_sink = 0;

  for (uint32_t c = 0; c < g.nchunks; ++c) {
    const uint32_t W = c * (uint32_t)kMaxStoreBlockSize;
    const uint64_t rem_len = g.len - W;
    const int n = rem_len < (uint64_t)kMaxStoreBlockSize ? (int)rem_len : kMaxStoreBlockSize;
    const uint8_t *src = g.stream + W;
    uint2 *mout = P.matches + g.mbase + (uint64_t)c * kMatchCapPerChunk;
    const int s_limit = n - kInputMargin;
    uint32_t acc_len = 0;  // per-lane partial sums of match lengths

    if (tid == 0) {
      vs->t[0].s = -1;
      vs->t[0].nm = 0;
      vs->t[0].flags = 0;
      vs->t[0].next_sweep = sweep_at;
    }

    // ---- front end: everything about 64 positions that does not depend on the parse ----
    auto front = [&](const int G) -> TeamFront {
      TeamFront f;
      f.G = G;
      const int q = G + lane;
      f.e1 = q >= 0 && q + 1 <= s_limit;
      f.e2 = q >= 0 && q + 2 <= s_limit;
      int qa = q < 0 ? 0 : q;
      if (qa > n - 16) qa = n - 16;  // (a chunk has at least 128 bytes)
      pf_sink ^= pf_val;             // retire the previous look-ahead load
      f.own = ld128(src + qa);
      f.h = f.e1 ? hash4(f.own.x) : 0u;
      f.old = (uint32_t)table[f.h];
      const uint32_t A1 = W + (uint32_t)q + 1u;
      const uint32_t dist = MULTI ? ((A1 - f.old) & 0xffffu) : (A1 - f.old);
      const bool inr = f.e1 && (MULTI ? dist != 0 : f.old != 0) && dist <= (uint32_t)kMaxMatchOffset;
      const uint32_t cand_abs = inr ? A1 - 1u - dist : W + (uint32_t)qa;
      f.cb = ld128(g.stream + cand_abs);
      {  // look-ahead: pull the next lines of this stream towards L2
        int pq = G + 768 + 4 * lane;
        if (pq > n - 4) pq = n - 4;
        if (pq < 0) pq = 0;
        pf_val = ld32(src + pq);
      }
      // lanes with equal hashes, one ballot per hash bit (the table is not touched)
      uint64_t eq = __ballot(f.e1);
#pragma unroll
      for (int k = 0; k < kTableBits; ++k) {
        const bool bit = (f.h >> k) & 1u;
        const uint64_t m = __ballot(bit);
        eq &= bit ? m : ~m;
      }
      f.eq = f.e1 ? eq : 0ull;
      return f;
    };

    // Sweep (MULTI): slots whose position is more than 32768 behind R can never be candidates
    // again.  Run by the parsing wave alone; the partner's concurrent speculative reads see either
    // the old value or the marker, both out of range, and every slot is re-read after the hand-over.
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
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      sweep_at = R + kSweepEvery;
    };

    TeamFront F;
    F.G = 0;
    F.e1 = F.e2 = false;
    F.own = F.cb = make_uint4(0, 0, 0, 0);
    F.h = F.old = 0;
    F.eq = 0;
    bool f_valid = false;
    // match records of my last parse: stored after the hand-over, off the critical path
    uint64_t pend_M = 0;
    uint32_t pend_tok = 0, pend_q = 0, pend_nm = 0;
    auto flush_records = [&]() {
      if ((pend_M >> lane) & 1) {  // streaming store: read again only by the entropy kernels
        const unsigned long long rec = (unsigned long long)pend_q | ((unsigned long long)pend_tok << 32);
        __builtin_nontemporal_store(
            rec, reinterpret_cast<unsigned long long *>(mout + pend_nm + (uint32_t)__popcll(pend_M & lanes_below(lane))));
      }
      pend_M = 0;
    };
    int Gprev = 0;
    bool have_prev = false;
    uint32_t nm_final = 0;
#ifdef FLATE_LZ_STAMPS
    uint64_t st_wait = 0, st_parse = 0, st_front = 0, st_rounds = 0, st_usable = 0,
             st_events = 0, st_sparse = 0, st_refresh = 0, st_ev = 0;
#endif

    for (uint32_t r = 0;; ++r) {
      TSTAMP(tb0);
      team_barrier<GUEST>();
      TSTAMP(tb1);
      TADD(st_wait, tb1 - tb0);
      flush_records();
      const volatile TeamToken *tk = &vs->t[r & 1];
      int s = __builtin_amdgcn_readfirstlane(tk->s);
      uint32_t nm = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk->nm);
      const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk->flags);
      sweep_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk->next_sweep);
      if (flags & 1u) {
        nm_final = nm;
        break;
      }
      const int Gguess = Gprev + kTeamStride;
      const int a0g = s - 1 - Gguess;
      const bool usable = have_prev && a0g >= 0 && a0g <= kTeamA0Max;
      const int G = usable ? Gguess : s - 1;
      Gprev = G;
      have_prev = true;
      if ((int)(r & 1u) != wave) {
        // ------------------------------ prepare the next batch ------------------------------
        F = front(G + kTeamStride);
        f_valid = true;
#ifdef FLATE_LZ_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        TSTAMP(tf1);
        TADD(st_front, tf1 - tb1);
        continue;
      }
      // ---------------------------------- parse this batch ----------------------------------
      TADD(st_rounds, 1);
      TADD(st_usable, (usable && f_valid && F.G == G) ? 1 : 0);
      if (!(usable && f_valid && F.G == G)) F = front(G);
      f_valid = false;
      const int B = G;
      const int a0 = s - 1 - B;
      if (MULTI) {
        const uint32_t first = W + (uint32_t)(s - 1 < 0 ? 0 : s - 1);
        if (first >= sweep_at) sweep(first);
      }
      const int q = B + lane;
      const bool e1 = F.e1, e2 = F.e2;
      const uint64_t E1 = __ballot(e1), E2 = __ballot(e2);
      const uint32_t A1 = W + (uint32_t)q + 1u;
      const uint4 own = F.own;
      const uint32_t h = F.h;
      // refresh: the slots as the previous batches left them
      uint32_t old = F.old;
      uint4 cb = F.cb;
      {
        const uint32_t cur = e1 ? slot_now<GUEST>(table + h) : old;
        if (__ballot(cur != old)) {
          old = cur;
          const uint32_t d2 = MULTI ? ((A1 - old) & 0xffffu) : (A1 - old);
          const bool in2 = e1 && (MULTI ? d2 != 0 : old != 0) && d2 <= (uint32_t)kMaxMatchOffset;
          int qa = q < 0 ? 0 : q;
          if (qa > n - 16) qa = n - 16;
          cb = ld128(g.stream + (in2 ? A1 - 1u - d2 : W + (uint32_t)qa));
        }
      }
      const uint32_t dist = MULTI ? ((A1 - old) & 0xffffu) : (A1 - old);
      const bool inr = e1 && (MULTI ? dist != 0 : old != 0) && dist <= (uint32_t)kMaxMatchOffset;
      const uint32_t cand_abs = A1 - 1u - dist;
      // lanes below the entry lane belong to the previous batch: they take no part
      const uint64_t eq = (lane >= a0) ? (F.eq & ~lanes_below(a0)) : 0ull;
      const uint64_t DUPall = __ballot((eq & ~(1ull << lane)) != 0);
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
      TSTAMP(tr1);
      TADD(st_refresh, tr1 - tb1);
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
      // Lanes inserted by the fast events whose start lanes are in `vis` (see lz77_kernels.hip)
      auto fast_inserts = [&](uint64_t vis) -> uint64_t {
        const uint64_t below_me = vis & lanes_upto(lane);
        const int j = below_me ? 63 - __builtin_clzll(below_me) : 0;
        const int fvj = (int)(__shfl(ev, j) & 127u);
        const int d = lane - j, o = d - 2;
        const bool sched = (d <= 1 || o < 32) ? e1 : (((o & 1) == 0) ? e2 : false);
        return __ballot(below_me != 0 && sched && lane <= fvj);
      };
      TSTAMP(te0);
      TADD(st_ev, te0 - tr1);
      uint64_t INS = 0, M = 0, MF = 0;
      uint64_t VISall = 0;
      bool done = false, sparse = false;
      int scan_base = 0, e_idx = 0;
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
      TSTAMP(te1);
      TADD(st_events, te1 - te0);
      if ((MF >> lane) & 1) acc_len += (uint32_t)tot_self;
      // commit: the inserted lanes write, and where several share a slot only the last of them
      // (position order: the latest insert wins)
      if (e1 && ((INS >> lane) & 1) && (eq & INS & ~lanes_upto(lane)) == 0) table[h] = (E)A1;
      // match records of this batch (stored after the hand-over)
      pend_M = M;
      pend_tok = rec_tok;
      pend_q = (uint32_t)q;
      pend_nm = nm;
      nm += (uint32_t)__popcll(M);

      // =============================== sparse batches ==============================
      TSTAMP(ts0);
      while (sparse && !done) {
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
        if (nall == 0) {  // emit_remainder (:152-159)
          done = true;
          break;
        }
        if (MULTI && W + (uint32_t)p0 >= sweep_at) sweep(W + (uint32_t)p0);

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
        uint64_t eqs = __ballot(ins);
#pragma unroll
        for (int k = 0; k < kTableBits; ++k) {
          const bool bit = (hs >> k) & 1u;
          const uint64_t m = __ballot(bit);
          eqs &= bit ? m : ~m;
        }
        const uint64_t C = __ballot(ins && (eqs & ~(1ull << lane)) != 0);
        if (C == 0 && ins) vtable[hs] = (E)As;

        int f = f0;
        uint32_t cand = 0;  // absolute candidate position
        if (C == 0) {
          if (f0 < 64) cand = rdlane(cands, f0);
        } else {
          // two lanes of this batch share a slot: replay the batch in order
          f = 64;
          for (int e2i = 0; e2i < nexist; ++e2i) {
            const uint32_t he = rdlane(hs, e2i);
            const uint32_t pe1 = rdlane(As, e2i);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const uint32_t cur =
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vtable[he]);
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
          if (nexist == nall && nall < 64) {  // the scan ran into s_limit
            done = true;
            break;
          }
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

      TSTAMP(ts1);
      TADD(st_sparse, ts1 - ts0);
      TADD(st_parse, ts1 - tb1);
      if (lane == 0) {  // hand over
        volatile TeamToken *nt = &vs->t[(r + 1) & 1];
        nt->s = s;
        nt->nm = nm;
        nt->flags = done ? 1u : 0u;
        nt->next_sweep = sweep_at;
      }
    }
    flush_records();
    // chunk epilogue: token count = literals + matches (DeflateFast::encode's token array length)
    uint32_t sumlen = acc_len;
    for (int d = 32; d >= 1; d >>= 1) sumlen += __shfl_xor(sumlen, d);
    if (lane == 0) vs->sum[wave] = sumlen;
    if (pf_sink == 0x9e3779b9u && P.debug) P.debug[0] = pf_sink;  // keeps the look-ahead loads alive
#ifdef FLATE_LZ_STAMPS
    if (P.debug && lane == 0) {
      unsigned long long *d = reinterpret_cast<unsigned long long *>(P.debug + (uint64_t)(g.chunk0 + c) * 8);
      atomicAdd(d + 0, (unsigned long long)st_wait);
      atomicAdd(d + 1, (unsigned long long)st_parse);
      atomicAdd(d + 2, (unsigned long long)st_front);
      atomicAdd(d + 3, (unsigned long long)st_rounds);
      atomicAdd(d + 4, (unsigned long long)st_usable);
      atomicAdd(d + 5, (unsigned long long)st_refresh);
      atomicAdd(d + 6, (unsigned long long)st_events);
      atomicAdd(d + 7, (unsigned long long)st_ev);
    }
#endif
    __syncthreads();
    if (tid == 0) {
      P.chunk_nmatch[g.chunk0 + c] = nm_final;
      P.chunk_ntok[g.chunk0 + c] = (uint32_t)n - (vs->sum[0] + vs->sum[1]) + nm_final;
    }
    __syncthreads();
  }
}

// Resident team: table in LDS (32 KiB + the token => 4 teams per CU).
template <bool MULTI>
__global__ __launch_bounds__(128) void lz77_team_kernel(LzParams P) {
  __shared__ uint16_t table[kTableSize];
  __shared__ TeamShared sh;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (bool first = true;; first = false) {
    uint32_t q;
    if (P.queue) {
      if (threadIdx.x == 0) sh.q = atomicAdd(P.queue, 1u);
      __syncthreads();
      q = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&sh.q);
      if (q >= P.queue_end) break;
    } else {
      if (!first) break;
      q = blockIdx.x;
    }
    __syncthreads();
    lz77_team_stream<MULTI, false>(P, P.stream_ids ? P.stream_ids[q] : q, table, &sh, lane, wave);
    __syncthreads();
  }
}

// Guest team: the table is a 32 KiB slice of HBM scratch that stays in the XCD's L2.
template <bool MULTI>
__global__ __launch_bounds__(128) void lz77_team_guest_kernel(LzParams P) {
  __shared__ TeamShared sh;
  uint16_t *table = reinterpret_cast<uint16_t *>(P.gtables) + (size_t)blockIdx.x * kTableSize;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (;;) {
    if (threadIdx.x == 0) sh.q = atomicAdd(P.queue, 1u);
    __syncthreads();
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&sh.q);
    if (q >= P.queue_end) break;
    __syncthreads();
    lz77_team_stream<MULTI, true>(P, P.stream_ids[q], table, &sh, lane, wave);
    __syncthreads();
  }
}

template __global__ void lz77_team_kernel<false>(LzParams);
template __global__ void lz77_team_kernel<true>(LzParams);
template __global__ void lz77_team_guest_kernel<false>(LzParams);
template __global__ void lz77_team_guest_kernel<true>(LzParams);

}  // namespace flate
