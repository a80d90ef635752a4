// lz77_kernels.hip -- deflate-fast match finder for gfx950 (wave64).
//
// Replaces DeflateFast::encode (reference deflate-fast.mbt:123-270) for a batch of
// independent streams: one wavefront per stream, the 16384-slot hash table of
// deflate-fast.mbt:95-117 resident in LDS.  The table stores positions only
// ("absolute position in the stream + 1", 0 = empty): TableEntry.val is the four
// input bytes at that position and the input is immutable, so val is re-read from
// the stream instead of being stored (u16 slots = 32 KiB per stream when the stream
// has a single window, u32 slots = 64 KiB otherwise).  TableEntry.offset - cur is
// exactly that absolute position, so the `cur` bookkeeping of :107,156 vanishes.
//
// Output: match records {position in chunk, token (token.mbt:76)}.  Literal tokens
// (token.mbt:69) are implied by the gaps and materialised by the entropy kernel.
//
// Two kernels with identical results:
//   lz77_serial_kernel : one lane walks the reference control flow (debug / device-
//                        side cross-check).
//   lz77_wave_kernel   : 64 lanes evaluate the next 64 probe events of the skip
//                        schedule at once; the sequential insert-before-judge
//                        semantics (:191-196) are restored by committing only the
//                        lanes up to the first valid candidate and detecting
//                        same-slot collisions inside the batch with an LDS
//                        write/read-back (collisions fall back to an in-order replay
//                        of that batch).
#include <type_traits>

#include "lz77_device.h"

namespace flate {

// ---------------------------------------------------------------------------------
// serial kernel: lane 0 restates the control flow of deflate-fast.mbt:123-270.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lz77_serial_kernel(LzParams P) {
  __shared__ uint32_t table[kTableSize];
  const int lane = threadIdx.x;
  const uint32_t sid = P.stream_ids ? P.stream_ids[blockIdx.x] : blockIdx.x;
  for (int i = lane; i < kTableSize; i += 64) table[i] = 0;
  __syncthreads();
  if (lane != 0) return;

  const ChunkGeom g = stream_geom(P, sid);
  for (uint32_t c = 0; c < g.nchunks; ++c) {
    const uint32_t W = c * (uint32_t)kMaxStoreBlockSize;  // absolute start of the chunk
    const uint64_t rem = g.len - W;
    const int n = rem < (uint64_t)kMaxStoreBlockSize ? (int)rem : kMaxStoreBlockSize;
    const uint8_t *src = g.stream + W;
    uint2 *mout = P.matches + g.mbase + (uint64_t)c * kMatchCapPerChunk;
    uint32_t nm = 0, sumlen = 0;

    const int s_limit = n - kInputMargin;
    int s = 0;
    uint32_t cv = ld32(src);
    bool done = false;
    while (!done) {
      int skip = 32;
      int next_s = s;
      uint32_t cand = 0;  // absolute position + 1
      for (;;) {
        s = next_s;
        int step = skip >> 5;
        next_s = s + step;
        skip += step;
        if (next_s > s_limit) {
          done = true;
          break;
        }
        uint32_t h = hash4(cv);
        cand = table[h];
        uint32_t now = ld32(src + next_s);
        table[h] = W + (uint32_t)s + 1;
        bool ok = cand != 0 && (W + (uint32_t)s + 1 - cand) <= (uint32_t)kMaxMatchOffset &&
                  ld32(g.stream + (cand - 1)) == cv;
        if (!ok) {
          cv = now;
          continue;
        }
        break;
      }
      if (done) break;
      for (;;) {
        const int pf = s;
        s += 4;
        const uint32_t Ac = cand - 1;
        int limit = n - s;
        if (limit > kMaxMatchTail) limit = kMaxMatchTail;
        int l = 0;
        if (P.compat_go || Ac + 4 >= W) {  // MoonBit: prev is empty
          const uint8_t *a = src + s, *b = g.stream + Ac + 4;
          while (l < limit && a[l] == b[l]) ++l;
        }
        mout[nm] = make_uint2((uint32_t)pf, kMatchType | ((uint32_t)(l + 1) << kLengthShift) |
                                                ((W + (uint32_t)pf) - Ac - 1));
        ++nm;
        sumlen += (uint32_t)l + 4;
        s += l;
        if (s >= s_limit) {
          done = true;
          break;
        }
        uint32_t x0 = ld32(src + s - 1), x1 = ld32(src + s);
        table[hash4(x0)] = W + (uint32_t)s;  // position s-1, stored +1
        uint32_t h1 = hash4(x1);
        cand = table[h1];
        table[h1] = W + (uint32_t)s + 1;
        bool ok = cand != 0 && (W + (uint32_t)s + 1 - cand) <= (uint32_t)kMaxMatchOffset &&
                  ld32(g.stream + (cand - 1)) == x1;
        if (!ok) {
          cv = ld32(src + s + 1);
          s += 1;
          break;
        }
      }
    }
    P.chunk_nmatch[g.chunk0 + c] = nm;
    P.chunk_ntok[g.chunk0 + c] = (uint32_t)n - sumlen + nm;
  }
}

// ---------------------------------------------------------------------------------
// wave kernel
//
// Parser state (wave-uniform): POST(s) -- the previous event was a match ending at s
// (chunk start is POST(-1)): re-insert s-1, probe s, then scan from s+1 with the skip
// schedule restarted (:246-265, :178-202) -- or SPARSE(scan_base, e_idx): a scan that
// has already run e_idx probes without success.
//
// POST is served by a DENSE batch: lane L owns position s-1+L.  Every lane loads its
// 16 input bytes, hashes, reads its slot, checks its candidate and measures the match
// (up to 16 bytes) -- one table gather and one input gather for 64 positions.  A
// register-only event loop then walks the probe schedule: first lane with a valid
// candidate = match, jump to its end, repeat inside the same batch (about six matches
// per batch on text).  Lanes that share a table slot with another lane of the batch
// (DUP, found by an LDS write/read-back) are judged exactly against the positions
// already inserted by this batch.  Inserts are committed to LDS at the end of the batch.
//
// SPARSE is served by one probe event per lane (strides grow with the skip counter).
// tests/host_model/lz77_wave_model.cpp is the lane-accurate CPU model of this kernel.
// ---------------------------------------------------------------------------------

// Diagnostic build only (-DFLATE_LZ_STAMPS): per-chunk s_memtime sums of the batch phases.
#ifdef FLATE_LZ_STAMPS
#define STAMP(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t1, t0) acc += (t1) - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(acc, t1, t0)
#endif

// One stream, start to finish.  `table` is the 16384-slot position table of this stream: LDS for
// the resident kernel, a per-block slice of HBM (L2-resident) for the guest kernel.
// Table slots are 16 bits.  MULTI == false (the stream has one LZ77 window): slot = position + 1,
// 0 = empty.  MULTI == true (windows chained through the table, deflate-fast.mbt:156,193): slot =
// (absolute position + 1) mod 2^16 and the distance is taken mod 2^16; that is unambiguous because
// every kSweepEvery positions the table is swept and every slot older than 32768 (dead for the
// reference too, :195) is replaced by a marker that stays out of range until the next sweep, and
// because a sparse batch never spans more than kSpanMax positions.

// GUEST: the table is a slice of HBM scratch living in L2 (lz77_guest_kernel).  Same algorithm;
// only the detection of lanes that share a slot avoids the table there (see the dense batch).
// c_begin .. c_end: the LZ77 windows of the stream to run (default: all of them, starting from an
// empty table).  A caller that runs one window at a time keeps the table itself and passes the
// sweep clock through *sweep_io.
//
// tags (guest blocks of single-window launches, else null): 2 bits per slot in LDS, a function of
// the four bytes at the position the slot holds (bits 17:16 of the hash product; the slot index is
// bits 31:18).  A lane whose own tag differs from its slot's tag cannot have a valid candidate --
// the reference would read the slot and fail `cv == cand.val` (deflate-fast.mbt:196) -- so it
// skips both the table gather and the candidate gather: the guest tables live in the Infinity
// Cache (profiles/r02/lz77_traffic.json) and those gathers are what fills the vector L1's miss
// queue.  Exact: every table write also writes the tag.
// The lanes of `among` whose slot h equals mine: one ballot per hash bit.  Per bit and half of the
// mask, `eq &= ~(m ^ bm)` with bm = 0 / -1 from my own bit is one three-input VALU operation
// (v_bitop3_b32), 4 VALU per bit; written with `bit ? m : ~m` on the 64-bit mask the compiler
// spent 9 (SQ counters: 331 VALU per guest batch against 200 per LDS-table batch).
FLATE_D uint64_t same_slot_lanes(uint32_t h, uint64_t among) {
  uint32_t lo = (uint32_t)among, hi = (uint32_t)(among >> 32);
#pragma unroll
  for (int k = 0; k < kTableBits; ++k) {
    uint32_t bm = (uint32_t)((int32_t)(h << (31 - k)) >> 31);  // my bit k, as 0 / ~0 (v_bfe_i32)
    asm volatile("" : "+v"(bm));  // (or the compare is derived from h again: one more shift per bit)
    const uint64_t m = __ballot(bm != 0);
    lo &= ~((uint32_t)m ^ bm);
    hi &= ~((uint32_t)(m >> 32) ^ bm);
  }
  return ((uint64_t)hi << 32) | lo;
}

// Only the first kTagSlots slots carry a tag (the others always take the gathers).  LDS is handed out in granules of
// 1280 bytes on gfx950 (160 KiB / 128): an LDS-table block takes 26 of them, and a guest with tags for all 16384
// slots (4096 B) takes FOUR -- 4 x 26 + 6 x 4 = 128: six guests per CU are resident beside four tables, whatever
// the launch asks for (profiles/r05/README.md section 5).  15360 tagged slots are 3840 B = THREE granules: eight.
#ifndef FLATE_LZ_TAG_SLOTS
#define FLATE_LZ_TAG_SLOTS 16384
#endif
constexpr uint32_t kTagSlots = FLATE_LZ_TAG_SLOTS;
static_assert(kTagSlots % 1024 == 0 && kTagSlots <= (uint32_t)kTableSize, "whole rounds of 64 lanes x 16 slots");
FLATE_D uint32_t tag_of(uint32_t cv) { return ((cv * 0x1e35a7bdu) >> 16) & 3u; }
FLATE_D bool tag_says_no(const uint32_t *tags, uint32_t h, uint32_t cv) {
  if (kTagSlots < (uint32_t)kTableSize && h >= kTagSlots) return false;
  return ((tags[h >> 4] >> (2u * (h & 15u))) & 3u) != tag_of(cv);
}
FLATE_D void tag_set(uint32_t *tags, uint32_t h, uint32_t t) {
  if (kTagSlots < (uint32_t)kTableSize && h >= kTagSlots) return;
  const uint32_t sh = 2u * (h & 15u);
  atomicAnd(&tags[h >> 4], ~(3u << sh));
  atomicOr(&tags[h >> 4], t << sh);
}

template <bool MULTI, bool GUEST = false>
FLATE_D void lz77_stream(const LzParams &P, const uint32_t sid, uint16_t *table, const int lane,
                         const uint32_t c_begin = 0, const uint32_t c_end = 0xffffffffu,
                         uint32_t *sweep_io = nullptr, uint32_t *tags = nullptr) {
  using E = uint16_t;
  if (tags && c_begin == 0)
    for (int i = lane; i < (int)kTagSlots / 16; i += 64) tags[i] = 0;
  if (c_begin == 0) {
    uint4 *t4 = reinterpret_cast<uint4 *>(table);
    const uint32_t fill = MULTI ? (((0u - kMarkerBack + 1u) & 0xffffu) * 0x10001u) : 0u;
    const uint4 z = make_uint4(fill, fill, fill, fill);
    for (int i = lane; i < (int)(kTableSize * sizeof(E) / 16); i += 64) t4[i] = z;
  }
  __syncthreads();
  volatile E *vtable = table;
  constexpr uint32_t kEMask = 0xffffu;
  // MULTI: absolute position at which the next sweep is due
  uint32_t next_sweep = (sweep_io && c_begin != 0) ? *sweep_io : kSweepEvery;
  // Sweep: slots whose position is more than 32768 behind R can never be candidates again.
  auto sweep = [&](uint32_t R) {
    uint32_t *t32 = reinterpret_cast<uint32_t *>(table);
    const uint32_t marker = (R - kMarkerBack + 1u) & 0xffffu;
    // (sixteen words per lane in flight: a guest's table is in memory, and one load at a time, each
    // waited for before its store, was 128 round trips per sweep, eight sweeps per window)
    static_assert(kTableSize / 2 % (64 * 16) == 0, "table = whole rounds of 16 dwords per lane");
    for (int i0 = lane; i0 < kTableSize / 2; i0 += 64 * 16) {
      uint32_t w[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = t32[i0 + 64 * k];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t v = w[k];
        const uint32_t d0 = (R + 1u - v) & 0xffffu, d1 = (R + 1u - (v >> 16)) & 0xffffu;
        const uint32_t lo = (d0 == 0 || d0 > 32768u) ? marker : (v & 0xffffu);
        const uint32_t hi = (d1 == 0 || d1 > 32768u) ? marker : (v >> 16);
        t32[i0 + 64 * k] = lo | (hi << 16);
      }
    }
    __syncthreads();
    next_sweep = R + kSweepEvery;
  };

  const ChunkGeom g = stream_geom(P, sid);
  const uint16_t *scan_tab = P.scan_off;
  uint32_t pf_val = 0, pf_sink = 0;
#ifdef FLATE_LZ_FINISH  // diagnostic build: when did this stream start and end, on which block (100 MHz clock)
  const uint64_t fin_t0 = __builtin_amdgcn_s_memrealtime();
#endif

  // (w0 != 0 only in resumed launches: the stream's chunks of THIS launch are windows w0, w0+1, ...)
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.win0);
  const uint32_t c_stop = c_end < w0 + g.nchunks ? c_end : w0 + g.nchunks;
  for (uint32_t c = c_begin; c < c_stop; ++c) {
    const uint32_t W = c * (uint32_t)kMaxStoreBlockSize;
    const uint64_t rem_len = g.len - W;
    const int n = rem_len < (uint64_t)kMaxStoreBlockSize ? (int)rem_len : kMaxStoreBlockSize;
    const uint8_t *src = g.stream + W;
    uint2 *mout = P.matches + g.mbase + (uint64_t)(c - w0) * kMatchCapPerChunk;
    uint32_t nm = 0;
    uint32_t acc_len = 0;  // per-lane partial sums of match lengths
    const int s_limit = n - kInputMargin;

    int s = -1;
    bool sparse = false;
    int scan_base = 0, e_idx = 0;
    bool done = false;
    uint4 own_pre = make_uint4(0, 0, 0, 0);  // input bytes of the next dense batch, loaded early
    bool pre_valid = false;
#ifdef FLATE_LZ_STAMPS
    uint64_t st_load = 0, st_dup = 0, st_ev = 0, st_commit = 0, st_sparse = 0, st_nb = 0, st_ext = 0;
#endif

    // the probe lanes of an event starting at my lane when every lane of the batch is inside the chunk
    const uint64_t spec_full =
        (2ull << lane) | (lane + 2 < 64 ? ((0x00000000ffffffffull << (lane + 2)) | (0x5555555500000000ull << (lane + 2)))
                                        : 0ull);
    // Progress guard: every batch moves the parser (a dense batch raises s or turns sparse, a sparse batch raises
    // e_idx or finds a match) or ends the chunk.  A batch that leaves all of it where it was would repeat for ever
    // (the kDenseKeep = 62 build of round 5 did): the chunk is abandoned and the call ends FLATE_HIP_E_INTERNAL.
    int g_s = -2, g_e = -1;
    bool g_sp = false;
    uint32_t g_nb = 0;
    while (!done) {
      if (s == g_s && e_idx == g_e && sparse == g_sp) {
        if (lane == 0) atomicExch(P.status, kStatusNoProgress);
        break;
      }
      g_s = s, g_e = e_idx, g_sp = sparse;
      if (!sparse) {
        // =============================== dense batch ===============================
        auto dense_batch = [&](auto interior_tag) {
          constexpr bool INTERIOR = decltype(interior_tag)::value;
        STAMP(t0);
        const int B = s - 1;
        if (MULTI) {
          const uint32_t first = W + (uint32_t)(B < 0 ? 0 : B);
          if (first >= next_sweep) sweep(first);
        }
        const int q = B + lane;
        // INTERIOR: all 64 positions and their step-2 probes lie inside the chunk (every batch but
        // the last of a chunk): e1, e2 and everything derived from them fold away
        const bool e1 = INTERIOR || (q >= 0 && q + 1 <= s_limit);  // may be inserted / probed with step 1
        const bool e2 = INTERIOR || (q >= 0 && q + 2 <= s_limit);  // may be probed with step 2
        const uint64_t E1 = INTERIOR ? ~0ull : __ballot(e1), E2 = INTERIOR ? ~0ull : __ballot(e2);
        const uint32_t A1 = W + (uint32_t)q + 1;
        uint4 own = make_uint4(0, 0, 0, 0);
        uint32_t h = 0, old = 0;
        bool maybe = false;  // my slot may hold a position with my four bytes
        pf_sink ^= pf_val;  // retire the previous batch's look-ahead load
        if (e1) {
          own = pre_valid ? own_pre : ld128(src + q);
          h = hash4(own.x);
          maybe = !tags || !tag_says_no(tags, h, own.x);
          if (maybe) old = table[h];
        }
        pre_valid = false;
        // candidate = absolute position stored in my slot; valid if within 32768 (:195)
        const uint32_t dist = MULTI ? ((A1 - old) & 0xffffu) : (A1 - old);
        const bool inr = maybe && (MULTI ? dist != 0 : old != 0) && dist <= (uint32_t)kMaxMatchOffset;
        const uint32_t cand_abs = A1 - 1u - dist;
        uint4 cb = own;
        if (inr) cb = ld128(g.stream + cand_abs);
        {  // look-ahead: pull the next lines of this stream towards L2.  Issued after the
           // candidate gather so that no wait of this batch has to include it.
          int pq = B + 768 + 4 * lane;
          if (pq > n - 4) pq = n - 4;
          pf_val = ld32(src + pq);
        }
        // DUPall: every lane that shares its slot with another lane of the batch (the commit
        // below must order their writes).  DUP: the lanes whose candidate may be a position
        // inserted by this batch, i.e. all but the first member of each group -- for the common
        // two-lane group only the later lane.
        uint64_t DUPall = 0, DUP = 0;
        uint64_t eq = 0;  // GUEST: the lanes whose slot is mine (including me)
        if (GUEST) {
          // With the table in L2 a speculative insert + read-back costs two more memory round
          // trips per batch; the lanes with equal hashes are found with one ballot per hash bit
          // instead and the table is not touched before the commit.
          eq = same_slot_lanes(h, E1);
          if (!e1) eq = 0;
          DUPall = __ballot((eq & ~(1ull << lane)) != 0);
          DUP = __ballot((eq & lanes_below(lane)) != 0);
        }
        // LDS table: speculative insert of all 64 positions + read-back
        asm volatile("" ::: "memory");
        if (!GUEST && e1) table[h] = (E)A1;
        asm volatile("" ::: "memory");
        const uint32_t rb = (!GUEST && e1) ? (uint32_t)table[h] : (A1 & kEMask);
        asm volatile("" ::: "memory");
        const bool loser = rb != (A1 & kEMask);
        const uint64_t LM = GUEST ? 0ull : __ballot(loser);
        if (!GUEST) DUPall = LM;
        if (LM) {
          const int wl = (int)((rb - (W + (uint32_t)B + 1u)) & kEMask);  // lane that owns the slot
          uint64_t m = LM;
          while (m) {
            const int x = __builtin_ctzll(m);
            const int w = (int)rdlane((uint32_t)wl, x);
            const uint64_t Lw = __ballot(loser && wl == w);
            DUPall |= 1ull << w;
            if (__popcll(Lw) == 1)
              DUP |= 1ull << (w > x ? w : x);
            else
              DUP |= Lw | (1ull << w);
            m &= ~Lw;
          }
        }
        STAMP(t1);
        // probe lanes of the event that would start with s-1 == my lane (independent of the
        // candidate bytes still in flight)
        const int bsh = lane + 2;  // (lanes 62, 63 have no scan lanes left: shift counts stay < 64)
        uint64_t specR;
        if (E1 == ~0ull && E2 == ~0ull) {  // every batch but the last ones of a chunk
          specR = spec_full;
        } else {
          specR = (E1 & (2ull << lane)) | (bsh < 64 ? (((0x00000000ffffffffull << bsh) & E1) |
                                                       ((0x5555555500000000ull << bsh) & E2))
                                                    : 0ull);
        }
        const uint64_t dupR = DUP & specR;
        const int fd = dupR ? __builtin_ctzll(dupR) : 64;
        const int mlen = inr ? prefix16(own, cb) : 0;
        const uint64_t OK = __ballot(mlen >= 4);
        // per-lane facts about "a match at my position against my slot's old value"
        const bool cross = MULTI && !P.compat_go && cand_abs + 4 < W;  // MoonBit: prev window is empty
        const int tot_self = cross ? 4 : mlen;                   // 16 => needs extension (slow)
        uint32_t rec_tok = kMatchType | ((uint32_t)(tot_self - 3) << kLengthShift) | (dist - 1u);
        // speculative evaluation of the event that would start with s-1 == my lane
        // ev: [6:0] match lane fv, [15:8] total length, bit16 general path needed, bit17 the
        // match ends the chunk, bit18 stop chasing (chunk end or next start lane > kDenseKeep),
        // [31:24] lane of the next event start (s-1 of the next event)
        uint32_t ev;
        {
          const uint64_t okR = OK & specR & ~DUP;
          const int fv = okR ? __builtin_ctzll(okR) : 64;
          const int tf = __shfl(tot_self, fv & 63);
          const bool slow = !(fv < fd) || tf >= 16;
          const bool ends = B + fv + tf >= s_limit;
          const int nxt = fv + tf - 1;
#ifdef FLATE_LZ_TERM_SHORTCUT  // (A/B builds) bit 19: no candidate and no same-slot lane among my probe lanes of this batch
          const bool off_batch = INTERIOR && fv == 64 && fd == 64;
#else
          constexpr bool off_batch = false;
#endif
          ev = (uint32_t)(fv & 127) | ((uint32_t)tf << 8) | (slow ? 1u << 16 : 0u) |
               (ends ? 1u << 17 : 0u) | ((ends || nxt > kDenseKeep) ? 1u << 18 : 0u) | (off_batch ? 1u << 19 : 0u) |
               ((uint32_t)(nxt & 255) << 24);
        }
        STAMP(t2);

        // Lanes inserted by the fast events whose start lanes are in `vis`: lane L belongs to the
        // event starting at the last visited lane j <= L and was inserted iff it is on that
        // event's probe schedule (j, j+1, then j+2+{0..31, 32,34,..}) and not past its match lane.
        auto fast_inserts = [&](uint64_t vis) -> uint64_t {
          const uint64_t below_me = vis & lanes_upto(lane);
          const int j = below_me ? 63 - __builtin_clzll(below_me) : 0;
          const int fvj = (int)(__shfl(ev, j) & 127u);
          const int d = lane - j, o = d - 2;
          const bool sched = (d <= 1 || o < 32) ? e1 : (((o & 1) == 0) ? e2 : false);
          return __ballot(below_me != 0 && sched && lane <= fvj);
        };
        uint64_t INS = 0, M = 0, MF = 0;  // inserted lanes (general events), match lanes, fast match lanes
        uint64_t VISall = 0;              // start lanes of the fast events chased so far
        int a = 0;
        for (;;) {  // events inside this batch
          // Chase the fast events: x = ev[a]; visit a; a = next[a]; until an event needs the
          // general path or asks to stop.  Hand-written: a step is 9 scalar
          // instructions (the compiler's version of this loop cost ~290 cycles per match).
          uint32_t x, tmp;
          uint64_t VIS = 0, MFl = 0;
          int a_s = __builtin_amdgcn_readfirstlane(a);
          STAMP(tc0);
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
              "s_cbranch_scc1 2f\n\t"
              // (second copy: one taken branch per two events; the lane select written by s_lshr
              // needs its four wait states here too)
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
          STAMP(tc1);
          STAMP_ADD(st_sparse, tc1, tc0);
          VISall |= VIS;
          s = B + a + 1;
          if (!(x & (1u << 16))) {  // stopped after a fast event
            if (x & (1u << 17)) done = true;
            break;
          }
#ifdef FLATE_LZ_TERM_SHORTCUT
          // the batch's usual last event (it starts at a lane a > 0 and finds nothing before lane 64): the general
          // path would walk its probe lanes, find no match and leave without changing INS or M
          if (INTERIOR && (x & (1u << 19)) && a != 0) break;
#endif
          // ---- general event (shared slots, long matches, end of scan) ----
          STAMP(tg0);
          const uint64_t FINS = fast_inserts(VISall);
          const uint64_t a_ins = E1 & (1ull << a);
          const int b = a + 2;
          const uint64_t full = 0x55555555ffffffffull << b;  // scan probe lanes (steps 1 then 2)
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
            // lane fd shares its slot with other lanes of this batch: judge it against the
            // latest position this batch has already inserted into that slot
            T |= rem & lanes_below(fd);
            const uint32_t hfd = rdlane(h, fd);
            const uint64_t G = __ballot(e1 && h == hfd) & (INS | FINS | T | a_ins) & lanes_below(fd);
            bool v;
            uint32_t cnd;
            int ml;
            if (G) {
              const int i = 63 - __builtin_clzll(G);
              v = rdlane(own.x, i) == rdlane(own.x, fd);
              cnd = W + (uint32_t)(B + i);
              const uint4 oi = make_uint4(rdlane(own.x, i), rdlane(own.y, i), rdlane(own.z, i),
                                          rdlane(own.w, i));
              const uint4 of = make_uint4(rdlane(own.x, fd), rdlane(own.y, fd), rdlane(own.z, fd),
                                          rdlane(own.w, fd));
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
          STAMP(tg1);
          STAMP_ADD(st_ext, tg1, tg0);
          if (s >= s_limit) {
            done = true;
            break;
          }
          a = s - 1 - B;
          if (a > kDenseKeep) break;
        }
        M |= MF;
        INS |= fast_inserts(VISall);
        STAMP(t3);
        // the next dense batch starts at s - 1: issue its input load now so that its latency
        // overlaps the record store and the table commit below
        if (!done && !sparse) {
          const int qn = s - 1 + lane;
          if (qn + 1 <= s_limit) own_pre = ld128(src + qn);
          pre_valid = true;
        }
        // match records of this batch, in position order, one coalesced store
        if ((MF >> lane) & 1) acc_len += (uint32_t)tot_self;
        if ((M >> lane) & 1)
          {  // streaming store: the records are read again only by the entropy kernels, and
             // keeping them out of L2 leaves more of it to the guest blocks' hash tables
            const unsigned long long rec = (unsigned long long)(uint32_t)q | ((unsigned long long)rec_tok << 32);
            __builtin_nontemporal_store(
                rec, reinterpret_cast<unsigned long long *>(mout + nm + (uint32_t)__popcll(M & lanes_below(lane))));
          }
        nm += (uint32_t)__popcll(M);
        // commit: slots of non-DUP lanes already hold their position (speculative write);
        // un-inserted lanes and every DUP lane restore the old value, then the inserted DUP
        // lanes write in position order so that the latest one wins.
        asm volatile("" ::: "memory");
        if (GUEST) {
          // the table is still untouched: the inserted lanes write, and where several share a
          // slot only the last of them (position order: the latest insert wins)
          if (e1 && ((INS >> lane) & 1) && (eq & INS & ~lanes_upto(lane)) == 0) {
            table[h] = (E)A1;
            if (tags) tag_set(tags, h, tag_of(own.x));
          }
        } else {
          if (e1 && (((DUPall | ~INS) >> lane) & 1)) table[h] = (E)old;
          uint64_t dm = DUPall & INS;
          while (dm) {
            const int k = __builtin_ctzll(dm);
            asm volatile("" ::: "memory");
            if (lane == k) table[h] = (E)A1;
            dm &= dm - 1;
          }
        }
        asm volatile("" ::: "memory");
        STAMP(t4);
        STAMP_ADD(st_dup, t1, t0);
        STAMP_ADD(st_load, t2, t1);
        STAMP_ADD(st_ev, t3, t2);
        STAMP_ADD(st_commit, t4, t3);
#ifdef FLATE_LZ_STAMPS
        st_nb += 1;
#endif
        };
        if (s - 1 >= 0 && s - 1 + 65 <= s_limit)
          dense_batch(std::true_type{});
        else
          dense_batch(std::false_type{});
        // (test hook, option debug_stall_batch: the k-th dense batch of a chunk forgets what it did)
        if (P.inject_stall != 0 && ++g_nb == P.inject_stall) s = g_s, sparse = false, done = false, pre_valid = false;
      } else {
        // =============================== sparse batch ==============================
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
        // MULTI: keep the batch within kSpanMax positions (the later probes move to the next batch)
        const bool exists = exists_all && (!MULTI || (uint32_t)(p - p0) < kSpanMax);
        const int nall = __popcll(__ballot(exists_all));
        const int nexist = __popcll(__ballot(exists));  // events are a prefix of the lanes
        if (nall == 0) break;                           // emit_remainder (:152-159)
        if (MULTI && W + (uint32_t)p0 >= next_sweep) sweep(W + (uint32_t)p0);

        uint32_t cv = 0, h = 0, old = 0;
        if (exists) {
          cv = ld32(src + p);
          h = hash4(cv);
          old = vtable[h];
        }
        const uint32_t A1 = W + (uint32_t)p + 1;
        const uint32_t dist = MULTI ? ((A1 - old) & 0xffffu) : (A1 - old);
        const uint32_t cand_abs = A1 - 1u - dist;
        bool ok = false;
        if (exists && (MULTI ? dist != 0 : old != 0) && dist <= (uint32_t)kMaxMatchOffset)
          ok = ld32(g.stream + cand_abs) == cv;
        const uint64_t V = __ballot(ok);
        const int f0 = ffs64(V);

        // commit the inserts of lanes <= first valid lane; detect same-slot collisions
        const int lim = f0 < nexist - 1 ? f0 : nexist - 1;
        const bool ins = lane <= lim;
        uint64_t C;
        if (GUEST) {  // same-slot lanes among the inserted ones by ballots (see the dense batch)
          const uint64_t eqs = same_slot_lanes(h, __ballot(ins));
          C = __ballot(ins && (eqs & ~(1ull << lane)) != 0);
          if (C == 0 && ins) {
            vtable[h] = (E)A1;
            if (tags) tag_set(tags, h, tag_of(cv));
          }
        } else {
          if (ins) vtable[h] = (E)A1;
          const uint32_t rb = ins ? (uint32_t)vtable[h] : (A1 & kEMask);
          C = __ballot(rb != (A1 & kEMask));
        }

        int f = f0;
        uint32_t cand = 0;  // absolute candidate position
        if (C == 0) {
          if (f0 < 64) cand = rdlane(cand_abs, f0);
        } else {
          // two lanes of this batch share a slot: replay the batch in order
          if (!GUEST && ins) vtable[h] = (E)old;
          f = 64;
          for (int e2 = 0; e2 < nexist; ++e2) {
            const uint32_t he = rdlane(h, e2);
            const uint32_t pe1 = rdlane(A1, e2);
            const uint32_t cur =
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vtable[he]);
            vtable[he] = (E)pe1;
            if (tags && lane == e2) tag_set(tags, h, tag_of(cv));
            bool v;
            uint32_t cnd;
            if (cur == rdlane(old, e2)) {
              v = (V >> e2) & 1;
              cnd = rdlane(cand_abs, e2);
            } else {  // candidate was inserted by an earlier lane of this batch
              const uint64_t m = __ballot(exists && (A1 & kEMask) == cur);
              v = false;
              cnd = 0;
              if (m) {
                const int i = __builtin_ctzll(m);
                v = rdlane(cv, i) == rdlane(cv, e2);
                cnd = rdlane(A1, i) - 1u;
              }
            }
            if (v) {
              f = e2;
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
      P.chunk_nmatch[g.chunk0 + c - w0] = nm;
      P.chunk_ntok[g.chunk0 + c - w0] = (uint32_t)n - sumlen + nm;
#ifdef FLATE_LZ_STAMPS
      if (P.debug) {
        uint64_t *d = P.debug + (uint64_t)(g.chunk0 + c - w0) * 8;
        d[0] = st_dup; d[1] = st_load; d[2] = st_ev; d[3] = st_commit; d[4] = st_nb; d[5] = nm;
        d[6] = st_ext; d[7] = st_sparse;
      }
#endif
    }
  }
  if (sweep_io) *sweep_io = next_sweep;
#ifdef FLATE_LZ_FINISH
  if (lane == 0 && P.debug && c_begin == 0) {
    uint64_t *d = P.debug + (uint64_t)g.chunk0 * 8;
    d[0] = fin_t0;
    d[1] = __builtin_amdgcn_s_memrealtime();
    d[2] = ((uint64_t)(GUEST ? 1u : 0u) << 32) | blockIdx.x;
  }
#endif
}

// Every lane of the wavefront must still be in the persistent loop.  hipcc once peeled lane 0 off
// such a loop (see uq_pop): the blocks then ran every later stream with lane 0 masked off and
// produced wrong records silently.  The code shape that avoids it is kept, and this turns a
// recurrence (a compiler upgrade) into FLATE_HIP_E_INTERNAL instead of wrong bytes.
FLATE_D void all_lanes_here(const LzParams &P, int lane) {
  const uint64_t here = __ballot(true);
  if (here != ~0ull && lane == (int)__builtin_ctzll(here)) atomicExch(P.status, kStatusLanesLost);
}

// ---------------------------------------------------------------------------------
// Window-granular scheduling of multi-window streams.  A stream's windows are sequential (the
// table carries over, deflate-fast.mbt:156,193), so with one wavefront per STREAM a batch of
// n streams on S table slots takes ceil(n / S) whole-stream rounds -- 4096 streams of four windows
// on 2560 slots: two rounds, the second 60 % full.  Here the unit of work is one WINDOW: every
// block pops the next ready unit {stream, window} from one FIFO, runs that window, and pushes the
// stream's next window to the back.  All first windows are ready at the start, so the order is
// breadth-first and every slot stays busy until the last window time.  Between two of its
// windows a stream's table rests in global memory (uq_tables, 32 KiB per stream): every block loads
// it into its own working table (LDS, or the guest block's slice of gtables) and stores it back.
// ---------------------------------------------------------------------------------
struct UqUnit {
  uint32_t q;  // queue entry (index into stream_ids)
  uint32_t c;  // LZ77 window of that stream
  bool ok;
};

// Pop the next unit: a ticket from the head counter, then wait until the unit with that ticket
// has been pushed (its producer is a block that is running a window right now, so the wait is
// bounded by one window time; a bounded spin turns anything else into an error, not a hang).
FLATE_D UqUnit uq_pop(const LzParams &P, const uint32_t push_word, int lane) {
  // One lane-0 block per loop iteration: publish the window this block has just finished with
  // (push_word: 0 = nothing, bit 31 = that stream is complete, else the next window's ready word; its payload was stored write-through and drained by every storing lane, the
  // window's match records are read only by later kernels), then take a ticket.  (Two separate
  // lane-0 blocks, one at the end and one at the start of the loop body, made the compiler peel
  // lane 0 off the unit loop: after the first unit the guest blocks ran without it.)
  uint32_t t = 0;
  if (lane == 0) {
    if (push_word != 0 && !(push_word & 0x80000000u)) {  // (bit 31: the previous unit was its stream's last window)
      const uint32_t k = __hip_atomic_fetch_add(P.uq_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (test hook: option debug_drop_window_push loses one hand-over, so that the bounded wait
      // below can be shown to end in an error code and not in a hang)
      if (k + 1u - P.queue_end != P.inject_drop_push)
        __hip_atomic_store(P.uq_ready + k, push_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t = __hip_atomic_fetch_add(P.uq_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  const uint32_t units = P.uq_units;
  uint32_t v = 0;
  if (t < units) {
    // bounded by polls, not by wall time: a wave that is preempted or time-sliced with another
    // process does not run out its bound while its producer is not running either
    for (uint32_t polls = 0;; ++polls) {
      uint32_t x = 0;
      if (lane == 0) x = __hip_atomic_load(P.uq_ready + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
      if (v != 0) break;
      if (polls >= P.spin_limit) break;
      __builtin_amdgcn_s_sleep(16);
    }
    if (v == 0 && lane == 0) atomicExch(P.status, kStatusUqTimeout);  // gave up: FLATE_HIP_E_INTERNAL
    // acquire (one lane: the invalidate is per CU) before anyone reads the producer's payload
    if (lane == 0) (void)__hip_atomic_load(P.uq_ready + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  UqUnit u;
  v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  u.ok = v != 0;
  u.q = (v >> 15) - 1u;  // ready word = (queue entry + 1) << 15 | window (entry < 2^17 - 1, window < 2^15)
  u.c = v & 0x7fffu;
  return u;
}

__global__ void uq_init_kernel(uint32_t *ready, uint32_t *ctr, uint32_t n_streams, uint32_t n_units) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_units) ready[i] = i < n_streams ? ((i + 1u) << 15) : 0u;  // {entry i, window 0}
  if (i == 0) {
    ctr[0] = 0;
    ctr[1] = n_streams;
  }
}

// One unit.  The working table is the block's own (LDS, or the guest block's slice of HBM); the
// stream's table travels through uq_tables.  The hand-off follows the write-through form of the
// inter-workgroup rules for gfx950 (private, mutually incoherent L2 per XCD): every payload word
// (table, sweep clock) is stored and loaded with agent-scope relaxed atomics -- `sc1` accesses that
// go past L1 and are written through L2 -- every storing lane drains its stores, then one lane
// publishes the ready word; the consumer polls that word and acquires once (uq_pop).
template <bool GUEST>
FLATE_D uint32_t uq_run(const LzParams &P, const UqUnit u, uint16_t *table, int lane, uint32_t *tags = nullptr) {
  // (wave-uniform values made scalar explicitly: with a per-lane `nch` the branch around the push
  // below is divergent for the compiler, which then peels lane 0 off the unit loop)
  const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.stream_ids[u.q]);
  const uint32_t nch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(P.chunk_base[sid + 1] - P.chunk_base[sid]));
  uint32_t *home = reinterpret_cast<uint32_t *>(P.uq_tables + (size_t)u.q * kTableSize);
  uint32_t *work = reinterpret_cast<uint32_t *>(table);
  uint32_t clock = 0;
  if (u.c != 0) {
    clock = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)__hip_atomic_load(P.uq_sweep + u.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // The table arrives from another block (an LDS-table block keeps no tags, and a guest's tags stay
    // behind in its LDS): a guest rebuilds its slot tags from the bytes the slots point at while the
    // table passes through its registers.  A slot holds (position + 1) mod 2^16 of a position at
    // most kMarkerBack behind this window's start W (older ones were swept to a marker, which is out
    // of range for every lookup whatever its tag says).
    const uint8_t *stream = P.in + P.in_off[sid];
    const uint32_t W = u.c * (uint32_t)kMaxStoreBlockSize;
    if (tags) {
      for (int i = lane; i < (int)kTagSlots / 16; i += 64) tags[i] = 0;
      __syncthreads();
    }
    // (sixteen loads in flight per lane: one at a time, each waited for before its store, was 128
    // round trips through the fabric per window -- a tenth of the window's own time; the tags' gathers
    // likewise, sixteen at a time)
    static_assert(kTableSize / 2 % (64 * 16) == 0, "table = whole rounds of 16 dwords per lane");
    for (int i0 = lane; i0 < kTableSize / 2; i0 += 64 * 16) {
      uint32_t w[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = __hip_atomic_load(home + i0 + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int k = 0; k < 16; ++k) work[i0 + 64 * k] = w[k];
      if (tags) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint32_t cv[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {  // dword half * 8 + j / 2, its low or high slot
            const uint32_t slot = (w[half * 8 + j / 2] >> (16 * (j & 1))) & 0xffffu;
            const uint32_t back = (W + 1u - slot) & 0xffffu;  // W - position
            cv[j] = (back != 0u && back <= W) ? ld32(stream + (W - back)) : 0xffffffffu;
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const uint32_t slot = (w[half * 8 + j / 2] >> (16 * (j & 1))) & 0xffffu;
            const uint32_t back = (W + 1u - slot) & 0xffffu;
            const uint32_t h = 2u * (uint32_t)(i0 + 64 * (half * 8 + j / 2)) + (uint32_t)(j & 1);
            if (back != 0u && back <= W && h < kTagSlots) atomicOr(&tags[h >> 4], tag_of(cv[j]) << (2u * (h & 15u)));
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  lz77_stream<true, GUEST>(P, sid, table, lane, u.c, u.c + 1, &clock, tags);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (u.c + 1 < nch) {
    for (int i0 = lane; i0 < kTableSize / 2; i0 += 64 * 16) {  // (a guest's working table is in memory too)
      uint32_t w[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = work[i0 + 64 * k];
#pragma unroll
      for (int k = 0; k < 16; ++k) __hip_atomic_store(home + i0 + 64 * k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store(P.uq_sweep + u.q, clock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every storing lane drains
    __syncthreads();
    return ((u.q + 1u) << 15) | (u.c + 1u);  // the ready word of the stream's next window
  }
  return 0x80000000u | u.q;  // the stream is finished (entry < 2^17: the flag bit is free)
}

// Resident kernel: table in LDS (32 KiB per stream => 5 streams per CU).
template <bool MULTI>
__global__ __launch_bounds__(64) void lz77_wave_kernel(LzParams P) {
  __shared__ uint16_t table[kTableSize];
  const int lane = threadIdx.x;
  if (MULTI && P.uq_ready) {  // persistent, one window at a time (see uq_run)
    uint32_t push_word = 0;
    uint32_t mine = 0;
    for (;;) {
      const UqUnit u = uq_pop(P, push_word, lane);
      if (__builtin_amdgcn_readfirstlane((int)u.ok) == 0) break;
      push_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)uq_run<false>(P, u, table, lane));
      all_lanes_here(P, lane);
      ++mine;
    }
    if (P.taken && lane == 0) __hip_atomic_fetch_add(P.taken, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // Either one block per stream (P.queue == null) or persistent: resident and guest blocks
  // share one queue (dynamic balance).  One call site keeps a single copy of the parser.
  uint32_t mine = 0;  // streams taken
  for (bool first = true;; first = false) {
    uint32_t q;
    if (P.queue) {
      q = 0;
      if (lane == 0) q = atomicAdd(P.queue, 1u);  // (the ONE lane-0 block of the loop: see uq_pop)
      q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
      if (q >= P.queue_end) break;
    } else {
      if (!first) break;
      q = blockIdx.x;
    }
    __syncthreads();
    lz77_stream<MULTI>(P, P.stream_ids ? P.stream_ids[q] : q, table, lane);
    __syncthreads();
    all_lanes_here(P, lane);
    ++mine;
  }
  if (P.taken && lane == 0) __hip_atomic_fetch_add(P.taken, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Guest kernel: the LDS of a CU holds only five 32 KiB tables, but its SIMDs are mostly idle
// (the parser is latency-bound).  A small persistent grid of extra wavefronts runs the same
// parser with tables in HBM scratch -- few enough that those tables stay in the 4 MiB L2 of
// their XCD -- and pulls streams from a queue.
template <bool MULTI>
__global__ __launch_bounds__(64) void lz77_guest_kernel(LzParams P) {
  if (blockIdx.x >= P.gtable_blocks) return;  // (gtables holds one table per block of this launch)
  uint16_t *table = reinterpret_cast<uint16_t *>(P.gtables) + (size_t)blockIdx.x * kTableSize;
  const int lane = threadIdx.x;
  // (round 2 gave multi-window guests no tags: -7 % with them at equal geometry, but the window
  // units hand a stream from block to block and the tags did not travel; round 3 rebuilds them at the
  // start of a unit instead, see uq_run)
  __shared__ uint32_t tag_mem[kTagSlots / 16];  // see tag_of, kTagSlots
  uint32_t *tags = tag_mem;
  if (MULTI && P.uq_ready) {  // persistent, one window at a time, table in place (see uq_run)
    uint32_t push_word = 0;
    for (;;) {
      const UqUnit u = uq_pop(P, push_word, lane);
      if (__builtin_amdgcn_readfirstlane((int)u.ok) == 0) break;
      push_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)uq_run<true>(P, u, table, lane, tags));
      all_lanes_here(P, lane);
    }
    return;
  }
  for (;;) {
    uint32_t q = 0;
    if (lane == 0) q = atomicAdd(P.queue, 1u);  // (the ONE lane-0 block of the loop: see uq_pop)
    q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (q >= P.queue_end) break;
    __syncthreads();
    lz77_stream<MULTI, true>(P, P.stream_ids[q], table, lane, 0, 0xffffffffu, nullptr, tags);
    __syncthreads();
    all_lanes_here(P, lane);
  }
}

// A stream that continues across calls (flate_hip_stream_write): the table and the sweep clock rest
// in global memory between the calls, exactly as between two window units (uq_run) -- but the next
// launch is a later kernel on the same HIP stream, so plain loads and stores are enough.
// rebase: the host has moved the stream's origin up by that many bytes (a multiple of 65535) so that
// absolute positions stay small however long the stream grows -- the reference does the same to its
// table offsets when `cur` nears 2^31 (shift_offsets, deflate-fast.mbt:366-389).  The slots hold
// positions mod 2^16 (markers included), so every slot and the sweep clock move down by the same
// amount and all distances stay what they were.
// forget: the launch's first window starts on an EMPTY table -- what the reference's shift_offsets does
// when `prev` is empty, which in MoonBit it always is (deflate-fast.mbt:367-374): every slot
// becomes the marker that is out of range for the window's first position, exactly as a sweep that finds
// every slot too old would leave it.
__global__ __launch_bounds__(64) void lz77_resume_kernel(LzParams P, uint16_t *table_io, uint32_t *clock_io,
                                                          uint32_t nwin, uint32_t rebase, uint32_t forget) {
  __shared__ uint16_t table[kTableSize];
  const int lane = threadIdx.x;
  uint32_t clock = 0;
  if (P.win0 != 0 && forget) {
    const uint32_t W = P.win0 * (uint32_t)kMaxStoreBlockSize;
    const uint32_t fill = ((W - kMarkerBack + 1u) & 0xffffu) * 0x10001u;
    uint4 *dst = reinterpret_cast<uint4 *>(table);
    for (int i = lane; i < (int)(kTableSize * sizeof(uint16_t) / 16); i += 64) dst[i] = make_uint4(fill, fill, fill, fill);
    clock = W + kSweepEvery;
  } else if (P.win0 != 0) {
    const uint4 *src = reinterpret_cast<const uint4 *>(table_io);
    uint4 *dst = reinterpret_cast<uint4 *>(table);
    for (int i = lane; i < (int)(kTableSize * sizeof(uint16_t) / 16); i += 64) dst[i] = src[i];
    clock = *clock_io - rebase;
    if (rebase) {
      __syncthreads();
      uint32_t *t32 = reinterpret_cast<uint32_t *>(table);
      const uint32_t d16 = rebase & 0xffffu;
      for (int i = lane; i < kTableSize / 2; i += 64) {
        const uint32_t v = t32[i];
        t32[i] = ((v - d16) & 0xffffu) | ((((v >> 16) - d16) & 0xffffu) << 16);
      }
    }
  }
  __syncthreads();
  lz77_stream<true>(P, 0, table, lane, P.win0, P.win0 + nwin, &clock);
  __syncthreads();
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(table);
    uint4 *dst = reinterpret_cast<uint4 *>(table_io);
    for (int i = lane; i < (int)(kTableSize * sizeof(uint16_t) / 16); i += 64) dst[i] = src[i];
    if (lane == 0) *clock_io = clock;
  }
}

template __global__ void lz77_wave_kernel<false>(LzParams);
template __global__ void lz77_wave_kernel<true>(LzParams);
template __global__ void lz77_guest_kernel<false>(LzParams);
template __global__ void lz77_guest_kernel<true>(LzParams);

}  // namespace flate
