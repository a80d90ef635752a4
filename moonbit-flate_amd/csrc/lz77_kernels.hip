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
#include "flate_kernels.h"

namespace flate {

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
// ---------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(64) void lz77_wave_kernel(LzParams P) {
  __shared__ E table[kTableSize];
  const int lane = threadIdx.x;
  const uint32_t sid = P.stream_ids ? P.stream_ids[blockIdx.x] : blockIdx.x;
  {
    uint4 *t4 = reinterpret_cast<uint4 *>(table);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < (int)(kTableSize * sizeof(E) / 16); i += 64) t4[i] = z;
  }
  __syncthreads();
  volatile E *vtable = table;

  const ChunkGeom g = stream_geom(P, sid);
  const uint16_t *scan_tab = P.scan_off;

  for (uint32_t c = 0; c < g.nchunks; ++c) {
    const uint32_t W = c * (uint32_t)kMaxStoreBlockSize;
    const uint64_t rem = g.len - W;
    const int n = rem < (uint64_t)kMaxStoreBlockSize ? (int)rem : kMaxStoreBlockSize;
    const uint8_t *src = g.stream + W;
    uint2 *mout = P.matches + g.mbase + (uint64_t)c * kMatchCapPerChunk;
    uint32_t nm = 0, sumlen = 0;
    const int s_limit = n - kInputMargin;

    // Wave-uniform parser state.  post: the previous event was a match ending at s
    // (s < s_limit): lane 0 re-inserts s-1, lane 1 probes s, lanes 2.. run the scan
    // that restarts at s+1 (:246-265, then :178-202).  Otherwise lanes continue the
    // scan that started at scan_base with probe index e_idx.
    bool post = false;
    int s = 0, scan_base = 0, e_idx = 0;

    for (;;) {
      // ---- lane -> probe event -------------------------------------------------
      int p, step, e;
      bool probe = true;
      if (post) {
        e = lane - 2;
        if (lane < 2) {
          p = s - 1 + lane;
          step = 0;
          probe = lane == 1;
        } else {
          p = s + 1 + scan_off_small(e, &step);
        }
      } else {
        e = e_idx + lane;
        if (e < kScanClosedForm) {
          p = scan_base + scan_off_small(e, &step);
        } else {
          int ec = e < P.scan_len - 1 ? e : P.scan_len - 2;
          int o0 = scan_tab[ec], o1 = scan_tab[ec + 1];
          p = scan_base + o0 + (e - ec) * 65536;  // beyond the table => never exists
          step = o1 - o0;
        }
      }
      const bool exists = p + step <= s_limit;  // the `next_s > s_limit` test of :188
      const uint64_t exm = __ballot(exists);
      const int nexist = __popcll(exm);  // events are a prefix of the lanes
      if (nexist == 0) break;            // emit_remainder (:152-159)

      // ---- probe: hash, table read, candidate check -----------------------------
      uint32_t cv = 0, h = 0, old = 0;
      if (exists) {
        cv = ld32(src + p);
        h = hash4(cv);
        old = vtable[h];
      }
      const uint32_t A1 = W + (uint32_t)p + 1;
      bool ok = false;
      if (exists && probe && old != 0 && (A1 - old) <= (uint32_t)kMaxMatchOffset)
        ok = ld32(g.stream + (old - 1)) == cv;
      const uint64_t V = __ballot(ok);
      const int f0 = V ? __builtin_ctzll(V) : 64;

      // ---- commit the inserts of lanes <= first valid lane; detect collisions ----
      const int lim = f0 < nexist - 1 ? f0 : nexist - 1;
      const bool ins = lane <= lim;
      if (ins) vtable[h] = (E)A1;
      const E rb = ins ? vtable[h] : (E)A1;
      const uint64_t C = __ballot(rb != (E)A1);

      int f = f0;
      uint32_t cand1 = 0;  // candidate position + 1
      if (C == 0) {
        if (f0 < 64) cand1 = rdlane(old, f0);
      } else {
        // Two lanes of this batch share a slot: replay the batch in order.
        if (ins) vtable[h] = (E)old;
        f = 64;
        for (int e2 = 0; e2 < nexist; ++e2) {
          const uint32_t he = rdlane(h, e2);
          const uint32_t pe1 = rdlane(A1, e2);
          const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vtable[he]);
          vtable[he] = (E)pe1;
          if (post && e2 == 0) continue;  // insert-only event (s-1)
          bool v;
          if (cur == rdlane(old, e2)) {
            v = (V >> e2) & 1;
          } else {
            // candidate was inserted by an earlier lane of this batch
            const uint64_t m = __ballot(exists && A1 == cur);
            v = false;
            if (m) v = rdlane(cv, __builtin_ctzll(m)) == rdlane(cv, e2);
          }
          if (v) {
            f = e2;
            cand1 = cur;
            break;
          }
        }
      }

      if (f == 64) {  // no candidate in this batch
        if (nexist < 64) break;
        if (post) {
          post = false;
          scan_base = s + 1;
          e_idx = 62;
        } else {
          e_idx += 64;
        }
        continue;
      }

      // ---- match at lane f: extend (match_len, :286-342) -------------------------
      const int pf = (int)rdlane((uint32_t)p, f);
      const uint32_t Ac = cand1 - 1;  // absolute candidate position
      const int s2 = pf + 4;
      int limit = n - s2;
      if (limit > kMaxMatchTail) limit = kMaxMatchTail;
      int l = 0;
      if (limit > 0 && (P.compat_go || Ac + 4 >= W)) {
        const int o = 4 * lane;
        uint32_t x = 0;
        if (o < limit) {
          const int r = limit - o;
          const uint8_t *pa = src + s2 + o, *pb = g.stream + Ac + 4 + o;
          if (r >= 4) {
            x = ld32(pa) ^ ld32(pb);
          } else {
            x = ld_partial(pa, r) ^ ld_partial(pb, r);
          }
        }
        const uint64_t mm = __ballot(x != 0);
        if (mm) {
          const int k = __builtin_ctzll(mm);
          const uint32_t xk = rdlane(x, k);
          l = 4 * k + (__builtin_ctz(xk) >> 3);
        } else {
          l = limit;
        }
      }
      if (lane == 0)
        mout[nm] = make_uint2((uint32_t)pf, kMatchType | ((uint32_t)(l + 1) << kLengthShift) |
                                                ((W + (uint32_t)pf) - Ac - 1));
      ++nm;
      sumlen += (uint32_t)l + 4;
      s = s2 + l;
      if (s >= s_limit) break;
      post = true;
    }
    if (lane == 0) {
      P.chunk_nmatch[g.chunk0 + c] = nm;
      P.chunk_ntok[g.chunk0 + c] = (uint32_t)n - sumlen + nm;
    }
  }
}

template __global__ void lz77_wave_kernel<uint16_t>(LzParams);
template __global__ void lz77_wave_kernel<uint32_t>(LzParams);

}  // namespace flate
