// huff_pack_kernels.hip -- per-stream entropy stage for gfx950 (wave64).
//
// Replaces, for a batch of streams, the block writer of the reference:
//   Compressor::enc_speed / close      deflate.mbt:236-277,157-183 (block policy)
//   HuffmanBitWriter::write_block_dynamic / write_block_huff / write_stored_header
//                                       huffman-bit-writer.mbt:496-542,738-824,474-487
//   index_tokens / generate_codegen / dynamic_size / write_dynamic_header / write_tokens
//                                       huffman-bit-writer.mbt:550-593,241-330,335-360,421-471,596-731
//   HuffmanEncoder::generate            huffman-code.mbt:295-343
//
// Three kernels, one wavefront per stream each, split so that every phase runs at the
// occupancy its LDS footprint allows:
//   huff_hist_kernel  index_tokens: per-block symbol histograms          (1.8 KiB LDS; bound by the LDS pipe)
//   huff_code_kernel  code construction, header, exact block bit sizes   (11.8 KiB LDS: 13 wavefronts per CU;
//                     bound by one block's chain of dependent LDS round trips)
//   huff_pack_kernel  write_dynamic_header + write_tokens, straight into the final
//                     output at the offsets given by a scan of the exact sizes (4.3 KiB LDS; VALU-bound);
//                     the token walk itself is handed over by huff_hist_kernel (tile_meta)
// Everything that is a loop over tokens or symbols in the reference is a wave-parallel pass:
//  * the token sequence is never materialised: each lane owns four input positions of a
//    256-byte tile and decides from the (sorted) match records which of them are literals,
//    the start of a match, or covered by one;
//  * code lengths come from a level-parallel package-merge that yields the same counts
//    as the lazy boundary algorithm of huffman-code.mbt:112-244 (ties between a leaf
//    and a pair go to the pair, as `next_char_freq < next_pair_freq` at :187 dictates);
//  * bits are placed by a wavefront prefix scan of the per-lane code lengths and OR-ed
//    into an LDS ring that is drained to HBM with coalesced dword stores (the 48-bit
//    accumulator and 248-byte buffer of huffman-bit-writer.mbt:170-199 describe the
//    same LSB-first bit string).
#include "flate_kernels.h"

namespace flate {

namespace {

constexpr int kRing = 512;  // dwords in the LDS bit ring (a 256-position tile adds <= 186 dwords)
constexpr int kTile = 256;  // input positions per walk step: 4 consecutive positions per lane
constexpr int kHdrMax = 704;

// codegen_order, huffman-bit-writer.mbt:83-85 (RFC 1951 3.2.7)
__constant__ uint8_t kCodegenOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// LDS of huff_code_kernel.  The kernel is latency-bound and this struct decides how many wavefronts
// a CU holds (12 KiB: 13), so arrays whose lives do not overlap share their bytes: the codegen items
// sit in the sort keys (dead once a code's leaves are sorted; the 19-symbol codegen code built while
// the items are live touches the first 76 bytes only), the header entries in the second
// package-merge list (idle once the last code is built).
constexpr int kPmLevels = 14;  // package-merge levels 2 .. 15 keep a leaf bitmap
struct Shared {
  uint32_t lit_freq[288];
  uint32_t off_freq[32];
  uint32_t cg_freq[20];
  uint32_t lit_cl[288];  // (len << 16) | bit-reversed code
  uint32_t off_cl[32];
  uint32_t cg_cl[20];
  // code construction scratch
  uint32_t key[288];    // sort keys (freq << 9 | symbol) of the symbols in use
  alignas(8) uint32_t lv[2][576];  // package-merge level lists (ping-pong)
#ifdef FLATE_HUFF_PAIRS_OWN  // (A/B builds: the layout of rounds 3-4, 12088 B = 10 LDS granules of 1280 B = 12 wavefronts per CU)
  uint32_t pairs_mem[288];
  FLATE_D uint32_t *pairs() { return pairs_mem; }
#else
  // the pair sums of a level live in the sort keys, which are dead once the leaves are sorted (a level of the
  // 19-symbol codegen code has at most 9 pairs: 36 bytes, in front of the codegen items at key + 128): 10936 B =
  // 9 granules = 14 wavefronts per CU
  FLATE_D uint32_t *pairs() { return key; }
#endif
  uint32_t leaf_bits[kPmLevels][18];  // per level: bit r set <=> item r of the merged list is a leaf
  uint8_t slen[288];
  uint32_t cg_n;           // number of codegen items
  uint32_t hdr_n;
  // header
  FLATE_D uint8_t *codegen() { return reinterpret_cast<uint8_t *>(key) + 128; }   // [320] the codegen symbols (0..18), one per item
  FLATE_D uint8_t *cg_extra() { return reinterpret_cast<uint8_t *>(key) + 448; }  // [320] the repeat count carried by a 16 / 17 / 18 item
  FLATE_D uint16_t *hdr_val() { return reinterpret_cast<uint16_t *>(&lv[1][0]); }            // [kHdrMax]
  FLATE_D uint8_t *hdr_nb() { return reinterpret_cast<uint8_t *>(&lv[1][0]) + 2 * kHdrMax; }  // [kHdrMax]
};
static_assert(448 + 320 <= sizeof(uint32_t) * 288, "codegen items fit the sort keys");
static_assert(3 * kHdrMax <= (int)sizeof(uint32_t) * 576, "header entries fit a level list");

// LDS of huff_hist_kernel
struct SharedHist {
  uint32_t lit_freq[288];
  uint32_t off_freq[32];
  // per-tile scatter target, one byte per position (lane L reads its four as one dword): 1 = the
  // first position a match covers (start + 1), 2 = the last one, 4 = a match starts here.  Matches
  // are >= 4 long and do not overlap, so no position ever gets two marks: plain byte stores.
  uint32_t marks[kTile / 4];
  uint8_t len_code[256];   // length - 3 -> length code (token.mbt:30-44), filled once per wavefront
};
FLATE_D void fill_len_code(SharedHist &sh, int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k) sh.len_code[4 * lane + k] = (uint8_t)length_code_of((uint32_t)(4 * lane + k)).code;
  __syncthreads();
}

struct SharedPack {
  uint32_t lit_cl[288];
  uint32_t off_cl[32];
  uint32_t len_bits[256];  // per block, by length - 3: (bits << 24) | length code + extra bits (<= 15 + 5)
  uint32_t ring[kRing];
};

FLATE_D uint32_t rdlane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
// DPP row shifts / row broadcasts (gfx9): wave64 inclusive prefix sum without LDS traffic.
template <int CTRL, int ROW_MASK>
FLATE_D uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
FLATE_D uint32_t wave_incl_scan(uint32_t v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}
FLATE_D uint32_t wave_sum(uint32_t v) { return rdlane(wave_incl_scan(v), 63); }
FLATE_D uint32_t wave_max(uint32_t v) {
  for (int d = 32; d >= 1; d >>= 1) {
    uint32_t o = __shfl_xor(v, d);
    v = o > v ? o : v;
  }
  return v;
}

// ---- bit sink ----------------------------------------------------------------------
// The stream's bytes start at an arbitrary address: out32 is that address rounded down to a
// dword and bitpos starts at 8 * misalignment.  Dwords that contain a byte of a neighbouring
// stream (the first one if misaligned, the last one if the stream ends inside it) are written
// byte by byte, everything else with coalesced dword stores.
struct BitSink {
  uint32_t *ring;    // LDS, zero outside the pending window
  uint32_t *out32;   // 4-byte aligned
  uint64_t bitpos;   // bit position relative to out32
  uint32_t flushed;  // dwords already drained to out32
  uint32_t head;     // bytes of dword 0 that belong to the previous stream (0..3)
  uint32_t last_dw;  // dword holding the stream's last byte
  uint32_t tail;     // bytes of last_dw that belong to this stream (1..4)
  // dword 0 / last_dw shared at BIT granularity with what is written in front of / behind this sink
  // (the neighbouring streams of a spliced batch; the neighbouring blocks of the same stream when
  // every block has its own wavefront): OR-ed atomically into a destination zeroed beforehand
  bool or_first, or_last;
};

FLATE_D void sink_store(const BitSink &S, uint32_t idx, uint32_t v) {
  if ((S.or_first && idx == 0) || (S.or_last && idx == S.last_dw)) {
    if (v) atomicOr(&S.out32[idx], v);
    return;
  }
  const bool first = idx == 0 && S.head != 0;
  const bool last = idx == S.last_dw && S.tail != 4;
  if (!first && !last) {
    S.out32[idx] = v;
    return;
  }
  uint8_t *b = reinterpret_cast<uint8_t *>(S.out32 + idx);
  const uint32_t lo = first ? S.head : 0u, hi = last ? S.tail : 4u;
  for (uint32_t k = lo; k < hi; ++k) b[k] = (uint8_t)(v >> (8 * k));
}

// Every lane appends nb (<= 48) bits, in lane order.
FLATE_D void sink_emit(BitSink &S, uint64_t bits, uint32_t nb, int lane) {
  const uint32_t incl = wave_incl_scan(nb);
  const uint32_t total = rdlane(incl, 63);
  if (nb) {
    const uint64_t q = S.bitpos + (incl - nb);
    const uint32_t w = (uint32_t)(q >> 5);
    const uint32_t sh = (uint32_t)q & 31u;
    const uint32_t d0 = (uint32_t)bits << sh;
    const uint64_t rest = bits >> (32u - sh);  // sh == 0 -> bits >> 32
    const uint32_t d1 = sh ? (uint32_t)rest : (uint32_t)(bits >> 32);
    const uint32_t d2 = sh ? (uint32_t)(rest >> 32) : 0u;
    if (d0) atomicOr(&S.ring[w & (kRing - 1)], d0);
    if (d1) atomicOr(&S.ring[(w + 1) & (kRing - 1)], d1);
    if (d2) atomicOr(&S.ring[(w + 2) & (kRing - 1)], d2);
  }
  S.bitpos += total;
  __syncthreads();
  while ((uint32_t)(S.bitpos >> 5) - S.flushed >= 64u) {
    const uint32_t idx = S.flushed + (uint32_t)lane;
    sink_store(S, idx, S.ring[idx & (kRing - 1)]);
    S.ring[idx & (kRing - 1)] = 0;
    S.flushed += 64;
  }
  __syncthreads();
}

// Wide form: every lane appends nb (<= 96) bits held in (lo, hi), in lane order.
FLATE_D void sink_emit_wide(BitSink &S, uint64_t lo, uint32_t hi, uint32_t nb, int lane) {
  const uint32_t incl = wave_incl_scan(nb);
  const uint32_t total = rdlane(incl, 63);
  if (nb) {
    const uint64_t q = S.bitpos + (incl - nb);
    const uint32_t w = (uint32_t)(q >> 5);
    const uint32_t sh = (uint32_t)q & 31u;
    // the 96 bits shifted left by sh < 32 into four dwords; (x >> 1) >> (31 - sh) is x >> (32 - sh)
    // without the special case sh == 0
    const uint64_t v0 = lo << sh;
    const uint32_t d0 = (uint32_t)v0, d1 = (uint32_t)(v0 >> 32);
    const uint32_t d2 = (hi << sh) | (((uint32_t)(lo >> 32) >> 1) >> (31u - sh));
    const uint32_t d3 = (hi >> 1) >> (31u - sh);
    if (d0) atomicOr(&S.ring[w & (kRing - 1)], d0);
    if (d1) atomicOr(&S.ring[(w + 1) & (kRing - 1)], d1);
    if (d2) atomicOr(&S.ring[(w + 2) & (kRing - 1)], d2);
    if (d3) atomicOr(&S.ring[(w + 3) & (kRing - 1)], d3);
  }
  S.bitpos += total;
  __syncthreads();
  while ((uint32_t)(S.bitpos >> 5) - S.flushed >= 64u) {
    const uint32_t idx = S.flushed + (uint32_t)lane;
    sink_store(S, idx, S.ring[idx & (kRing - 1)]);
    S.ring[idx & (kRing - 1)] = 0;
    S.flushed += 64;
  }
  __syncthreads();
}

// flush(): pad with zero bits to a byte boundary (huffman-bit-writer.mbt:139-158)
FLATE_D void sink_pad_to_byte(BitSink &S) { S.bitpos = (S.bitpos + 7) & ~7ull; }

// drain everything (stream end; bitpos is byte aligned)
FLATE_D void sink_finish(BitSink &S, int lane) {
  const uint32_t end = (uint32_t)((S.bitpos + 31) >> 5);
  while (S.flushed < end) {
    const uint32_t idx = S.flushed + (uint32_t)lane;
    if (idx < end) {
      sink_store(S, idx, S.ring[idx & (kRing - 1)]);
      S.ring[idx & (kRing - 1)] = 0;
    }
    S.flushed += 64;
  }
  __syncthreads();
}

// ---- canonical length-limited Huffman code (huffman-code.mbt:295-343) ----------------
// freq[0..nsym) -> cl[i] = (len << 16) | reversed code; len = 0 for absent symbols.
//
// The kernel is bound by the length of one block's dependent chain of LDS round trips (15 KiB of
// scratch allow few wavefronts per CU), so everything that is a chain of dependent LDS reads in
// the textbook form is kept short here: the binary searches of a level are branch-free with a
// wave-uniform trip count, all of a lane's searches (its <= NT leaves and <= NT pairs) advance
// together -- one round trip per halving, not one per search and halving -- and the per-level
// counts, the canonical first codes and the length classes live in registers (one level per lane)
// instead of LDS arrays walked by lane 0.
//
// build_code_sorted<NT>: the n > 2 symbols in use are in sh.key[0..n) as (freq << 9 | symbol);
// a lane owns items lane, lane + 64, ... (NT >= ceil(n / 64) of them).
template <int NT>
FLATE_D void build_code_sorted(Shared &sh, int n, int nsym, int max_bits, uint32_t *cl, int lane) {
  // The leaves in ascending order live in the OUTPUT array until the codes are written: cl[] is not read before
  // the last loop below, which writes every entry (10936 -> 9784 B of LDS = eight granules of 1280 B = 16
  // wavefronts per CU, what the kernel's 120 VGPRs allow: 16384 streams are four full rounds instead of 4.6).
  uint32_t *sfreq = cl;
  // rank sort (keys are distinct): item j of the unsorted list is leaf rank[j] of the sorted one
  uint32_t mine[NT];
  int rank[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = lane + 64 * t;
    mine[t] = j < n ? sh.key[j] : 0xffffffffu;
    rank[t] = 0;
  }
  for (int k = 0; k < n; ++k) {
    const uint32_t kk = sh.key[k];
#pragma unroll
    for (int t = 0; t < NT; ++t) rank[t] += kk < mine[t];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (lane + 64 * t < n) {
      sfreq[rank[t]] = mine[t] >> 9;      // leaves ascending by (freq, symbol), by_frequency :346
      sh.lv[0][rank[t]] = mine[t] >> 9;   // level 1: the leaves themselves
    }
  for (int i = lane; i < kPmLevels * 18; i += 64) (&sh.leaf_bits[0][0])[i] = 0;
  __syncthreads();

  const int mb = max_bits < n - 1 ? max_bits : n - 1;  // :126-129
  uint32_t lf[NT];  // my leaves of the SORTED list: i = lane + 64 t
#pragma unroll
  for (int t = 0; t < NT; ++t) lf[t] = lane + 64 * t < n ? sfreq[lane + 64 * t] : 0u;
  const int s0 = 1 << (31 - __builtin_clz(n));  // first step of a search over <= n elements

  int lp = n;  // length of the previous level's list
  int cur = 0;
  for (int lvl = 2; lvl <= mb; ++lvl) {
    const uint32_t *prev = sh.lv[cur];
    uint32_t *next = sh.lv[cur ^ 1];
    const int np = lp >> 1;  // >= 1, < n
    uint32_t ps[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = lane + 64 * t;
      const uint2 two = j < np ? *reinterpret_cast<const uint2 *>(prev + 2 * j) : make_uint2(0u, 0u);
      ps[t] = two.x + two.y;
      if (j < np) sh.pairs()[j] = ps[t];
    }
    __syncthreads();
    // leaf i goes to i + #pairs with sum <= leaf (a pair wins a tie, :187);
    // pair j goes to j + #leaves with freq < sum.  Both lists are ascending: the count is built
    // bit by bit from the top (element c + s - 1 satisfies the test <=> the count is >= c + s).
    int cl_[NT], cp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cl_[t] = cp[t] = 0;
    for (int s = s0; s >= 1; s >>= 1) {
      uint32_t pv[NT], lv_[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int a = cl_[t] + s, c = cp[t] + s;
        pv[t] = sh.pairs()[(a <= np ? a : np) - 1];
        lv_[t] = sfreq[(c <= n ? c : n) - 1];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int a = cl_[t] + s, c = cp[t] + s;
        cl_[t] = (a <= np && pv[t] <= lf[t]) ? a : cl_[t];
        cp[t] = (c <= n && lv_[t] < ps[t]) ? c : cp[t];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = lane + 64 * t;
      if (i < n) {
        const int r = i + cl_[t];
        next[r] = lf[t];
        atomicOr(&sh.leaf_bits[lvl - 2][r >> 5], 1u << (r & 31));
      }
      if (i < np) next[i + cp[t]] = ps[t];
    }
    __syncthreads();
    lp = n + np;
    cur ^= 1;
  }

  // top-down: how many leaves sit in the needed prefix of each level (:168, :234-243).
  // Lane w < 18 holds word w of every level's leaf bitmap; lane L collects counts[L].
  uint32_t lb[kPmLevels];
#pragma unroll
  for (int k = 0; k < kPmLevels; ++k) lb[k] = (k + 2 <= mb && lane < 18) ? sh.leaf_bits[k][lane] : 0u;
  uint32_t cnt_v = 0;  // lane L: counts[L] (counts[0] = 0)
  {
    uint32_t m = 2u * (uint32_t)n - 2u;
#pragma unroll
    for (int lvl = kPmLevels + 1; lvl >= 2; --lvl)
      if (lvl <= mb) {
        const int lo = lane * 32;
        const uint32_t keep = (int)m >= lo + 32 ? 0xffffffffu : ((int)m <= lo ? 0u : ((1u << (m - lo)) - 1u));
        const uint32_t a = wave_sum((uint32_t)__popc(lb[lvl - 2] & keep));
        if (lane == lvl) cnt_v = a;
        m = 2u * (m - a);
      }
    const uint32_t a1 = m < (uint32_t)n ? m : (uint32_t)n;  // level 1 holds only leaves
    if (lane == 1) cnt_v = a1;
  }
  // bit_count[b] = counts[mb-b+1] - counts[mb-b] in lane b; len_base[b] = symbols with length <= b,
  // counted from the most frequent; canonical first codes (:250-280)
  const bool is_len = lane >= 1 && lane <= mb;
  const uint32_t c_hi = __shfl(cnt_v, is_len ? mb - lane + 1 : 0);
  const uint32_t c_lo = __shfl(cnt_v, is_len ? mb - lane : 0);
  const uint32_t bc_v = is_len ? c_hi - c_lo : 0u;
  const uint32_t len_base_v = wave_incl_scan(bc_v);
  uint32_t first_code[16];  // (wave-uniform)
  {
    uint32_t code = 0;
    first_code[0] = 0;
#pragma unroll
    for (int b = 1; b < 16; ++b) {
      code <<= 1;
      first_code[b] = code;
      code += rdlane(bc_v, b);  // 0 beyond mb
    }
  }
  // the len_base[b] - len_base[b-1] most frequent leaves not yet served get length b
  {
    int bl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bl[t] = 1;
#pragma unroll
    for (int k = 1; k < 15; ++k)
      if (k < mb) {
        const uint32_t lbk = rdlane(len_base_v, k);
#pragma unroll
        for (int t = 0; t < NT; ++t) bl[t] += (uint32_t)(n - 1 - rank[t]) >= lbk;  // 0 = most frequent
      }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (lane + 64 * t < n) sh.slen[mine[t] & 511u] = (uint8_t)bl[t];
  }
  __syncthreads();
  // codes in symbol order within each length
  uint32_t running[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) running[b] = 0;
  for (int base = 0; base < nsym; base += 64) {
    const int i = base + lane;
    const uint32_t L = i < nsym ? sh.slen[i] : 0u;
    uint32_t code = 0;
#pragma unroll
    for (int b = 1; b < 16; ++b) {
      const uint64_t m = __ballot(L == (uint32_t)b);
      if (L == (uint32_t)b) code = first_code[b] + running[b] + __popcll(m & ((1ull << lane) - 1));
      running[b] += __popcll(m);
    }
    if (i < nsym) cl[i] = L ? ((L << 16) | (__brev(code) >> (32 - L))) : 0u;  // (every entry: cl held the sorted leaves)
  }
  __syncthreads();
}

// MAXT = ceil(nsym / 64)
template <int MAXT>
FLATE_D void build_code(Shared &sh, const uint32_t *freq, int nsym, int max_bits, uint32_t *cl,
                        int lane) {
  // compact the symbols with non-zero frequency, in symbol order
  int n = 0;
  for (int base = 0; base < nsym; base += 64) {
    const int i = base + lane;
    const uint32_t f = i < nsym ? freq[i] : 0u;
    const uint64_t m = __ballot(f != 0);
    if (f != 0) {
      const int r = n + __popcll(m & ((1ull << lane) - 1));
      sh.key[r] = (f << 9) | (uint32_t)i;  // (freq, symbol) order, by_frequency :346
    }
    if (i < nsym) {
      sh.slen[i] = 0;
      cl[i] = 0;
    }
    n += __popcll(m);
  }
  __syncthreads();
  if (n <= 2) {  // :326-336: lengths 1, codes 0/1 in symbol order
    if (lane < n) cl[sh.key[lane] & 511u] = (1u << 16) | (uint32_t)lane;
    __syncthreads();
    return;
  }
  if constexpr (MAXT == 1) {
    build_code_sorted<1>(sh, n, nsym, max_bits, cl, lane);
  } else {
    if (n <= 128)
      build_code_sorted<2>(sh, n, nsym, max_bits, cl, lane);
    else if (n <= 192)
      build_code_sorted<3>(sh, n, nsym, max_bits, cl, lane);
    else
      build_code_sorted<MAXT>(sh, n, nsym, max_bits, cl, lane);
  }
}

// ---- tile walk over the implied token sequence ----------------------------------------
// A tile is 256 consecutive input positions; lane L owns positions P0 + 4L .. P0 + 4L + 3.
// At most one match can start inside a lane's four positions (matches are >= 4 long).
struct TileTok {
  uint32_t bytes;  // the lane's four input bytes (little endian)
  uint32_t lit;    // byte k = 1: position k is a literal
  uint32_t start;  // byte k = 1: a match starts at position k (at most one byte set)
  uint32_t tok;    // its token
  // What huff_pack_kernel needs to know about a lane's four positions, in one byte: the walk is
  // done once, by huff_hist_kernel, and handed over through HuffParams::tile_meta (64 bytes per
  // tile): bits 0..3 = literal at position k, bit 4 = a match starts in the lane, bits 5..6 = at
  // which position.  The j-th lane of a tile with bit 4 set starts the j-th match record of that
  // tile.  Positions after a match start are covered (matches are >= 4 long), so the literals of a
  // lane all precede its match.  Two byte dot products (v_dot4_u32_u8) build it from the flags.
  FLATE_D uint32_t pack() const {
    return __builtin_amdgcn_udot4(lit, 0x08040201u, __builtin_amdgcn_udot4(start, 0x70503010u, 0u, false), false);
  }
};
// Where a block's tile rows (64 bytes per 256-position tile) start in HuffParams::tile_meta: by
// the block's position in the input plus its index, so that the whole array is input / 4 bytes
// plus one row per block (a block of n bytes has ceil(n / 256) <= n / 256 + 1 rows).
FLATE_D uint64_t tile_meta_at(uint64_t block_start_in_input, uint32_t gb) {
  return ((block_start_in_input >> 8) + gb) * 64u;
}

struct Walker {
  const uint8_t *src;
  const uint2 *recs;
  uint32_t nm;
  int n;
  uint32_t mp;         // next match record
  uint32_t cov_until;  // positions < cov_until are covered by an earlier match
  // software pipeline: data of the tile about to be processed, loaded one tile ahead
  uint2 rec;           // recs[mp + lane]  (<= 64 matches start in a tile)
  uint32_t bytes;      // src[P0 + 4 lane .. +3]
};

FLATE_D uint2 load_rec(const Walker &w, uint32_t mp, int lane) {
  uint2 r = make_uint2(0xffffffffu, 0);
  if (mp + (uint32_t)lane < w.nm) r = w.recs[mp + lane];
  return r;
}
FLATE_D uint32_t load_bytes4(const Walker &w, int pos) {
#ifdef FLATE_EXP_NO_INPUT  // TIMING EXPERIMENT ONLY (wrong bytes on purpose): the entropy kernels without their reads of the input
  return 0x20746165u + 0x01010101u * (((uint32_t)pos * 2654435761u) >> 30);
#endif
  if (pos + 4 <= w.n) {
    uint32_t v;
    __builtin_memcpy(&v, w.src + pos, 4);
    return v;
  }
  uint32_t v = 0;  // chunk tail: never read past the chunk
  for (int k = 0; k < 4; ++k)
    if (pos + k < w.n) v |= (uint32_t)w.src[pos + k] << (8 * k);
  return v;
}

FLATE_D Walker walker_init(const uint8_t *src, const uint2 *recs, uint32_t nm, int n, int lane) {
  Walker w;
  w.src = src;
  w.recs = recs;
  w.nm = nm;
  w.n = n;
  w.mp = 0;
  w.cov_until = 0;
  w.rec = load_rec(w, 0, lane);
  w.bytes = load_bytes4(w, 4 * lane);
  return w;
}

FLATE_D int clamp04(int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }  // (v_med3_i32)
// Byte-parallel form: the marks of a lane's four positions are the four bytes of a dword, so a
// multiply by 0x01010101 is their inclusive prefix sum (sums <= 4: no carry between bytes) and the
// classification of the four positions is straight-line dword arithmetic.  (v_mul_lo_u32 is a
// quarter-rate instruction: the sums the walk needs come from v_sad_u8, the prefix sums from two
// shift-adds, the packed result from v_dot4_u32_u8.)
FLATE_D TileTok walk_tile(SharedHist &sh, Walker &w, int P0, int lane) {
  TileTok t;
  const int pos = P0 + 4 * lane;
  const uint2 rec = w.rec;
  t.bytes = w.bytes;
  const bool mine = rec.x < (uint32_t)(P0 + kTile);
  const int cnt = __popcll(__ballot(mine));
  // issue the next tile's loads now; they are consumed one iteration later
  w.mp += (uint32_t)cnt;
  w.rec = load_rec(w, w.mp, lane);
  w.bytes = load_bytes4(w, pos + kTile);
  sh.marks[lane] = 0;
  __syncthreads();
  const uint32_t mlen = ((rec.y >> kLengthShift) & 0xffu) + 3u;
  if (mine) {
    uint8_t *mb = reinterpret_cast<uint8_t *>(sh.marks);
    const uint32_t o = rec.x - (uint32_t)P0;
    mb[o] = 4;
    const uint32_t o1 = o + 1u, o2 = o + mlen - 1u;
    if (o1 < (uint32_t)kTile) mb[o1] = 1;
    if (o2 < (uint32_t)kTile) mb[o2] = 2;
  }
  __syncthreads();
  const uint32_t m = sh.marks[lane];
  const uint32_t first = m & 0x01010101u, last = (m >> 1) & 0x01010101u;
  // net coverage change of this lane: marks are bytes 0 / 1, v_sad_u8 against 0 sums four of them
  const uint32_t tot = __builtin_amdgcn_sad_u8(first, 0u, 0u) - __builtin_amdgcn_sad_u8(last, 0u, 0u);
  const uint32_t base = wave_incl_scan(tot) - tot;  // coverage entering this lane: 0 or 1
  // positions below cov_until are covered by a match of an earlier tile: the t lowest bytes
  const int t_lo = clamp04((int)w.cov_until - pos);
  const uint32_t low = (uint32_t)((0x01010101ull << (8 * t_lo)) >> 32);
  if (cnt) w.cov_until = rdlane(rec.x, cnt - 1) + rdlane(mlen, cnt - 1);
  // coverage of position k = coverage entering the lane + firsts up to k - lasts before k: per-byte
  // inclusive prefix sums of first - (last << 8), x * 0x01010101 as two shift-adds (the borrows
  // between the bytes cancel in the sums; the coverage itself is 0 or 1 in every byte).
  // (inline asm: written in C the compiler folds the two steps back into a quarter-rate multiply)
  const uint32_t e = first - (last << 8) + base;
  uint32_t e2, e4;
  asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(e2) : "v"(e));
  asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(e4) : "v"(e2));
  const uint32_t cov = e4 | low;
  const int t_act = clamp04(w.n - pos);  // my positions inside the chunk
  const uint32_t act = (uint32_t)((0x01010101ull << (8 * t_act)) >> 32);
  t.start = (m >> 2) & act;
  t.lit = act & ~(cov | t.start);
  // token of the match that starts here: the j-th lane with a start owns the j-th record of the tile
  // (records are sorted, a lane holds at most one start), fetched from the lane that loaded it
  const uint64_t sb = __ballot(t.start != 0);
  const uint32_t tk = (uint32_t)__shfl((int)rec.y, (int)__popcll(sb & ((1ull << lane) - 1ull)));
  t.tok = t.start ? tk : 0u;
  return t;
}

// stored block: header, pad, LEN, ~LEN, raw bytes (huffman-bit-writer.mbt:474-487,202-225)
FLATE_D void emit_stored(BitSink &S, const uint8_t *src, int n, bool eof, int lane) {
  sink_emit(S, eof ? 1u : 0u, lane == 0 ? 3u : 0u, lane);
  sink_pad_to_byte(S);
  const uint32_t lenw = (uint32_t)n | (((~(uint32_t)n) & 0xffffu) << 16);
  sink_emit(S, lenw, lane == 0 ? 32u : 0u, lane);
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    sink_emit(S, i < n ? src[i] : 0u, i < n ? 8u : 0u, lane);
  }
}

// generate_codegen (:241-330) + codegen code + dynamic_size (:335-360) + header items.
// Returns size in bits (for the stored decision) and leaves the header in hdr_*.
// Wave-parallel: the reference's loops over the ~300 code lengths are (1) runs of equal lengths
// found with ballots, (2) per run the closed form of the greedy loops of :262-327 -- a non-zero
// length v repeated c times is v, then (c-1)/6 times "16, 3", then "16, rem-3" if the remainder
// is >= 3 or the remainder as plain v's; a zero run is c/138 times "18, 127", then "18, rem-11"
// if rem >= 11, else "17, rem-3" if rem >= 3, else rem plain zeros -- written at the offset a
// prefix sum over the runs gives, (3) after the codegen code is built, one header entry per item
// plus one per repeat count, again placed by a prefix sum.  (A lane-0 loop with two dependent LDS
// reads per element cost about half of huff_code_kernel's time.)
FLATE_D uint32_t make_header(Shared &sh, int num_literals, int num_offsets, int lane) {
  const int total = num_literals + num_offsets;  // <= 316
  auto len_at = [&](int i) -> uint32_t {
    return i < num_literals ? (sh.lit_cl[i] >> 16) : (sh.off_cl[i - num_literals] >> 16);
  };
  if (lane < 20) sh.cg_freq[lane] = 0;
  // (1) run starts, in order (+ the end).  The list lives in the package-merge scratch, which is
  // idle between two code constructions (one more KiB of LDS would cost a wavefront per CU).
  uint16_t *run_pos = reinterpret_cast<uint16_t *>(&sh.lv[0][0]);
  uint32_t nruns = 0;
  for (int t = 0; t < 5; ++t) {
    const int i = t * 64 + lane;
    const bool start = i < total && (i == 0 || len_at(i) != len_at(i - 1));
    const uint64_t m = __ballot(start);
    if (start) run_pos[nruns + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
    nruns += (uint32_t)__popcll(m);
  }
  if (lane == 0) run_pos[nruns] = (uint16_t)total;
  __syncthreads();
  // (2) the items of every run
  uint32_t nitems = 0;
  for (uint32_t r0 = 0; r0 < nruns; r0 += 64) {
    const uint32_t r = r0 + (uint32_t)lane;
    uint32_t v = 0, c = 0, k_big = 0, tail_sym = 0, tail_extra = 0, plain = 0;
    bool has_tail = false;
    if (r < nruns) {
      const int p = run_pos[r];
      c = (uint32_t)run_pos[r + 1] - (uint32_t)p;
      v = len_at(p);
      if (v != 0) {
        const uint32_t rem = c - 1u;
        k_big = rem / 6u;  // "16, 3" (repeat 6)
        const uint32_t lo = rem - 6u * k_big;
        has_tail = lo >= 3u;
        tail_sym = 16u;
        tail_extra = lo - 3u;
        plain = 1u + (has_tail ? 0u : lo);
      } else {
        k_big = c / 138u;  // "18, 127" (repeat 138)
        uint32_t lo = c - 138u * k_big;
        if (lo >= 11u) {
          has_tail = true;
          tail_sym = 18u;
          tail_extra = lo - 11u;
          lo = 0;
        } else if (lo >= 3u) {
          has_tail = true;
          tail_sym = 17u;
          tail_extra = lo - 3u;
          lo = 0;
        }
        plain = lo;
      }
    }
    const uint32_t mine = r < nruns ? plain + k_big + (has_tail ? 1u : 0u) : 0u;
    const uint32_t incl = wave_incl_scan(mine);
    uint32_t at = nitems + incl - mine;
    nitems += rdlane(incl, 63);
    if (r < nruns) {
      const uint32_t big_sym = v != 0 ? 16u : 18u, big_extra = v != 0 ? 3u : 127u;
      if (v != 0) {  // the first occurrence is always written plainly (:276-279)
        sh.codegen()[at] = (uint8_t)v;
        sh.cg_extra()[at++] = 0;
      }
      for (uint32_t k = 0; k < k_big; ++k) {
        sh.codegen()[at] = (uint8_t)big_sym;
        sh.cg_extra()[at++] = (uint8_t)big_extra;
      }
      if (has_tail) {
        sh.codegen()[at] = (uint8_t)tail_sym;
        sh.cg_extra()[at++] = (uint8_t)tail_extra;
      }
      for (uint32_t k = v != 0 ? 1u : 0u; k < plain; ++k) {
        sh.codegen()[at] = (uint8_t)v;
        sh.cg_extra()[at++] = 0;
      }
      if (plain) atomicAdd(&sh.cg_freq[v], plain);
      const uint32_t n_big = k_big + ((has_tail && tail_sym == big_sym) ? 1u : 0u);
      if (n_big) atomicAdd(&sh.cg_freq[big_sym], n_big);
      if (has_tail && tail_sym == 17u) atomicAdd(&sh.cg_freq[17], 1u);
    }
  }
  if (lane == 0) sh.cg_n = nitems;
  __syncthreads();
  build_code<1>(sh, sh.cg_freq, kCodegenCodeCount, 7, sh.cg_cl, lane);

  // (3) header entries: HLIT, HDIST, HCLEN, the code lengths of the codegen code in codegen_order
  // (trailing zeros dropped, at least four), then the items (:421-471)
  const bool in_order = lane < kCodegenCodeCount;
  const uint32_t ord_freq = in_order ? sh.cg_freq[kCodegenOrder[lane]] : 0u;
  const uint64_t nz = __ballot(ord_freq != 0);
  int ncg = nz ? 64 - (int)__builtin_clzll(nz) : 0;
  if (ncg < 4) ncg = 4;
  uint32_t term = 0;
  if (in_order) {
    const uint32_t f = sh.cg_freq[lane];
    term = f * (sh.cg_cl[lane] >> 16) + (lane == 16 ? 2u * f : (lane == 17 ? 3u * f : (lane == 18 ? 7u * f : 0u)));
  }
  const uint32_t size = 3u + 5u + 5u + 4u + 3u * (uint32_t)ncg + wave_sum(term);
  if (lane == 0) {
    sh.hdr_val()[0] = 4;  // BFINAL=0, BTYPE=10 (callers never pass eof, deflate.mbt:251,267,269)
    sh.hdr_nb()[0] = 3;
    sh.hdr_val()[1] = (uint16_t)(num_literals - 257);
    sh.hdr_nb()[1] = 5;
    sh.hdr_val()[2] = (uint16_t)(num_offsets - 1);
    sh.hdr_nb()[2] = 5;
    sh.hdr_val()[3] = (uint16_t)(ncg - 4);
    sh.hdr_nb()[3] = 4;
  }
  if (lane < ncg) {
    sh.hdr_val()[4 + lane] = (uint16_t)(sh.cg_cl[kCodegenOrder[lane]] >> 16);
    sh.hdr_nb()[4 + lane] = 3;
  }
  uint32_t h = 4u + (uint32_t)ncg;
  for (uint32_t k0 = 0; k0 < nitems; k0 += 64) {
    const uint32_t k = k0 + (uint32_t)lane;
    const bool live = k < nitems;
    const uint32_t cw = live ? sh.codegen()[k] : 0u;
    const uint32_t ents = live ? (cw >= 16u ? 2u : 1u) : 0u;
    const uint32_t incl = wave_incl_scan(ents);
    const uint32_t at = h + incl - ents;
    h += rdlane(incl, 63);
    if (live) {
      const uint32_t c = sh.cg_cl[cw];
      sh.hdr_val()[at] = (uint16_t)(c & 0xffffu);
      sh.hdr_nb()[at] = (uint8_t)(c >> 16);
      if (cw >= 16u) {
        sh.hdr_val()[at + 1] = sh.cg_extra()[k];
        sh.hdr_nb()[at + 1] = cw == 16u ? 2 : (cw == 17u ? 3 : 7);
      }
    }
  }
  if (lane == 0) sh.hdr_n = h;
  __syncthreads();
  return size;
}

FLATE_D bool prefer_stored(uint32_t compat_go, int n, uint32_t size) {
  // stored_size (:375-384): n in 1..65535 is always storable here.
  const uint32_t ssize = ((uint32_t)n + 5u) * 8u;
  if (compat_go) return ssize < size + (size >> 4);  // Go 1.23.1
  return ssize < ((size + size) >> 4);               // huffman-bit-writer.mbt:527,780
}

constexpr int kBlkStride = 320;  // u32 per block in blk_hist / blk_cl: 288 lit + 32 off

struct BlockGeom {
  const uint8_t *stream;
  uint64_t len, full;
  int r;
  uint32_t nblocks, blk0, chunk0;
};
FLATE_D BlockGeom block_geom(const HuffParams &P, uint32_t sid) {
  BlockGeom g;
  const uint64_t a = P.in_off[sid];
  g.len = P.in_off[sid + 1] - a;
  g.stream = P.in + a;
  g.full = g.len / (uint64_t)kMaxStoreBlockSize;
  g.r = (int)(g.len % (uint64_t)kMaxStoreBlockSize);
  g.nblocks = (uint32_t)g.full + (g.r > 0 ? 1u : 0u);
  g.blk0 = P.blk_base[sid];
  g.chunk0 = P.chunk_base[sid];
  return g;
}

// extra bits carried by length code c / offset code c (huffman-bit-writer.mbt:49-54,67-70)
FLATE_D uint32_t len_extra_of_code(uint32_t c) { return (c < 8 || c >= 28) ? 0u : (c >> 2) - 1u; }
FLATE_D uint32_t off_extra_of_code(uint32_t c) { return c < 4 ? 0u : (c >> 1) - 1u; }

}  // namespace

// ---------------------------------------------------------------------------------------
// index_tokens (:550-593) / histogram (:831): per-block symbol histograms and the enc_speed
// block policy (deflate.mbt:243-269): kind 0 = stored (<= 16 B tail), 1 = Huffman-only,
// 2 = dynamic.
// ---------------------------------------------------------------------------------------
// index_tokens / histogram of ONE block (and the enc_speed policy that picks its kind).
FLATE_D void hist_block(const HuffParams &P, SharedHist &sh, const BlockGeom &g, uint32_t b, int lane) {
    const uint8_t *src = g.stream + (uint64_t)b * kMaxStoreBlockSize;
    const int n = b < g.full ? kMaxStoreBlockSize : g.r;
    const uint32_t gb = g.blk0 + b;
    int kind;
    if (n < kSmallLzMin) {
      kind = n <= 16 ? 0 : 1;
    } else {
      kind = P.chunk_ntok[g.chunk0 + b] > (uint32_t)(n - (n >> 4)) ? 1 : 2;  // deflate.mbt:266
    }
    if (lane == 0) P.blk_meta[gb] = make_uint4((uint32_t)kind, 0u, 0u, 0u);
    if (kind == 0) return;
    for (int i = lane; i < 288; i += 64) sh.lit_freq[i] = 0;
    if (lane < 32) sh.off_freq[lane] = 0;
    // (four copies of the histograms at an odd stride, one per 16 lanes, were slower: 0.91 vs 0.84 ms)
    uint32_t *lit_freq = sh.lit_freq, *off_freq = sh.off_freq;
    __syncthreads();
    if (kind == 1) {
      Walker w = walker_init(src, nullptr, 0u, n, lane);
      for (int base = 0; base < n; base += kTile) {
        const int i = base + 4 * lane;
        const uint32_t bt = w.bytes;
        w.bytes = load_bytes4(w, i + kTile);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i + k < n) atomicAdd(&lit_freq[(bt >> (8 * k)) & 0xffu], 1u);
      }
      __syncthreads();
      if (lane == 0) {
        sh.lit_freq[kEndBlockMarker] = 1;
        sh.off_freq[0] = 1;
      }
    } else {
      const uint32_t chunk = g.chunk0 + b;
      Walker w = walker_init(src, P.matches + (uint64_t)chunk * kMatchCapPerChunk, P.chunk_nmatch[chunk],
                             n, lane);
      uint8_t *tmeta = P.tile_meta + tile_meta_at((uint64_t)(src - P.in), gb);
      for (int P0 = 0; P0 < n; P0 += kTile) {
        const TileTok t = walk_tile(sh, w, P0, lane);
        tmeta[(P0 >> 2) + lane] = (uint8_t)t.pack();
        if (t.start) {
          const uint32_t lcode = sh.len_code[(t.tok >> kLengthShift) & 0xffu];
          const CodeBits oc = offset_code_of(t.tok & ((1u << kLengthShift) - 1u));
          atomicAdd(&lit_freq[kLengthCodesStart + lcode], 1u);
          atomicAdd(&off_freq[oc.code], 1u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((t.lit >> (8 * k)) & 1u) atomicAdd(&lit_freq[(t.bytes >> (8 * k)) & 0xffu], 1u);
      }
      if (lane == 0) atomicAdd(&lit_freq[kEndBlockMarker], 1u);  // tokens.push(EOB), :507
    }
    __syncthreads();
    uint32_t *h = P.blk_hist + (uint64_t)gb * kBlkStride;
    for (int i = lane; i < 288; i += 64) h[i] = sh.lit_freq[i];
    if (lane < 32) h[288 + lane] = sh.off_freq[lane];
    __syncthreads();
  }

__global__ __launch_bounds__(64) void huff_hist_kernel(HuffParams P) {
  __shared__ SharedHist sh;
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= P.n_streams) return;
  const BlockGeom g = block_geom(P, sid);
  fill_len_code(sh, lane);
  for (uint32_t b = 0; b < g.nblocks; ++b) hist_block(P, sh, g, b, lane);
}

// The same with one wavefront per BLOCK (multi-window streams: a batch of few long streams has few
// wavefronts per stream-kernel; blk_sid maps a global block index to its stream).
__global__ __launch_bounds__(64) void huff_hist_block_kernel(HuffParams P) {
  __shared__ SharedHist sh;
  const int lane = threadIdx.x;
  const uint32_t gb = blockIdx.x;
  const uint32_t sid = P.blk_sid[gb];
  const BlockGeom g = block_geom(P, sid);
  fill_len_code(sh, lane);
  hist_block(P, sh, g, gb - g.blk0, lane);
}

// ---------------------------------------------------------------------------------------
// Per block: Huffman codes, header items, stored-vs-Huffman decision and the exact bit size,
// hence every block's start bit inside its stream and the stream's byte length.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void huff_code_kernel(HuffParams P) {
  __shared__ Shared sh;
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= P.n_streams) return;
  const BlockGeom g = block_geom(P, sid);
  uint64_t bitpos = 0;
  // spliced mode (splice_kernels.hip): x -> x + sum_a, or x -> align8(x + sum_a) + sum_b once a
  // stored block has been seen
  uint64_t sum_a = 0, sum_b = ~0ull;
  for (uint32_t b = 0; b < g.nblocks; ++b) {
    const int n = b < g.full ? kMaxStoreBlockSize : g.r;
    const uint32_t gb = g.blk0 + b;
    int kind = (int)P.blk_meta[gb].x;
    uint32_t hdr_bits = 0;
    uint64_t bits = 0;
    if (kind != 0) {
      const uint32_t *h = P.blk_hist + (uint64_t)gb * kBlkStride;
      for (int i = lane; i < 288; i += 64) sh.lit_freq[i] = h[i];
      if (lane < 32) sh.off_freq[lane] = h[288 + lane];
      __syncthreads();
      int num_literals, num_offsets;
      build_code<5>(sh, sh.lit_freq, kMaxNumLit, 15, sh.lit_cl, lane);
      if (kind == 1) {  // write_block_huff (:738-776): huff_offset = code 0 of length 1
        num_literals = kEndBlockMarker + 1;
        num_offsets = 1;
        if (lane < 32) sh.off_cl[lane] = lane == 0 ? (1u << 16) : 0u;
        __syncthreads();
      } else {  // index_tokens tail (:575-591)
        uint32_t hi = 0;
        for (int i = lane; i < kMaxNumLit; i += 64)
          if (sh.lit_freq[i]) hi = (uint32_t)i + 1;
        num_literals = (int)wave_max(hi);
        const uint32_t ho = (lane < kOffsetCodeCount && sh.off_freq[lane]) ? (uint32_t)lane + 1 : 0u;
        num_offsets = (int)wave_max(ho);
        if (num_offsets == 0) {
          if (lane == 0) sh.off_freq[0] = 1;
          num_offsets = 1;
        }
        __syncthreads();
        build_code<1>(sh, sh.off_freq, kOffsetCodeCount, 15, sh.off_cl, lane);
      }
      hdr_bits = make_header(sh, num_literals, num_offsets, lane);
      uint32_t part = 0, extra = 0;
      for (int i = lane; i < kMaxNumLit; i += 64) {
        part += sh.lit_freq[i] * (sh.lit_cl[i] >> 16);
        if (i >= kLengthCodesStart) extra += sh.lit_freq[i] * len_extra_of_code((uint32_t)(i - kLengthCodesStart));
      }
      if (lane < kOffsetCodeCount) {
        part += sh.off_freq[lane] * (sh.off_cl[lane] >> 16);
        extra += sh.off_freq[lane] * off_extra_of_code((uint32_t)lane);
      }
      if (kind == 1) extra = 0;
      const uint32_t payload = wave_sum(part);           // what dynamic_size counts (:355-358)
      const uint32_t extra_bits = wave_sum(extra);        // ... and what it leaves out (:519-523)
      const uint32_t size = hdr_bits + payload;
      // In Huffman-only blocks the (unused) offset code still counts 1 bit in `size` but no
      // offset symbol is ever written.
      const uint32_t written = kind == 1 ? size - 1u : size + extra_bits;
      if (prefer_stored(P.compat_go, n, size)) {
        kind = 0;
      } else {
        bits = written;
        uint32_t *cl = P.blk_cl + (uint64_t)gb * kBlkStride;
        for (int i = lane; i < 288; i += 64) cl[i] = sh.lit_cl[i];
        if (lane < 32) cl[288 + lane] = sh.off_cl[lane];
        uint32_t *hd = P.blk_hdr + (uint64_t)gb * kHdrMax;
        const int hn = (int)sh.hdr_n;
        for (int i = lane; i < hn; i += 64) hd[i] = ((uint32_t)sh.hdr_nb()[i] << 16) | sh.hdr_val()[i];
      }
    }
    const uint64_t start = bitpos;
    if (kind == 0) {  // write_stored_header + write_bytes (:474-487,202-225)
      bitpos = ((bitpos + 3 + 7) & ~7ull) + 32 + 8ull * (uint64_t)n;
      if (sum_b == ~0ull) {
        sum_a += 3;
        sum_b = 32 + 8ull * (uint64_t)n;
      } else {
        sum_b = ((sum_b + 3 + 7) & ~7ull) + 32 + 8ull * (uint64_t)n;
      }
    } else {
      bitpos += bits;
      if (sum_b == ~0ull) sum_a += bits; else sum_b += bits;
    }
    if (lane == 0)
      P.blk_meta[gb] = make_uint4((uint32_t)kind, kind ? sh.hdr_n : 0u, (uint32_t)start, (uint32_t)(start >> 32));
    __syncthreads();
  }
  // Compressor::close: empty stored block with BFINAL, then flush (deflate.mbt:171-176)
  bitpos = ((bitpos + 3 + 7) & ~7ull) + 32;
  if (lane == 0) {
    P.out_len[sid] = bitpos >> 3;
    if (P.spliced) {
      P.stream_sum[2 * (uint64_t)sid] = sum_a;
      P.stream_sum[2 * (uint64_t)sid + 1] = sum_b;
    }
  }
}

// One block (write_block_dynamic / write_block_huff / a stored block) appended to the sink.
FLATE_D void pack_block(const HuffParams &P, SharedPack &sh, BitSink &S, const BlockGeom &g, uint32_t b, int lane) {
    const uint8_t *src = g.stream + (uint64_t)b * kMaxStoreBlockSize;
    const int n = b < g.full ? kMaxStoreBlockSize : g.r;
    const uint32_t gb = g.blk0 + b;
    const uint4 meta = P.blk_meta[gb];
    const int kind = (int)meta.x;
    if (kind == 0) {
      emit_stored(S, src, n, false, lane);
      return;
    }
    __syncthreads();
    const uint32_t *cl = P.blk_cl + (uint64_t)gb * kBlkStride;
    for (int i = lane; i < 288; i += 64) sh.lit_cl[i] = cl[i];
    if (lane < 32) sh.off_cl[lane] = cl[288 + lane];
    __syncthreads();
    if (kind == 2) {  // everything a match length contributes, by length - 3
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const uint32_t x = 4u * (uint32_t)lane + (uint32_t)k;
        const CodeBits lc = length_code_of(x);
        const uint32_t c1 = sh.lit_cl[kLengthCodesStart + lc.code];
        sh.len_bits[x] = (((c1 >> 16) + lc.nextra) << 24) | (c1 & 0xffffu) | (lc.extra << (c1 >> 16));
      }
      __syncthreads();
    }
    {  // header items written by huff_code_kernel
      const uint32_t *hd = P.blk_hdr + (uint64_t)gb * kHdrMax;
      const int hn = (int)meta.y;
      for (int base = 0; base < hn; base += 64) {
        const int i = base + lane;
        const uint32_t it = i < hn ? hd[i] : 0u;
        sink_emit(S, it & 0xffffu, it >> 16, lane);
      }
    }
    if (kind == 1) {
      Walker w = walker_init(src, nullptr, 0u, n, lane);
      for (int base = 0; base < n; base += kTile) {
        const int i = base + 4 * lane;
        const uint32_t bt = w.bytes;
        w.bytes = load_bytes4(w, i + kTile);
        uint64_t lo = 0;  // four codes of <= 15 bits
        uint32_t nb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i + k < n) {
            const uint32_t c = sh.lit_cl[(bt >> (8 * k)) & 0xffu];
            lo |= (uint64_t)(c & 0xffffu) << nb;
            nb += c >> 16;
          }
        sink_emit_wide(S, lo, 0u, nb, lane);
      }
    } else {
      // The walk over the implied token sequence was done by huff_hist_kernel: per tile, one byte
      // per lane (TileTok::pack) says which of the lane's four positions are literals and whether
      // a match starts there; the lane's rank among the match lanes of the tile is its record.
      // Everything a tile needs is loaded one tile ahead (its meta byte two ahead).
      const uint32_t chunk = g.chunk0 + b;
      const uint2 *recs = P.matches + (uint64_t)chunk * kMatchCapPerChunk;
      const uint8_t *tmeta = P.tile_meta + tile_meta_at((uint64_t)(src - P.in), gb);
      Walker w = walker_init(src, nullptr, 0u, n, lane);  // (input bytes only)
      const int ntiles = (n + kTile - 1) / kTile;
      uint32_t mp = 0;
      auto tok_of = [&](uint32_t m) -> uint32_t {
        const bool has = (m & 0x10u) != 0;
        const uint64_t B = __ballot(has);
        uint32_t tok = 0;
        if (has) tok = recs[mp + (uint32_t)__popcll(B & ((1ull << lane) - 1ull))].y;
        mp += (uint32_t)__popcll(B);
        return tok;
      };
      uint32_t m_cur = tmeta[lane];
      uint32_t tok_cur = tok_of(m_cur);
      uint32_t m_next = ntiles > 1 ? tmeta[64 + lane] : 0u;
      for (int t = 0; t < ntiles; ++t) {
        const uint32_t bt = w.bytes;
        w.bytes = load_bytes4(w, (t + 1) * kTile + 4 * lane);
        const uint32_t tok_next = tok_of(m_next);
        const uint32_t m_next2 = t + 2 < ntiles ? tmeta[(t + 2) * 64 + lane] : 0u;
        uint64_t lo;
        uint32_t hi = 0, nb;
        {
          // The codes of all four bytes are looked up at once (one LDS round trip, no branches) and
          // those of covered positions masked off; pairs are joined in 32 bits (<= 30 each), the two
          // pairs by one 64-bit shift.  At most three literals when a match follows: <= 45 bits.
          uint32_t c[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) c[k] = sh.lit_cl[(bt >> (8 * k)) & 0xffu];
#pragma unroll
          for (int k = 0; k < 4; ++k) c[k] = ((m_cur >> k) & 1u) ? c[k] : 0u;
          const uint32_t n0 = c[0] >> 16, n1 = c[1] >> 16, n2 = c[2] >> 16, n3 = c[3] >> 16;
          const uint32_t p01 = (c[0] & 0xffffu) | ((c[1] & 0xffffu) << n0);
          const uint32_t p23 = (c[2] & 0xffffu) | ((c[3] & 0xffffu) << n2);
          lo = (uint64_t)p01 | ((uint64_t)p23 << (n0 + n1));
          nb = n0 + n1 + n2 + n3;
        }
        if (m_cur & 0x10u) {
          const uint32_t lb = sh.len_bits[(tok_cur >> kLengthShift) & 0xffu];
          const CodeBits oc = offset_code_of(tok_cur & ((1u << kLengthShift) - 1u));
          const uint32_t c2 = sh.off_cl[oc.code];
          // length code + extra (<= 15 + 5 bits, from the block's table) and offset code + extra
          // (<= 15 + 13) each in one dword, joined by a single 64-bit shift
          const uint32_t n1 = lb >> 24;
          const uint32_t part1 = lb & 0xffffffu;
          const uint32_t part2 = (c2 & 0xffffu) | (oc.extra << (c2 >> 16));
          const uint64_t bits = part1 | ((uint64_t)part2 << n1);
          const uint32_t mb = n1 + (c2 >> 16) + oc.nextra;  // <= 48
          hi = nb ? (uint32_t)(bits >> (64u - nb)) : 0u;  // (< 2^29: 48 + 45 - 64 bits)
          lo |= bits << nb;
          nb += mb;
        }
        sink_emit_wide(S, lo, hi, nb, lane);
        m_cur = m_next;
        tok_cur = tok_next;
        m_next = m_next2;
      }
    }
    const uint32_t eob = sh.lit_cl[kEndBlockMarker];
    sink_emit(S, eob & 0xffffu, lane == 0 ? (eob >> 16) : 0u, lane);
  }

// ---------------------------------------------------------------------------------------
// write_dynamic_header (:421-471), write_tokens (:596-731), write_block_huff's byte loop
// (:788-823), stored blocks: the stream's bits go to out + out_off[sid].
// ---------------------------------------------------------------------------------------
// (amdgpu_num_sgpr: a SIMD's 800 scalar registers admit floor(800 / (ceil(sgpr / 16) * 16 + 16)) wavefronts -- 6 at
// the 105 the compiler would take, 8 at 80; the kernel is VALU-bound and wants all eight: 1.09 -> 0.93 ms)
#ifndef FLATE_HUFF_PACK_SGPR
#define FLATE_HUFF_PACK_SGPR 80
#endif
__attribute__((amdgpu_num_sgpr(FLATE_HUFF_PACK_SGPR)))
__global__ __launch_bounds__(64) void huff_pack_kernel(HuffParams P) {
  __shared__ SharedPack sh;
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= P.n_streams || *P.status != 0) return;
  const BlockGeom g = block_geom(P, sid);

  for (int i = lane; i < kRing; i += 64) sh.ring[i] = 0;
  __syncthreads();
  BitSink S;
  S.ring = sh.ring;
  S.flushed = 0;
  S.or_first = S.or_last = P.spliced != 0;
  uint64_t out_bits;  // what this stream must write
  if (!P.spliced) {
    uint8_t *dst = P.out + P.out_off[sid];
    const uint64_t out_bytes = P.out_len[sid];
    out_bits = 8 * out_bytes;
    S.head = (uint32_t)((uintptr_t)dst & 3u);
    S.out32 = reinterpret_cast<uint32_t *>(dst - S.head);
    S.bitpos = 8ull * S.head;
    const uint64_t end_byte = S.head + out_bytes;  // exclusive, relative to out32
    S.last_dw = (uint32_t)((end_byte - 1) >> 2);
    S.tail = (uint32_t)(((end_byte - 1) & 3u) + 1u);
  } else {
    // one DEFLATE stream for the whole batch (splice_kernels.hip): this stream's blocks start at
    // bit stream_bit[sid] of the output; only the last stream writes the closing block.  The
    // sink's bit position is congruent to the position in the spliced stream mod 8, so the
    // padding of stored blocks comes out relative to the spliced stream.
    const uint64_t mis = (uint64_t)((uintptr_t)P.out & 3u);
    const uint64_t g0 = P.stream_bit[sid] + 8 * mis;  // relative to the aligned dword grid of out
    uint64_t g1 = P.stream_bit[sid + 1] + 8 * mis;
    const bool last_stream = sid + 1 == P.n_streams && !P.no_close;
    if (last_stream) g1 = ((g1 + 3 + 7) & ~7ull) + 32;  // + closing block
    out_bits = g1 - g0;
    S.head = 0;
    S.out32 = reinterpret_cast<uint32_t *>(P.out - mis) + (g0 >> 5);
    S.bitpos = g0 & 31u;
    S.last_dw = g1 > g0 ? (uint32_t)(((g1 - 1) >> 5) - (g0 >> 5)) : 0u;
    S.tail = 4;  // (the batch's very last dword may reach 3 bytes past the result: out_cap covers it)
  }
  const uint64_t bit0 = S.bitpos;

  for (uint32_t b = 0; b < g.nblocks; ++b) pack_block(P, sh, S, g, b, lane);
  if (!P.spliced || (sid + 1 == P.n_streams && !P.no_close))
    emit_stored(S, g.stream, 0, true, lane);  // Compressor::close (deflate.mbt:171-176)
  sink_finish(S, lane);
  // the sizes computed by huff_code_kernel and the bits actually written must agree
  if (lane == 0 && S.bitpos - bit0 != out_bits) atomicExch(P.status, -(int)(0x100000u + (sid & 0xfffffu)));  // E_INTERNAL + stream
}

// ---------------------------------------------------------------------------------------
// One wavefront per BLOCK (multi-window streams).  The blocks of a stream follow one another at bit
// granularity, so the dword two of them share is zeroed first (huff_zero_edges_kernel, own bytes
// only) and OR-ed into by both; the stream's first and last dword are shared with OTHER streams at
// byte granularity (plain byte stores, as in huff_pack_kernel).  Batch form only: in a spliced batch
// the bit a block starts at depends on where its stream starts (stored blocks pad to a byte of the
// spliced stream), which the per-stream kernel finds as it walks.
// ---------------------------------------------------------------------------------------
namespace {
struct StreamSpan {
  uint32_t *origin;  // dword grid the stream's bit positions are counted on
  uint64_t g0, g1;   // first bit of the stream's first block, end of the stream (closing block included)
  uint32_t head;     // non-spliced: bytes of the first dword that belong to the previous stream
};
FLATE_D StreamSpan stream_span(const HuffParams &P, uint32_t sid) {
  StreamSpan sp;
  if (!P.spliced) {
    uint8_t *dst = P.out + P.out_off[sid];
    sp.head = (uint32_t)((uintptr_t)dst & 3u);
    sp.origin = reinterpret_cast<uint32_t *>(dst - sp.head);
    sp.g0 = 8ull * sp.head;
    sp.g1 = sp.g0 + 8ull * P.out_len[sid];
  } else {
    const uint64_t mis = (uint64_t)((uintptr_t)P.out & 3u);
    sp.head = 0;
    sp.origin = reinterpret_cast<uint32_t *>(P.out - mis);
    sp.g0 = P.stream_bit[sid] + 8 * mis;
    sp.g1 = P.stream_bit[sid + 1] + 8 * mis;
    if (sid + 1 == P.n_streams && !P.no_close) sp.g1 = ((sp.g1 + 3 + 7) & ~7ull) + 32;  // + closing block
  }
  return sp;
}
FLATE_D uint64_t block_start_bit(const HuffParams &P, uint32_t gb) {
  const uint4 m = P.blk_meta[gb];
  return (uint64_t)m.z | ((uint64_t)m.w << 32);
}
}  // namespace

// One thread per block: the dword that holds a block's first bit, and the one in front of it, are
// shared with the previous block of the same stream.
__global__ __launch_bounds__(256) void huff_zero_edges_kernel(HuffParams P, uint32_t n_blocks) {
  const uint32_t gb = blockIdx.x * 256u + threadIdx.x;
  if (gb >= n_blocks || *P.status != 0) return;
  const uint32_t sid = P.blk_sid[gb];
  if (gb == P.blk_base[sid]) return;  // the stream's first block starts the stream
  const StreamSpan sp = stream_span(P, sid);
  const uint64_t x = sp.g0 + block_start_bit(P, gb);
  uint8_t *bytes = reinterpret_cast<uint8_t *>(sp.origin);
  // own bytes only: the stream's first and last dword may hold bytes of its neighbours
  const uint64_t own_lo = P.spliced ? 0ull : sp.g0 >> 3, own_hi = P.spliced ? ~0ull : sp.g1 >> 3;
  const uint64_t d = x >> 5;
  for (uint64_t k = (d ? d - 1 : 0) * 4; k < (d + 1) * 4; ++k)
    if (k >= own_lo && k < own_hi) bytes[k] = 0;
}

__attribute__((amdgpu_num_sgpr(FLATE_HUFF_PACK_SGPR)))
__global__ __launch_bounds__(64) void huff_pack_block_kernel(HuffParams P) {
  __shared__ SharedPack sh;
  const int lane = threadIdx.x;
  const uint32_t gb = blockIdx.x;
  if (*P.status != 0) return;
  const uint32_t sid = P.blk_sid[gb];
  const BlockGeom g = block_geom(P, sid);
  const uint32_t b = gb - g.blk0;
  const bool last = b + 1 == g.nblocks;
  const StreamSpan sp = stream_span(P, sid);
  const uint64_t x0 = sp.g0 + block_start_bit(P, gb);
  const uint64_t x1 = last ? sp.g1 : sp.g0 + block_start_bit(P, gb + 1);

  for (int i = lane; i < kRing; i += 64) sh.ring[i] = 0;
  __syncthreads();
  BitSink S;
  S.ring = sh.ring;
  S.flushed = 0;
  S.out32 = sp.origin + (x0 >> 5);
  S.bitpos = x0 & 31u;
  S.last_dw = x1 > x0 ? (uint32_t)(((x1 - 1) >> 5) - (x0 >> 5)) : 0u;
  S.head = (b == 0 && !P.spliced) ? sp.head : 0u;
  S.tail = (last && !P.spliced) ? (uint32_t)((((x1 >> 3) - 1) & 3u) + 1u) : 4u;
  S.or_first = b > 0 || P.spliced != 0;
  S.or_last = !last || P.spliced != 0;
  const uint64_t bit0 = S.bitpos;

  pack_block(P, sh, S, g, b, lane);
  if (last && (!P.spliced || (sid + 1 == P.n_streams && !P.no_close)))
    emit_stored(S, g.stream, 0, true, lane);  // Compressor::close (deflate.mbt:171-176)
  sink_finish(S, lane);
#ifdef FLATE_PB_DEBUG
  if (lane == 0 && S.bitpos - bit0 != x1 - x0)
    printf("pb mismatch sid %u b %u/%u kind %u wrote %llu want %llu x0 %llu x1 %llu g0 %llu g1 %llu\n", sid, b, g.nblocks,
           P.blk_meta[gb].x, (unsigned long long)(S.bitpos - bit0), (unsigned long long)(x1 - x0), (unsigned long long)x0,
           (unsigned long long)x1, (unsigned long long)sp.g0, (unsigned long long)sp.g1);
#endif
  if (lane == 0 && S.bitpos - bit0 != x1 - x0) atomicExch(P.status, -(int)(0x100000u + (sid & 0xfffffu)));
}

}  // namespace flate
