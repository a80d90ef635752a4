// inflate_kernels.hip -- batch inflater for gfx950: one wavefront per DEFLATE stream.
//
// Replaces, for N independent streams, the reference decoder:
//   Decompressor::next_block / read_huffman / read_literal / huff_sym / data_block
//                                   inflate.mbt:345-379, 429-548, 565-684, 803-854, 708-766
//   HuffmanDecoder::initialize      inflate.mbt:100-223
//   DictDecoder::write_copy         dict-decoder.mbt:114-185  (the history IS the output buffer)
//
// Decoding one stream is serial (every code starts where the previous one ended), so lane 0
// walks the bit stream; all 64 lanes build the decode tables (one symbol per lane) and perform
// the LZ77 / stored-block copies 64 bytes at a time.  Parallelism comes from the batch:
// thousands of streams, up to 8 waves per SIMD.
//
// Tables are not the reference's chunk/link tables: a 9-bit primary table (symbol, length) plus
// canonical first-code/count arrays for the (rare) longer codes -- same symbols, same accept /
// reject decisions.  Error reporting follows the reference exactly: the byte offset in
// corrupt_input_error (inflate.mbt:38) is the number of input bytes the byte-at-a-time reader
// (more_bits :789, huff_sym :818-831) has consumed, which is a function of the bits requested so
// far: roffset = max(roffset, ceil((consumed_bits + requested) / 8)).
#include "flate_kernels.h"

namespace flate {

namespace {

constexpr int kPrimBits = 9;  // huffman_chunk_bits, inflate.mbt:69
constexpr int kPrimSize = 1 << kPrimBits;
constexpr int kMaxLit = 286, kMaxDist = 30, kNumCodes = 19;
constexpr uint16_t kLongCode = 0x000f;

constexpr int E_OUT_SMALL = -2, E_CORRUPT = -4, E_EOF = -7;


// one Huffman decoder (inflate.mbt:81-86), canonical form
struct Dec {
  uint16_t prim[kPrimSize];  // (sym << 4) | len for len <= 9; kLongCode = longer code; 0 = invalid
  uint16_t sorted[288];      // symbols ordered by (len, symbol)
  uint32_t count[16];        // codes per length
  uint32_t first[16];        // first canonical code of each length
  uint32_t offs[16];         // index into sorted of the first symbol of each length
  int32_t min;               // minimum code length (h.min), 0 = empty tree
  int32_t max;
  int32_t ok;
};

__constant__ uint8_t kCodeOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

FLATE_D uint32_t ld32g(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// HuffmanDecoder::initialize (inflate.mbt:100-223) over lens[0..n), n <= 288.  All lanes.
// d.ok = 0 for an over- or under-subscribed code (:161).
__device__ __noinline__ void dec_init(Dec &d, const uint8_t *lens, int n, int lane) {
  for (int i = lane; i < kPrimSize; i += 64) d.prim[i] = 0;
  if (lane < 16) d.count[lane] = 0;
  __syncthreads();
  for (int i = lane; i < n; i += 64)
    if (lens[i]) atomicAdd(&d.count[lens[i]], 1u);
  __syncthreads();
  if (lane == 0) {
    int mn = 0, mx = 0;
    for (int l = 1; l < 16; ++l)
      if (d.count[l]) {
        if (!mn) mn = l;
        mx = l;
      }
    d.min = mn;
    d.max = mx;
    d.ok = 1;
    uint32_t code = 0, off = 0;
    for (int l = 0; l < 16; ++l) {
      d.first[l] = 0;
      d.offs[l] = off;
    }
    for (int l = mn; l <= mx && mx; ++l) {  // :148-154
      code <<= 1;
      d.first[l] = code;
      d.offs[l] = off;
      code += d.count[l];
      off += d.count[l];
    }
    if (mx && code != (1u << mx) && !(code == 1 && mx == 1)) d.ok = 0;  // :161
  }
  __syncthreads();
  if (d.max == 0 || !d.ok) return;  // empty tree is valid (:143-145)
  // canonical rank of every symbol inside its length, in symbol order (running counts are
  // wave-uniform, so every lane keeps its own copy in registers)
  uint32_t run[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) run[l] = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint32_t L = i < n ? lens[i] : 0u;
    uint32_t rank = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      const uint64_t m = __ballot(L == (uint32_t)l);
      if (L == (uint32_t)l) rank = run[l] + __popcll(m & ((1ull << lane) - 1));
      run[l] += __popcll(m);
    }
    if (L) {
      const uint32_t code = d.first[L] + rank;
      d.sorted[d.offs[L] + rank] = (uint16_t)i;
      if (L <= (uint32_t)kPrimBits) {
        const uint32_t rev = __brev(code) >> (32 - L);
        const uint16_t e = (uint16_t)((i << 4) | L);
        for (uint32_t k = rev; k < (uint32_t)kPrimSize; k += 1u << L) d.prim[k] = e;
      } else {
        const uint32_t top = code >> (L - kPrimBits);
        d.prim[__brev(top) >> (32 - kPrimBits)] = kLongCode;
      }
    }
  }
  __syncthreads();
}

// Wave-uniform bit reader over the LDS stage of the compressed input.  Every value below is
// the same in all 64 lanes and lives in SGPRs (LDS look-ups go through v_readfirstlane), so the
// serial symbol loop runs on the scalar unit; the vector lanes are used for the copies.
constexpr int kStage = 2048;      // bytes of compressed input staged in LDS
constexpr int kStageMargin = 64;  // restage when fewer bytes than this remain staged
constexpr int kWin = 32768;       // history window (max_match_offset, inflate.mbt:330)
constexpr int kFlushAt = 8192;    // flush the window to HBM when this much output is pending

FLATE_D uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct InfShared {
  Dec lit, dist;
  uint8_t lens[kMaxLit + kMaxDist + 2 + 288];
  uint32_t stage[kStage / 4];
  uint8_t win[kWin];  // DictDecoder.hist (dict-decoder.mbt:29-35): the last 32 KiB of output
};

struct Bits {      // 32-bit offsets: the API rejects streams >= 2 GiB
  uint32_t in_len;
  uint32_t roff;   // bytes the reference's reader has consumed (roffset, inflate.mbt:260)
  uint32_t ipos;   // next byte of the stream to load into buf
  uint32_t sbase;  // stream offset of stage[0]
  uint64_t buf;    // physical look-ahead, LSB first
  int32_t cnt;     // valid bits in buf
  int32_t avail;   // bits the reference's reader holds: 8 * roff - consumed bits
};

FLATE_D void bits_refill(Bits &b, const uint32_t *stage) {
  while (b.cnt <= 32) {
    const uint32_t idx = (b.ipos - b.sbase) >> 2;
    const uint32_t w = idx < (uint32_t)(kStage / 4) ? uni(stage[idx]) : 0u;  // past the stage: zeros
    b.buf |= (uint64_t)w << b.cnt;
    b.cnt += 32;
    b.ipos += 4;
  }
}
// the reader asks for n more bits: false = the input ends first
FLATE_D bool bits_need(Bits &b, int n) {
  if (n > b.avail) {
    const int bytes = (n - b.avail + 7) >> 3;
    if (b.roff + (uint32_t)bytes > b.in_len) {
      b.roff = b.in_len;
      return false;
    }
    b.roff += (uint32_t)bytes;
    b.avail += 8 * bytes;
  }
  return true;
}
FLATE_D uint32_t bits_peek(const Bits &b, uint32_t n) { return (uint32_t)b.buf & ((1u << n) - 1u); }
FLATE_D void bits_drop(Bits &b, int n, const uint32_t *stage) {
  b.buf >>= n;
  b.cnt -= n;
  b.avail -= n;
  if (b.cnt <= 32) bits_refill(b, stage);
}

// huff_sym (inflate.mbt:803-854).  Returns the symbol, or -1 with *err set.
FLATE_D int huff_sym(Bits &b, const Dec &d, int dmin, int dmax, const uint32_t *stage, int *err) {
  if (!bits_need(b, dmin)) {
    *err = E_EOF;
    return -1;
  }
  const uint32_t e = uni(d.prim[(uint32_t)b.buf & (kPrimSize - 1)]);
  if (e == 0) {
    *err = E_CORRUPT;
    return -1;
  }
  if (e != kLongCode) {
    const int len = (int)(e & 15u);
    if (!bits_need(b, len)) {
      *err = E_EOF;
      return -1;
    }
    bits_drop(b, len, stage);
    return (int)(e >> 4);
  }
  uint32_t code = __brev((uint32_t)b.buf & (kPrimSize - 1)) >> (32 - kPrimBits);
  for (int L = kPrimBits + 1; L <= dmax; ++L) {
    code = (code << 1) | ((uint32_t)(b.buf >> (L - 1)) & 1u);
    const uint32_t idx = code - uni(d.first[L]);
    if (idx < uni(d.count[L])) {
      if (!bits_need(b, L)) {
        *err = E_EOF;
        return -1;
      }
      bits_drop(b, L, stage);
      return (int)uni(d.sorted[uni(d.offs[L]) + idx]);
    }
  }
  *err = E_CORRUPT;
  return -1;
}

}  // namespace

__global__ __launch_bounds__(64) void inflate_kernel(InfParams P) {
  __shared__ InfShared sh;
  const int lane = threadIdx.x;
  const uint32_t sid = blockIdx.x;
  if (sid >= P.n_streams) return;
  // size_only: the same decoding (the 32 KiB window in LDS is all the history it needs), nothing
  // stored to HBM and no capacity limit: out_len becomes the size the stream inflates to
  const bool size_only = P.size_only != 0;
  uint8_t *out = size_only ? nullptr : P.out + P.out_off[sid];
  const uint64_t cap64 = size_only ? ~0ull : P.out_off[sid + 1] - P.out_off[sid];
  const uint32_t out_cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
  const uint8_t *in = P.in + P.in_off[sid];
  const uint32_t in_len = (uint32_t)(P.in_off[sid + 1] - P.in_off[sid]);

  Bits b;
  b.in_len = in_len;
  b.roff = 0;
  b.ipos = 0;
  b.sbase = 0;
  b.buf = 0;
  b.cnt = 0;
  b.avail = 0;
  uint32_t opos = 0;  // bytes produced
  uint32_t fpos = 0;  // bytes already flushed to HBM
  int err = 0;
  bool final_block = false;

  // (re)load the LDS stage of the compressed input at b.ipos
  auto restage = [&]() {
    __syncthreads();
    const uint32_t base = b.ipos;
    for (int k = lane; k < kStage / 4; k += 64) {
      const uint32_t p = base + 4u * k;
      uint32_t w = 0;
      if (p + 4 <= in_len) {
        w = ld32g(in + p);
      } else {
        for (uint32_t q = p; q < in_len; ++q) w |= (uint32_t)in[q] << (8 * (q - p));
      }
      sh.stage[k] = w;
    }
    b.sbase = base;
    __syncthreads();
    bits_refill(b, sh.stage);
  };
  auto stage_low = [&]() { return b.ipos + kStageMargin > b.sbase + kStage && b.sbase + kStage < in_len; };
  // read_flush (dict-decoder.mbt:200-209): window -> HBM, coalesced
  auto flush = [&]() {
    __syncthreads();
    if (!size_only)
      for (uint32_t i = fpos + lane; i < opos; i += 64) out[i] = sh.win[i & (kWin - 1)];
    fpos = opos;
  };

  // Re-assert wave-uniformity of the decoder state (no-ops at run time: every lane already holds
  // the same values) so that the symbol loop is compiled for the scalar unit.
  auto pin = [&]() {
    b.roff = uni(b.roff);
    b.ipos = uni(b.ipos);
    b.sbase = uni(b.sbase);
    b.buf = ((uint64_t)uni((uint32_t)(b.buf >> 32)) << 32) | uni((uint32_t)b.buf);
    b.cnt = (int)uni((uint32_t)b.cnt);
    b.avail = (int)uni((uint32_t)b.avail);
    opos = uni(opos);
    fpos = uni(fpos);
    err = (int)uni((uint32_t)err);
  };

  restage();
  while (!final_block && !err) {  // next_block (inflate.mbt:345-379)
    pin();
    if (stage_low()) restage();
    if (!bits_need(b, 3)) {
      err = E_EOF;
      break;
    }
    const uint32_t h = bits_peek(b, 3);
    final_block = h & 1;
    const uint32_t typ = h >> 1;
    bits_drop(b, 3, sh.stage);
    if (typ == 3) {
      err = E_CORRUPT;  // reserved (:375-377)
      break;
    }
    if (typ == 0) {  // data_block (:708-766): discard the partial byte, LEN, ~LEN, raw bytes
      const uint32_t p = b.roff;
      if (in_len - p < 4) {
        b.roff = in_len;
        err = E_EOF;
        break;
      }
      b.roff = p + 4;
      const uint32_t n = uni((uint32_t)in[p] | ((uint32_t)in[p + 1] << 8));
      const uint32_t nn = uni((uint32_t)in[p + 2] | ((uint32_t)in[p + 3] << 8));
      if ((nn & 0xffffu) != ((~n) & 0xffffu)) {
        err = E_CORRUPT;
        break;
      }
      flush();
      const uint32_t ip = b.roff, avail = in_len - ip;
      const uint32_t cnt = avail < n ? avail : n;
      if (cnt > out_cap - opos) {
        err = E_OUT_SMALL;
        break;
      }
      for (uint32_t i = lane; i < cnt; i += 64) {
        const uint8_t v = in[ip + i];
        if (!size_only) out[opos + i] = v;
        sh.win[(opos + i) & (kWin - 1)] = v;
      }
      opos += cnt;
      fpos = opos;
      b.roff += cnt;
      if (cnt < n) {
        err = E_EOF;
        break;
      }
      b.ipos = b.roff;  // the bit reader restarts at the byte after the block
      b.buf = 0;
      b.cnt = 0;
      b.avail = 0;
      restage();
      continue;
    }
    int lit_min, lit_max, dist_min, dist_max;
    if (typ == 1) {  // fixed_huffman_decoder (:886-939); distances are 5-bit codes
      __syncthreads();
      uint8_t *fl = sh.lens + 32;
      for (int i = lane; i < 288; i += 64) fl[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
      if (lane < 32) fl[288 + lane] = 5;
      __syncthreads();
      dec_init(sh.lit, fl, 288, lane);
      dec_init(sh.dist, fl + 288, 32, lane);
    } else {  // read_huffman (:429-548)
      if (!bits_need(b, 14)) {
        err = E_EOF;
        break;
      }
      const uint32_t v = bits_peek(b, 14);
      const int nlit = (int)(v & 31u) + 257, ndist = (int)((v >> 5) & 31u) + 1;
      const int nclen = (int)((v >> 10) & 15u) + 4;
      if (nlit > kMaxLit || ndist > kMaxDist) {
        err = E_CORRUPT;
        break;
      }
      bits_drop(b, 14, sh.stage);
      __syncthreads();
      if (lane < kNumCodes) sh.lens[lane] = 0;
      __syncthreads();
      for (int i = 0; i < nclen && !err; ++i) {
        if (!bits_need(b, 3)) {
          err = E_EOF;
          break;
        }
        if (lane == 0) sh.lens[kCodeOrder[i]] = (uint8_t)bits_peek(b, 3);
        bits_drop(b, 3, sh.stage);
      }
      if (err) break;
      __syncthreads();
      dec_init(sh.dist, sh.lens, kNumCodes, lane);  // code-length code
      if (!uni((uint32_t)sh.dist.ok)) {
        err = E_CORRUPT;
        break;
      }
      const int cmin = (int)uni((uint32_t)sh.dist.min), cmax = (int)uni((uint32_t)sh.dist.max);
      uint8_t *cl = sh.lens + 32;
      const int total = nlit + ndist;
      int i = 0;
      while (i < total) {  // :471-530
        pin();
        i = (int)uni((uint32_t)i);
        if (stage_low()) restage();
        const int x = huff_sym(b, sh.dist, cmin, cmax, sh.stage, &err);
        if (x < 0) break;
        if (x < 16) {
          if (lane == 0) cl[i] = (uint8_t)x;
          ++i;
          continue;
        }
        int rep, nb;
        uint32_t fill = 0;
        if (x == 16) {
          rep = 3;
          nb = 2;
          if (i == 0) {
            err = E_CORRUPT;
            break;
          }
          __syncthreads();
          fill = uni(cl[i - 1]);
        } else if (x == 17) {
          rep = 3;
          nb = 3;
        } else {
          rep = 11;
          nb = 7;
        }
        if (!bits_need(b, nb)) {
          err = E_EOF;
          break;
        }
        rep += (int)bits_peek(b, (uint32_t)nb);
        bits_drop(b, nb, sh.stage);
        if (i + rep > total) {
          err = E_CORRUPT;
          break;
        }
        if (lane < rep) cl[i + lane] = (uint8_t)fill;
        if (lane + 64 < rep) cl[i + lane + 64] = (uint8_t)fill;
        if (lane + 128 < rep) cl[i + lane + 128] = (uint8_t)fill;
        i += rep;
      }
      if (err) break;
      __syncthreads();
      dec_init(sh.lit, cl, nlit, lane);
      dec_init(sh.dist, cl + nlit, ndist, lane);
      if (!uni((uint32_t)sh.lit.ok) || !uni((uint32_t)sh.dist.ok)) {
        err = E_CORRUPT;
        break;
      }
    }
    lit_min = (int)uni((uint32_t)sh.lit.min);
    lit_max = (int)uni((uint32_t)sh.lit.max);
    dist_min = (int)uni((uint32_t)sh.dist.min);
    dist_max = (int)uni((uint32_t)sh.dist.max);
    if (typ == 2) {  // read at least the end-of-block code's length (:542-544)
      const int eob = (int)uni(sh.lens[32 + 256]);
      if (lit_min < eob) lit_min = eob;
    }

    for (;;) {  // read_literal (:565-684)
      pin();
      if (stage_low()) restage();
      if (opos - fpos >= (uint32_t)kFlushAt) flush();
      const int v = huff_sym(b, sh.lit, lit_min, lit_max, sh.stage, &err);
      if (v < 0) break;
      if (v < 256) {
        if (opos >= out_cap) {
          err = E_OUT_SMALL;
          break;
        }
        if (lane == 0) sh.win[opos & (kWin - 1)] = (uint8_t)v;  // DictDecoder::write_byte
        ++opos;
        continue;
      }
      if (v == 256) break;  // finish_block
      int length, n;
      if (v < 265) {
        length = v - (257 - 3);
        n = 0;
      } else if (v < 269) {
        length = v * 2 - (265 * 2 - 11);
        n = 1;
      } else if (v < 273) {
        length = v * 4 - (269 * 4 - 19);
        n = 2;
      } else if (v < 277) {
        length = v * 8 - (273 * 8 - 35);
        n = 3;
      } else if (v < 281) {
        length = v * 16 - (277 * 16 - 67);
        n = 4;
      } else if (v < 285) {
        length = v * 32 - (281 * 32 - 131);
        n = 5;
      } else if (v < kMaxLit) {
        length = 258;
        n = 0;
      } else {
        err = E_CORRUPT;
        break;
      }
      if (n > 0) {
        if (!bits_need(b, n)) {
          err = E_EOF;
          break;
        }
        length += (int)bits_peek(b, (uint32_t)n);
        bits_drop(b, n, sh.stage);
      }
      int dist = huff_sym(b, sh.dist, dist_min, dist_max, sh.stage, &err);
      if (dist < 0) break;
      if (dist < 4) {
        dist += 1;
      } else if (dist < kMaxDist) {
        const int nb = (dist - 2) >> 1;
        int extra = (dist & 1) << nb;
        if (!bits_need(b, nb)) {
          err = E_EOF;
          break;
        }
        extra |= (int)bits_peek(b, (uint32_t)nb);
        bits_drop(b, nb, sh.stage);
        dist = (1 << (nb + 1)) + 1 + extra;
      } else {
        err = E_CORRUPT;
        break;
      }
      const uint32_t hist = opos < (uint32_t)kWin ? opos : (uint32_t)kWin;  // hist_size
      if ((uint32_t)dist > hist) {
        err = E_CORRUPT;
        break;
      }
      if ((uint32_t)length > out_cap - opos) {
        err = E_OUT_SMALL;
        break;
      }
      // copy_history (:689) / write_copy (dict-decoder.mbt:114-154); overlap = periodic extension
      __syncthreads();
      const uint32_t op = opos;
      for (int i = lane; i < length; i += 64) {
        const int o = dist >= length ? i : i % dist;
        sh.win[(op + (uint32_t)i) & (kWin - 1)] = sh.win[(op - (uint32_t)dist + (uint32_t)o) & (kWin - 1)];
      }
      __syncthreads();
      opos += (uint32_t)length;
    }
  }
  flush();
  if (lane == 0) {
    P.out_len[sid] = opos;
    P.status[sid] = err;
    P.err_off[sid] = err == E_CORRUPT ? (long long)b.roff : -1;
  }
}

#include "inflate_stream_kernel.inc"
#include "inflate_spec_kernel.inc"
template __global__ void inflate_spec_kernel<FLATE_SPEC_SMALL>(InfParams);
template __global__ void inflate_spec_kernel<FLATE_SPEC_LARGE>(InfParams);

}  // namespace flate

// =======================================================================================
// SIMT inflater: one LANE per stream (64 streams per wavefront).
//
// The wave-per-stream kernel above keeps 63 lanes idle during symbol decoding and is limited
// to four wavefronts per CU by its 32 KiB LDS window.  For large batches it is better to give
// every lane its own stream: each lane runs the same decoder state machine (inflate.mbt's step
// functions) on its own bit stream, the decode tables of the 64 streams are interleaved in LDS
// (entry i of lane L at u16 index i*64+L) and the history window is the output buffer itself
// (a lane reads back only bytes it has written).  Divergence is bounded by design: one step of
// the loop lets every lane either decode one symbol or copy up to eight bytes.
// =======================================================================================
namespace flate {

namespace {

// per-lane LDS layout, in u16 entries: 320 bytes per lane, so EIGHT 64-lane wavefronts fit one CU (two
// per SIMD; 8 x 64 x 320 B = the CU's 160 KiB exactly) and 131072 streams -- 8 GiB of 64 KiB streams,
// BASELINE config 5 -- are one round of 512 lanes per CU.  There is no lookup table: a code is
// resolved canonically (canon_decode), which needs only the symbols sorted by (length, symbol) -- one
// byte each -- and small per-length arrays.  LDS holds what the literal/length code needs on every
// symbol: its sorted list (288 B) and delta[16] (32 B).  The rest lives in REGISTERS: the distance
// code's sorted list (30 symbols, four per register), and the two per-length arrays that a lookup
// reaches through the length comparison itself (PerLen: no dynamic register index, no select chain).
// Round 3 kept everything, and the header's code lengths, in LDS: 592 B per lane, four wavefronts.
constexpr int kOffLitSorted = 0;                      // 288 symbols & 0xff, two per u16
constexpr int kOffLitMeta = kOffLitSorted + 144;      // delta[16]
constexpr int kLaneWords = kOffLitMeta + 16;          // 160
// The 352 code lengths of a block header while it is parsed (32 + 286 + 30 used, four bits each) do NOT
// live in LDS: they are touched by the header states only -- once per block, a few hundred steps
// against thousands of symbol steps -- and their 176 bytes per lane were what held a CU to four
// wavefronts.  Every lane has a slice of global scratch
// instead (InfParams::simt_lens, kLensDwords dwords per lane).
constexpr int kLensDwords = 48;

enum SState { S_BLOCK = 0, S_DYN_LENS, S_SYM, S_DIST, S_STORED, S_DONE };
// literal/length symbols one lane may decode per step (measured on config 5: 2 -4 %, 3 = 4, 6 -9 %)
constexpr int kSymPerStep = 4;

// The 15 code-length limits of one Huffman code, two per register (see shuff_sym).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
struct Limits {
  us2 p[8];
};
// A per-length array V[1..15] (16-bit values) in eight registers, read WITHOUT an index: the length of
// a code is 16 - #{k : lim[k] > c15}, and with e[k] = V[k+1] - V[k+2] (V[16] = 0) the sum of e[k] over
// exactly those k telescopes to V[len] (mod 2^16) -- one packed multiply-add per pair of limits, next
// to the compare that counts them (canon_decode).
struct PerLen {
  us2 e[8];
};
// the distance code (and, while a header is read, the code-length code): sorted symbols + delta[]
struct DistRegs {
  uint32_t sorted[8];  // 32 symbols of 8 bits
  PerLen delta;
};
// sorted[i & 31] without a dynamic register index
FLATE_D uint32_t dist_sorted_get(const DistRegs &D, uint32_t i) {
  // (selects over VALUES read first and made opaque: a select between two fields becomes a load from
  // a selected address -- dynamic indexing -- and that keeps the whole struct in scratch memory)
  uint32_t s0 = D.sorted[0], s1 = D.sorted[1], s2 = D.sorted[2], s3 = D.sorted[3];
  uint32_t s4 = D.sorted[4], s5 = D.sorted[5], s6 = D.sorted[6], s7 = D.sorted[7];
  asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));
  const uint32_t w = (i >> 2) & 7u;
  const uint32_t a = (w & 1u) ? s1 : s0, b = (w & 1u) ? s3 : s2;
  const uint32_t c = (w & 1u) ? s5 : s4, d = (w & 1u) ? s7 : s6;
  const uint32_t e = (w & 2u) ? b : a, f = (w & 2u) ? d : c;
  return (((w & 4u) ? f : e) >> ((i & 3u) * 8u)) & 0xffu;
}

template <int LPW>
struct LaneLds {
  uint16_t *base;  // &lds[lane]; entry i of this lane is base[i * LPW]
  FLATE_D uint32_t get(int i) const { return base[i * LPW]; }
  FLATE_D void set(int i, uint32_t v) const { base[i * LPW] = (uint16_t)v; }
  FLATE_D uint32_t get8(int off, uint32_t i) const { return (base[(off + (int)(i >> 1)) * LPW] >> ((i & 1u) * 8u)) & 0xffu; }
  FLATE_D void set8(int off, uint32_t i, uint32_t v) const {
    uint16_t &w = base[(off + (int)(i >> 1)) * LPW];
    w = (uint16_t)((i & 1u) ? ((w & 0x00ffu) | (v << 8)) : ((w & 0xff00u) | v));
  }
};

struct LaneLens {
  uint32_t *g;  // this lane's kLensDwords dwords
  FLATE_D uint32_t len_get(int i) const { return (g[i >> 3] >> ((i & 7) * 4)) & 15u; }
  FLATE_D void len_set(int i, uint32_t v) const {
    const int sh = (i & 7) * 4;
    const uint32_t w = g[i >> 3];
    g[i >> 3] = (w & ~(15u << sh)) | (v << sh);
  }
};

// Bit reader of one lane: the consumed-bit position plus the three dwords under it; the two
// dwords after those are always requested one step ahead.  The reference reads its input byte-wise
// (inflate.mbt:771 more_bits); its roffset -- reported by corrupt_input_error -- equals
// ceil(hi / 8), hi = the highest bit position any read has asked for.
struct SBits {
  const uint8_t *in;
  uint32_t in_len, in_bits;
  uint32_t bitpos, hi;
  uint32_t w0, w1, w2, widx;  // the window: dwords widx, widx+1, widx+2 of the stream
  uint32_t n0, n1;            // dwords widx+3, widx+4, requested one step ahead
};

FLATE_D uint32_t sb_load(const SBits &b, uint32_t widx) {
  const uint32_t pos = widx * 4u;
  uint32_t w = 0;
  if (pos + 4 <= b.in_len) {
    w = ld32g(b.in + pos);
  } else {
    for (uint32_t k = pos; k < b.in_len; ++k) w |= (uint32_t)b.in[k] << (8 * (k - pos));
  }
  return w;
}
// dwords widx and widx+1 in one request
FLATE_D uint64_t sb_load2(const SBits &b, uint32_t widx) {
  const uint32_t pos = widx * 4u;
  uint64_t v;
  if (pos + 8 <= b.in_len) {
    __builtin_memcpy(&v, b.in + pos, 8);
  } else {
    v = sb_load(b, widx) | ((uint64_t)sb_load(b, widx + 1) << 32);
  }
  // Hide that the halves come from one register pair: otherwise the two field stores are merged
  // into one 8-byte store, and that keeps the whole lane state in scratch memory instead of
  // registers (scalar replacement gives up on the mixed-width accesses).
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  return ((uint64_t)hi << 32) | lo;
}
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt 0, expcnt/lgkmcnt untouched).  The block-header paths
// load straight into the window registers; waiting there keeps the compiler from guarding every
// use of the window in the hot states with a wait that would also cover the prefetches.
FLATE_D void vm_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }

FLATE_D void sb_start(SBits &b, uint32_t byte_pos) {
  b.bitpos = b.hi = byte_pos * 8u;
  b.widx = byte_pos >> 2;
  // (pairs as the struct lays them out: a merged 8-byte store across two pairs would keep the
  // whole lane state from being promoted to registers)
  const uint64_t a = sb_load2(b, b.widx), c = sb_load2(b, b.widx + 3);
  b.w0 = (uint32_t)a;
  b.w1 = (uint32_t)(a >> 32);
  b.w2 = sb_load(b, b.widx + 2);
  b.n0 = (uint32_t)c;
  b.n1 = (uint32_t)(c >> 32);
  vm_wait_all();
}
// The next 32 bits at window offset off = bitpos - 32*widx (0..95); past offset 64 only the
// first 96 - off of them are real, the callers check what they use.
FLATE_D uint32_t sb_window(const SBits &b, uint32_t off) {
  // (selects over values, not over the fields themselves: a load from a selected field address
  // is dynamic indexing, and that would keep the bit reader in scratch memory)
  const uint32_t w0 = b.w0, w1 = b.w1, w2 = b.w2;
  const uint32_t lo = off < 32u ? w0 : (off < 64u ? w1 : w2);
  const uint32_t hi = off < 32u ? w1 : (off < 64u ? w2 : 0u);
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (off & 31u));
}
FLATE_D uint32_t sb_peek(const SBits &b) { return sb_window(b, b.bitpos - b.widx * 32u); }
FLATE_D bool sb_need(SBits &b, uint32_t n) {
  const uint32_t t = b.bitpos + n;
  b.hi = b.hi > t ? b.hi : t;
  return t <= b.in_bits;
}
FLATE_D void sb_take(SBits &b, uint32_t n) { b.bitpos += n; }
// blocking catch-up for the block-header states (the hot states advance in the step loop)
FLATE_D void sb_sync(SBits &b) {
  while ((b.bitpos >> 5) != b.widx) {
    b.w0 = b.w1;
    b.w1 = b.w2;
    b.w2 = b.n0;
    b.n0 = b.n1;
    ++b.widx;
    b.n1 = sb_load(b, b.widx + 4);
    vm_wait_all();
  }
}
FLATE_D uint32_t sb_roffset(const SBits &b) { return (b.hi + 7u) >> 3; }

// HuffmanDecoder::initialize for one lane (inflate.mbt:118-213): lens[lens_at .. +n) -> the
// canonical decoding data.  Returns false for an over/under-subscribed code (:161); *mn = min
// length.  lim[k-1] (k = 1..15) = exclusive upper bound of the codes of length <= k, left-justified
// to 15 bits; meta[k] = offs[k] - first[k] (mod 2^16), so sorted index = code + meta[k]; with THR,
// meta[16+k] = first code of length k whose symbol is >= 256 (the sorted list keeps only the low
// byte; inside one length symbols ascend, so those come last).
// (results by value: handing out addresses of the caller's lane state would pin it in memory)
struct DecInit {
  Limits lim;
  int mn;
  bool ok;
  PerLen aux;          // LIT: thr[] (first code of each length whose symbol is >= 256); else delta[]
  uint32_t sorted[8];  // !LIT: the sorted symbols (registers); LIT: the list is in LDS
};
template <bool LIT, class LL>
FLATE_D DecInit sdec_init(const LL &L, const LaneLens &N, int lens_at, int n) {
  DecInit R;
  R.ok = true;
  uint32_t lim[16];
  uint32_t cnt[16], low[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) cnt[k] = low[k] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    R.sorted[k] = 0;
    R.aux.e[k] = us2{0, 0};
  }
  uint32_t lw = 0;  // the dword of eight lengths under the cursor (one global load per eight symbols)
  for (int i = 0; i < n; ++i) {
    const int idx = lens_at + i;
    if (i == 0 || (idx & 7) == 0) lw = N.g[idx >> 3];
    const uint32_t l = (lw >> ((idx & 7) * 4)) & 15u;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      cnt[k] += l == (uint32_t)k ? 1u : 0u;
      if (LIT) low[k] += (l == (uint32_t)k && i < 256) ? 1u : 0u;
    }
  }
  int mn = 0, mx = 0;
#pragma unroll
  for (int k = 15; k >= 1; --k) {
    if (cnt[k]) mn = k;
    if (cnt[k] && !mx) mx = k;
  }
  R.mn = mn;
#pragma unroll
  for (int k = 0; k < 8; ++k) R.lim.p[k] = us2{0, 0};
  if (mx == 0) return R;  // empty tree (:143-145): every lookup is corrupt
  lim[15] = 0;  // pad: never above the code bits
  uint32_t next_off[16];
  // the per-length array V[k] that goes to registers (thr / delta) as ev[j] = V[j + 1] - V[j + 2],
  // j = 0 .. 14 (limit j separates lengths <= j + 1 from the longer ones), V[16] = 0
  uint32_t ev[16];
  ev[15] = 0;
  uint32_t code = 0, off = 0, vprev = 0;
#pragma unroll
  for (int k = 1; k < 16; ++k) {  // :148-154
    code <<= 1;
    const uint32_t delta = (off - code) & 0xffffu;
    if (LIT) L.set(kOffLitMeta + k, delta);
    const uint32_t v = LIT ? code + low[k] : delta;
    if (k >= 2) ev[k - 2] = (vprev - v) & 0xffffu;
    vprev = v;
    next_off[k] = off;
    code += cnt[k];
    off += cnt[k];
    lim[k - 1] = code << (15 - k);
  }
  ev[14] = vprev & 0xffffu;
  {  // completeness (:161), from min to max as the reference computes it
    uint32_t cc = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k)
      if (k >= mn && k <= mx) cc = (cc << 1) + cnt[k];
    if (cc != (1u << mx) && !(cc == 1 && mx == 1)) {
      R.ok = false;
      return R;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    R.lim.p[k] = us2{(unsigned short)lim[2 * k], (unsigned short)lim[2 * k + 1]};
    R.aux.e[k] = us2{(unsigned short)ev[2 * k], (unsigned short)ev[2 * k + 1]};
  }
  lw = 0;
  for (int i = 0; i < n; ++i) {
    const int idx = lens_at + i;
    if (i == 0 || (idx & 7) == 0) lw = N.g[idx >> 3];
    const uint32_t l = (lw >> ((idx & 7) * 4)) & 15u;
    if (!l) continue;
    uint32_t at = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k)
      if (l == (uint32_t)k) at = next_off[k]++;
    if (LIT) {
      L.set8(kOffLitSorted, at, (uint32_t)i & 0xffu);
    } else {  // (at < 32: the distance code has 30 symbols, the code-length code 19)
      const uint32_t put = ((uint32_t)i & 0xffu) << ((at & 3u) * 8u);
#pragma unroll
      for (int w = 0; w < 8; ++w) R.sorted[w] |= (at >> 2) == (uint32_t)w ? put : 0u;
    }
  }
  return R;
}

// Canonical decode of one code, no loop and no branch: returns the code length (16 = no such
// code, also for an empty tree) and the symbol.  The length is 16 - the number of limits above the
// next 15 bits (MSB first), counted two at a time with packed 16-bit arithmetic (c15 - lim is
// negative exactly when lim is above); the same 0/1 pairs, multiplied into a PerLen, deliver that
// length's entry of a per-length array.  The symbol sits at code + delta[len] in the sorted list.
// w = the bit window (LSB = next bit).
FLATE_D uint32_t canon_len(uint32_t w, const Limits &lim, const PerLen &aux, uint32_t *c15_out, uint32_t *aux_out) {
  const uint32_t c15 = __brev(w) >> 17;
  const us2 c2 = us2{(unsigned short)c15, (unsigned short)c15};
  us2 above = us2{0, 0}, acc = us2{0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const us2 ind = (c2 - lim.p[k]) >> 15;
    above += ind;
    acc += ind * aux.e[k];
  }
  *c15_out = c15;
  *aux_out = ((uint32_t)acc.x + (uint32_t)acc.y) & 0xffffu;
  return 16u - above.x - above.y;
}
// literal/length code: sorted list and delta[] in LDS, thr[] in registers
template <class LL>
FLATE_D uint32_t canon_decode_lit(const LL &L, uint32_t w, const Limits &lim, const PerLen &thr, uint32_t *sym_out) {
  uint32_t c15, t;
  const uint32_t len = canon_len(w, lim, thr, &c15, &t);
  const uint32_t lc = len > 15u ? 15u : len;
  const uint32_t code = c15 >> (15u - lc);
  uint32_t sym = L.get8(kOffLitSorted, (code + L.get(kOffLitMeta + (int)lc)) & 0xffffu);
  sym |= code >= t ? 256u : 0u;
  *sym_out = sym;
  return len;
}
// distance code / code-length code: everything in registers
FLATE_D uint32_t canon_decode_dist(uint32_t w, const Limits &lim, const DistRegs &D, uint32_t *sym_out) {
  uint32_t c15, delta;
  const uint32_t len = canon_len(w, lim, D.delta, &c15, &delta);
  const uint32_t lc = len > 15u ? 15u : len;
  const uint32_t code = c15 >> (15u - lc);
  *sym_out = dist_sorted_get(D, code + delta);
  return len;
}

// huff_sym (inflate.mbt:803-854) for the block-header states; the hot states inline the same
// checks without branches.  Returns the symbol, or -1 with *err set.
FLATE_D int shuff_sym(SBits &b, const DistRegs &D, int dmin, const Limits &lim, int *err) {
  uint32_t sym;
  const uint32_t len = canon_decode_dist(sb_peek(b), lim, D, &sym);
  if (!sb_need(b, (uint32_t)dmin)) {
    *err = E_EOF;
    return -1;
  }
  if (len > 15u) {
    *err = E_CORRUPT;
    return -1;
  }
  if (!sb_need(b, len)) {
    *err = E_EOF;
    return -1;
  }
  sb_take(b, len);
  return (int)sym;
}



// ---- the output ROW of a lane (round 5) ---------------------------------------------------------------
// Every lane stores into its own stream: a lane-store is its own cache line, and with 65536 streams in
// flight L2 cannot keep their half-written lines (8 MiB per XCD), so every 8- or 16-byte store left L2
// as its own sector: 70 GB written for 8.6 GB of output (profiles/r04/inflate_traffic.json).  With
// ROWD > 0 a lane collects its output in ROWD registers -- bytes [rbase, rbase + 4 ROWD) of its slot, rbase
// a multiple of 4 ROWD -- and stores a row when it is full: whole aligned pieces, written once.
// Everything is indexed statically (a dynamic register index would put the lane's state into scratch
// memory): the bytes of a step -- the copy chunk, then the literals -- are joined in six registers, shifted
// to the row's byte offset by one v_perm per dword and to its dword offset by a barrel shifter of
// log2(ROWD) select stages, and OR-ed into the row (whose bytes at and behind the write position are zero).

// keep the first k (0..16) bytes of the 16 in d[0..3]
FLATE_D void row_keep_first(uint32_t (&d)[4], uint32_t k) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = (int)k - 4 * j;
    const uint32_t nb = (uint32_t)(t < 0 ? 0 : (t > 4 ? 4 : t));
    const uint32_t m = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
    d[j] &= m;
  }
}
// x shifted left by s (0..3) bytes, the bytes of `below` coming in from underneath
FLATE_D uint32_t row_shl_bytes(uint32_t x, uint32_t below, uint32_t sel) { return __builtin_amdgcn_perm(x, below, sel); }
// a[0..5] |= the literals (lo, hi; zero beyond their count) at byte offset k (0..16)
FLATE_D void row_join(uint32_t (&a)[6], uint32_t lo, uint32_t hi, uint32_t k) {
  const uint32_t sel = 0x07060504u - 0x01010101u * (k & 3u);
  const uint32_t l0 = row_shl_bytes(lo, 0u, sel), l1 = row_shl_bytes(hi, lo, sel), l2 = row_shl_bytes(0u, hi, sel);
  const uint32_t q = k >> 2;  // 0..4 (4 only with k == 16: l2 == 0 then)
  const bool q0 = q == 0u, q1 = q == 1u, q2 = q == 2u, q3 = q == 3u, q4 = q == 4u;
  a[0] |= q0 ? l0 : 0u;
  a[1] |= q1 ? l0 : (q0 ? l1 : 0u);
  a[2] |= q2 ? l0 : (q1 ? l1 : (q0 ? l2 : 0u));
  a[3] |= q3 ? l0 : (q2 ? l1 : (q1 ? l2 : 0u));
  a[4] |= q4 ? l0 : (q3 ? l1 : (q2 ? l2 : 0u));
  a[5] |= q4 ? l1 : (q3 ? l2 : 0u);
}
// f[0 .. ROWD + 6) = the 24 bytes of a[0..5] at byte offset o (0 .. 4 ROWD - 1), zero elsewhere
template <int ROWD>
FLATE_D void row_place(uint32_t (&f)[ROWD + 6], const uint32_t (&a)[6], uint32_t o) {
  constexpr int F = ROWD + 6;
  const uint32_t sel = 0x07060504u - 0x01010101u * (o & 3u);
  uint32_t v[F];
#pragma unroll
  for (int j = 0; j < F; ++j) {
    const uint32_t x = j < 6 ? a[j] : 0u, below = (j >= 1 && j <= 6) ? a[j - 1] : 0u;
    v[j] = j <= 6 ? row_shl_bytes(x, below, sel) : 0u;
  }
  const uint32_t q = o >> 2;
#pragma unroll
  for (int bit = 1, live = 7; bit < ROWD; bit <<= 1) {  // live = entries that can be non-zero so far
    const bool on = (q & (uint32_t)bit) != 0u;
    live = live + bit < F ? live + bit : F;
#pragma unroll
    for (int j = F - 1; j >= 0; --j)
      if (j < live) v[j] = on ? (j >= bit ? v[j - bit] : 0u) : v[j];
  }
#pragma unroll
  for (int j = 0; j < F; ++j) f[j] = v[j];
}

extern __shared__ uint16_t simt_lds[];  // kLaneWords * LPW entries

}  // namespace

// LPW = streams (active lanes) per wavefront; eight 64-lane wavefronts fit the LDS of a CU, and a
// batch too small to give every SIMD one of those runs with 32 or 16 lanes per wavefront.
template <int LPW, int ROWD>
__global__ __launch_bounds__(64) void inflate_simt_kernel(InfParams P) {
  const int lane = threadIdx.x;
  const int lds_lane = lane < LPW ? lane : 0;
  const uint32_t sid = P.sid0 + blockIdx.x * (uint32_t)LPW + (uint32_t)lane;
  const bool have = lane < LPW && sid < P.n_streams;
  const LaneLds<LPW> L = {simt_lds + lds_lane};
  const LaneLens N = {P.simt_lens + ((size_t)blockIdx.x * 64 + (size_t)lane) * kLensDwords};

  // the lane's state (plain locals: they must live in registers)
  SBits b;
  uint8_t *out;
  uint32_t out_cap, opos;
  int state, err;
  bool final_block;
  int lit_min, dist_min, cl_min;
  int hdr_i, hdr_n, hdr_nlit, hdr_ndist;
  Limits lit_lim, dist_lim;       // dist_lim also serves the code-length code
  PerLen lit_thr;                 // thr[] of the literal/length code
  DistRegs dreg;                  // the distance code (the code-length code while a header is read)
  uint32_t match_len;             // S_DIST: the length decoded by S_SYM
  uint32_t copy_len, copy_dist;   // LZ77 copy in flight (S_STORED: raw bytes left, in copy_len)
  // its next (up to) 16 source bytes, requested a step ahead.  16 rather than 8: one step (and one
  // 64-byte sector each way) per match up to 16 bytes; +11 % on config 5 (profiles/r02/README.md)
  uint32_t pend_lo, pend_hi, pend_2 = 0, pend_3 = 0;
  constexpr uint32_t kChunk = 16;
  uint32_t lit_lo, lit_hi, lit_n; // up to eight literals decoded but not stored yet (they end at opos)
  // ROWD > 0: the output row (see row_place).  Bytes [rbase, wpos) of the output are in row[], not in memory
  // yet; wpos = opos minus the literals still in lit_lo / lit_hi.
  constexpr int RD = ROWD > 0 ? ROWD : 1;
  uint32_t row[RD];
  uint32_t rbase = 0, wpos = 0;
#pragma unroll
  for (int j = 0; j < RD; ++j) row[j] = 0;
  // the row to memory: ONE aligned piece while it lies inside the slot (bytes behind wpos are not final yet
  // and are written again by the row's real store), else exactly the bytes below wpos
  auto row_store = [&]() {
    if constexpr (ROWD > 0) {
      uint8_t *dst = out + rbase;
      if (rbase + 4u * ROWD <= out_cap) {
#pragma unroll
        for (int j = 0; j < ROWD; j += 4) {
          const uint4 v4 = make_uint4(row[j], row[j + 1], row[j + 2], row[j + 3]);
          __builtin_memcpy(dst + 4 * j, &v4, 16);
        }
      } else {
#pragma unroll
        for (int j = 0; j < ROWD; ++j) {
          const uint32_t p = rbase + 4u * j, w = row[j];
          if (p + 4u <= wpos) {
            __builtin_memcpy(dst + 4 * j, &w, 4);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (p + t < wpos) dst[4 * j + t] = (uint8_t)(w >> (8 * t));
          }
        }
      }
    }
  };
  out = P.out;
  out_cap = 0;
  b.in = P.in;
  b.in_len = 0;
  // spliced input: the lane's piece ends at this bit (relative to b.in); ~0 = runs to BFINAL
  uint32_t stop_bit = ~0u;
  uint64_t in_base = 0;  // byte offset of b.in inside P.in (for the reported error offset)
  uint32_t start_bit = 0;
  if (have) {
    out = P.out + P.out_off[sid];
    const uint64_t cap64 = P.out_off[sid + 1] - P.out_off[sid];
    out_cap = cap64 > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)cap64;
    if (P.bit_off == nullptr) {
      b.in = P.in + P.in_off[sid];
      b.in_len = (uint32_t)(P.in_off[sid + 1] - P.in_off[sid]);  // < 2^28: checked by the host
    } else {
      // bit positions are kept relative to the dword the piece starts in (32-bit arithmetic; byte
      // alignment relative to the whole stream is preserved, as stored blocks need it)
      const uint64_t g0 = P.bit_off[sid], g1 = P.bit_off[sid + 1];
      in_base = (g0 >> 5) * 4;
      b.in = P.in + in_base;
      start_bit = (uint32_t)(g0 - 8 * in_base);
      const uint64_t rest = P.in_len - in_base;
      b.in_len = rest < (1ull << 28) ? (uint32_t)rest : (1u << 28);
      if (sid + 1 != P.n_streams) stop_bit = (uint32_t)(g1 - 8 * in_base);  // piece < 2^28 B: host
    }
  }
  b.in_bits = b.in_len * 8u;
  b.bitpos = b.hi = b.widx = 0;
  b.w0 = b.w1 = b.w2 = b.n0 = b.n1 = 0;
  if (have) {
    sb_start(b, 0);
    b.bitpos = b.hi = start_bit;  // < 32: still inside the window's first dword
  }
  opos = 0;
  state = have ? S_BLOCK : S_DONE;
  err = 0;
  final_block = false;
  lit_min = dist_min = cl_min = 0;
  hdr_i = hdr_n = hdr_nlit = hdr_ndist = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    lit_lim.p[k] = dist_lim.p[k] = lit_thr.e[k] = dreg.delta.e[k] = us2{0, 0};
    dreg.sorted[k] = 0;
  }
  match_len = copy_len = copy_dist = 0;
  pend_lo = pend_hi = 0;
  lit_lo = lit_hi = lit_n = 0;

  // One step of a lane in S_BLOCK, S_DYN_LENS or S_STORED: block headers, table construction, raw
  // blocks -- a few hundred steps per block against thousands of symbol steps.  Loads here block.
  auto slow_step = [&]() {
    int serr = 0;
    if (state == S_DYN_LENS) {  // inflate.mbt:471-536, one code-length symbol per step
      if (hdr_i >= hdr_n) {
        // (the code-length decoder sits in the distance slots: the literal code is built first)
        const DecInit lit = sdec_init<true>(L, N, 32, hdr_nlit);
        const DecInit dst = sdec_init<false>(L, N, 32 + hdr_nlit, hdr_ndist);
        lit_lim = lit.lim;
        lit_min = lit.mn;
        lit_thr = lit.aux;
        dist_lim = dst.lim;
        dist_min = dst.mn;
        dreg.delta = dst.aux;
#pragma unroll
        for (int k = 0; k < 8; ++k) dreg.sorted[k] = dst.sorted[k];
        if (!lit.ok || !dst.ok) {
          serr = E_CORRUPT;
        } else {
          const int eob = (int)N.len_get(32 + 256);
          if (lit_min < eob) lit_min = eob;  // :542-544
          state = S_SYM;
        }
      } else {
        sb_sync(b);
        const int x = shuff_sym(b, dreg, cl_min, dist_lim, &serr);
        if (x >= 0) {
          if (x < 16) {
            N.len_set(32 + hdr_i, (uint32_t)x);
            ++hdr_i;
          } else {
            int rep = x == 18 ? 11 : 3;
            const uint32_t nb = x == 16 ? 2u : (x == 17 ? 3u : 7u);
            uint32_t fill = 0;
            if (x == 16 && hdr_i == 0) {
              serr = E_CORRUPT;
            } else {
              if (x == 16) fill = N.len_get(32 + hdr_i - 1);
              const uint32_t w = sb_peek(b);
              if (!sb_need(b, nb)) {
                serr = E_EOF;
              } else {
                sb_take(b, nb);
                rep += (int)(w & ((1u << nb) - 1u));
                if (hdr_i + rep > hdr_n) {
                  serr = E_CORRUPT;
                } else {
                  for (int j = 0; j < rep; ++j) N.len_set(32 + hdr_i + j, fill);
                  hdr_i += rep;
                }
              }
            }
          }
        }
      }
    } else if (state == S_BLOCK) {  // next_block (inflate.mbt:345-379)
      sb_sync(b);
      if (final_block) {
        state = S_DONE;
        if (stop_bit != ~0u) serr = E_CORRUPT;  // BFINAL inside a spliced stream
      } else if (b.bitpos >= stop_bit) {        // spliced input: the next piece starts here
        state = S_DONE;
        if (b.bitpos != stop_bit) serr = E_CORRUPT;  // the index does not point at a block boundary
      } else if (!sb_need(b, 3)) {
        serr = E_EOF;
      } else {
        const uint32_t h = sb_peek(b) & 7u;
        final_block = h & 1;
        const uint32_t typ = h >> 1;
        sb_take(b, 3);
        if (typ == 3) {
          serr = E_CORRUPT;
        } else if (typ == 0) {  // data_block (:708-737): header bytes follow the bytes read so far
          const uint32_t p = sb_roffset(b);
          if (b.in_len - p < 4) {
            b.hi = b.in_bits;
            serr = E_EOF;
          } else {
            b.hi = (p + 4) * 8u;
            const uint32_t n = (uint32_t)b.in[p] | ((uint32_t)b.in[p + 1] << 8);
            const uint32_t nn = (uint32_t)b.in[p + 2] | ((uint32_t)b.in[p + 3] << 8);
            if ((nn & 0xffffu) != ((~n) & 0xffffu)) {
              serr = E_CORRUPT;
            } else {
              copy_len = n;
              state = S_STORED;
            }
          }
        } else if (typ == 1) {  // fixed tables (:886-939); distances are 5-bit codes
          // lengths 8 x 144, 9 x 112, 7 x 24, 8 x 8, then 5 x 32: whole dwords of eight
          for (int i = 0; i < 40; ++i)
            N.g[i] = i < 18 ? 0x88888888u : (i < 32 ? 0x99999999u : (i < 35 ? 0x77777777u : (i < 36 ? 0x88888888u : 0x55555555u)));
          const DecInit lit = sdec_init<true>(L, N, 0, 288);
          const DecInit dst = sdec_init<false>(L, N, 288, 32);
          lit_lim = lit.lim;
          lit_min = lit.mn;
          lit_thr = lit.aux;
          dist_lim = dst.lim;
          dist_min = dst.mn;
          dreg.delta = dst.aux;
#pragma unroll
          for (int k = 0; k < 8; ++k) dreg.sorted[k] = dst.sorted[k];
          state = S_SYM;
        } else if (!sb_need(b, 14)) {  // read_huffman (:429-470)
          serr = E_EOF;
        } else {
          const uint32_t v = sb_peek(b) & 0x3fffu;
          hdr_nlit = (int)(v & 31u) + 257;
          hdr_ndist = (int)((v >> 5) & 31u) + 1;
          const int nclen = (int)((v >> 10) & 15u) + 4;
          if (hdr_nlit > kMaxLit || hdr_ndist > kMaxDist) {
            serr = E_CORRUPT;
          } else {
            sb_take(b, 14);
            sb_sync(b);
            N.g[0] = N.g[1] = N.g[2] = 0;  // the 19 lengths of the code-length code
            for (int i = 0; i < nclen && !serr; ++i) {
              if (!sb_need(b, 3)) {
                serr = E_EOF;
              } else {
                N.len_set(kCodeOrder[i], sb_peek(b) & 7u);
                sb_take(b, 3);
                sb_sync(b);
              }
            }
            if (!serr) {
              const DecInit cl = sdec_init<false>(L, N, 0, kNumCodes);
              dist_lim = cl.lim;
              cl_min = cl.mn;
              dreg.delta = cl.aux;
#pragma unroll
              for (int k = 0; k < 8; ++k) dreg.sorted[k] = cl.sorted[k];
              if (!cl.ok) {
                serr = E_CORRUPT;
              } else {
                hdr_i = 0;
                hdr_n = hdr_nlit + hdr_ndist;
                state = S_DYN_LENS;
              }
            }
          }
        }
      }
      sb_sync(b);
    } else if (state == S_STORED) {  // copy_data (:742-766): 8 raw bytes per step
      if (copy_len == 0) {
        sb_start(b, sb_roffset(b));  // restart the bit reader at the byte after the block
        state = S_BLOCK;
      } else {
        const uint32_t p = sb_roffset(b);
        const uint32_t avail = b.in_len - p;
        uint32_t n = copy_len < 8u ? copy_len : 8u;
        if (n > avail) n = avail;
        if (n > out_cap - opos) {
          serr = E_OUT_SMALL;
        } else if (n == 0) {
          serr = E_EOF;
        } else {
          if constexpr (ROWD > 0) {
            // through the literal registers (empty here: the end-of-block symbol flushed them), so that the
            // raw bytes reach the row in order in this step's phase (2)
            uint64_t raw = 0;
            for (uint32_t i = 0; i < n; ++i) raw |= (uint64_t)b.in[p + i] << (8u * i);
            lit_lo = (uint32_t)raw;
            lit_hi = (uint32_t)(raw >> 32);
            lit_n = n;
          } else {
            for (uint32_t i = 0; i < n; ++i) out[opos + i] = b.in[p + i];
          }
          opos += n;
          b.hi = (p + n) * 8u;
          copy_len -= n;
        }
      }
    }
    if (serr) {
      err = serr;
      state = S_DONE;
      copy_len = 0;  // a raw-block count must not keep the lane alive
    }
    vm_wait_all();
  };

  // One step: (1) decode -- ALU and LDS only, reads the bit window but never global memory;
  // (2) the stores, which consume the copy bytes requested at the end of the previous step, and
  // the advance of the bit window into the dword requested then; (3) the loads for the next
  // step.  The wavefront has one vmcnt counter for all lanes, so this order gives one memory
  // wait per step, overlapped with (1).  A step takes at most 95 bits minus the window offset
  // (<= 31 at its start, every decode checks what is left), so the 96-bit window never runs dry
  // and advances by at most two dwords.
  // A lane decodes its next symbol in the same step that stores the last chunk of its copy.
  for (uint32_t guard = 0; guard < 0x20000000u; ++guard) {
    if (__ballot(state != S_DONE || copy_len != 0) == 0) break;

    if ((state == S_BLOCK || state == S_DYN_LENS) ? copy_len == 0 : state == S_STORED) {
      slow_step();
    }

    uint32_t k = 0;  // bytes of the copy in flight that go out this step
    if (state != S_STORED && copy_len != 0) {
      k = copy_len < copy_dist ? copy_len : copy_dist;  // source bytes that already exist
      if (k > kChunk) k = kChunk;
    }
    const bool last_chunk = copy_len == k;
    const uint32_t copy_dst = opos;  // where phase (2) stores those k bytes
    opos += k;                       // from here on opos is the logical end of the output
    bool new_match = false;
    uint32_t new_dist = 0;

    // read_literal (inflate.mbt:565-630): literal/length symbols + length extra bits, up to
    // kSymPerStep per step -- the memory round trip at the end of a step is the same for one
    // symbol or three.  A lane goes on while it decodes literals, has room for them in lit_acc and
    // the window still holds 15+5 bits.  The checks of huff_sym and more_bits are evaluated
    // without branches, in the reference's order.
    for (int r = 0; r < kSymPerStep; ++r) {
      const bool go = state == S_SYM && last_chunk && lit_n < 8u && b.bitpos - b.widx * 32u <= 75u;
      if (__ballot(go) == 0) break;
      if (go) {
        const uint32_t w = sb_peek(b);  // >= 21 real bits
        uint32_t sym;
        const uint32_t len = canon_decode_lit(L, w, lit_lim, lit_thr, &sym);
        const bool nocode = len > 15u;
        // :590-617 in closed form: 257..264 -> 3..10; 265..284 -> ((4|(x&3)) << n) + 3, x = sym-261
        const uint32_t x = sym - 261u;
        uint32_t n = sym >= 265u ? x >> 2 : 0u;
        uint32_t base = sym >= 265u ? ((4u | (x & 3u)) << n) + 3u : sym - 254u;
        if (sym >= 285u) {
          n = 0;
          base = 258;
        }
        if (sym <= 256u) n = 0;
        const uint32_t t1 = b.bitpos + (uint32_t)lit_min, t2 = b.bitpos + len, t3 = t2 + n;
        int e = 0;
        if (sym < 256u && opos >= out_cap) e = E_OUT_SMALL;
        if (t3 > b.in_bits) e = E_EOF;
        if (sym > 285u) e = E_CORRUPT;
        if (t2 > b.in_bits) e = E_EOF;
        if (nocode) e = E_CORRUPT;
        if (t1 > b.in_bits) e = E_EOF;
        const uint32_t asked = nocode ? t1 : (sym > 285u ? t2 : t3);
        b.hi = b.hi > t1 ? b.hi : t1;
        b.hi = b.hi > asked ? b.hi : asked;
        if (e) {
          err = e;
          state = S_DONE;
        } else {
          b.bitpos = t3;
          if (sym < 256u) {
            const uint64_t put = (uint64_t)sym << (8u * lit_n);
            lit_lo |= (uint32_t)put;
            lit_hi |= (uint32_t)(put >> 32);
            ++lit_n;
            ++opos;
          }
          match_len = base + ((w >> len) & ((1u << n) - 1u));
          state = sym < 256u ? S_SYM : (sym == 256u ? S_BLOCK : S_DIST);  // 256: finish_block
        }
      }
    }
    // read_literal (:631-684): distance symbol + extra -- in the step that decoded the length
    // whenever the window still holds the 28 bits this may take (one dword crossing per step)
    const uint32_t off2 = b.bitpos - b.widx * 32u;
    if (state == S_DIST && last_chunk && off2 <= 67u) {
      const uint32_t w = sb_window(b, off2);  // >= 29 real bits
      uint32_t d;
      const uint32_t len = canon_decode_dist(w, dist_lim, dreg, &d);
      const bool nocode = len > 15u;
      const uint32_t nb = d < 4u ? 0u : (d - 2u) >> 1;
      const uint32_t dist =
          d < 4u ? d + 1u : (1u << (nb + 1u)) + 1u + ((d & 1u) << nb) + ((w >> len) & ((1u << nb) - 1u));
      const uint32_t t1 = b.bitpos + (uint32_t)dist_min, t2 = b.bitpos + len, t3 = t2 + nb;
      const uint32_t hist = opos < 32768u ? opos : 32768u;  // hist_size
      int e = 0;
      if (match_len > out_cap - opos) e = E_OUT_SMALL;
      if (dist > hist) e = E_CORRUPT;
      if (t3 > b.in_bits) e = E_EOF;
      if (d >= (uint32_t)kMaxDist) e = E_CORRUPT;
      if (t2 > b.in_bits) e = E_EOF;
      if (nocode) e = E_CORRUPT;
      if (t1 > b.in_bits) e = E_EOF;
      const uint32_t asked = nocode ? t1 : (d >= (uint32_t)kMaxDist ? t2 : t3);
      b.hi = b.hi > t1 ? b.hi : t1;
      b.hi = b.hi > asked ? b.hi : asked;
      if (e) {
        err = e;
        state = S_DONE;
      } else {
        b.bitpos = t3;
        new_match = true;
        new_dist = dist;
        state = S_SYM;
      }
    }

    // (2) consume what the previous step requested (the only wait on global memory): advance the
    // bit window, then the stores -- copy (copy_history :689 / write_copy), literals
    const uint32_t crossed = (b.bitpos >> 5) - b.widx;  // 0, 1 or 2 dwords
    if (crossed != 0) {
      const uint32_t o1 = b.w1, o2 = b.w2, o3 = b.n0, o4 = b.n1;  // (values first, see sb_window)
      b.w0 = crossed == 1u ? o1 : o2;
      b.w1 = crossed == 1u ? o2 : o3;
      b.w2 = crossed == 1u ? o3 : o4;
      b.widx += crossed;
    }
    if constexpr (ROWD > 0) {
      // this step's bytes -- the copy chunk, then the literals if they are due -- as ONE run at wpos
      const bool flush_lits = lit_n != 0 && (lit_n >= 4u || state != S_SYM || new_match);
      if (__ballot(k != 0 || flush_lits) != 0) {
        uint32_t a[6];
        {
          uint32_t d[4] = {pend_lo, pend_hi, pend_2, pend_3};
          row_keep_first(d, k);
          a[0] = d[0], a[1] = d[1], a[2] = d[2], a[3] = d[3], a[4] = 0u, a[5] = 0u;
        }
        row_join(a, flush_lits ? lit_lo : 0u, flush_lits ? lit_hi : 0u, k);
        uint32_t f[ROWD + 6];
        row_place<ROWD>(f, a, wpos - rbase);
#pragma unroll
        for (int j = 0; j < ROWD; ++j) row[j] |= f[j];
        wpos += k + (flush_lits ? lit_n : 0u);
        copy_len -= k;
        if (flush_lits) {
          lit_n = 0;
          lit_lo = lit_hi = 0;
        }
        if (wpos - rbase >= 4u * ROWD) {  // the row is full: store it, go on with what ran over
          row_store();
#pragma unroll
          for (int j = 0; j < ROWD; ++j) row[j] = j < 6 ? f[ROWD + j] : 0u;
          rbase += 4u * ROWD;
        }
      }
    } else {
    // Every lane-store is its own cache line, so stores are kept few and wide.  A store may
    // write (inside the stream's slot) past the bytes that are final: the lane's next store starts
    // right after the final ones and overwrites the rest, and nothing reads them before that.
    if (k != 0) {
      uint8_t *dst = out + copy_dst;
      if (copy_dst + 16u <= out_cap) {
        const uint4 v4 = make_uint4(pend_lo, pend_hi, pend_2, pend_3);
        __builtin_memcpy(dst, &v4, 16);
      } else if (k & 8u) {  // (only near the end of the slot)
        const uint64_t v = ((uint64_t)pend_hi << 32) | pend_lo;
        __builtin_memcpy(dst, &v, 8);
        dst += 8;
        uint32_t w = pend_2;
        if (k & 4u) {
          __builtin_memcpy(dst, &w, 4);
          dst += 4;
          w = pend_3;
        }
        if (k & 2u) {
          const uint16_t h = (uint16_t)w;
          __builtin_memcpy(dst, &h, 2);
          dst += 2;
          w >>= 16;
        }
        if (k & 1u) *dst = (uint8_t)w;
      } else if (copy_dst + 8u <= out_cap) {
        const uint64_t v = ((uint64_t)pend_hi << 32) | pend_lo;
        __builtin_memcpy(dst, &v, 8);
      } else {
        uint32_t v = pend_lo;
        if (k & 4u) {
          __builtin_memcpy(dst, &v, 4);
          dst += 4;
          v = pend_hi;
        }
        if (k & 2u) {
          const uint16_t h = (uint16_t)v;
          __builtin_memcpy(dst, &h, 2);
          dst += 2;
          v >>= 16;
        }
        if (k & 1u) *dst = (uint8_t)v;  // (k == 8 always has room: the copy was checked to fit)
      }
      copy_len -= k;
    }
    // literals collect in a register pair: one 8-byte store once there are four or more, or when
    // something else follows
    if (lit_n != 0 && (lit_n >= 4u || state != S_SYM || new_match)) {
      uint8_t *dst = out + opos - lit_n;
      const uint64_t acc = ((uint64_t)lit_hi << 32) | lit_lo;
      if (opos - lit_n + 8u <= out_cap) {
        __builtin_memcpy(dst, &acc, 8);
      } else {
        for (uint32_t i = 0; i < lit_n; ++i) dst[i] = (uint8_t)(acc >> (8u * i));
      }
      lit_n = 0;
      lit_lo = lit_hi = 0;
    }
    }
    if (new_match) {
      copy_len = match_len;
      copy_dist = new_dist;
    }
    // (3) loads for the next step
    if (crossed != 0) {
      const uint64_t v = sb_load2(b, b.widx + 3);
      b.n0 = (uint32_t)v;
      b.n1 = (uint32_t)(v >> 32);
    }
    if (state != S_STORED && copy_len != 0) {
      const uint8_t *src = out + opos - copy_dist;
      if constexpr (ROWD > 0) {
        // the 16 bytes about to be requested may reach into the row: memory must hold them first (the same
        // lane's store -> load order is the hardware's, as for every copy that reads what the lane has just
        // written)
        if (opos - copy_dist + 16u > rbase && wpos > rbase) row_store();
      }
      // (streaming loads: the history is not read again soon, and the lines they would displace
      // in L2 are the output lines the lanes are still filling)
      if (opos - copy_dist + 16u <= out_cap) {  // the read stays inside this stream's slot
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 u128u __attribute__((aligned(1)));
#ifdef FLATE_EXP_NO_HISTORY_LOAD  // TIMING EXPERIMENT ONLY (wrong bytes on purpose): what the kernel costs without its history fetches
        pend_lo = pend_hi = pend_2 = pend_3 = (uint32_t)(uintptr_t)src;
#else
        const u32x4 hv = __builtin_nontemporal_load(reinterpret_cast<const u128u *>(src));
        pend_lo = hv.x;
        pend_hi = hv.y;
        pend_2 = hv.z;
        pend_3 = hv.w;
#endif
      } else {
        uint32_t n = copy_len < copy_dist ? copy_len : copy_dist;
        if (n > 16u) n = 16u;
        uint64_t v = 0, v2 = 0;
        for (uint32_t i = 0; i < n && i < 8u; ++i) v |= (uint64_t)src[i] << (8 * i);
        for (uint32_t i = 8; i < n; ++i) v2 |= (uint64_t)src[i] << (8 * (i - 8));
        pend_lo = (uint32_t)v;
        pend_hi = (uint32_t)(v >> 32);
        pend_2 = (uint32_t)v2;
        pend_3 = (uint32_t)(v2 >> 32);
      }
    }
  }
  if constexpr (ROWD > 0) {
    if (have && wpos > rbase) row_store();  // what the row still holds (every path ends here: errors too)
  }
  if (have) {
    P.out_len[sid] = opos;
    P.status[sid] = err;
    P.err_off[sid] = err == E_CORRUPT ? (long long)(in_base + sb_roffset(b)) : -1;
  }
}
template __global__ void inflate_simt_kernel<64, 0>(InfParams);
template __global__ void inflate_simt_kernel<64, 8>(InfParams);
template __global__ void inflate_simt_kernel<64, 16>(InfParams);
template __global__ void inflate_simt_kernel<32, 0>(InfParams);
template __global__ void inflate_simt_kernel<16, 0>(InfParams);

size_t inflate_simt_lds_bytes(int lanes_per_wave) { return (size_t)kLaneWords * lanes_per_wave * sizeof(uint16_t); }
size_t inflate_simt_lens_bytes(uint32_t blocks) { return (size_t)blocks * 64 * kLensDwords * sizeof(uint32_t); }

}  // namespace flate
