// flate_common.h -- shared constants and small helpers for the gfx950 kernels.
// Constants mirror reference deflate-fast.mbt:12-55,89-92, token.mbt:13-24 and
// huffman-bit-writer.mbt:11-85; nothing here is a translation of reference code:
// length/offset codes are computed arithmetically instead of via the 256-entry LUTs.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FLATE_HD __host__ __device__ __forceinline__
#define FLATE_D __device__ __forceinline__
#else
#define FLATE_HD inline
#endif

// Timing experiments that produce WRONG BYTES on purpose (-DFLATE_EXP_NO_INPUT, -DFLATE_EXP_NO_HISTORY_LOAD) compile
// only in a build that says what it is: -DFLATE_EXPERIMENT_BUILD, which flate_hip_build_id() reports as ";exp".
#if (defined(FLATE_EXP_NO_INPUT) || defined(FLATE_EXP_NO_HISTORY_LOAD)) && !defined(FLATE_EXPERIMENT_BUILD)
#error "FLATE_EXP_* switches corrupt the output on purpose: build them with -DFLATE_EXPERIMENT_BUILD only"
#endif

namespace flate {

// deflate-fast.mbt:12.  A scaling probe may build the library with another table size
// (-DFLATE_EXPERIMENT_TABLE_BITS=n: NOT bit-exact); such a build says so in flate_hip_build_id().
#ifdef FLATE_EXPERIMENT_TABLE_BITS
constexpr int kTableBits = FLATE_EXPERIMENT_TABLE_BITS;
#else
constexpr int kTableBits = 14;
static_assert(kTableBits == 14, "table_bits of the reference");
#endif
constexpr int kTableSize = 1 << kTableBits;     // :15
constexpr int kTableShift = 32 - kTableBits;    // :21
constexpr int kMaxMatchOffset = 1 << 15;        // :40
constexpr int kMaxStoreBlockSize = 65535;       // :46  (one LZ77 "chunk" = one window)
constexpr int kInputMargin = 15;                // :89
constexpr int kMaxMatchTail = 258 - 4;          // match_len limit (:292)
constexpr uint32_t kMatchType = 1u << 30;       // token.mbt:24
constexpr int kLengthShift = 22;                // token.mbt:13
constexpr int kMaxNumLit = 286;                 // inflate.mbt:28
constexpr int kOffsetCodeCount = 30;            // huffman-bit-writer.mbt:11
constexpr int kCodegenCodeCount = 19;           // :26
constexpr int kEndBlockMarker = 256;            // :16
constexpr int kLengthCodesStart = 257;          // :21
constexpr int kSmallHuffMin = 17;               // enc_speed: 17..127 -> write_block_huff
constexpr int kSmallLzMin = 128;                // enc_speed: >= 128 -> LZ77 (deflate.mbt:243)
constexpr int kMatchCapPerChunk = 16384;        // >= floor(65535 / 4) match records per chunk

// hash, deflate-fast.mbt:78
FLATE_HD uint32_t hash4(uint32_t u) { return (u * 0x1e35a7bdu) >> kTableShift; }

FLATE_HD int ilog2(uint32_t x) {  // x > 0
#if defined(__HIP_DEVICE_COMPILE__)
  return 31 - __clz((int)x);
#else
  return 31 - __builtin_clz(x);
#endif
}

// Length code of x = length - 3 (token.mbt:30-44,107), with the number of extra
// bits and their value (huffman-bit-writer.mbt:49-62), computed from the RFC 1951
// structure: for x >= 4, b = floor(log2 x): code = 4(b-1) + ((x >> (b-2)) & 3).
struct CodeBits {
  uint32_t code;   // length code 0..28 / offset code 0..29
  uint32_t nextra; // number of extra bits
  uint32_t extra;  // extra-bits value
};

FLATE_HD CodeBits length_code_of(uint32_t x) {
  CodeBits r;
  if (x < 8) {
    r.code = x;
    r.nextra = 0;
    r.extra = 0;
  } else if (x == 255) {
    r.code = 28;
    r.nextra = 0;
    r.extra = 0;
  } else {
    int b = ilog2(x);
    r.code = 4u * (uint32_t)(b - 1) + ((x >> (b - 2)) & 3u);
    r.nextra = (uint32_t)(b - 2);
    r.extra = x & ((1u << (b - 2)) - 1u);
  }
  return r;
}

// Offset code of d = distance - 1 (token.mbt:47-61,112-123; extra bits
// huffman-bit-writer.mbt:67-78): for d >= 4, b = floor(log2 d):
// code = 2b + ((d >> (b-1)) & 1), b-1 extra bits.
FLATE_HD CodeBits offset_code_of(uint32_t d) {
  CodeBits r;
  if (d < 4) {
    r.code = d;
    r.nextra = 0;
    r.extra = 0;
  } else {
    int b = ilog2(d);
    r.code = 2u * (uint32_t)b + ((d >> (b - 1)) & 1u);
    r.nextra = (uint32_t)(b - 1);
    r.extra = d & ((1u << (b - 1)) - 1u);
  }
  return r;
}

// Probe offsets of the skip heuristic (deflate-fast.mbt:178-187): the e-th probe
// after a scan (re)starts with skip = 32 sits at start + scan_off(e) and is
// followed by a step of scan_step(e).  Closed form for the first 67 probes.
constexpr int kScanClosedForm = 67;
FLATE_HD int scan_off_small(int e, int *step) {
  if (e < 32) {
    *step = 1;
    return e;
  }
  if (e < 48) {
    *step = 2;
    return 32 + 2 * (e - 32);
  }
  if (e < 59) {
    *step = 3;
    return 64 + 3 * (e - 48);
  }
  *step = 4;
  return 97 + 4 * (e - 59);
}

}  // namespace flate
