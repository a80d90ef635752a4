// splice_kernels.hip -- SURVEY.md section 8(f)-3: the streams of a batch as ONE legal DEFLATE
// stream.  The pack kernel writes every stream at its bit position in the spliced stream (and the
// closing block of Writer::close, deflate.mbt:171-176, only once); this file holds the step in
// between: from the per-stream summaries of huff_code_kernel to those bit positions.
//
// A stream without stored blocks just adds its bit count: f(x) = x + a.  A stored block pads to a
// byte boundary of the *spliced* stream (write_stored_header -> flush,
// huffman-bit-writer.mbt:474-487,139-158), so a stream with one maps the position as
// f(x) = align8(x + a) + b: a = bits in front of the first stored block + its 3 header bits,
// b = everything after that boundary (itself independent of x).  These maps are closed under
// composition, so the positions are an exclusive scan:
//   (a1,-) then (a2,-)   = (a1+a2, -)          (a1,-)  then (a2,b2) = (a1+a2, b2)
//   (a1,b1) then (a2,-)  = (a1, b1+a2)         (a1,b1) then (a2,b2) = (a1, align8(b1+a2) + b2)
#include <hip/hip_runtime.h>

#include "flate_common.h"
#include "flate_kernels.h"

namespace flate {

namespace {

constexpr uint64_t kNoStored = ~0ull;

struct PosMap {
  uint64_t a, b;  // b == kNoStored: x -> x + a; else x -> align8(x + a) + b
};
__device__ inline uint64_t align8(uint64_t x) { return (x + 7) & ~7ull; }
__device__ inline uint64_t apply(const PosMap &f, uint64_t x) {
  return f.b == kNoStored ? x + f.a : align8(x + f.a) + f.b;
}
__device__ inline PosMap then(const PosMap &f, const PosMap &g) {  // first f, then g
  PosMap r;
  if (f.b == kNoStored) {
    r.a = f.a + g.a;
    r.b = g.b;
  } else {
    r.a = f.a;
    r.b = g.b == kNoStored ? f.b + g.a : align8(f.b + g.a) + g.b;
  }
  return r;
}

}  // namespace

// stream_bit[i] = bit position of stream i's first block, stream_bit[n] = end of the last stream;
// *total_bytes = size of the spliced stream including the closing block.  One block of 1024.
__global__ __launch_bounds__(1024) void splice_scan_kernel(SpliceParams P) {
  __shared__ PosMap part[1024];
  const uint32_t t = threadIdx.x;
  const uint64_t n = P.n_streams;
  const uint64_t lo = n * t / 1024, hi = n * (t + 1) / 1024;
  PosMap f = {0, kNoStored};
  for (uint64_t i = lo; i < hi; ++i) f = then(f, PosMap{P.sum[2 * i], P.sum[2 * i + 1]});
  part[t] = f;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // inclusive scan of the chunk maps
    PosMap v = part[t];
    if (t >= d) v = then(part[t - d], v);
    __syncthreads();
    part[t] = v;
    __syncthreads();
  }
  uint64_t x = t ? apply(part[t - 1], P.start_bit) : P.start_bit;  // position in front of this thread's chunk
  for (uint64_t i = lo; i < hi; ++i) {
    P.stream_bit[i] = x;
    x = apply(PosMap{P.sum[2 * i], P.sum[2 * i + 1]}, x);
  }
  if (t == 1023) {
    P.stream_bit[n] = x;
    // closing block: 3 bits, padding, LEN, NLEN (not behind a stream that continues in a later call)
    const uint64_t bytes = P.no_close ? (align8(x) >> 3) : (align8(x + 3) >> 3) + 4;
    *P.total_bytes = bytes;
    if (bytes + 3 > P.out_cap) atomicExch(P.status, -2);  // + the rest of the last dword
  }
}

// The pack kernel ORs into the dwords a stream shares with its neighbours (its first and its last
// one) and plainly stores all others, so only those need to be zero beforehand: the dword holding
// bit stream_bit[i] and the one in front of it, for every i, plus the closing block's.
__global__ __launch_bounds__(256) void splice_zero_kernel(SpliceParams P, uint8_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i > P.n_streams || *P.status != 0) return;
  const uint64_t mis = (uint64_t)(reinterpret_cast<uintptr_t>(out) & 3u);
  uint32_t *dst = reinterpret_cast<uint32_t *>(out - mis);
  const uint64_t g = P.stream_bit[i] + 8 * mis;
  if ((g >> 5) == 0) {  // the grid's first dword starts mis bytes in front of out: not ours to write
    for (uint64_t k = mis; k < 4; ++k) out[k - mis] = 0;
  } else {
    dst[g >> 5] = 0;
    if ((g >> 5) == 1) {
      for (uint64_t k = mis; k < 4; ++k) out[k - mis] = 0;
    } else {
      dst[(g >> 5) - 1] = 0;
    }
  }
  if (i == P.n_streams && !P.no_close) {  // closing block: 3 bits, padding, LEN, NLEN -- at most 3 more dwords
    const uint64_t end = (*P.total_bytes + mis) * 8;
    for (uint64_t d = (g >> 5) + 1; d * 32 < end; ++d) dst[d] = 0;
  }
}

}  // namespace flate
