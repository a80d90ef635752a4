// compact_kernels.hip -- the (N+1)-entry offset index of the output: an exclusive scan of the
// exact per-stream sizes, so that huff_pack_kernel can write every stream in its final place.
// The reference has no container format: each stream is an independent DEFLATE
// stream ending in BFINAL; streams are laid out back to back.
#include "flate_kernels.h"

namespace flate {

// Exclusive scan of out_len (one workgroup; the index is tiny next to the payload).
__global__ __launch_bounds__(1024) void scan_sizes_kernel(CompactParams P) {
  __shared__ uint64_t wtot[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t lo = 0u, hi = P.n_streams;
  if (tid == 0) carry_s = 0ull;
  __syncthreads();
  for (uint32_t base = lo; base < hi; base += 1024) {
    const uint32_t i = base + (uint32_t)tid;
    const uint64_t v = i < hi ? P.out_len[i] : 0ull;
    uint64_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(x, d);
      if (lane >= d) x += o;
    }
    if (lane == 63) wtot[wid] = x;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    const uint64_t carry = carry_s;
    if (i < hi) P.out_off[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    P.out_off[P.n_streams] = carry_s;
    if (carry_s > P.out_cap) *P.status = -2;  // FLATE_HIP_E_OUT_TOO_SMALL
  }
}

// The small index arrays of a call (stream offsets in, sizes and statuses out) travel by THIS kernel,
// between the ctx's pinned staging memory and the device, not by hipMemcpyAsync: a copy command, however
// small, queues on a DMA engine behind whatever bulk transfer another stream of the same process has
// in flight there -- the host-pointer pipelines keep both engines busy with 100+ MiB pieces, and every
// group's kernels then sat 5-40 ms behind them waiting for 128 KiB of offsets (flate_api.hip: ctl_up /
// ctl_down).  A kernel reads and writes page-locked host memory directly.
__global__ __launch_bounds__(256) void copy_ctl_kernel(uint32_t *dst, const uint32_t *src, size_t nwords) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace flate
