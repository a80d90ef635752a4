// compact_kernels.hip -- gathers the per-stream output slots into one contiguous
// buffer and produces the (N+1)-entry offset index.  The reference has no container
// format: each stream is an independent DEFLATE stream ending in BFINAL.
#include "flate_kernels.h"

namespace flate {

// Exclusive scan of out_len (one workgroup; the index is tiny next to the payload).
__global__ __launch_bounds__(1024) void scan_sizes_kernel(CompactParams P) {
  __shared__ uint64_t wtot[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < P.n_streams; base += 1024) {
    const uint32_t i = base + (uint32_t)tid;
    const uint64_t v = i < P.n_streams ? P.out_len[i] : 0ull;
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
    if (i < P.n_streams) P.out_off[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    P.out_off[P.n_streams] = carry_s;
    if (carry_s > P.out_cap) *P.status = -2;  // FLATE_HIP_E_OUT_TOO_SMALL
  }
}

// One workgroup per stream: slot (16-byte aligned) -> out + out_off[i] (any alignment).
__global__ __launch_bounds__(256) void compact_kernel(CompactParams P) {
  if (*P.status != 0) return;
  const uint32_t sid = blockIdx.x;
  const uint64_t len = P.out_len[sid];
  const uint8_t *src = P.slots + P.slot_off[sid];
  uint8_t *dst = P.out + P.out_off[sid];
  const int tid = threadIdx.x;
  // head: bytes until dst is 4-byte aligned
  uint64_t head = (4 - ((uintptr_t)dst & 3)) & 3;
  if (head > len) head = len;
  if ((uint64_t)tid < head) dst[tid] = src[tid];
  const uint64_t body = (len - head) >> 2;  // dwords
  const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + head);
  const uint32_t shift = (uint32_t)head;  // src byte offset of d32[0] (0..3); src is aligned
  for (uint64_t j = tid; j < body; j += 256) {
    const uint32_t lo = s32[j], hi = shift ? s32[j + 1] : 0u;
    d32[j] = shift ? __builtin_amdgcn_alignbyte(hi, lo, shift) : lo;
  }
  const uint64_t done = head + (body << 2);
  if ((uint64_t)tid < len - done) dst[done + tid] = src[done + tid];
}

}  // namespace flate
