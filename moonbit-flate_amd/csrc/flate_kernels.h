// flate_kernels.h -- kernel parameter blocks and launch entry points (internal).
#pragma once

#include "flate_common.h"

namespace flate {

// A stream is cut into LZ77 chunks as Compressor::enc_speed does (reference
// deflate.mbt:236-277): every full 65535-byte window plus a final partial window
// of >= 128 bytes.  chunk_base[i] .. chunk_base[i+1] are stream i's chunks; chunk c
// owns match records [c * kMatchCapPerChunk, (c+1) * kMatchCapPerChunk).
struct LzParams {
  const uint8_t *in;
  const uint64_t *in_off;      // n_streams + 1
  const uint32_t *chunk_base;  // n_streams + 1
  const uint32_t *stream_ids;  // streams handled by this launch (or null = identity)
  const uint16_t *scan_off;    // probe offsets of the skip schedule
  int scan_len;
  uint2 *matches;          // {pos in chunk, token}
  uint32_t *chunk_nmatch;  // per chunk
  uint32_t *chunk_ntok;    // per chunk: literals + matches (DeflateFast::encode's token count)
  uint32_t compat_go;
  uint64_t *debug;  // diagnostic builds only (8 u64 per chunk), else null
  // guest kernel only
  void *gtables;        // one table per guest block
  uint32_t *queue;      // next index into stream_ids
  uint32_t queue_end;
};

struct HuffParams {
  const uint8_t *in;
  const uint64_t *in_off;
  const uint32_t *chunk_base;
  const uint2 *matches;
  const uint32_t *chunk_nmatch;
  const uint32_t *chunk_ntok;
  uint8_t *slots;            // per-stream output slots (16-byte aligned)
  const uint64_t *slot_off;  // n_streams + 1
  uint64_t *out_len;         // bytes produced per stream
  uint32_t n_streams;
  uint32_t compat_go;
  uint64_t *debug;  // diagnostic builds only (8 u64 per stream), else null
};

struct CompactParams {
  const uint8_t *slots;
  const uint64_t *slot_off;
  const uint64_t *out_len;
  uint64_t *out_off;  // n_streams + 1 (device), exclusive scan of out_len
  uint8_t *out;
  uint64_t out_cap;
  uint32_t n_streams;
  int *status;  // set to FLATE_HIP_E_OUT_TOO_SMALL if the total exceeds out_cap
};

struct InfParams {
  const uint8_t *in;
  const uint64_t *in_off;   // n_streams + 1
  uint8_t *out;
  const uint64_t *out_off;  // n_streams + 1: slot of every stream's output (capacity)
  uint64_t *out_len;
  int32_t *status;
  int64_t *err_off;
  uint32_t n_streams;
};

__global__ void lz77_serial_kernel(LzParams P);
template <typename E>
__global__ void lz77_wave_kernel(LzParams P);
template <typename E>
__global__ void lz77_guest_kernel(LzParams P);
__global__ void huff_pack_kernel(HuffParams P);
__global__ void scan_sizes_kernel(CompactParams P);
__global__ void compact_kernel(CompactParams P);
__global__ void inflate_kernel(InfParams P);

}  // namespace flate
