// flate_kernels.h -- kernel parameter blocks and launch entry points (internal).
#pragma once

#include "flate_common.h"

namespace flate {

// A stream is cut into LZ77 chunks as Compressor::enc_speed does (reference
// deflate.mbt:236-277): every full 65535-byte window plus a final partial window
// of >= 128 bytes.  chunk_base[i] .. chunk_base[i+1] are stream i's chunks; chunk c
// owns match records [c * kMatchCapPerChunk, (c+1) * kMatchCapPerChunk).
// device-side status words that the host turns into FLATE_HIP_E_INTERNAL with a message
// (a range of their own: no device-only word may alias a public FLATE_HIP_E_* code, which some status
// words are -- FLATE_HIP_E_OUT_TOO_SMALL from the scan kernels -- and which a caller may be handed raw;
// every value <= kStatusUqTimeout is an internal condition, the pack self-check's -(0x100000 + stream) too)
constexpr int kStatusUqTimeout = -0x1001;    // uq_pop: the unit with my ticket was never pushed
constexpr int kStatusGateTimeout = -0x1002;  // (retired with the overlapped entropy stage; the number stays reserved)
constexpr int kStatusBadIndex = -0x1003;     // an index outside the scratch it addresses (never expected)
constexpr int kStatusLanesLost = -0x1004;    // a persistent loop is running without all 64 lanes
constexpr int kStatusNoProgress = -0x1005;   // lz77_stream: a batch left the parser where it found it
static_assert(kStatusUqTimeout < -64 && kStatusNoProgress > -0x100000, "device-only status words: their own range");

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
  uint32_t gtable_blocks;  // tables behind `gtables` (a guest block beyond them does nothing)
  uint32_t spin_limit;     // polls before a bounded wait gives up
  uint32_t inject_drop_push;  // test hook (option debug_drop_window_push): drop that hand-over
  uint32_t inject_stall;      // test hook (option debug_stall_batch): that dense batch of every chunk undoes its progress
  // resumable single-stream launches (flate_hip_stream_write): the stream's first LZ77 window in this
  // launch has absolute index win0 (in / in_off then describe the stream through a virtual base:
  // in + absolute position is valid for the 32 KiB of history and the new bytes); 0 otherwise
  uint32_t win0;
  // measurement aid: the launch counts the streams it took from the queue here (null = off)
  uint32_t *taken;
  // Window-granular scheduling of multi-window streams (persistent MULTI launches): the unit of
  // work is one LZ77 window of one stream.  uq_ready[k] = (queue entry + 1) << 15 | window of the
  // k-th unit to run (0 = not pushed yet); uq_ctr = {head, tail}; the table of a stream between
  // two of its windows lives in uq_tables (32 KiB per queue entry), its sweep clock in uq_sweep.
  uint32_t *uq_ready;
  uint32_t *uq_ctr;
  uint32_t uq_units;
  uint16_t *uq_tables;
  uint32_t *uq_sweep;
  int *status;
};

// Entropy stage.  Blocks are the units enc_speed writes: every full 65535-byte window plus the
// tail (blk_base[i] .. blk_base[i+1] are stream i's blocks; the first chunk_base[i+1]-chunk_base[i]
// of them are LZ77 chunks).
struct HuffParams {
  const uint8_t *in;
  const uint64_t *in_off;
  const uint32_t *chunk_base;
  const uint32_t *blk_base;  // n_streams + 1
  const uint2 *matches;
  const uint32_t *chunk_nmatch;
  const uint32_t *chunk_ntok;
  uint32_t *blk_hist;  // per block 320 u32: literal/length histogram [0,286), offsets [288,318)
  uint32_t *blk_cl;    // per block 320 u32: (len << 16) | bit-reversed code, same layout
  uint32_t *blk_hdr;   // per block 704 u32: dynamic-header items (nbits << 16) | value
  uint8_t *tile_meta;  // input / 4 bytes: one byte per lane and 256-position tile, written by
                       // huff_hist_kernel and read by huff_pack_kernel (see TileTok::pack)
  uint4 *blk_meta;     // per block {kind 0 stored / 1 huffman-only / 2 dynamic, header items, start bit lo, hi}
  // spliced mode (one DEFLATE stream for the whole batch, splice_kernels.hip); 0/NULL otherwise
  uint32_t spliced;
  uint64_t *stream_sum;          // per stream {a, b}: written by huff_code_kernel
  const uint64_t *stream_bit;    // n_streams + 1: read by huff_pack_kernel
  uint64_t *out_len;   // exact compressed bytes per stream (huff_code_kernel)
  const uint64_t *out_off;  // exclusive scan of out_len
  uint8_t *out;
  int *status;
  uint32_t n_streams;
  uint32_t compat_go;
  const uint32_t *blk_sid;  // per-block launches (huff_*_block_kernel): stream of every block, else null
  uint32_t no_close;  // spliced mode: the batch's last stream does not write Writer::close's block
                      // (a stream that continues in a later call: flate_hip_stream_write)
};

struct CompactParams {
  const uint64_t *out_len;
  uint64_t *out_off;  // n_streams + 1 (device), exclusive scan of out_len
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
  // spliced input (inflate_simt_kernel, inflate_spec_kernel): the n_streams pieces of ONE DEFLATE stream in[0, in_len);
  // piece i starts at bit bit_off[i] and ends where piece i+1 starts; the last one runs to BFINAL.
  // NULL: independent streams given by in_off.
  const uint64_t *bit_off;
  uint64_t in_len;
  uint32_t size_only;  // inflate_kernel: decode and count, store nothing (FLATE_HIP_SIZE_ONLY)
  uint32_t *simt_lens; // inflate_simt_kernel: per-lane scratch of the header being parsed (inflate_simt_lens_bytes)
  uint32_t sid0;       // inflate_simt_kernel: first stream of this launch (a batch of several rounds)
};

__global__ void lz77_serial_kernel(LzParams P);
template <bool MULTI>
__global__ void lz77_wave_kernel(LzParams P);
template <bool MULTI>
__global__ void lz77_guest_kernel(LzParams P);
// one stream continued from an earlier launch: table and sweep clock come from / go back to `table_io`,
// `clock_io`; runs the nwin windows from P.win0 on
// rebase != 0: every position the table and the clock hold is first moved down by that many bytes
// (the stream's origin was moved up: what shift_offsets does for the reference, deflate-fast.mbt:366-389)
// forget != 0: the first window of the launch starts on an empty table (shift_offsets with an empty
// `prev`, deflate-fast.mbt:367-374: what the reference does in its default compat mode)
__global__ void lz77_resume_kernel(LzParams P, uint16_t *table_io, uint32_t *clock_io, uint32_t nwin,
                                   uint32_t rebase, uint32_t forget);
__global__ void huff_hist_kernel(HuffParams P);
__global__ void huff_code_kernel(HuffParams P);
__global__ void huff_pack_kernel(HuffParams P);
// one wavefront per block instead of per stream (multi-window streams)
__global__ void huff_hist_block_kernel(HuffParams P);
__global__ void huff_zero_edges_kernel(HuffParams P, uint32_t n_blocks);
__global__ void huff_pack_block_kernel(HuffParams P);
__global__ void uq_init_kernel(uint32_t *ready, uint32_t *ctr, uint32_t n_streams, uint32_t n_units);
__global__ void scan_sizes_kernel(CompactParams P);
// small index arrays between pinned host staging and the device (see compact_kernels.hip)
__global__ void copy_ctl_kernel(uint32_t *dst, const uint32_t *src, size_t nwords);
// spins (bounded) until *counter >= target: gates a sub-batch of the entropy stage on the match
// finder that is still running on another stream
__global__ void inflate_kernel(InfParams P);
// one wavefront per stream, 64 sub-blocks of the bit stream decoded at once (inflate_spec_kernel.inc):
// <bits per sub-block, tokens per list, bytes of history ring>
template <int SUB, int CAP, int RING>
__global__ void inflate_spec_kernel(InfParams P);
#ifndef FLATE_SPEC_SMALL
#define FLATE_SPEC_SMALL 288, 61, 8192  // batches up to one wavefront per SIMD (31.4 KiB of LDS)
#endif
#ifndef FLATE_SPEC_LARGE
#define FLATE_SPEC_LARGE 224, 47, 512  // 19.9 KiB of LDS: two wavefronts per SIMD
#endif
template <int LPW, int ROWD>  // ROWD: dwords of the lane's output row (0 = none), see inflate_kernels.hip
__global__ void inflate_simt_kernel(InfParams P);
// one long stream decoded in pieces (inflate_stream_kernel.inc): the decoder's state -- the 32 KiB
// window, the tables of the block in progress, the bit carry, a copy that did not fit -- rests in
// `state` (inflate_stream_state_bytes()) between launches; its first 64 bytes are InfStreamResult
struct InfStreamResult {
  uint64_t total_in, total_out;
  int64_t err_off;
  int32_t status;  // 0 = call again, 1 = the final block is done, < 0 = error (sticky)
  uint32_t in_used, out_len, bit_in_byte;
};
size_t inflate_stream_state_bytes();
__global__ void inflate_stream_init_kernel(void *state, const uint8_t *dict, uint32_t dict_len);
__global__ void inflate_stream_kernel(void *state, const uint8_t *in, uint32_t in_len, uint32_t final_in,
                                      uint8_t *out, uint32_t out_cap);

// splice (splice_kernels.hip): bit positions of the streams inside one spliced DEFLATE stream
struct SpliceParams {
  const uint64_t *sum;    // per stream {a, b} written by huff_code_kernel (see splice_kernels.hip)
  uint64_t *stream_bit;   // n_streams + 1
  uint64_t *total_bytes;  // size of the spliced stream
  uint64_t out_cap;
  int *status;            // -2 if total_bytes > out_cap
  uint32_t n_streams;
  uint64_t start_bit;     // bit position of the first stream's first block (0; a continued stream: its carry)
  uint32_t no_close;      // no closing block behind the last stream: total_bytes = ceil(end bit / 8)
};
__global__ void splice_scan_kernel(SpliceParams P);
__global__ void splice_zero_kernel(SpliceParams P, uint8_t *out);
size_t inflate_simt_lds_bytes(int lanes_per_wave);  // dynamic LDS of that launch
size_t inflate_simt_lens_bytes(uint32_t blocks);    // global scratch of that launch (InfParams::simt_lens)

}  // namespace flate

// the ctx as the other host-side files of the library see it (gather.hip)
#include <string>
struct flate_hip_ctx;
namespace flate {
hipStream_t ctx_stream(flate_hip_ctx *c);
int ctx_device(flate_hip_ctx *c);
void ctx_set_error(flate_hip_ctx *c, const std::string &msg);
uint32_t ctx_num_cus(flate_hip_ctx *c);
// one stage's kernels between two events of the ctx (when profiling is on); ctx_stage_collect after the
// stream has been synchronised: stage_ms[stage] from them, every other stage zero
void ctx_stage_begin(flate_hip_ctx *c, int stage);
void ctx_stage_end(flate_hip_ctx *c, int stage);
int ctx_stage_collect(flate_hip_ctx *c, int stage);
// grow-only device scratch owned by the ctx (slot 0: a call's index arrays and partial results, slot 1: a
// staged copy of host input): no hipMalloc / hipFree per call -- hipFree drains the whole device
int ctx_scratch(flate_hip_ctx *c, int slot, size_t bytes, void **p);
// the ctx's pinned staging + copy kernel for small index arrays (see ctl_begin in flate_api.hip);
// ctx_ctl_finish after the stream has been synchronised
int ctx_ctl_begin(flate_hip_ctx *c, size_t up_bytes, size_t down_bytes);
int ctx_ctl_up(flate_hip_ctx *c, void *dev_dst, const void *host_src, size_t bytes);
int ctx_ctl_down(flate_hip_ctx *c, void *host_dst, const void *dev_src, size_t bytes);
void ctx_ctl_finish(flate_hip_ctx *c);
}  // namespace flate
