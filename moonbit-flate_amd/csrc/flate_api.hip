// flate_api.hip -- C ABI of libflate_hip.so (see include/flate_hip.h).
//
// Host-side driver: plans the chunking exactly as Compressor::write / enc_speed /
// close stage their 65535-byte window (reference deflate.mbt:222-294,157-183), owns
// the HBM scratch (match records, output slots, index arrays) and launches the
// kernels on one HIP stream.  There is no CPU compression path in this library.
#include "flate_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <stdexcept>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "flate_kernels.h"

using namespace flate;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct flate_hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string hip_err;
  bool profiling = false;
  float stage_ms[FLATE_HIP_STAGE_COUNT] = {0, 0, 0, 0};
  hipEvent_t ev[2 * FLATE_HIP_STAGE_COUNT] = {};
  // persistent device data
  DevBuf scan_tab;
  int scan_len = 0;
  // grow-only scratch
  DevBuf d_in, d_out, d_in_off, d_chunk_base, d_ids16, d_ids32, d_matches, d_nmatch, d_ntok;
  DevBuf d_slot_off, d_out_len, d_out_off, d_status;
  DevBuf d_blk_base, d_blk_hist, d_blk_cl, d_blk_hdr, d_blk_meta, d_tile_meta, d_blk_sid;
  // entropy stage with one wavefront per BLOCK instead of per stream: -1 = when the batch's streams
  // have three or more blocks on average (multi-window streams), 0 = never, 1 = whenever possible
  int entropy_per_block = -1;
  DevBuf d_istatus, d_ierr, d_debug, d_gtables, d_queue, d_simt_lens;
  hipStream_t guest_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int guest_blocks = 0;      // 0 = guest kernel off
  uint32_t guest_min = 1280; // below this many streams (5 per CU) the guests stay idle: one block per stream
  int32_t h_status_word = 0;  // landing pads of small async D2H copies
  uint64_t h_total_bytes = 0;
  uint32_t num_cus = 256;
  int inflate_lanes = 0;  // streams per wavefront of that inflater: 0 = by batch size, or 16/32/64
  int inflate_row = 8;    // dwords of a lane's output row in the 64-lane form (0 = stores go straight to memory, 8, 16)
  // batches at least this large use the lane-per-stream inflater: it takes ~30 ms for 64 KiB
  // streams whatever the batch size, the wave-per-stream one ~13 ms per 1024 streams (measured:
  // tools/inflate_crossover.py)
  uint32_t inflate_simt_min = 2049;
  // the speculative wave-per-stream decoder (inflate_spec_kernel): 0 = never, 1 = for batches below
  // inflate_spec_max streams (where it beats both other decoders), 2 = always (tests)
  int inflate_spec = 1;
  int inflate_spec_shape = 0;  // 0 = by batch size, 1 / 2 = always the small-batch / large-batch build (tests, tuning)
  uint32_t inflate_spec_max = 45056;  // measured (tools/inflate_crossover.py, ms per batch of 64 KiB text streams,
                                      // sub-block decoder against lane per stream; profiles/r04/inflate_crossover.txt):
                                      // 8192: 6.7 / 29.3; 16384: 13.1 / 30.8; 32768: 26.0 / 32.4; 40960: 32.4 / 33.7;
                                      // 49152: 38.8 / 35.3; 65536: 51.6 / 39.3 -- the lane-per-stream decoder wins
                                      // from ~44 k streams on (round 3, with four of its wavefronts per CU: ~37 k)
  uint32_t resident_blocks = 1024;  // persistent LDS-table blocks (4 per CU x 256 CUs)
  // (Rounds 2-4 carried an entropy stage OVERLAPPED with the match finder -- sub-batches gated on counters
  // the persistent launch incremented, in an even and an uneven form.  Never faster than running the two one
  // after the other (profiles/r02, r04), and a soak run of round 4 once saw the pack kernel's self-check fire
  // in the even form, not reproduced in 115 000 stress runs: removed)
  // window-granular scheduling of multi-window streams (lz77_kernels.hip, uq_*): on by default
  int window_units = 1;
  DevBuf d_uq_ready, d_uq_tables, d_uq_sweep;
  DevBuf d_aux[2];  // ctx_scratch (checksum.hip)
  // measurement aids (flate_hip_last_resident_share, option "profile_split_streams")
  uint32_t profile_split = 0;        // > 0: LDS-table blocks take exactly the first K queue entries,
                                     // the guest blocks the rest (two queues instead of one)
  uint32_t last_count[2] = {0, 0};   // queue lengths of the last persistent launches (16-bit, multi)
  uint32_t queue_init = 0;
  uint32_t debug_chunks = 0;
  // Host-pointer calls of the batch encoder: the batch is cut into host_groups groups of streams
  // and group g is compressed while group g+1 is copied in and the output of group g-1 is copied
  // out (two copy threads on two non-blocking streams).  0 = one copy in, compress, one copy out.
  int host_groups = 8;
  uint32_t host_group_streams = 2048;  // a group holds at least this many streams (inflate: four times as many)
  // bounded waits of the persistent kernels (uq_pop): polls before giving up
  // (a poll is one relaxed load + s_sleep, >= 0.4 us; a wave that is not running does not count)
  uint32_t spin_limit = 8u << 20;
  uint32_t inject_drop_push = 0;  // test hook: the k-th window hand-over (1-based) is dropped
  uint32_t inject_stall = 0;      // test hook: the k-th dense batch (1-based) of every chunk makes no progress
  uint64_t stream_rebase = 1ull << 30;  // flate_hip_stream: origin moved up past this many bytes
  int64_t debug_buffer_reset = 0;       // test hook: buffer_reset (deflate-fast.mbt:55) of streams opened from now on
  hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
  // Host-pointer batches run their groups on TWO lanes (sub-contexts with their own streams and
  // scratch, driven by two host threads): the persistent match-finder launch of group g+1 fills the
  // chip while group g's last streams, its entropy kernels and its size read-back drain, which a single
  // lane leaves idle (4 groups of 4096 streams: 19.9 ms of match finding against 16.2 for the batch
  // as one launch).  0 = one lane (round 3's behaviour).
  int host_lanes = 2;
  flate_hip_ctx *lane[2] = {nullptr, nullptr};
  // pinned staging of a call's small index arrays (ctl_up / ctl_down): they travel by a copy KERNEL,
  // never through the DMA engines the bulk transfers of the host-pointer pipelines occupy
  struct CtlStage {
    uint8_t *p = nullptr;
    size_t cap = 0, used = 0;
  } ctl_up_buf, ctl_down_buf;
  struct CtlPending {
    void *host_dst;
    size_t off, bytes;
  };
  std::vector<CtlPending> ctl_pending;
};

namespace flate {
hipStream_t ctx_stream(flate_hip_ctx *c) { return c->stream; }
int ctx_device(flate_hip_ctx *c) { return c->device; }
void ctx_set_error(flate_hip_ctx *c, const std::string &msg) { c->hip_err = msg; }
uint32_t ctx_num_cus(flate_hip_ctx *c) { return c->num_cus; }
void ctx_stage_begin(flate_hip_ctx *c, int stage) {
  if (c->profiling) (void)hipEventRecord(c->ev[2 * stage], c->stream);
}
void ctx_stage_end(flate_hip_ctx *c, int stage) {
  if (c->profiling) (void)hipEventRecord(c->ev[2 * stage + 1], c->stream);
}
int ctx_stage_collect(flate_hip_ctx *c, int stage) {
  for (int s = 0; s < FLATE_HIP_STAGE_COUNT; ++s) c->stage_ms[s] = 0.f;
  if (c->profiling) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[2 * stage], c->ev[2 * stage + 1]) != hipSuccess) return FLATE_HIP_E_HIP;
    c->stage_ms[stage] = ms;
  }
  return FLATE_HIP_OK;
}
}  // namespace flate

namespace {

#define HIP_TRY(ctx, expr)                                                         \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      (ctx)->hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
      return FLATE_HIP_E_HIP;                                                      \
    }                                                                              \
  } while (0)

int ensure(flate_hip_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return FLATE_HIP_OK;
  if (b.p) HIP_TRY(c, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + (bytes >> 3) + 256;
  HIP_TRY(c, hipMalloc(&b.p, want));
  b.cap = want;
  return FLATE_HIP_OK;
}

void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// ---- small index arrays: host <-> device through pinned staging and a copy kernel (copy_ctl_kernel) ----
// ctl_begin: room for the call's uploads / downloads (a staging buffer only grows between calls: the
// stream is drained first).  ctl_up: stage + launch.  ctl_down: launch into the staging; the bytes reach
// the caller's array in ctl_finish, after the stream has been synchronised.
int ctl_begin(flate_hip_ctx *c, size_t up_bytes, size_t down_bytes) {
  auto grow = [&](flate_hip_ctx::CtlStage &b, size_t need) -> int {
    b.used = 0;
    if (need <= b.cap) return FLATE_HIP_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b.p) HIP_TRY(c, hipHostFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = need + (need >> 2) + 4096;
    HIP_TRY(c, hipHostMalloc((void **)&b.p, want, hipHostMallocDefault));
    b.cap = want;
    return FLATE_HIP_OK;
  };
  c->ctl_pending.clear();
  int rc;
  if ((rc = grow(c->ctl_up_buf, up_bytes + 16 * 256))) return rc;
  return grow(c->ctl_down_buf, down_bytes + 16 * 256);
}

void ctl_launch(flate_hip_ctx *c, void *dst, const void *src, size_t bytes) {
  const size_t nwords = (bytes + 3) / 4;
  uint32_t blocks = (uint32_t)((nwords + 255) / 256);
  if (blocks > 512) blocks = 512;
  if (blocks == 0) return;
  hipLaunchKernelGGL(copy_ctl_kernel, dim3(blocks), dim3(256), 0, c->stream, (uint32_t *)dst, (const uint32_t *)src, nwords);
}

// (bytes: a multiple of 4 or rounded up to one -- every device buffer here has that slack)
int ctl_up(flate_hip_ctx *c, void *dev_dst, const void *host_src, size_t bytes) {
  if (!bytes) return FLATE_HIP_OK;
  auto &b = c->ctl_up_buf;
  const size_t at = (b.used + 255) & ~(size_t)255;
  if (at + bytes + 4 > b.cap) return FLATE_HIP_E_INTERNAL;  // (ctl_begin was given too little)
  memcpy(b.p + at, host_src, bytes);
  b.used = at + bytes;
  ctl_launch(c, dev_dst, b.p + at, bytes);
  return FLATE_HIP_OK;
}

int ctl_down(flate_hip_ctx *c, void *host_dst, const void *dev_src, size_t bytes) {
  if (!bytes) return FLATE_HIP_OK;
  auto &b = c->ctl_down_buf;
  const size_t at = (b.used + 255) & ~(size_t)255;
  if (at + bytes + 4 > b.cap) return FLATE_HIP_E_INTERNAL;
  b.used = at + bytes;
  ctl_launch(c, b.p + at, dev_src, bytes);
  c->ctl_pending.push_back({host_dst, at, bytes});
  return FLATE_HIP_OK;
}

// after hipStreamSynchronize(c->stream)
void ctl_finish(flate_hip_ctx *c) {
  for (const auto &p : c->ctl_pending) memcpy(p.host_dst, c->ctl_down_buf.p + p.off, p.bytes);
  c->ctl_pending.clear();
}

}  // namespace
namespace flate {
int ctx_scratch(flate_hip_ctx *c, int slot, size_t bytes, void **p) {
  if (slot < 0 || slot > 1) return FLATE_HIP_E_INVALID;
  const int rc = ensure(c, c->d_aux[slot], bytes + 16);
  *p = c->d_aux[slot].p;
  return rc;
}
int ctx_ctl_begin(flate_hip_ctx *c, size_t up_bytes, size_t down_bytes) { return ctl_begin(c, up_bytes, down_bytes); }
int ctx_ctl_up(flate_hip_ctx *c, void *dev_dst, const void *host_src, size_t bytes) { return ctl_up(c, dev_dst, host_src, bytes); }
int ctx_ctl_down(flate_hip_ctx *c, void *host_dst, const void *dev_src, size_t bytes) { return ctl_down(c, host_dst, dev_src, bytes); }
void ctx_ctl_finish(flate_hip_ctx *c) { ctl_finish(c); }
}  // namespace flate
namespace {

// Probe offsets of the skip heuristic (deflate-fast.mbt:178-187) from skip = 32.
std::vector<uint16_t> make_scan_table() {
  std::vector<uint16_t> t;
  uint32_t skip = 32, pos = 0;
  while (pos <= 65535) {
    t.push_back((uint16_t)pos);
    uint32_t step = skip >> 5;
    pos += step;
    skip += step;
  }
  t.push_back(65535);  // sentinel: never a legal probe
  return t;
}

struct StagePlan {
  uint32_t n_streams = 0;
  std::vector<uint32_t> chunk_base;  // n+1
  std::vector<uint32_t> ids16, ids32;
  std::vector<uint32_t> blk_base;  // n+1
  uint32_t n_chunks = 0;
  uint32_t n_blocks = 0;
};

int make_plan(const uint64_t *in_off, uint32_t n, StagePlan &pl, uint32_t flags) {
  pl.n_streams = n;
  pl.chunk_base.resize((size_t)n + 1);
  pl.blk_base.resize((size_t)n + 1);
  uint64_t chunks = 0, blocks = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (in_off[i + 1] < in_off[i]) return FLATE_HIP_E_INVALID;
    const uint64_t len = in_off[i + 1] - in_off[i];
    if (len >= 0x7ffe0000ull) return FLATE_HIP_E_TOO_LARGE;
    const uint64_t full = len / kMaxStoreBlockSize, r = len % kMaxStoreBlockSize;
    const uint64_t nch = full + (r >= (uint64_t)kSmallLzMin ? 1 : 0);
    // The reference's `cur` reaches buffer_reset at a Writer's window 32 766 (deflate-fast.mbt:55,130):
    // shift_offsets then CLEARS the table in MoonBit (`prev` is empty, :367-374).  Batch streams keep
    // their table from start to end, so a stream with an LZ77 window that far in is refused here
    // (flate_hip_stream_write follows the reference past that point); in Go's semantics the shift
    // changes no distance and the 32-bit positions above are the only limit.
    if (!(flags & FLATE_HIP_COMPAT_GO) && nch > 32766) return FLATE_HIP_E_TOO_LARGE;
    pl.chunk_base[i] = (uint32_t)chunks;
    pl.blk_base[i] = (uint32_t)blocks;
    if (nch == 1) {
      pl.ids16.push_back(i);  // one LZ77 window (it starts at 0): positions fit a 16-bit slot
    } else if (nch > 0) {
      pl.ids32.push_back(i);
    }
    chunks += nch;
    if (chunks > 0xffffffffull) return FLATE_HIP_E_TOO_LARGE;
    blocks += full + (r > 0 ? 1 : 0);
    if (blocks > 0xffffffffull) return FLATE_HIP_E_TOO_LARGE;
  }
  pl.chunk_base[n] = (uint32_t)chunks;
  pl.blk_base[n] = (uint32_t)blocks;
  pl.n_chunks = (uint32_t)chunks;
  pl.n_blocks = (uint32_t)blocks;
  return FLATE_HIP_OK;
}

struct StageTimer {
  flate_hip_ctx *c;
  int stage;
  StageTimer(flate_hip_ctx *ctx, int s) : c(ctx), stage(s) {
    if (c->profiling) (void)hipEventRecord(c->ev[2 * stage], c->stream);
  }
  ~StageTimer() {
    if (c->profiling) (void)hipEventRecord(c->ev[2 * stage + 1], c->stream);
  }
};

int collect_timing(flate_hip_ctx *c, const bool used[FLATE_HIP_STAGE_COUNT]) {
  for (int s = 0; s < FLATE_HIP_STAGE_COUNT; ++s) {
    c->stage_ms[s] = 0.f;
    if (c->profiling && used[s]) {
      float ms = 0.f;
      HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2 * s], c->ev[2 * s + 1]));
      c->stage_ms[s] = ms;
    }
  }
  return FLATE_HIP_OK;
}

// Upload the index arrays and run the match finder over every LZ77 chunk.
// (the caller has checked that the launch is one persistent resident+guest launch in stream order)
int run_lz77(flate_hip_ctx *c, const uint8_t *d_in, const uint64_t *in_off, const StagePlan &pl,
             uint32_t flags) {
  const uint32_t n = pl.n_streams;
  int rc;
  if ((rc = ensure(c, c->d_in_off, ((size_t)n + 1) * 8))) return rc;
  if ((rc = ensure(c, c->d_chunk_base, ((size_t)n + 1) * 4))) return rc;
  if ((rc = ensure(c, c->d_ids16, pl.ids16.size() * 4 + 4))) return rc;
  if ((rc = ensure(c, c->d_ids32, pl.ids32.size() * 4 + 4))) return rc;
  if ((rc = ensure(c, c->d_matches, (size_t)pl.n_chunks * kMatchCapPerChunk * sizeof(uint2) + 16)))
    return rc;
  if ((rc = ensure(c, c->d_nmatch, (size_t)pl.n_chunks * 4 + 4))) return rc;
  if ((rc = ensure(c, c->d_ntok, (size_t)pl.n_chunks * 4 + 4))) return rc;
  if ((rc = ctl_up(c, c->d_in_off.p, in_off, ((size_t)n + 1) * 8))) return rc;
  if ((rc = ctl_up(c, c->d_chunk_base.p, pl.chunk_base.data(), ((size_t)n + 1) * 4))) return rc;
  if ((rc = ctl_up(c, c->d_ids16.p, pl.ids16.data(), pl.ids16.size() * 4))) return rc;
  if ((rc = ctl_up(c, c->d_ids32.p, pl.ids32.data(), pl.ids32.size() * 4))) return rc;

  LzParams P{};  // (value-initialised: a field added later must never reach a kernel as stack garbage)
  P.in = d_in;
  P.in_off = (const uint64_t *)c->d_in_off.p;
  P.chunk_base = (const uint32_t *)c->d_chunk_base.p;
  P.stream_ids = nullptr;
  P.scan_off = (const uint16_t *)c->scan_tab.p;
  P.scan_len = c->scan_len;
  P.matches = (uint2 *)c->d_matches.p;
  P.chunk_nmatch = (uint32_t *)c->d_nmatch.p;
  P.chunk_ntok = (uint32_t *)c->d_ntok.p;
  P.compat_go = (flags & FLATE_HIP_COMPAT_GO) ? 1u : 0u;
  P.debug = nullptr;
  P.gtables = nullptr;
  P.queue = nullptr;
  P.queue_end = 0;
  P.gtable_blocks = 0;
  P.spin_limit = c->spin_limit;
  P.inject_drop_push = c->inject_drop_push;
  P.inject_stall = c->inject_stall;
  P.taken = nullptr;
  P.uq_ready = nullptr;
  P.uq_ctr = nullptr;
  P.uq_units = 0;
  P.uq_tables = nullptr;
  P.uq_sweep = nullptr;
  P.status = (int *)c->d_status.p;
  c->last_count[0] = c->last_count[1] = 0;
  // multi-window streams of a persistent launch run one window at a time (see uq_run): the
  // streams' tables rest in global memory between windows (32 KiB each; the ready word limits it
  // to 2^17 - 2 streams; more than that, or no memory for the scratch: whole-stream scheduling)
  uint32_t uq_units = 0;
  const size_t n32 = pl.ids32.size();
  const bool use_uq = c->window_units && c->guest_blocks > 0 && n32 >= c->guest_min &&
                      n32 < (1u << 17) - 1u && !(flags & FLATE_HIP_LZ_SERIAL);
  if (use_uq) {
    uint64_t units = 0;
    for (uint32_t sid : pl.ids32) units += pl.chunk_base[sid + 1] - pl.chunk_base[sid];
    if (units < 0xffffffffull) uq_units = (uint32_t)units;
  }
  if (uq_units) {
    // (grow-only scratch, 32 KiB per multi-window stream: when the device cannot give it, the
    // launch falls back to whole-stream scheduling, which needs none)
    if (ensure(c, c->d_uq_ready, (size_t)uq_units * 4 + 64) != FLATE_HIP_OK ||
        ensure(c, c->d_uq_tables, n32 * (size_t)kTableSize * 2 + 64) != FLATE_HIP_OK ||
        ensure(c, c->d_uq_sweep, n32 * 4 + 64) != FLATE_HIP_OK) {
      (void)hipGetLastError();
      c->hip_err.clear();
      uq_units = 0;
    }
  }
  if (c->guest_blocks > 0) {
    if ((rc = ensure(c, c->d_gtables, (size_t)c->guest_blocks * kTableSize * 2 + 64))) return rc;
    if ((rc = ensure(c, c->d_queue, 64))) return rc;
    // words: [0..1] stream queues (single-, multi-window), [2..3] the guests' queues of a fixed
    // profiling split, [4..5] what the LDS-table launches took, [6..7] window-unit head / tail
    HIP_TRY(c, hipMemsetAsync(c->d_queue.p, 0, 32, c->stream));
  }
#if defined(FLATE_LZ_STAMPS) || defined(FLATE_LZ_FINISH)
  if ((rc = ensure(c, c->d_debug, (size_t)pl.n_chunks * 64 + 64))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_debug.p, 0, (size_t)pl.n_chunks * 64, c->stream));
  P.debug = (uint64_t *)c->d_debug.p;
  c->debug_chunks = pl.n_chunks;
#endif
  {
    StageTimer t(c, FLATE_HIP_STAGE_LZ77);
    if (flags & FLATE_HIP_LZ_SERIAL) {
      if (!pl.ids16.empty()) {
        P.stream_ids = (const uint32_t *)c->d_ids16.p;
        hipLaunchKernelGGL(lz77_serial_kernel, dim3((uint32_t)pl.ids16.size()), dim3(64), 0,
                           c->stream, P);
      }
      if (!pl.ids32.empty()) {
        P.stream_ids = (const uint32_t *)c->d_ids32.p;
        hipLaunchKernelGGL(lz77_serial_kernel, dim3((uint32_t)pl.ids32.size()), dim3(64), 0,
                           c->stream, P);
      }
    } else {
      // single-window streams (ids16) and multi-window streams (ids32) use the same 32 KiB
      // 16-bit tables; the latter add the periodic sweep (MULTI)
      auto launch = [&](const DevBuf &ids, uint32_t count, bool multi, uint32_t queue_slot) {
        if (!count) return;
        P.stream_ids = (const uint32_t *)ids.p;
        const bool guests = c->guest_blocks > 0 && count >= c->guest_min;
        if (!guests) {
          if (multi)
            hipLaunchKernelGGL(lz77_wave_kernel<true>, dim3(count), dim3(64), 0, c->stream, P);
          else
            hipLaunchKernelGGL(lz77_wave_kernel<false>, dim3(count), dim3(64), 0, c->stream, P);
          return;
        }
        // fork: resident (LDS-table) and guest (L2-table) blocks pull streams from one queue
        LzParams G = P;
        G.gtables = c->d_gtables.p;
        G.gtable_blocks = (uint32_t)c->guest_blocks;  // d_gtables holds exactly this many tables
        G.queue = (uint32_t *)c->d_queue.p + queue_slot;
        G.queue_end = count;
        c->last_count[queue_slot] = count;
        if (multi && uq_units) {
          G.uq_ready = (uint32_t *)c->d_uq_ready.p;
          G.uq_ctr = (uint32_t *)c->d_queue.p + 6;  // {head, tail}
          G.uq_units = uq_units;
          G.uq_tables = (uint16_t *)c->d_uq_tables.p;
          G.uq_sweep = (uint32_t *)c->d_uq_sweep.p;
          c->last_count[queue_slot] = uq_units;
          hipLaunchKernelGGL(uq_init_kernel, dim3((uq_units + 255) / 256), dim3(256), 0, c->stream,
                             G.uq_ready, G.uq_ctr, count, uq_units);
        }
        LzParams R = G;  // the LDS-table launch counts what it takes
        R.taken = (uint32_t *)c->d_queue.p + 4 + queue_slot;
        if (c->profile_split > 0 && c->profile_split < count && !G.uq_ready) {
          // measurement aid: a fixed split instead of the shared queue, so that a profiler that
          // serialises the two kernels still sees each of them do its share of the work
          R.queue_end = c->profile_split;
          G.queue = (uint32_t *)c->d_queue.p + 2 + queue_slot;
          G.stream_ids = P.stream_ids + c->profile_split;
          G.queue_end = count - c->profile_split;
        }
        (void)hipEventRecord(c->ev_fork, c->stream);
        (void)hipStreamWaitEvent(c->guest_stream, c->ev_fork, 0);
        uint32_t resident = c->resident_blocks < count ? c->resident_blocks : count;
        // The LDS-table kernel is submitted FIRST: its blocks need 26 contiguous LDS granules each, and guests that
        // reach a CU before them can leave it with room for three (measured: -1.4 % with this order, section 7 of
        // profiles/r05/README.md).
        if (multi) {
          hipLaunchKernelGGL(lz77_wave_kernel<true>, dim3(resident), dim3(64), 0, c->stream, R);
          hipLaunchKernelGGL(lz77_guest_kernel<true>, dim3((uint32_t)c->guest_blocks), dim3(64), 0,
                             c->guest_stream, G);
          (void)hipEventRecord(c->ev_join, c->guest_stream);
        } else {
          hipLaunchKernelGGL(lz77_wave_kernel<false>, dim3(resident), dim3(64), 0, c->stream, R);
          hipLaunchKernelGGL(lz77_guest_kernel<false>, dim3((uint32_t)c->guest_blocks), dim3(64), 0,
                             c->guest_stream, G);
          (void)hipEventRecord(c->ev_join, c->guest_stream);
        }
        (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
      };
      launch(c->d_ids16, (uint32_t)pl.ids16.size(), false, 0);
      launch(c->d_ids32, (uint32_t)pl.ids32.size(), true, 1);
    }
  }
  // (test hook: it loses one hand-over of THIS launch, not of every later one)
  if (uq_units) c->inject_drop_push = 0;
  HIP_TRY(c, hipGetLastError());
  return FLATE_HIP_OK;
}

}  // namespace

extern "C" {

const char *flate_hip_strerror(int code) {
  switch (code) {
    case FLATE_HIP_OK: return "ok";
    case FLATE_HIP_E_INVALID: return "invalid argument";
    case FLATE_HIP_E_OUT_TOO_SMALL: return "output buffer too small";
    case FLATE_HIP_E_HIP: return "HIP runtime error";
    case FLATE_HIP_E_CORRUPT: return "flate: corrupt input";
    case FLATE_HIP_E_NO_DEVICE: return "no usable HIP device (this engine has no CPU path)";
    case FLATE_HIP_E_TOO_LARGE: return "stream too large";
    case FLATE_HIP_E_UNEXPECTED_EOF: return "unexpected EOF";
    case FLATE_HIP_E_INTERNAL: return "internal error: encoder self-check failed";
    case FLATE_HIP_E_AGAIN: return "a shard outgrew the agreed plan (pad or stream count): repeat this batch with the blocking exchange";
    default: return "unknown error";
  }
}

#ifndef FLATE_HIP_BUILD_ID
#define FLATE_HIP_BUILD_ID "unknown"
#endif
#define FLATE_STR2(x) #x
#define FLATE_STR(x) FLATE_STR2(x)
#ifdef FLATE_EXPERIMENT_BUILD  // (a build that may contain FLATE_EXP_* switches: never a measurement's or a product's id)
#define FLATE_ID_EXP ";exp"
#else
#define FLATE_ID_EXP ""
#endif
#ifdef FLATE_EXPERIMENT_TABLE_BITS
const char *flate_hip_build_id(void) {
  return FLATE_HIP_BUILD_ID FLATE_ID_EXP ";NOT-BIT-EXACT:table_bits=" FLATE_STR(FLATE_EXPERIMENT_TABLE_BITS);
}
#else
const char *flate_hip_build_id(void) { return FLATE_HIP_BUILD_ID FLATE_ID_EXP; }
#endif

const char *flate_hip_last_hip_error(const flate_hip_ctx *ctx) {
  return ctx ? ctx->hip_err.c_str() : "";
}

const char *flate_hip_stage_name(int stage) {
  switch (stage) {
    case FLATE_HIP_STAGE_LZ77: return "lz77_match";
    case FLATE_HIP_STAGE_HUFF_PACK: return "huff_pack";
    case FLATE_HIP_STAGE_CHECKSUM: return "checksum";
    case FLATE_HIP_STAGE_INFLATE: return "inflate";
    default: return "?";
  }
}

int flate_hip_init(int device, flate_hip_ctx **out) {
  if (!out) return FLATE_HIP_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return FLATE_HIP_E_NO_DEVICE;
  flate_hip_ctx *c = new flate_hip_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess) {
    flate_hip_destroy(c);
    return FLATE_HIP_E_NO_DEVICE;
  }
  c->stream = c->own_stream;
  // The match finder is TWO kernels that must run side by side (LDS-table blocks on c->stream, guest blocks
  // on guest_stream, one queue of streams between them).  HIP spreads its streams over a few hardware
  // queues (four by default) round robin; two streams that land on the same one run their kernels one
  // after the other -- measured: the match finder of a sub-context 40 % slower (19.8 -> 27.9 ms per GiB in
  // 4096-stream launches) when the number of streams created before it shifted by one.  Streams of a
  // different PRIORITY come from a different pool of hardware queues, so the guest stream asks for one:
  // it can never share a queue with the (normal-priority) stream of the kernel it runs beside.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const char *gp = getenv("FLATE_HIP_GUEST_STREAM_PRIORITY");  // (developer A/B: "normal" = as before)
  const bool plain_guest = (gp && gp[0] == 'n') || prio_greatest == prio_least;
  if ((plain_guest ? hipStreamCreateWithFlags(&c->guest_stream, hipStreamNonBlocking)
                   : hipStreamCreateWithPriority(&c->guest_stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    flate_hip_destroy(c);
    return FLATE_HIP_E_HIP;
  }
  {  // 4 resident (LDS-table) + 6 guest (L2-table, 4 KiB of LDS slot tags each) match-finder waves per CU:
     // what a CU's LDS granules hold, and the measured optimum (profiles/r05/README.md sections 6-7)
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    c->num_cus = (uint32_t)cus;
    c->guest_min = 5u * (uint32_t)cus;  // measured: 1024 streams 1.72 ms as one block per stream vs 2.15 ms
                                        // persistent; 1280: 3.41 vs 2.35; 2048: 3.61 vs 2.76; 3072: 5.55 vs 3.96
    // LDS comes in 128 granules of 1280 B per CU: an LDS-table block (32768 B) takes 26, a guest (4096 B of slot
    // tags) 4, so 4 + 6 blocks fill a CU exactly.  Launching MORE guests than fit (rounds 2-4 asked for 6.5 per CU)
    // lets guests that arrive first take the granules of an LDS-table block: most processes then ran 3.5 + 6.5
    // blocks per CU and the match finder 4 % slower (profiles/r05/README.md section 7).  Ask for what fits.
    c->resident_blocks = 4u * (uint32_t)cus;
    c->guest_blocks = 6 * cus;
  }
  if (const char *e = getenv("FLATE_HIP_GUEST_BLOCKS")) c->guest_blocks = atoi(e) < 0 ? 0 : atoi(e);
  if (const char *e = getenv("FLATE_HIP_GUEST_MIN")) c->guest_min = (uint32_t)atoi(e);
  if (const char *e = getenv("FLATE_HIP_RESIDENT_BLOCKS")) c->resident_blocks = atoi(e) < 1 ? 1u : (uint32_t)atoi(e);
  for (auto &e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      flate_hip_destroy(c);
      return FLATE_HIP_E_HIP;
    }
  std::vector<uint16_t> tab = make_scan_table();
  c->scan_len = (int)tab.size();
  if (ensure(c, c->scan_tab, tab.size() * 2) != FLATE_HIP_OK ||
      hipMemcpy(c->scan_tab.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
      ensure(c, c->d_status, 16) != FLATE_HIP_OK) {
    flate_hip_destroy(c);
    return FLATE_HIP_E_HIP;
  }
  *out = c;
  return FLATE_HIP_OK;
}

void flate_hip_destroy(flate_hip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  for (auto &l : c->lane) {
    flate_hip_destroy(l);
    l = nullptr;
  }
  if (c->ctl_up_buf.p) (void)hipHostFree(c->ctl_up_buf.p);
  if (c->ctl_down_buf.p) (void)hipHostFree(c->ctl_down_buf.p);
  if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  for (DevBuf *b : {&c->scan_tab, &c->d_in, &c->d_out, &c->d_in_off, &c->d_chunk_base, &c->d_ids16,
                    &c->d_ids32, &c->d_matches, &c->d_nmatch, &c->d_ntok, &c->d_blk_base,
                    &c->d_blk_hist, &c->d_blk_cl, &c->d_blk_hdr, &c->d_blk_meta, &c->d_tile_meta, &c->d_blk_sid, &c->d_slot_off, &c->d_out_len, &c->d_out_off, &c->d_status, &c->d_istatus,
                    &c->d_ierr, &c->d_debug, &c->d_gtables, &c->d_queue, &c->d_simt_lens})
    release(*b);
  for (auto &e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->guest_stream) (void)hipStreamDestroy(c->guest_stream);
  release(c->d_uq_ready);
  release(c->d_uq_tables);
  release(c->d_uq_sweep);
  release(c->d_aux[0]);
  release(c->d_aux[1]);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int flate_hip_set_stream(flate_hip_ctx *c, void *hip_stream) {
  if (!c) return FLATE_HIP_E_INVALID;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return FLATE_HIP_OK;
}

int flate_hip_set_option(flate_hip_ctx *c, const char *name, int64_t value) {
  if (!c || !name) return FLATE_HIP_E_INVALID;
  const std::string k(name);
  if (k == "guest_blocks" && value >= 0 && value <= 65536) {
    c->guest_blocks = (int)value;
  } else if (k == "guest_min_streams" && value >= 0) {
    c->guest_min = (uint32_t)value;
  } else if (k == "inflate_lanes" && (value == 0 || value == 16 || value == 32 || value == 64)) {
    c->inflate_lanes = (int)value;
  } else if (k == "inflate_row_dwords" && (value == 0 || value == 8 || value == 16)) {
    c->inflate_row = (int)value;
  } else if (k == "inflate_simt_min_streams" && value >= 0) {
    c->inflate_simt_min = (uint32_t)value;
  } else if (k == "inflate_spec" && value >= 0 && value <= 2) {
    c->inflate_spec = (int)value;
  } else if (k == "inflate_spec_shape" && value >= 0 && value <= 2) {
    c->inflate_spec_shape = (int)value;
  } else if (k == "inflate_spec_max_streams" && value >= 0 && value <= 0x7fffffff) {
    c->inflate_spec_max = (uint32_t)value;
  } else if (k == "resident_blocks" && value > 0 && value <= 65536) {
    c->resident_blocks = (uint32_t)value;
  } else if (k == "host_pipeline_groups" && value >= 0 && value <= 64) {
    c->host_groups = (int)value;
  } else if (k == "host_pipeline_group_streams" && value > 0 && value <= 0x7fffffff) {
    c->host_group_streams = (uint32_t)value;
  } else if (k == "host_pipeline_lanes" && value >= 1 && value <= 2) {
    c->host_lanes = (int)value;
  } else if (k == "profile_split_streams" && value >= 0 && value <= 0x7fffffff) {
    c->profile_split = (uint32_t)value;
  } else if (k == "window_units" && (value == 0 || value == 1)) {
    c->window_units = (int)value;
  } else if (k == "spin_limit_polls" && value > 0 && value <= 0x7fffffff) {
    c->spin_limit = (uint32_t)value;
  } else if (k == "entropy_per_block" && value >= -1 && value <= 1) {
    c->entropy_per_block = (int)value;
  } else if (k == "stream_rebase_bytes" && value >= 65535 && value <= (1ll << 30)) {
    c->stream_rebase = (uint64_t)value;
  } else if (k == "debug_drop_window_push" && value >= 0 && value <= 0x7fffffff) {
    c->inject_drop_push = (uint32_t)value;
  } else if (k == "debug_stall_batch" && value >= 0 && value <= 0x7fffffff) {
    c->inject_stall = (uint32_t)value;
  } else if (k == "debug_buffer_reset" && value >= 0 && value <= 0x7fffffff) {
    c->debug_buffer_reset = value;

  } else {
    return FLATE_HIP_E_INVALID;
  }
  return FLATE_HIP_OK;
}

// Host buffers the caller keeps across calls (a Writer's input buffer, a Reader's output buffer): once
// page-locked, the copies of the host-pointer calls are DMA transfers at the link's rate; from pageable
// memory the runtime stages every copy through bounce buffers of its own.  Nothing else changes: the
// copy paths hand the same pointers to hipMemcpyAsync, which knows the registered ranges.
int flate_hip_host_register(flate_hip_ctx *c, void *p, size_t bytes) {
  if (!c || !p || !bytes) return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
  return FLATE_HIP_OK;
}

int flate_hip_host_unregister(flate_hip_ctx *c, void *p) {
  if (!c || !p) return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipHostUnregister(p));
  return FLATE_HIP_OK;
}

int flate_hip_host_alloc(flate_hip_ctx *c, size_t bytes, void **out) {
  if (!c || !out || !bytes) return FLATE_HIP_E_INVALID;
  *out = nullptr;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return FLATE_HIP_OK;
}

int flate_hip_host_free(flate_hip_ctx *c, void *p) {
  if (!c) return FLATE_HIP_E_INVALID;
  if (!p) return FLATE_HIP_OK;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipHostFree(p));
  return FLATE_HIP_OK;
}

int flate_hip_last_resident_share(flate_hip_ctx *c, uint32_t *resident_streams, uint32_t *queued_streams) {
  if (!c || !resident_streams || !queued_streams) return FLATE_HIP_E_INVALID;
  *resident_streams = *queued_streams = 0;
  if (!c->d_queue.p) return FLATE_HIP_OK;
  uint32_t q[8] = {0};
  HIP_TRY(c, hipMemcpy(q, c->d_queue.p, 32, hipMemcpyDeviceToHost));
  *resident_streams = q[4] + q[5];
  *queued_streams = c->last_count[0] + c->last_count[1];
  return FLATE_HIP_OK;
}

int flate_hip_set_profiling(flate_hip_ctx *c, int on) {
  if (!c) return FLATE_HIP_E_INVALID;
  c->profiling = on != 0;
  return FLATE_HIP_OK;
}

int flate_hip_last_timing(flate_hip_ctx *c, float *ms, int n) {
  if (!c || !ms) return FLATE_HIP_E_INVALID;
  for (int i = 0; i < n && i < FLATE_HIP_STAGE_COUNT; ++i) ms[i] = c->stage_ms[i];
  return FLATE_HIP_OK;
}

// Worst case of one stream: every window Huffman-coded with matches (< 15 bits per
// byte), a <= 320-byte dynamic header per window, 5 bytes per stored block.
size_t flate_hip_deflate_bound(size_t n) {
  const size_t windows = n / kMaxStoreBlockSize + 1;
  return n * 2 + windows * 320 + 16;
}

}  // extern "C"

// Both encode entry points.  spliced: the whole batch becomes one DEFLATE stream; out_off then
// receives the bit position of every stream (may be NULL) and *total_bytes the size.
static int deflate_common(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                          uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t flags,
                          bool spliced, uint64_t *total_bytes) {
  HIP_TRY(c, hipSetDevice(c->device));
  StagePlan pl;
  int rc = make_plan(in_off, n, pl, flags);
  if (rc) return rc;
  const bool dev = (flags & FLATE_HIP_DEVICE_PTRS) != 0;
  const uint64_t in_bytes = in_off[n];
  // the call's index arrays: in_off, chunk_base, blk_base (n + 1 each), the two stream lists, blk_sid
  if ((rc = ctl_begin(c, ((size_t)n + 1) * 16 + (pl.ids16.size() + pl.ids32.size() + (size_t)pl.n_blocks) * 4,
                      ((size_t)n + 1) * 8 + 64)))
    return rc;

  const uint8_t *d_in = in;
  uint8_t *d_out = out;
  if (!dev) {
    if ((rc = ensure(c, c->d_in, in_bytes + 16))) return rc;
    if ((rc = ensure(c, c->d_out, out_cap + 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in.p, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    d_in = (const uint8_t *)c->d_in.p;
    d_out = (uint8_t *)c->d_out.p;
  }
  const size_t nb = (size_t)pl.n_blocks + 1;
  if ((rc = ensure(c, c->d_blk_base, ((size_t)n + 1) * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_hist, nb * 320 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_cl, nb * 320 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_hdr, nb * 704 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_meta, nb * 16))) return rc;
  if ((rc = ensure(c, c->d_tile_meta, ((in_bytes >> 8) + nb + 2) * 64))) return rc;  // (tile_meta_at)
  if ((rc = ensure(c, c->d_out_len, (size_t)n * 8 + 8))) return rc;
  if ((rc = ensure(c, c->d_out_off, ((size_t)n + 1) * 8))) return rc;
  if (spliced) {
    if ((rc = ensure(c, c->d_slot_off, ((size_t)n + 1) * 16))) return rc;  // stream summaries {a, b}
  }
  if ((rc = ctl_up(c, c->d_blk_base.p, pl.blk_base.data(), ((size_t)n + 1) * 4))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_status.p, 0, 4, c->stream));
  // One wavefront per block in the histogram and pack kernels when the streams have many blocks
  // (4096 streams of four windows are 4096 wavefronts per stream-kernel, a quarter of what fills
  // the chip).  Every stream needs at least one block (a stream without any has nobody to write
  // its closing block in that form).  Not for spliced output: where a block starts then depends on
  // the bit its stream starts at (a stored block pads to a byte of the SPLICED stream), which only the
  // stream's own walk knows.
  bool per_block = c->entropy_per_block != 0 && pl.n_blocks > 0 && !spliced &&
                   (c->entropy_per_block == 1 || (uint64_t)pl.n_blocks >= 3ull * n);
  for (uint32_t i = 0; i < n && per_block; ++i) per_block = pl.blk_base[i + 1] > pl.blk_base[i];
  std::vector<uint32_t> blk_sid;
  if (per_block) {
    blk_sid.resize(pl.n_blocks);
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t b = pl.blk_base[i]; b < pl.blk_base[i + 1]; ++b) blk_sid[b] = i;
    if ((rc = ensure(c, c->d_blk_sid, (size_t)pl.n_blocks * 4 + 4))) return rc;
    if ((rc = ctl_up(c, c->d_blk_sid.p, blk_sid.data(), (size_t)pl.n_blocks * 4))) return rc;
  }

  if ((rc = run_lz77(c, d_in, in_off, pl, flags))) return rc;

  HuffParams H{};
  H.in = d_in;
  H.in_off = (const uint64_t *)c->d_in_off.p;
  H.chunk_base = (const uint32_t *)c->d_chunk_base.p;
  H.blk_base = (const uint32_t *)c->d_blk_base.p;
  H.matches = (const uint2 *)c->d_matches.p;
  H.chunk_nmatch = (const uint32_t *)c->d_nmatch.p;
  H.chunk_ntok = (const uint32_t *)c->d_ntok.p;
  H.blk_hist = (uint32_t *)c->d_blk_hist.p;
  H.blk_cl = (uint32_t *)c->d_blk_cl.p;
  H.blk_hdr = (uint32_t *)c->d_blk_hdr.p;
  H.blk_meta = (uint4 *)c->d_blk_meta.p;
  H.tile_meta = (uint8_t *)c->d_tile_meta.p;
  H.spliced = spliced ? 1u : 0u;
  H.stream_sum = spliced ? (uint64_t *)c->d_slot_off.p : nullptr;
  H.stream_bit = spliced ? (const uint64_t *)c->d_out_off.p : nullptr;
  H.out_len = (uint64_t *)c->d_out_len.p;
  H.out_off = (const uint64_t *)c->d_out_off.p;
  H.out = d_out;
  H.status = (int *)c->d_status.p;
  H.n_streams = n;
  H.compat_go = (flags & FLATE_HIP_COMPAT_GO) ? 1u : 0u;
  H.blk_sid = per_block ? (const uint32_t *)c->d_blk_sid.p : nullptr;
  CompactParams C{};
  C.out_len = (const uint64_t *)c->d_out_len.p;
  C.out_off = (uint64_t *)c->d_out_off.p;
  C.out_cap = out_cap;
  C.n_streams = n;
  C.status = (int *)c->d_status.p;
  {
    StageTimer t(c, FLATE_HIP_STAGE_HUFF_PACK);
    if (per_block)
      hipLaunchKernelGGL(huff_hist_block_kernel, dim3(pl.n_blocks), dim3(64), 0, c->stream, H);
    else
      hipLaunchKernelGGL(huff_hist_kernel, dim3(n), dim3(64), 0, c->stream, H);
    hipLaunchKernelGGL(huff_code_kernel, dim3(n), dim3(64), 0, c->stream, H);
    if (!spliced) {
      hipLaunchKernelGGL(scan_sizes_kernel, dim3(1), dim3(1024), 0, c->stream, C);
    } else {
      SpliceParams S{};
      S.sum = (const uint64_t *)c->d_slot_off.p;
      S.stream_bit = (uint64_t *)c->d_out_off.p;
      S.total_bytes = (uint64_t *)c->d_out_len.p + n;  // (d_out_len has n + 1 slots)
      S.out_cap = out_cap;
      S.status = (int *)c->d_status.p;
      S.n_streams = n;
      hipLaunchKernelGGL(splice_scan_kernel, dim3(1), dim3(1024), 0, c->stream, S);
      hipLaunchKernelGGL(splice_zero_kernel, dim3(n / 256 + 1), dim3(256), 0, c->stream, S, d_out);
    }
    if (per_block) {
      hipLaunchKernelGGL(huff_zero_edges_kernel, dim3(pl.n_blocks / 256 + 1), dim3(256), 0, c->stream, H,
                         pl.n_blocks);
      hipLaunchKernelGGL(huff_pack_block_kernel, dim3(pl.n_blocks), dim3(64), 0, c->stream, H);
    } else {
      hipLaunchKernelGGL(huff_pack_kernel, dim3(n), dim3(64), 0, c->stream, H);
    }
  }
  HIP_TRY(c, hipGetLastError());

  uint64_t produced = 0;
  if (out_off && (rc = ctl_down(c, out_off, c->d_out_off.p, ((size_t)n + 1) * 8))) return rc;
  if (spliced && (rc = ctl_down(c, &c->h_total_bytes, (uint64_t *)c->d_out_len.p + n, 8))) return rc;
  if ((rc = ctl_down(c, &c->h_status_word, c->d_status.p, 4))) return rc;
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  ctl_finish(c);
  if (c->h_status_word) {
    if (c->h_status_word == kStatusNoProgress) {
      c->hip_err = "a match-finder batch made no progress (the chunk was abandoned)";
      return FLATE_HIP_E_INTERNAL;
    }
    if (c->h_status_word == kStatusLanesLost) {
      c->hip_err = "a persistent match-finder loop lost lanes of its wavefront (miscompiled loop?)";
      return FLATE_HIP_E_INTERNAL;
    }
    if (c->h_status_word == kStatusUqTimeout || c->h_status_word == kStatusBadIndex) {
      c->hip_err = c->h_status_word == kStatusUqTimeout
                       ? "a match-finder block waited for a window that was never handed over"
                       : "a match-finder block was handed an index outside its scratch";
      return FLATE_HIP_E_INTERNAL;
    }
    if (c->h_status_word <= -0x100000) {  // encoder self-check (huff_pack_kernel): -(0x100000 + stream)
      c->hip_err = "packed bits differ from the computed block size in stream " +
                   std::to_string((uint32_t)(-c->h_status_word) - 0x100000u) + " (mod 2^20)";
      return FLATE_HIP_E_INTERNAL;
    }
    return c->h_status_word;
  }
  produced = spliced ? c->h_total_bytes : out_off[n];
  if (total_bytes) *total_bytes = produced;
  if (!dev) {
    HIP_TRY(c, hipMemcpyAsync(out, c->d_out.p, produced, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  const bool used[FLATE_HIP_STAGE_COUNT] = {true, true, false, false};
  return collect_timing(c, used);
}

// FLATE_HIP_TRACE_HOST=1: timestamps of the host-pointer pipeline's stages on stderr (developer aid)
static bool host_trace_on() {
  static const bool on = getenv("FLATE_HIP_TRACE_HOST") != nullptr;
  return on;
}
static double host_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void host_trace(double t0, const char *what, unsigned g, double a, double b) {
  if (host_trace_on()) fprintf(stderr, "[host-pipe] %-8s g=%u  %.2f .. %.2f ms\n", what, g, a - t0, b - t0);
}

// ---- host-pointer batches, pipelined over groups of streams (flate_hip_ctx::host_groups) ----
// Two copy threads beside the calling thread: one brings the groups' input to the device in
// order, the other takes every group's output back as soon as the caller posts it.  Each uses its
// own non-blocking HIP stream, so the copies run beside the kernels of the group in between.
namespace {
struct CopyJob {
  void *dst;
  const void *src;
  size_t bytes;
};
class CopyPipe {
 public:
  CopyPipe(size_t n_in, size_t n_out) : in_ready_(n_in, 0), out_state_(n_out, 0), out_jobs_(n_out) {}
  // (not in the constructor: if the second thread cannot be created, the destructor must still run
  // to join the first)
  void start(int device, hipStream_t s_in, hipStream_t s_out, std::vector<CopyJob> in_jobs) {
    const size_t n_out = out_jobs_.size();
    t_in_ = std::thread([this, device, s_in, in_jobs] {
      (void)hipSetDevice(device);
      for (size_t g = 0; g < in_jobs.size(); ++g) {
        {
          std::lock_guard<std::mutex> l(mu_);
          if (stop_) return;
        }
        const double a = host_now_ms();
        const bool ok = run(in_jobs[g], hipMemcpyHostToDevice, s_in, "host-to-device copy: ");
        host_trace(t0_, "h2d", (unsigned)g, a, host_now_ms());
        std::lock_guard<std::mutex> l(mu_);
        in_ready_[g] = ok ? 1 : -1;
        cv_.notify_all();
        if (!ok) return;
      }
    });
    t_out_ = std::thread([this, device, s_out, n_out] {
      (void)hipSetDevice(device);
      for (size_t g = 0; g < n_out; ++g) {
        CopyJob j;
        {
          std::unique_lock<std::mutex> l(mu_);
          cv_.wait(l, [&] { return out_state_[g] != 0; });
          if (out_state_[g] < 0) return;
          j = out_jobs_[g];
        }
        const double a = host_now_ms();
        if (!run(j, hipMemcpyDeviceToHost, s_out, "device-to-host copy: ")) return;
        host_trace(t0_, "d2h", (unsigned)g, a, host_now_ms());
      }
    });
  }
  // blocks until group g's input is on the device; false = its copy failed
  bool wait_in(size_t g) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return in_ready_[g] != 0; });
    return in_ready_[g] > 0;
  }
  void post_out(size_t g, CopyJob j) {
    std::lock_guard<std::mutex> l(mu_);
    out_jobs_[g] = j;
    out_state_[g] = 1;
    cv_.notify_all();
  }
  // ends both threads (outputs not posted yet are dropped) and returns the first copy error
  std::string finish() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
      for (auto &st : out_state_)
        if (st == 0) st = -1;
      cv_.notify_all();
    }
    if (t_in_.joinable()) t_in_.join();
    if (t_out_.joinable()) t_out_.join();
    return err_;
  }
  ~CopyPipe() { (void)finish(); }

 private:
  bool run(const CopyJob &j, hipMemcpyKind kind, hipStream_t s, const char *what) {
    hipError_t e = hipSuccess;
    if (j.bytes) e = hipMemcpyAsync(j.dst, j.src, j.bytes, kind, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) return true;
    std::lock_guard<std::mutex> l(mu_);
    if (err_.empty()) err_ = std::string(what) + hipGetErrorString(e);
    return false;
  }
  double t0_ = host_now_ms();
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<int> in_ready_, out_state_;  // 0 pending, 1 done / posted, -1 failed / dropped
  std::vector<CopyJob> out_jobs_;
  std::string err_;
  bool stop_ = false;
  std::thread t_in_, t_out_;
};

// Group boundaries of a host-pointer batch: group g ends at the first stream where the running byte
// count (a, plus b when given) reaches g / G of the total -- streams of very different sizes still
// give groups of equal work.  lo has G + 1 entries, lo[0] = 0, lo[G] = n, non-decreasing.
void cut_by_bytes(const uint64_t *a, const uint64_t *b, uint32_t n, uint32_t G, std::vector<uint32_t> &lo) {
  auto at = [&](uint32_t i) { return (a[i] - a[0]) + (b ? b[i] - b[0] : 0ull); };
  const uint64_t total = at(n);
  lo[0] = 0;
  uint32_t i = 0;
  for (uint32_t g = 1; g < G; ++g) {
    const uint64_t want = total / G * g;
    while (i < n && at(i) < want) ++i;
    lo[g] = i;
  }
  lo[G] = n;
}

// The two copy streams of the host-pointer pipelines: LOW priority, i.e. hardware queues of a pool of their
// own (see the guest stream in flate_hip_init): a copy that shared a hardware queue with a lane's kernels
// would wait behind them and the pipeline would run in lock step.
int host_pipe_streams(flate_hip_ctx *c) {
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const char *gp = getenv("FLATE_HIP_GUEST_STREAM_PRIORITY");
  const bool plain = (gp && gp[0] == 'n') || least == greatest;
  auto mk = [&](hipStream_t *s) {
    return plain ? hipStreamCreateWithFlags(s, hipStreamNonBlocking) : hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
  };
  if (!c->h2d_stream) HIP_TRY(c, mk(&c->h2d_stream));
  if (!c->d2h_stream) HIP_TRY(c, mk(&c->d2h_stream));
  return FLATE_HIP_OK;
}
}  // namespace

// The launch options of the parent, as they are now, for a lane's sub-context.
static void lane_options(flate_hip_ctx *dst, const flate_hip_ctx *src) {
  dst->guest_blocks = src->guest_blocks;
  dst->guest_min = src->guest_min;
  dst->resident_blocks = src->resident_blocks;
  dst->window_units = src->window_units;
  dst->entropy_per_block = src->entropy_per_block;
  dst->spin_limit = src->spin_limit;
  dst->profile_split = src->profile_split;
  dst->profiling = src->profiling;
  dst->host_groups = 0;
}

// The streams are independent, so the bytes are those of one call over the whole batch.
// Group g is compressed on lane g % lanes into its own slot of the device output (the slots are
// sized by the groups' bounds: where a group's bytes end up in `out` depends on the sizes of the
// groups before it, which the host only learns as they finish); the calling thread takes the groups
// in order, fills the index and posts each group's bytes to the copy-out thread.
static int deflate_host_pipelined(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                                  uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t flags,
                                  uint32_t G) {
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  const int lanes = c->host_lanes > 1 ? 2 : 1;
  for (int k = 0; k < lanes; ++k) {
    if (!c->lane[k] && (rc = flate_hip_init(c->device, &c->lane[k]))) return rc;
    lane_options(c->lane[k], c);
  }
  std::vector<uint32_t> lo(G + 1);
  std::vector<CopyJob> in_jobs(G);
  cut_by_bytes(in_off, nullptr, n, G, lo);  // equal BYTES per group: copy and compute stages stay balanced
  // device slots of the groups' output
  std::vector<uint64_t> slot(G + 1, 0);
  for (uint32_t g = 0; g < G; ++g) {
    uint64_t bound = 0;
    for (uint32_t i = lo[g]; i < lo[g + 1]; ++i) bound += flate_hip_deflate_bound((size_t)(in_off[i + 1] - in_off[i]));
    if (bound > out_cap) bound = out_cap;  // (a group that needs more than that fails the call anyway)
    slot[g + 1] = slot[g] + ((bound + 255) & ~255ull);
  }
  if ((rc = ensure(c, c->d_in, in_off[n] + 16))) return rc;
  if ((rc = ensure(c, c->d_out, slot[G] + 16))) return rc;
  if ((rc = host_pipe_streams(c))) return rc;
  uint8_t *d_in = (uint8_t *)c->d_in.p, *d_out = (uint8_t *)c->d_out.p;
  for (uint32_t g = 0; g < G; ++g)
    in_jobs[g] = {d_in + in_off[lo[g]], in + in_off[lo[g]], (size_t)(in_off[lo[g + 1]] - in_off[lo[g]])};
  const double t_call = host_now_ms();
  CopyPipe pipe(G, G);
  pipe.start(c->device, c->h2d_stream, c->d2h_stream, in_jobs);

  struct GroupResult {
    std::vector<uint64_t> off;
    int rc = FLATE_HIP_OK;
    bool done = false;
    bool threw = false;  // an exception was caught in the lane's thread
    float stage[FLATE_HIP_STAGE_COUNT] = {0, 0, 0, 0};
    std::string err;
  };
  std::vector<GroupResult> res(G);
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  auto run_lane = [&](int k) {
    flate_hip_ctx *lc = c->lane[k];
    std::vector<uint64_t> gin;
    for (uint32_t g = (uint32_t)k; g < G; g += (uint32_t)lanes) {
      GroupResult &r = res[g];
      {
        std::lock_guard<std::mutex> l(mu);
        if (stop) r.rc = FLATE_HIP_E_INTERNAL;
      }
      // (this is a worker thread: an exception that left it would end the process.  std::bad_alloc /
      // length_error from the vectors here or inside deflate_common are recorded instead; the calling thread
      // rethrows after the join, and its caller runs the batch as one pass, as before the lanes existed)
      try {
        if (r.rc == FLATE_HIP_OK && !pipe.wait_in(g)) r.rc = FLATE_HIP_E_HIP;
        const uint32_t cnt = lo[g + 1] - lo[g];
        r.off.assign((size_t)cnt + 1, 0);
        if (r.rc == FLATE_HIP_OK && cnt) {
          gin.resize((size_t)cnt + 1);
          const uint64_t base = in_off[lo[g]];
          for (uint32_t i = 0; i <= cnt; ++i) gin[i] = in_off[lo[g] + i] - base;
          lc->hip_err.clear();
          const double a = host_now_ms();
          r.rc = deflate_common(lc, d_in + base, gin.data(), cnt, d_out + slot[g], slot[g + 1] - slot[g], r.off.data(),
                                flags | FLATE_HIP_DEVICE_PTRS, false, nullptr);
          host_trace(t_call, "compute", g, a, host_now_ms());
          for (int s = 0; s < FLATE_HIP_STAGE_COUNT; ++s) r.stage[s] = lc->stage_ms[s];
          if (r.rc != FLATE_HIP_OK) r.err = lc->hip_err;
        }
      } catch (const std::exception &e) {
        r.rc = FLATE_HIP_E_INTERNAL;
        r.threw = true;
        try {
          r.err = std::string("host pipeline lane: ") + e.what();
        } catch (...) {
        }
      } catch (...) {
        r.rc = FLATE_HIP_E_INTERNAL;
        r.threw = true;
      }
      std::lock_guard<std::mutex> l(mu);
      r.done = true;
      if (r.rc != FLATE_HIP_OK) stop = true;
      cv.notify_all();
    }
  };
  // (the calling thread only collects; a thread that cannot be started ends the others before the
  // exception travels on to the caller, which then runs the batch as one pass)
  std::thread workers[2];
  try {
    for (int k = 0; k < lanes; ++k) workers[k] = std::thread(run_lane, k);
  } catch (...) {
    {
      std::lock_guard<std::mutex> l(mu);
      stop = true;
    }
    (void)pipe.finish();
    for (auto &w : workers)
      if (w.joinable()) w.join();
    throw;
  }

  float stage_sum[FLATE_HIP_STAGE_COUNT] = {0, 0, 0, 0};
  uint64_t at = 0;
  bool lane_threw = false;
  rc = FLATE_HIP_OK;
  out_off[0] = 0;
  for (uint32_t g = 0; g < G && rc == FLATE_HIP_OK; ++g) {
    GroupResult &r = res[g];
    {
      std::unique_lock<std::mutex> l(mu);
      cv.wait(l, [&] { return r.done; });
    }
    if (r.rc != FLATE_HIP_OK) {
      rc = r.rc;
      lane_threw = r.threw;
      if (c->hip_err.empty()) c->hip_err = r.err;
      break;
    }
    const uint32_t cnt = lo[g + 1] - lo[g];
    const uint64_t bytes = r.off[cnt];
    if (at + bytes > out_cap) {
      rc = FLATE_HIP_E_OUT_TOO_SMALL;
      break;
    }
    for (uint32_t i = 1; i <= cnt; ++i) out_off[lo[g] + i] = at + r.off[i];
    pipe.post_out(g, {out + at, d_out + slot[g], (size_t)bytes});
    at += bytes;
    for (int k = 0; k < FLATE_HIP_STAGE_COUNT; ++k) stage_sum[k] += r.stage[k];
  }
  {
    std::lock_guard<std::mutex> l(mu);
    if (rc != FLATE_HIP_OK) stop = true;
  }
  for (auto &w : workers)
    if (w.joinable()) w.join();
  const std::string err = pipe.finish();
  // A lane that threw may sit behind the group the collector stopped at (lane 1 throws in group 3 and raises `stop`
  // before lane 0 has started group 2: group 2 then carries E_INTERNAL without a message): look at every group.
  for (uint32_t g = 0; g < G; ++g)
    if (res[g].threw) {
      lane_threw = true;
      if (!res[g].err.empty()) c->hip_err = res[g].err;
      break;
    }
  if (lane_threw) {
    // the one-pass fallback reuses c->d_in / c->d_out: nothing of the lanes may still be writing there
    for (int k = 0; k < lanes; ++k) {
      (void)hipStreamSynchronize(c->lane[k]->stream);
      if (c->lane[k]->guest_stream) (void)hipStreamSynchronize(c->lane[k]->guest_stream);
    }
    throw std::runtime_error(c->hip_err);  // every thread has ended: the caller falls back to one pass
  }
  if (rc == FLATE_HIP_OK && !err.empty()) rc = FLATE_HIP_E_HIP;
  if (rc == FLATE_HIP_E_HIP && c->hip_err.empty()) c->hip_err = err;
  for (int k = 0; k < FLATE_HIP_STAGE_COUNT; ++k) c->stage_ms[k] = stage_sum[k];
  return rc;
}

extern "C" {

int flate_hip_deflate_fast_batch(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off,
                                 uint32_t n, uint8_t *out, uint64_t out_cap, uint64_t *out_off,
                                 uint32_t flags) {
  if (!c || !in_off || !out_off || (n && (!in || !out))) return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  if (n == 0) {
    out_off[0] = 0;
    return FLATE_HIP_OK;
  }
  // host pointers and a batch large enough that every group still fills the persistent launch
  if (!(flags & FLATE_HIP_DEVICE_PTRS) && c->host_groups > 1 && in_off[n] >= (64ull << 20)) {
    uint32_t G = (uint32_t)c->host_groups;
    // (a group of 4096 64-KiB streams still runs at 80 %; option host_pipeline_group_streams)
    const uint32_t per = c->guest_min > c->host_group_streams ? c->guest_min : c->host_group_streams;
    if (n / per < G) G = n / per;
    if (G > 1) {
      StagePlan pl;  // validate the whole index first (the same checks as the one-call path)
      const int rc = make_plan(in_off, n, pl, flags);
      if (rc) return rc;
      try {
        return deflate_host_pipelined(c, in, in_off, n, out, out_cap, out_off, flags, G);
      } catch (const std::exception &e) {  // (no copy threads, out of host memory): one pass instead
        c->hip_err.clear();
      }
    }
  }
  return deflate_common(c, in, in_off, n, out, out_cap, out_off, flags, false, nullptr);
}

int flate_hip_deflate_fast_spliced(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off,
                                   uint32_t n, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                                   uint64_t *bit_off, uint32_t flags) {
  if (!c || !in_off || !out || !out_len || (n && !in)) return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  if (n == 0) {  // nothing but the closing block of Writer::close
    static const uint8_t closing[5] = {0x01, 0x00, 0x00, 0xff, 0xff};
    if (out_cap < 5) return FLATE_HIP_E_OUT_TOO_SMALL;
    if (flags & FLATE_HIP_DEVICE_PTRS) {
      HIP_TRY(c, hipSetDevice(c->device));
      HIP_TRY(c, hipMemcpyAsync(out, closing, 5, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
      memcpy(out, closing, 5);
    }
    *out_len = 5;
    if (bit_off) bit_off[0] = 0;
    return FLATE_HIP_OK;
  }
  return deflate_common(c, in, in_off, n, out, out_cap, bit_off, flags, true, out_len);
}

// ---- one long stream, written in pieces (Writer::write as the reference behaves: output leaves
// ---- while later input is still to come, deflate.mbt:280-294) ---------------------------------
}  // extern "C"

struct flate_hip_stream {
  flate_hip_ctx *ctx = nullptr;
  uint32_t flags = 0;
  DevBuf table, clock, hist, stage, io, out;  // io: {lz77 in_off[2], huff in_off[2]} (u64) + chunk/blk bases
  uint64_t abs = 0;        // bytes of the stream consumed so far (a multiple of 65535 until the end)
  uint64_t pos = 0;        // the same, counted from the stream's current origin (see rebase_at)
  uint64_t rebase_at = 1ull << 30;  // origin moved up when pos passes this (option stream_rebase_bytes)
  // DeflateFast.cur as the reference counts it (deflate-fast.mbt:107,115,156): 65535 at the start,
  // + the window's length after every encode; when it reaches buffer_reset (:55,130-132) shift_offsets
  // runs -- in MoonBit `prev` is always empty, so that CLEARS the table (:367-374); in Go
  // the offsets move down and every distance stays what it was
  int64_t ref_cur = kMaxStoreBlockSize;
  int64_t buffer_reset = 2147483647ll - 2 * kMaxStoreBlockSize;
  uint32_t carry_bits = 0; // bits of the last, incomplete output byte (0..7) ...
  uint8_t carry = 0;       // ... and their value
  bool closed = false;
  int err = 0;             // sticky (Compressor.err, deflate.mbt:74)
};

namespace {
constexpr uint64_t kHist = 32768;  // max_match_offset: what a later window can still reference

int stream_write_impl(flate_hip_stream *st, const uint8_t *in, uint64_t n, bool final, uint8_t *out,
                      uint64_t out_cap, uint64_t *out_len) {
  flate_hip_ctx *c = st->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  // Positions inside the kernels are 32-bit and counted from the stream's origin.  A long stream
  // moves its origin up (the reference's shift_offsets, deflate-fast.mbt:366-389: same distances,
  // smaller numbers), so its length is not limited; one piece is (< 1 GiB).
  if (n >= (1ull << 30)) return FLATE_HIP_E_TOO_LARGE;
  uint32_t rebase = 0;
  if (st->pos >= st->rebase_at && st->pos > (uint64_t)kMaxStoreBlockSize) {
    rebase = (uint32_t)(st->pos - (uint64_t)kMaxStoreBlockSize);  // new origin: one window in front
    st->pos = kMaxStoreBlockSize;
  }
  const uint64_t W0 = st->pos;
  const uint64_t full = n / kMaxStoreBlockSize, r = n % kMaxStoreBlockSize;
  const uint32_t nch = (uint32_t)(full + (r >= (uint64_t)kSmallLzMin ? 1 : 0));
  const uint32_t nblk = (uint32_t)(full + (r > 0 ? 1 : 0));
  const uint32_t win0 = (uint32_t)(W0 / kMaxStoreBlockSize);
  int rc;
  // device staging: [the last 32 KiB of what came before][the new bytes]
  if ((rc = ensure(c, st->table, kTableSize * 2 + 64))) return rc;
  if ((rc = ensure(c, st->clock, 64))) return rc;
  if ((rc = ensure(c, st->hist, kHist + 64))) return rc;
  if ((rc = ensure(c, st->stage, kHist + n + 64))) return rc;
  if ((rc = ensure(c, st->io, 256))) return rc;
  const uint64_t cap_need = flate_hip_deflate_bound(n) + 16;
  if ((rc = ensure(c, st->out, cap_need + 16))) return rc;
  uint8_t *stage = (uint8_t *)st->stage.p;
  if (W0) HIP_TRY(c, hipMemcpyAsync(stage, st->hist.p, kHist, hipMemcpyDeviceToDevice, c->stream));
  if (n) HIP_TRY(c, hipMemcpyAsync(stage + kHist, in, n, hipMemcpyHostToDevice, c->stream));
  // index arrays: the match finder sees the stream through a virtual base (absolute positions, the
  // table's mod-2^16 arithmetic needs them), the entropy stage sees the new bytes only
  uint64_t h_io[8] = {0, W0 + n, 0, n, 0, 0, 0, 0};
  uint32_t *h32 = reinterpret_cast<uint32_t *>(h_io + 4);
  h32[0] = 0; h32[1] = nch;   // chunk_base
  h32[2] = 0; h32[3] = nblk;  // blk_base
  HIP_TRY(c, hipMemcpyAsync(st->io.p, h_io, sizeof h_io, hipMemcpyHostToDevice, c->stream));
  const uint64_t *d_off_abs = (const uint64_t *)st->io.p, *d_off_loc = d_off_abs + 2;
  const uint32_t *d_chunk_base = (const uint32_t *)((const uint64_t *)st->io.p + 4), *d_blk_base = d_chunk_base + 2;
  if ((rc = ensure(c, c->d_matches, (size_t)(nch ? nch : 1) * kMatchCapPerChunk * sizeof(uint2) + 16))) return rc;
  if ((rc = ensure(c, c->d_nmatch, (size_t)nch * 4 + 4))) return rc;
  if ((rc = ensure(c, c->d_ntok, (size_t)nch * 4 + 4))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_status.p, 0, 4, c->stream));
  if (nch) {
    LzParams P{};
    P.in = stage + kHist - W0;  // virtual: only positions >= W0 - 32768 are ever dereferenced
    P.in_off = d_off_abs;
    P.chunk_base = d_chunk_base;
    P.scan_off = (const uint16_t *)c->scan_tab.p;
    P.scan_len = c->scan_len;
    P.matches = (uint2 *)c->d_matches.p;
    P.chunk_nmatch = (uint32_t *)c->d_nmatch.p;
    P.chunk_ntok = (uint32_t *)c->d_ntok.p;
    P.compat_go = (st->flags & FLATE_HIP_COMPAT_GO) ? 1u : 0u;
    P.status = (int *)c->d_status.p;
    P.spin_limit = c->spin_limit;
    // One launch per run of windows between two shift_offsets of the reference (one launch, except
    // for the piece in which `cur` passes buffer_reset: window 32 766 of a Writer, then every 32 767).
    const bool forgets = !(st->flags & FLATE_HIP_COMPAT_GO);
    uint32_t k0 = 0;        // first window (of this piece) of the launch being collected
    bool forget0 = false;   // ... and whether it starts on a cleared table
    auto flush = [&](uint32_t k1) {
      if (k1 == k0) return;
      LzParams Q = P;
      Q.win0 = win0 + k0;
      Q.matches = P.matches + (size_t)k0 * kMatchCapPerChunk;  // the kernel indexes both by window - win0
      Q.chunk_nmatch = P.chunk_nmatch + k0;
      Q.chunk_ntok = P.chunk_ntok + k0;
      hipLaunchKernelGGL(lz77_resume_kernel, dim3(1), dim3(64), 0, c->stream, Q, (uint16_t *)st->table.p,
                         (uint32_t *)st->clock.p, k1 - k0, k0 == 0 ? rebase : 0u, forget0 ? 1u : 0u);
    };
    for (uint32_t k = 0; k < nch; ++k) {
      if (st->ref_cur >= st->buffer_reset) {  // deflate-fast.mbt:130-132
        st->ref_cur = kMaxMatchOffset + 1;    // :372,388
        if (forgets) {
          flush(k);
          k0 = k;
          forget0 = true;
        }
      }
      st->ref_cur += k < full ? (int64_t)kMaxStoreBlockSize : (int64_t)r;  // :156
    }
    flush(nch);
  } else if (rebase) {
    st->pos += rebase;  // (nothing ran: the table still counts from the old origin)
  }
  const size_t nb = (size_t)nblk + 1;
  if ((rc = ensure(c, c->d_blk_hist, nb * 320 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_cl, nb * 320 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_hdr, nb * 704 * 4))) return rc;
  if ((rc = ensure(c, c->d_blk_meta, nb * 16))) return rc;
  if ((rc = ensure(c, c->d_tile_meta, ((n >> 8) + nb + 2) * 64))) return rc;
  if ((rc = ensure(c, c->d_out_len, 4 * 8))) return rc;
  if ((rc = ensure(c, c->d_out_off, 4 * 8))) return rc;
  if ((rc = ensure(c, c->d_slot_off, 4 * 16))) return rc;
  uint8_t *d_out = (uint8_t *)st->out.p;
  HuffParams H{};
  H.in = stage + kHist;
  H.in_off = d_off_loc;
  H.chunk_base = d_chunk_base;
  H.blk_base = d_blk_base;
  H.matches = (const uint2 *)c->d_matches.p;
  H.chunk_nmatch = (const uint32_t *)c->d_nmatch.p;
  H.chunk_ntok = (const uint32_t *)c->d_ntok.p;
  H.blk_hist = (uint32_t *)c->d_blk_hist.p;
  H.blk_cl = (uint32_t *)c->d_blk_cl.p;
  H.blk_hdr = (uint32_t *)c->d_blk_hdr.p;
  H.blk_meta = (uint4 *)c->d_blk_meta.p;
  H.tile_meta = (uint8_t *)c->d_tile_meta.p;
  H.spliced = 1u;
  H.stream_sum = (uint64_t *)c->d_slot_off.p;
  H.stream_bit = (const uint64_t *)c->d_out_off.p;
  H.out_len = (uint64_t *)c->d_out_len.p;
  H.out_off = (const uint64_t *)c->d_out_off.p;
  H.out = d_out;
  H.status = (int *)c->d_status.p;
  H.n_streams = 1;
  H.compat_go = (st->flags & FLATE_HIP_COMPAT_GO) ? 1u : 0u;
  H.no_close = final ? 0u : 1u;
  SpliceParams S{};
  S.sum = (const uint64_t *)c->d_slot_off.p;
  S.stream_bit = (uint64_t *)c->d_out_off.p;
  S.total_bytes = (uint64_t *)c->d_out_len.p + 1;
  S.out_cap = cap_need;
  S.status = (int *)c->d_status.p;
  S.n_streams = 1;
  S.start_bit = st->carry_bits;
  S.no_close = H.no_close;
  hipLaunchKernelGGL(huff_hist_kernel, dim3(1), dim3(64), 0, c->stream, H);
  hipLaunchKernelGGL(huff_code_kernel, dim3(1), dim3(64), 0, c->stream, H);
  hipLaunchKernelGGL(splice_scan_kernel, dim3(1), dim3(1024), 0, c->stream, S);
  hipLaunchKernelGGL(splice_zero_kernel, dim3(1), dim3(256), 0, c->stream, S, d_out);
  // the bits left over from the previous piece share the first byte with this piece's first block
  if (st->carry_bits) HIP_TRY(c, hipMemcpyAsync(d_out, &st->carry, 1, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(huff_pack_kernel, dim3(1), dim3(64), 0, c->stream, H);
  HIP_TRY(c, hipGetLastError());
  uint64_t h_pos[2] = {0, 0};  // {end bit of the piece, total bytes}
  HIP_TRY(c, hipMemcpyAsync(&h_pos[0], (uint64_t *)c->d_out_off.p + 1, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&h_pos[1], (uint64_t *)c->d_out_len.p + 1, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&c->h_status_word, c->d_status.p, 4, hipMemcpyDeviceToHost, c->stream));
  if (!final && n >= kHist)  // what the next piece may still reference
    HIP_TRY(c, hipMemcpyAsync(st->hist.p, stage + n, kHist, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->h_status_word) return c->h_status_word <= kStatusUqTimeout ? FLATE_HIP_E_INTERNAL : c->h_status_word;
  const uint64_t whole = final ? h_pos[1] : (h_pos[0] >> 3);
  if (whole > out_cap) return FLATE_HIP_E_OUT_TOO_SMALL;
  if (whole) HIP_TRY(c, hipMemcpyAsync(out, d_out, whole, hipMemcpyDeviceToHost, c->stream));
  st->carry_bits = final ? 0u : (uint32_t)(h_pos[0] & 7u);
  if (st->carry_bits) HIP_TRY(c, hipMemcpyAsync(&st->carry, d_out + whole, 1, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (st->carry_bits) st->carry &= (uint8_t)((1u << st->carry_bits) - 1u);
  st->pos += n;
  st->abs += n;
  *out_len = whole;
  return FLATE_HIP_OK;
}
}  // namespace

extern "C" {

int flate_hip_stream_open(flate_hip_ctx *c, uint32_t flags, flate_hip_stream **out) {
  if (!c || !out || (flags & ~FLATE_HIP_COMPAT_GO)) return FLATE_HIP_E_INVALID;
  flate_hip_stream *st = new flate_hip_stream();
  st->ctx = c;
  st->flags = flags;
  st->rebase_at = c->stream_rebase;
  if (c->debug_buffer_reset > 0) st->buffer_reset = c->debug_buffer_reset;
  *out = st;
  return FLATE_HIP_OK;
}

void flate_hip_stream_free(flate_hip_stream *st) {
  if (!st) return;
  (void)hipSetDevice(st->ctx->device);
  for (DevBuf *b : {&st->table, &st->clock, &st->hist, &st->stage, &st->io, &st->out}) release(*b);
  delete st;
}

size_t flate_hip_stream_bound(size_t n) { return flate_hip_deflate_bound(n) + 8; }

int flate_hip_stream_write(flate_hip_stream *st, const uint8_t *in, uint64_t n, int final, uint8_t *out,
                           uint64_t out_cap, uint64_t *out_len) {
  if (!st || !out_len || (n && !in) || !out) return FLATE_HIP_E_INVALID;
  *out_len = 0;
  if (st->err) return st->err;
  if (st->closed) return FLATE_HIP_E_INVALID;
  // a piece that is not the last one is whole windows: the 65535-byte staging window of
  // Compressor::fill_store (deflate.mbt:222-229) is what enc_speed compresses at a time
  if (!final && (n == 0 || n % kMaxStoreBlockSize != 0)) return FLATE_HIP_E_INVALID;
  st->ctx->hip_err.clear();
  const int rc = stream_write_impl(st, in, n, final != 0, out, out_cap, out_len);
  if (rc != FLATE_HIP_OK && rc != FLATE_HIP_E_OUT_TOO_SMALL) st->err = rc;  // sticky, as Compressor.err
  if (rc == FLATE_HIP_E_OUT_TOO_SMALL) st->err = rc;  // (the piece's state is gone with the call)
  if (rc == FLATE_HIP_OK && final) st->closed = true;
  return rc;
}

}  // extern "C"

// ---- one long stream decoded in pieces (Decompressor::read as the reference behaves: the caller
// ---- holds a piece of input and a piece of output, never the whole stream; inflate.mbt:382-407) ----
struct flate_hip_inflate_stream {
  flate_hip_ctx *ctx = nullptr;
  DevBuf state, in, out;
  int status = 0;           // sticky: 1 = the final block is done, < 0 = error
  int64_t err_off = -1;
  uint32_t bit_in_byte = 0; // of the byte the next call's input starts with
  uint64_t total_in = 0, total_out = 0;
};

extern "C" {

int flate_hip_inflate_stream_open(flate_hip_ctx *c, flate_hip_inflate_stream **out) {
  if (!c || !out) return FLATE_HIP_E_INVALID;
  *out = nullptr;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  flate_hip_inflate_stream *st = new flate_hip_inflate_stream();
  st->ctx = c;
  int rc = ensure(c, st->state, inflate_stream_state_bytes() + 64);
  if (rc == FLATE_HIP_OK) {
    hipLaunchKernelGGL(inflate_stream_init_kernel, dim3(1), dim3(64), 0, c->stream, st->state.p,
                       (const uint8_t *)nullptr, 0u);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = FLATE_HIP_E_HIP;
  }
  if (rc != FLATE_HIP_OK) {
    release(st->state);
    delete st;
    return rc;
  }
  *out = st;
  return FLATE_HIP_OK;
}

// Decompressor::reset(r, dict) (inflate.mbt:862-884) / &Reader::new_dict (:315-317): a fresh decoder on
// the same handle, with the last 32768 bytes of `dict` as history that has already been read
// (DictDecoder::new, dict-decoder.mbt:40-60).
int flate_hip_inflate_stream_reset(flate_hip_inflate_stream *st, const uint8_t *dict, uint64_t dict_len) {
  if (!st || (dict_len && !dict)) return FLATE_HIP_E_INVALID;
  flate_hip_ctx *c = st->ctx;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  if (dict_len > (uint64_t)kMaxMatchOffset) {
    dict += dict_len - (uint64_t)kMaxMatchOffset;
    dict_len = (uint64_t)kMaxMatchOffset;
  }
  int rc;
  if ((rc = ensure(c, st->in, dict_len + 16))) return rc;
  if (dict_len) HIP_TRY(c, hipMemcpyAsync(st->in.p, dict, dict_len, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(inflate_stream_init_kernel, dim3(1), dim3(256), 0, c->stream, st->state.p,
                     (const uint8_t *)st->in.p, (uint32_t)dict_len);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  st->status = 0;
  st->err_off = -1;
  st->bit_in_byte = 0;
  st->total_in = st->total_out = 0;
  return FLATE_HIP_OK;
}

void flate_hip_inflate_stream_free(flate_hip_inflate_stream *st) {
  if (!st) return;
  (void)hipSetDevice(st->ctx->device);
  for (DevBuf *b : {&st->state, &st->in, &st->out}) release(*b);
  delete st;
}

int flate_hip_inflate_stream_read(flate_hip_inflate_stream *st, const uint8_t *in, uint64_t in_len, int final_in,
                                  uint8_t *out, uint64_t out_cap, uint64_t *in_used, uint64_t *out_len,
                                  int64_t *err_off) {
  if (!st || !in_used || !out_len || (in_len && !in) || (out_cap && !out)) return FLATE_HIP_E_INVALID;
  *in_used = *out_len = 0;
  if (err_off) *err_off = st->err_off;
  if (st->status) return st->status == 1 ? FLATE_HIP_STREAM_END : st->status;  // sticky (Decompressor.err, inflate.mbt:285,398)
  // the byte that holds the next unconsumed bit was reported as unused: it has to be here again
  if (st->bit_in_byte && in_len == 0) return final_in ? FLATE_HIP_E_UNEXPECTED_EOF : FLATE_HIP_OK;
  if (in_len == 0 && !final_in) return FLATE_HIP_OK;  // nothing to decode from
  flate_hip_ctx *c = st->ctx;
  c->hip_err.clear();
  HIP_TRY(c, hipSetDevice(c->device));
  // one call takes at most 1 GiB each way (32-bit positions inside the kernel); more input than that is
  // simply not all used, and not final
  const uint64_t kPiece = 1ull << 30;
  if (in_len > kPiece) {
    in_len = kPiece;
    final_in = 0;
  }
  if (out_cap > kPiece) out_cap = kPiece;
  int rc;
  if ((rc = ensure(c, st->in, in_len + 16))) return rc;
  if ((rc = ensure(c, st->out, out_cap + 16))) return rc;
  if (in_len) HIP_TRY(c, hipMemcpyAsync(st->in.p, in, in_len, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(inflate_stream_kernel, dim3(1), dim3(64), 0, c->stream, st->state.p, (const uint8_t *)st->in.p,
                     (uint32_t)in_len, final_in ? 1u : 0u, (uint8_t *)st->out.p, (uint32_t)out_cap);
  HIP_TRY(c, hipGetLastError());
  InfStreamResult r{};
  HIP_TRY(c, hipMemcpyAsync(&r, st->state.p, sizeof r, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (r.out_len > out_cap || r.in_used > in_len) return FLATE_HIP_E_INTERNAL;
  if (r.out_len) {
    HIP_TRY(c, hipMemcpyAsync(out, st->out.p, r.out_len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  *in_used = r.in_used;
  *out_len = r.out_len;
  st->bit_in_byte = r.bit_in_byte;
  st->total_in = r.total_in;
  st->total_out = r.total_out;
  if (r.status) {
    st->status = r.status;
    st->err_off = r.err_off;
    if (err_off) *err_off = r.err_off;
    return r.status == 1 ? FLATE_HIP_STREAM_END : r.status;
  }
  return FLATE_HIP_OK;
}

int flate_hip_lz77_matches(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                           uint32_t flags, uint32_t *n_chunks, uint64_t *n_recs_cap,
                           uint32_t *chunk_nmatch, uint64_t *chunk_rec_off, uint32_t *recs) {
  if (!c || !in_off || !n_chunks || !n_recs_cap) return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  StagePlan pl;
  int rc = make_plan(in_off, n, pl, flags);
  if (rc) return rc;
  *n_chunks = pl.n_chunks;
  *n_recs_cap = (uint64_t)pl.n_chunks * kMatchCapPerChunk;
  if (!recs) return FLATE_HIP_OK;
  if (!in || !chunk_nmatch || !chunk_rec_off) return FLATE_HIP_E_INVALID;
  if (pl.n_chunks == 0) return FLATE_HIP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool dev = (flags & FLATE_HIP_DEVICE_PTRS) != 0;
  const uint8_t *d_in = in;
  if (!dev) {
    if ((rc = ensure(c, c->d_in, in_off[n] + 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in.p, in, in_off[n], hipMemcpyHostToDevice, c->stream));
    d_in = (const uint8_t *)c->d_in.p;
  }
  HIP_TRY(c, hipMemsetAsync(c->d_status.p, 0, 4, c->stream));
  if ((rc = ctl_begin(c, ((size_t)n + 1) * 12 + (pl.ids16.size() + pl.ids32.size()) * 4, 64))) return rc;
  if ((rc = run_lz77(c, d_in, in_off, pl, flags))) return rc;
  HIP_TRY(c, hipMemcpyAsync(chunk_nmatch, c->d_nmatch.p, (size_t)pl.n_chunks * 4,
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&c->h_status_word, c->d_status.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->h_status_word) return c->h_status_word <= kStatusUqTimeout ? FLATE_HIP_E_INTERNAL : c->h_status_word;
  for (uint32_t k = 0; k <= pl.n_chunks; ++k) chunk_rec_off[k] = (uint64_t)k * kMatchCapPerChunk;
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIP_TRY(c, hipMemcpyAsync(recs, c->d_matches.p, (size_t)pl.n_chunks * kMatchCapPerChunk * 8, kind,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const bool used[FLATE_HIP_STAGE_COUNT] = {true, false, false, false};
  return collect_timing(c, used);
}

#if defined(FLATE_LZ_STAMPS) || defined(FLATE_LZ_FINISH)
// diagnostic builds only: per-chunk phase cycle sums (or start / finish times) of the last match-finder launch
int flate_hip_debug_lz_stamps(flate_hip_ctx *c, uint64_t *out, uint32_t max_chunks) {
  uint32_t k = c->debug_chunks < max_chunks ? c->debug_chunks : max_chunks;
  if (hipMemcpy(out, c->d_debug.p, (size_t)k * 64, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  return (int)k;
}
#endif

}  // extern "C"

// Both decode entry points.  spliced_len != 0: `in` is ONE stream of that many bytes and in_off
// holds the bit positions of its n pieces (flate_hip_inflate_spliced).
static int inflate_common(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                          uint8_t *out, const uint64_t *out_off, uint64_t *out_len, int32_t *status,
                          int64_t *err_off, uint32_t flags, uint64_t spliced_len) {
  const bool spliced = spliced_len != 0;
  const bool size_only = (flags & FLATE_HIP_SIZE_ONLY) != 0 && !spliced;
  const uint64_t in_bytes = spliced ? spliced_len : in_off[n];
  HIP_TRY(c, hipSetDevice(c->device));
  const bool dev = (flags & FLATE_HIP_DEVICE_PTRS) != 0;
  int rc;
  const uint8_t *d_in = in;
  uint8_t *d_out = out;
  std::vector<uint64_t> no_slots;
  if (size_only) {  // nothing is stored: no output buffer, no slots
    no_slots.assign((size_t)n + 1, 0);
    out_off = no_slots.data();
  }
  if (!dev) {
    if ((rc = ensure(c, c->d_in, in_bytes + 16))) return rc;
    if ((rc = ensure(c, c->d_out, out_off[n] + 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in.p, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    d_in = (const uint8_t *)c->d_in.p;
    d_out = (uint8_t *)c->d_out.p;
  }
  if ((rc = ensure(c, c->d_in_off, ((size_t)n + 1) * 8))) return rc;
  if ((rc = ensure(c, c->d_slot_off, ((size_t)n + 1) * 8))) return rc;
  if ((rc = ensure(c, c->d_out_len, (size_t)n * 8 + 8))) return rc;
  if ((rc = ensure(c, c->d_istatus, (size_t)n * 4 + 4))) return rc;
  if ((rc = ensure(c, c->d_ierr, (size_t)n * 8 + 8))) return rc;
  if ((rc = ctl_begin(c, ((size_t)n + 1) * 16, (size_t)n * 20 + 64))) return rc;
  if ((rc = ctl_up(c, c->d_in_off.p, in_off, ((size_t)n + 1) * 8))) return rc;
  if ((rc = ctl_up(c, c->d_slot_off.p, out_off, ((size_t)n + 1) * 8))) return rc;
  InfParams I{};
  I.in = d_in;
  I.in_off = (const uint64_t *)c->d_in_off.p;
  I.out = d_out;
  I.out_off = (const uint64_t *)c->d_slot_off.p;
  I.out_len = (uint64_t *)c->d_out_len.p;
  I.status = (int32_t *)c->d_istatus.p;
  I.err_off = (int64_t *)c->d_ierr.p;
  I.n_streams = n;
  I.bit_off = spliced ? (const uint64_t *)c->d_in_off.p : nullptr;
  I.in_len = in_bytes;
  I.size_only = size_only ? 1u : 0u;
  {
    StageTimer t(c, FLATE_HIP_STAGE_INFLATE);
    // large batches: one lane per stream (64 streams per wavefront); small ones: one wavefront
    // per stream
    // (its bit positions are 32-bit: every compressed stream must be < 256 MiB)
    // (size-only passes never use the lane-per-stream decoder: it reads its history back from the
    // output it has written)
    bool simt = (spliced || n >= c->inflate_simt_min) && !size_only;
    for (uint32_t i = 0; i < n && simt && !spliced; ++i) simt = in_off[i + 1] - in_off[i] < (1ull << 28);
    // (a size-only pass needs token lengths only: the sub-block decoder at any batch size, unless switched off)
    bool spec = c->inflate_spec == 2 || (c->inflate_spec == 1 && (size_only || n < c->inflate_spec_max));
    // 32-bit bit positions: a stream (a piece of a spliced stream: in_off holds bit offsets then) below 256 MiB
    for (uint32_t i = 0; i < n && spec; ++i) spec = in_off[i + 1] - in_off[i] < (spliced ? (1ull << 31) : (1ull << 28));
    if (spec) {
      // (two builds of the same kernel: long token lists and a 16 KiB history ring while a SIMD holds
      // one wavefront, the small footprint beyond)
      const int shape = c->inflate_spec_shape ? c->inflate_spec_shape : (n <= 4u * c->num_cus ? 1 : 2);
      if (shape == 1)
        hipLaunchKernelGGL(inflate_spec_kernel<FLATE_SPEC_SMALL>, dim3(n), dim3(64), 0, c->stream, I);
      else
        hipLaunchKernelGGL(inflate_spec_kernel<FLATE_SPEC_LARGE>, dim3(n), dim3(64), 0, c->stream, I);
    } else if (simt) {
      // streams per wavefront
      int lpw = c->inflate_lanes;
      // (measured, same file: 16 lanes per wavefront up to ~20 k streams, 32 up to ~36 k, 64 beyond)
      if (lpw == 0) lpw = n >= 144u * c->num_cus ? 64 : (n >= 80u * c->num_cus ? 32 : 16);
      const uint32_t sblocks = (n + (uint32_t)lpw - 1) / (uint32_t)lpw;
      // A CU holds eight of these wavefronts (320 B of LDS per lane): a batch of more blocks than
      // that runs in ROUNDS, and a lane's rate depends little on how full the chip is -- so the rounds
      // are made equal (196608 streams: two launches of 98304 = 94 ms, against 60 + 47 for a full
      // round and a third of one).
      const uint32_t slots = 8u * c->num_cus;
      const uint32_t rounds = (sblocks + slots - 1) / slots;
      const uint32_t per = (sblocks + rounds - 1) / (rounds ? rounds : 1u);
      if ((rc = ensure(c, c->d_simt_lens, inflate_simt_lens_bytes(per)))) return rc;
      I.simt_lens = (uint32_t *)c->d_simt_lens.p;
      for (uint32_t b0 = 0; b0 < sblocks; b0 += per) {
        const uint32_t nb = sblocks - b0 < per ? sblocks - b0 : per;
        I.sid0 = b0 * (uint32_t)lpw;
        // (the output row -- a lane's output collected in registers and stored as whole aligned pieces -- pays
        // where the chip is full of lanes: the 64-lane form only)
        if (lpw == 64 && c->inflate_row == 16)
          hipLaunchKernelGGL((inflate_simt_kernel<64, 16>), dim3(nb), dim3(64), inflate_simt_lds_bytes(64), c->stream, I);
        else if (lpw == 64 && c->inflate_row == 8)
          hipLaunchKernelGGL((inflate_simt_kernel<64, 8>), dim3(nb), dim3(64), inflate_simt_lds_bytes(64), c->stream, I);
        else if (lpw == 64)
          hipLaunchKernelGGL((inflate_simt_kernel<64, 0>), dim3(nb), dim3(64), inflate_simt_lds_bytes(64), c->stream, I);
        else if (lpw == 32)
          hipLaunchKernelGGL((inflate_simt_kernel<32, 0>), dim3(nb), dim3(64), inflate_simt_lds_bytes(32), c->stream, I);
        else
          hipLaunchKernelGGL((inflate_simt_kernel<16, 0>), dim3(nb), dim3(64), inflate_simt_lds_bytes(16), c->stream, I);
      }
    }
    else
      hipLaunchKernelGGL(inflate_kernel, dim3(n), dim3(64), 0, c->stream, I);
  }
  HIP_TRY(c, hipGetLastError());
  if ((rc = ctl_down(c, out_len, c->d_out_len.p, (size_t)n * 8))) return rc;
  if ((rc = ctl_down(c, status, c->d_istatus.p, (size_t)n * 4))) return rc;
  if ((rc = ctl_down(c, err_off, c->d_ierr.p, (size_t)n * 8))) return rc;
  if (!dev && !size_only)
    HIP_TRY(c, hipMemcpyAsync(out, c->d_out.p, out_off[n], hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  ctl_finish(c);
  const bool used[FLATE_HIP_STAGE_COUNT] = {false, false, false, true};
  if ((rc = collect_timing(c, used))) return rc;
  // A size-only pass has no capacity -- but the kernels count output in 32 bits: a stream that inflates
  // to 4 GiB or more stops there with "slot too small", which for a call without slots means "too large"
  if (size_only)
    for (uint32_t i = 0; i < n; ++i)
      if (status[i] == FLATE_HIP_E_OUT_TOO_SMALL) status[i] = FLATE_HIP_E_TOO_LARGE;
  for (uint32_t i = 0; i < n; ++i)
    if (status[i]) return status[i];
  return FLATE_HIP_OK;
}

// Host-pointer inflate of independent streams, pipelined like deflate_host_pipelined: every
// stream has its own input range and output slot, so a group is a contiguous range of both.
static int inflate_host_pipelined(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                                  uint8_t *out, const uint64_t *out_off, uint64_t *out_len,
                                  int32_t *status, int64_t *err_off, uint32_t flags, uint32_t G) {
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->d_in, in_off[n] + 16))) return rc;
  if ((rc = ensure(c, c->d_out, out_off[n] + 16))) return rc;
  if ((rc = host_pipe_streams(c))) return rc;
  uint8_t *d_in = (uint8_t *)c->d_in.p, *d_out = (uint8_t *)c->d_out.p;
  std::vector<uint32_t> lo(G + 1);
  std::vector<CopyJob> in_jobs(G);
  cut_by_bytes(in_off, out_off, n, G, lo);  // by input + output bytes (both cross PCIe)
  for (uint32_t g = 0; g < G; ++g)
    in_jobs[g] = {d_in + in_off[lo[g]], in + in_off[lo[g]], (size_t)(in_off[lo[g + 1]] - in_off[lo[g]])};
  const double t_call = host_now_ms();
  CopyPipe pipe(G, G);
  pipe.start(c->device, c->h2d_stream, c->d2h_stream, in_jobs);
  float stage_sum[FLATE_HIP_STAGE_COUNT] = {0, 0, 0, 0};
  std::vector<uint64_t> gin, gout;
  rc = FLATE_HIP_OK;
  int first_status = FLATE_HIP_OK;
  for (uint32_t g = 0; g < G; ++g) {
    if (!pipe.wait_in(g)) {
      rc = FLATE_HIP_E_HIP;
      break;
    }
    const uint32_t a = lo[g], cnt = lo[g + 1] - lo[g];
    gin.resize((size_t)cnt + 1);
    gout.resize((size_t)cnt + 1);
    for (uint32_t i = 0; i <= cnt; ++i) {
      gin[i] = in_off[a + i] - in_off[a];
      gout[i] = out_off[a + i] - out_off[a];
    }
    if (cnt) {
      const double ta = host_now_ms();
      const int r = inflate_common(c, d_in + in_off[a], gin.data(), cnt, d_out + out_off[a], gout.data(),
                                   out_len + a, status + a, err_off + a, flags | FLATE_HIP_DEVICE_PTRS, 0);
      host_trace(t_call, "compute", g, ta, host_now_ms());
      // a stream's own failure (its status, also the return value) does not stop the batch: as in
      // one pass, every stream is decoded and the first failing status is what the call returns
      const bool stream_status = r == FLATE_HIP_E_CORRUPT || r == FLATE_HIP_E_UNEXPECTED_EOF ||
                                 r == FLATE_HIP_E_OUT_TOO_SMALL;
      if (r != FLATE_HIP_OK && !stream_status) {
        rc = r;
        break;
      }
      if (first_status == FLATE_HIP_OK) first_status = r;
      for (int k = 0; k < FLATE_HIP_STAGE_COUNT; ++k) stage_sum[k] += c->stage_ms[k];
    }
    pipe.post_out(g, {out + out_off[a], d_out + out_off[a], (size_t)gout[cnt]});
  }
  const std::string err = pipe.finish();
  if (rc == FLATE_HIP_OK) rc = first_status;
  if ((rc == FLATE_HIP_OK || rc == first_status) && !err.empty()) rc = FLATE_HIP_E_HIP;
  if (rc == FLATE_HIP_E_HIP && c->hip_err.empty()) c->hip_err = err;
  for (int k = 0; k < FLATE_HIP_STAGE_COUNT; ++k) c->stage_ms[k] = stage_sum[k];
  return rc;
}

extern "C" {

int flate_hip_inflate_batch(flate_hip_ctx *c, const uint8_t *in, const uint64_t *in_off, uint32_t n,
                            uint8_t *out, const uint64_t *out_off, uint64_t *out_len,
                            int32_t *status, int64_t *err_off, uint32_t flags) {
  const bool size_only = (flags & FLATE_HIP_SIZE_ONLY) != 0;
  if (!c || !in_off || !out_len || !status || !err_off || (n && !in) ||
      (!size_only && (!out_off || (n && !out))))
    return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  if (n == 0) return FLATE_HIP_OK;
  for (uint32_t i = 0; i < n; ++i)
    if (in_off[i + 1] < in_off[i] || (!size_only && out_off[i + 1] < out_off[i])) return FLATE_HIP_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)
    if (in_off[i + 1] - in_off[i] >= 0x7ffe0000ull) return FLATE_HIP_E_TOO_LARGE;
  if (size_only)
    return inflate_common(c, in, in_off, n, nullptr, nullptr, out_len, status, err_off, flags, 0);
  // host pointers and a large batch: decode group g while g+1 is copied in and g-1 out
  if (!(flags & FLATE_HIP_DEVICE_PTRS) && c->host_groups > 1 &&
      in_off[n] - in_off[0] + out_off[n] - out_off[0] >= (64ull << 20)) {
    uint32_t G = (uint32_t)c->host_groups;
    const uint32_t iper = 4u * c->host_group_streams;  // (a group should still fill the lane-per-stream launch: 16384)
    if (n / iper < G) G = n / iper;
    if (G > 1) {
      try {
        return inflate_host_pipelined(c, in, in_off, n, out, out_off, out_len, status, err_off, flags, G);
      } catch (const std::exception &e) {  // (no copy threads, out of host memory): one pass instead
        c->hip_err.clear();
      }
    }
  }
  return inflate_common(c, in, in_off, n, out, out_off, out_len, status, err_off, flags, 0);
}

int flate_hip_inflate_spliced(flate_hip_ctx *c, const uint8_t *in, uint64_t in_len,
                              const uint64_t *bit_off, uint32_t n, uint8_t *out,
                              const uint64_t *out_off, uint64_t *out_len, int32_t *status,
                              int64_t *err_off, uint32_t flags) {
  if (!c || !in || !in_len || !bit_off || !out_off || !out_len || !status || !err_off || (n && !out))
    return FLATE_HIP_E_INVALID;
  c->hip_err.clear();
  if (n == 0) return FLATE_HIP_OK;
  for (uint32_t i = 0; i < n; ++i) {
    if (bit_off[i + 1] < bit_off[i] || out_off[i + 1] < out_off[i] || bit_off[i + 1] > in_len * 8)
      return FLATE_HIP_E_INVALID;
    if (bit_off[i + 1] - bit_off[i] >= (1ull << 30)) return FLATE_HIP_E_TOO_LARGE;  // piece < 128 MiB
  }
  return inflate_common(c, in, bit_off, n, out, out_off, out_len, status, err_off, flags, in_len);
}

}  // extern "C"
