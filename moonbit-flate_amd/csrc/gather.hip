// gather.hip -- the exchange step of the multi-GPU row behind the
// C ABI: every rank has compressed its own contiguous range of streams (no collective in the
// compress path); one exchange concatenates the compressed shards on every rank.  The reference
// has no counterpart (single-threaded MoonBit, no communication layer): this is the step
// BASELINE.json's north_star names ("RCCL all-gather over xGMI to concatenate the compressed stream"),
// made callable from the host language through include/flate_hip.h instead of only from Python.
//
// Two forms, as SURVEY section 5 asks:
//   allgather: every rank's payload padded to a common size `pad` (sticky: agreed once, raised on
//              overflow), one ncclAllGather of the payloads + one of the fused metadata
//              {bytes, stream count, stream offsets}; rank r's payload lands at out + r * pad.
//   sendrecv:  the exact sizes to and from every peer in one ncclGroup (all xGMI links at once
//              instead of a ring); the shards land back to back (rank_base = prefix sums).
// RCCL is bound at the first use with dlopen (librccl.so.1, the copy torch has already mapped when
// the host is Python): a host that never gathers needs no RCCL.
#include "flate_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "flate_kernels.h"

namespace {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
#ifdef FLATE_HIP_TEST_BUILD
    // TEST BUILD ONLY (moonbit-flate_amd/build.py: build_test -> libflate_hip_testbuild.so; the product
    // library is compiled without this block and binds RCCL and nothing else).
    // FLATE_HIP_TEST_TRANSPORT = path of a library with the same nine entry points: the tests'
    // rehearsal transport (tests/rehearsal_transport/, ranks as processes on ONE GPU exchanging
    // through host shared memory), so that the multi-rank branches below can run on a one-GPU box.
    if (const char *alt = getenv("FLATE_HIP_TEST_TRANSPORT")) {
      r.h = dlopen(alt, RTLD_NOW | RTLD_LOCAL);
      if (!r.h) {
        r.err = std::string("cannot load FLATE_HIP_TEST_TRANSPORT: ") + dlerror();
        return;
      }
    }
#endif
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.h) break;
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.h) {
      r.err = std::string("cannot load RCCL: ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(r.h, n);
      if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}

struct Buf {
  void *p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct flate_hip_comm {
  flate_hip_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  bool owned = false;
  int rank = 0, world = 1;
  // the sticky plan (see GatherPlan in moonbit-flate_amd/shard.py): pad and kmax only grow, and every
  // rank raises them to the same values because they come from the same gathered metadata
  uint64_t pad = 0, pad_to = 1ull << 20;
  uint32_t kmax = 0;
  // the exchange runs on its own stream behind an event of the ctx's stream, so that
  // flate_hip_gather_begin returns at once and the next batch compresses beside it
  hipStream_t gstream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  Buf d_meta, d_metas, d_stage;
  // landing pads of the metadata copies: PINNED host memory, so that the asynchronous copies of
  // flate_hip_gather_begin really are asynchronous (a copy to or from pageable memory makes the
  // calling thread wait for everything queued in front of it: the compression and the exchange)
  Buf h_meta, h_metas;
  // the gather in flight (begin .. end)
  bool in_flight = false;
  uint32_t fl_mode = 0;
  std::string err;
};

namespace {

#define G_HIP(cm, expr)                                                     \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) {                                                 \
      (cm)->err = std::string(#expr) + ": " + hipGetErrorString(e_);        \
      flate::ctx_set_error((cm)->ctx, (cm)->err);                           \
      return FLATE_HIP_E_HIP;                                               \
    }                                                                       \
  } while (0)
#define G_NCCL(cm, expr)                                                    \
  do {                                                                      \
    ncclResult_t r_ = (expr);                                               \
    if (r_ != ncclSuccess) {                                                \
      (cm)->err = std::string(#expr) + ": " + rccl()->GetErrorString(r_);   \
      flate::ctx_set_error((cm)->ctx, (cm)->err);                           \
      return FLATE_HIP_E_HIP;                                               \
    }                                                                       \
  } while (0)

int grow(flate_hip_comm *cm, Buf &b, size_t bytes) {
  if (bytes <= b.cap) return FLATE_HIP_OK;
  if (b.p) G_HIP(cm, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  G_HIP(cm, hipMalloc(&b.p, bytes + (bytes >> 2) + 256));
  b.cap = bytes + (bytes >> 2) + 256;
  return FLATE_HIP_OK;
}

int grow_pinned(flate_hip_comm *cm, Buf &b, size_t bytes) {
  if (bytes <= b.cap) return FLATE_HIP_OK;
  if (b.p) G_HIP(cm, hipHostFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  G_HIP(cm, hipHostMalloc(&b.p, bytes + (bytes >> 2) + 256, hipHostMallocDefault));
  b.cap = bytes + (bytes >> 2) + 256;
  return FLATE_HIP_OK;
}

uint64_t round_up(uint64_t v, uint64_t to) {
  const uint64_t r = (v + to - 1) / to * to;
  return r < to ? to : r;
}

int check_offsets(const uint64_t *off, uint32_t k) {
  if (!off || off[0] != 0) return FLATE_HIP_E_INVALID;
  for (uint32_t i = 0; i < k; ++i)
    if (off[i + 1] < off[i]) return FLATE_HIP_E_INVALID;
  return FLATE_HIP_OK;
}

// metadata of one rank: {payload bytes, stream count, off[0 .. kmax]} as u64
size_t meta_words(uint32_t kmax) { return (size_t)kmax + 3; }

// (k may exceed kmax in the overlapped form: the count travels, the offsets that do not fit are left
// out, and every rank then raises the plan and repeats -- see resolve)
void fill_meta(uint64_t *m, const uint64_t *local_off, uint32_t k, uint32_t kmax) {
  memset(m, 0, meta_words(kmax) * 8);
  m[0] = local_off[k];
  m[1] = k;
  for (uint32_t i = 0; i <= k && i <= kmax; ++i) m[2 + i] = local_off[i];
}

// issue the metadata all-gather and the payload exchange on `s`; the sticky plan is final here
int issue(flate_hip_comm *cm, const uint8_t *local, uint64_t local_cap, const uint64_t *local_off, uint32_t k,
          uint8_t *out, uint32_t mode, const uint64_t *peer_bytes, const uint64_t *rank_base, hipStream_t s) {
  Rccl *R = rccl();
  const int W = cm->world;
  const uint64_t clen = local_off[k];
  int rc;
  const size_t mw = meta_words(cm->kmax);
  if ((rc = grow(cm, cm->d_meta, mw * 8))) return rc;
  if ((rc = grow(cm, cm->d_metas, mw * 8 * (size_t)W))) return rc;
  if ((rc = grow_pinned(cm, cm->h_meta, mw * 8))) return rc;
  if ((rc = grow_pinned(cm, cm->h_metas, mw * 8 * (size_t)W))) return rc;
  // (h_meta is free again: at most one exchange is in flight and the last one has been waited for)
  fill_meta((uint64_t *)cm->h_meta.p, local_off, k, cm->kmax);
  G_HIP(cm, hipMemcpyAsync(cm->d_meta.p, cm->h_meta.p, mw * 8, hipMemcpyHostToDevice, s));
  G_NCCL(cm, R->AllGather(cm->d_meta.p, cm->d_metas.p, mw, ncclUint64, cm->comm, s));
  if (mode == FLATE_HIP_GATHER_ALLGATHER) {
    const uint8_t *src = local;
    if (local_cap < cm->pad) {  // the collective reads `pad` bytes from every rank: stage a short buffer
      if ((rc = grow(cm, cm->d_stage, cm->pad))) return rc;
      if (clen) G_HIP(cm, hipMemcpyAsync(cm->d_stage.p, local, clen, hipMemcpyDeviceToDevice, s));
      src = (const uint8_t *)cm->d_stage.p;
    }
    G_NCCL(cm, R->AllGather(src, out, cm->pad, ncclUint8, cm->comm, s));
  } else {
    G_NCCL(cm, R->GroupStart());
    ncclResult_t bad = ncclSuccess;  // (a group that was started is always ended)
    for (int r = 0; r < W && bad == ncclSuccess; ++r) {
      if (r == cm->rank) continue;
      if (clen) bad = R->Send(local, clen, ncclUint8, r, cm->comm, s);
      if (bad == ncclSuccess && peer_bytes[r])
        bad = R->Recv(out + rank_base[r], peer_bytes[r], ncclUint8, r, cm->comm, s);
    }
    const ncclResult_t ended = R->GroupEnd();
    G_NCCL(cm, bad);
    G_NCCL(cm, ended);
    if (clen)
      G_HIP(cm, hipMemcpyAsync(out + rank_base[cm->rank], local, clen, hipMemcpyDeviceToDevice, s));
  }
  return FLATE_HIP_OK;
}

// after the stream has drained: the global index from the gathered metadata
int resolve(flate_hip_comm *cm, uint32_t mode, uint64_t *stream_off, uint64_t *stream_len, uint64_t index_cap,
            uint64_t *total_streams, bool *overflow) {
  const int W = cm->world;
  const size_t mw = meta_words(cm->kmax);
  const uint64_t *m = (const uint64_t *)cm->h_metas.p;
  uint64_t total = 0, max_bytes = 0, max_k = 0;
  for (int r = 0; r < W; ++r) {
    total += m[r * mw + 1];
    if (m[r * mw] > max_bytes) max_bytes = m[r * mw];
    if (m[r * mw + 1] > max_k) max_k = m[r * mw + 1];
  }
  if (total_streams) *total_streams = total;
  // A shard that outgrew the pad, or a rank with more streams than the metadata block holds: both are
  // data-dependent, so neither may be refused by one rank alone (its peers would wait in the
  // collective).  Every rank has taken part, every rank sees the same metadata, every rank raises
  // the plan alike and reports FLATE_HIP_E_AGAIN.
  *overflow = (mode == FLATE_HIP_GATHER_ALLGATHER && max_bytes > cm->pad) || max_k > cm->kmax;
  if (*overflow) {
    if (max_bytes > cm->pad) cm->pad = round_up(max_bytes, cm->pad_to);
    if (max_k > cm->kmax) cm->kmax = (uint32_t)max_k;
    return FLATE_HIP_E_AGAIN;
  }
  if (total > index_cap || !stream_off || !stream_len) return total ? FLATE_HIP_E_OUT_TOO_SMALL : FLATE_HIP_OK;
  uint64_t j = 0, base = 0;
  for (int r = 0; r < W; ++r) {
    const uint64_t *o = m + r * mw + 2;
    const uint64_t kr = m[r * mw + 1];
    if (mode == FLATE_HIP_GATHER_ALLGATHER) base = (uint64_t)r * cm->pad;
    for (uint64_t i = 0; i < kr; ++i, ++j) {
      stream_off[j] = base + o[i];
      stream_len[j] = o[i + 1] - o[i];
    }
    if (mode != FLATE_HIP_GATHER_ALLGATHER) base += m[r * mw];
  }
  return FLATE_HIP_OK;
}

}  // namespace

extern "C" {

int flate_hip_gather_layout(uint32_t world, const uint64_t *rank_bytes, uint64_t pad_to, uint32_t mode,
                            uint64_t *pad, uint64_t *rank_base, uint64_t *out_bytes) {
  if (!world || !rank_bytes || !pad_to || mode > FLATE_HIP_GATHER_SENDRECV) return FLATE_HIP_E_INVALID;
  uint64_t mx = 0, sum = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (rank_bytes[r] > mx) mx = rank_bytes[r];
    sum += rank_bytes[r];
  }
  const uint64_t p = round_up(mx, pad_to);
  if (pad) *pad = p;
  uint64_t at = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (rank_base) rank_base[r] = mode == FLATE_HIP_GATHER_ALLGATHER ? (uint64_t)r * p : at;
    at += rank_bytes[r];
  }
  if (out_bytes) *out_bytes = mode == FLATE_HIP_GATHER_ALLGATHER ? (uint64_t)world * p : sum;
  return FLATE_HIP_OK;
}

int flate_hip_comm_unique_id(uint8_t id[FLATE_HIP_UNIQUE_ID_BYTES]) {
  static_assert(FLATE_HIP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  Rccl *R = rccl();
  if (!id) return FLATE_HIP_E_INVALID;
  if (!R->err.empty()) return FLATE_HIP_E_HIP;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return FLATE_HIP_E_HIP;
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return FLATE_HIP_OK;
}

static int comm_new(flate_hip_ctx *ctx, int rank, int world, flate_hip_comm **out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return FLATE_HIP_E_INVALID;
  *out = nullptr;
  Rccl *R = rccl();
  if (!R->err.empty()) {
    flate::ctx_set_error(ctx, R->err);
    return FLATE_HIP_E_HIP;
  }
  flate_hip_comm *cm = new flate_hip_comm();
  cm->ctx = ctx;
  cm->rank = rank;
  cm->world = world;
  if (hipSetDevice(flate::ctx_device(ctx)) != hipSuccess ||
      hipStreamCreateWithFlags(&cm->gstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&cm->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&cm->ev_done, hipEventDisableTiming) != hipSuccess) {
    flate_hip_comm_destroy(cm);
    return FLATE_HIP_E_HIP;
  }
  *out = cm;
  return FLATE_HIP_OK;
}

int flate_hip_comm_init(flate_hip_ctx *ctx, const uint8_t id[FLATE_HIP_UNIQUE_ID_BYTES], int rank, int world,
                        flate_hip_comm **out) {
  if (!id) return FLATE_HIP_E_INVALID;
  int rc = comm_new(ctx, rank, world, out);
  if (rc) return rc;
  flate_hip_comm *cm = *out;
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = rccl()->CommInitRank(&cm->comm, world, u, rank);
  if (r != ncclSuccess) {
    flate::ctx_set_error(ctx, std::string("ncclCommInitRank: ") + rccl()->GetErrorString(r));
    flate_hip_comm_destroy(cm);
    *out = nullptr;
    return FLATE_HIP_E_HIP;
  }
  cm->owned = true;
  return FLATE_HIP_OK;
}

int flate_hip_comm_wrap(flate_hip_ctx *ctx, void *nccl_comm, int rank, int world, flate_hip_comm **out) {
  if (!nccl_comm) return FLATE_HIP_E_INVALID;
  int rc = comm_new(ctx, rank, world, out);
  if (rc) return rc;
  (*out)->comm = (ncclComm_t)nccl_comm;
  (*out)->owned = false;
  return FLATE_HIP_OK;
}

void flate_hip_comm_destroy(flate_hip_comm *cm) {
  if (!cm) return;
  if (cm->gstream) (void)hipStreamSynchronize(cm->gstream);
  if (cm->owned && cm->comm) (void)rccl()->CommDestroy(cm->comm);
  for (Buf *b : {&cm->d_meta, &cm->d_metas, &cm->d_stage})
    if (b->p) (void)hipFree(b->p);
  for (Buf *b : {&cm->h_meta, &cm->h_metas})
    if (b->p) (void)hipHostFree(b->p);
  if (cm->ev_ready) (void)hipEventDestroy(cm->ev_ready);
  if (cm->ev_done) (void)hipEventDestroy(cm->ev_done);
  if (cm->gstream) (void)hipStreamDestroy(cm->gstream);
  delete cm;
}

int flate_hip_comm_plan(flate_hip_comm *cm, uint64_t *pad, uint32_t *max_streams) {
  if (!cm) return FLATE_HIP_E_INVALID;
  if (pad) *pad = cm->pad;
  if (max_streams) *max_streams = cm->kmax;
  return FLATE_HIP_OK;
}

// Blocking form.  Agrees on the sizes first (24 bytes per rank), so it needs no plan; it leaves the
// plan (pad, largest stream count) behind for flate_hip_gather_begin.
int flate_hip_gather_compressed(flate_hip_comm *cm, const uint8_t *local, uint64_t local_cap,
                                const uint64_t *local_off, uint32_t k, uint8_t *out, uint64_t out_cap,
                                uint64_t *stream_off, uint64_t *stream_len, uint64_t index_cap,
                                uint64_t *total_streams, uint32_t mode) {
  if (!cm || !out || mode > FLATE_HIP_GATHER_SENDRECV || cm->in_flight) return FLATE_HIP_E_INVALID;
  int rc = check_offsets(local_off, k);
  if (rc) return rc;
  const uint64_t clen = local_off[k];
  if ((clen && !local) || local_cap < clen) return FLATE_HIP_E_INVALID;
  cm->err.clear();
  G_HIP(cm, hipSetDevice(flate::ctx_device(cm->ctx)));
  Rccl *R = rccl();
  const int W = cm->world;
  hipStream_t s = flate::ctx_stream(cm->ctx);
  // 1. sizes: {payload bytes, stream count, room in my out} of every rank
  if ((rc = grow(cm, cm->d_meta, 3 * 8))) return rc;
  if ((rc = grow(cm, cm->d_metas, 3 * 8 * (size_t)W))) return rc;
  const uint64_t mine[3] = {clen, k, out_cap};
  G_HIP(cm, hipMemcpyAsync(cm->d_meta.p, mine, 24, hipMemcpyHostToDevice, s));
  G_NCCL(cm, R->AllGather(cm->d_meta.p, cm->d_metas.p, 3, ncclUint64, cm->comm, s));
  std::vector<uint64_t> every(3 * (size_t)W), bytes(W), base(W);
  G_HIP(cm, hipMemcpyAsync(every.data(), cm->d_metas.p, 24 * (size_t)W, hipMemcpyDeviceToHost, s));
  G_HIP(cm, hipStreamSynchronize(s));
  uint64_t kmx = 0, min_cap = ~0ull;
  for (int r = 0; r < W; ++r) {
    bytes[r] = every[3 * r];
    if (every[3 * r + 1] > kmx) kmx = every[3 * r + 1];
    if (every[3 * r + 2] < min_cap) min_cap = every[3 * r + 2];
  }
  uint64_t pad = 0, need = 0;
  (void)flate_hip_gather_layout((uint32_t)W, bytes.data(), cm->pad_to, mode, &pad, base.data(), &need);
  if (pad > cm->pad) cm->pad = pad;
  if (kmx > cm->kmax) cm->kmax = (uint32_t)kmx;
  if (mode == FLATE_HIP_GATHER_ALLGATHER) {
    need = (uint64_t)W * cm->pad;
    for (int r = 0; r < W; ++r) base[r] = (uint64_t)r * cm->pad;
  }
  // (decided from gathered values only: every rank takes the same branch, nobody is left in a collective)
  if (min_cap < need) return FLATE_HIP_E_OUT_TOO_SMALL;
  // 2. metadata + payload
  if ((rc = issue(cm, local, local_cap, local_off, k, out, mode, bytes.data(), base.data(), s))) return rc;
  const size_t mw = meta_words(cm->kmax);
  G_HIP(cm, hipMemcpyAsync(cm->h_metas.p, cm->d_metas.p, mw * 8 * (size_t)W, hipMemcpyDeviceToHost, s));
  G_HIP(cm, hipStreamSynchronize(s));
  bool overflow = false;
  return resolve(cm, mode, stream_off, stream_len, index_cap, total_streams, &overflow);
}

// Overlapped form (padded all-gather with the sticky plan): nothing here waits for the GPU or for
// a peer.  The exchange starts when the work already queued on the ctx's stream (the compression
// that produced `local`) has finished, and runs on the communicator's own stream.
int flate_hip_gather_begin(flate_hip_comm *cm, const uint8_t *local, uint64_t local_cap, const uint64_t *local_off,
                           uint32_t k, uint8_t *out, uint64_t out_cap) {
  if (!cm || !out || cm->in_flight) return FLATE_HIP_E_INVALID;
  int rc = check_offsets(local_off, k);
  if (rc) return rc;
  const uint64_t clen = local_off[k];
  if ((clen && !local) || local_cap < clen) return FLATE_HIP_E_INVALID;
  // A plan must exist (one blocking call, or flate_hip_comm_set_plan on every rank).  The two
  // refusals here depend on the plan, the world size and out_cap only -- values the caller must keep
  // equal on all ranks -- so every rank takes the same branch and nobody is left in the collective.
  // What depends on the data (a shard larger than the pad, more streams than the plan holds) goes
  // through the exchange and comes back from flate_hip_gather_end as FLATE_HIP_E_AGAIN on every rank.
  if (cm->pad == 0 || cm->kmax == 0) return FLATE_HIP_E_INVALID;
  if (out_cap < (uint64_t)cm->world * cm->pad) return FLATE_HIP_E_OUT_TOO_SMALL;
  cm->err.clear();
  G_HIP(cm, hipSetDevice(flate::ctx_device(cm->ctx)));
  G_HIP(cm, hipEventRecord(cm->ev_ready, flate::ctx_stream(cm->ctx)));
  G_HIP(cm, hipStreamWaitEvent(cm->gstream, cm->ev_ready, 0));
  if ((rc = issue(cm, local, local_cap, local_off, k, out, FLATE_HIP_GATHER_ALLGATHER, nullptr, nullptr, cm->gstream)))
    return rc;
  const size_t mw = meta_words(cm->kmax);
  G_HIP(cm, hipMemcpyAsync(cm->h_metas.p, cm->d_metas.p, mw * 8 * (size_t)cm->world, hipMemcpyDeviceToHost,
                           cm->gstream));
  G_HIP(cm, hipEventRecord(cm->ev_done, cm->gstream));
  cm->in_flight = true;
  return FLATE_HIP_OK;
}

int flate_hip_gather_end(flate_hip_comm *cm, uint64_t *stream_off, uint64_t *stream_len, uint64_t index_cap,
                         uint64_t *total_streams) {
  if (!cm || !cm->in_flight) return FLATE_HIP_E_INVALID;
  cm->in_flight = false;
  G_HIP(cm, hipSetDevice(flate::ctx_device(cm->ctx)));
  G_HIP(cm, hipStreamSynchronize(cm->gstream));
  // later work on the ctx's stream may overwrite `local` / read `out`: order it behind the exchange
  G_HIP(cm, hipStreamWaitEvent(flate::ctx_stream(cm->ctx), cm->ev_done, 0));
  bool overflow = false;
  return resolve(cm, FLATE_HIP_GATHER_ALLGATHER, stream_off, stream_len, index_cap, total_streams, &overflow);
}

int flate_hip_comm_set_plan(flate_hip_comm *cm, uint64_t pad, uint32_t max_streams) {
  if (!cm || cm->in_flight) return FLATE_HIP_E_INVALID;
  if (pad) cm->pad = round_up(pad, cm->pad_to);
  if (max_streams > cm->kmax) cm->kmax = max_streams;
  return FLATE_HIP_OK;
}

}  // extern "C"
