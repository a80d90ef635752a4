// rehearsal_rccl.cpp -- TEST INFRASTRUCTURE, never part of libflate_hip.so.
//
// A stand-in for the nine RCCL entry points csrc/gather.hip binds with dlopen, for boxes with ONE
// GPU: the ranks are processes that share the card (RCCL itself refuses two ranks on one device) and
// exchange through a POSIX shared-memory segment named by the unique id.  It exists so that the
// multi-rank branches of the C-ABI exchange (rank_base placement, peer sizes, the grouped
// send/receive loop, the plan raised on every rank) run somewhere before a multi-GPU node is
// available; it measures nothing and overlaps nothing: every call drains the stream it is given,
// copies through the host and meets the other ranks at a barrier.
// Selected with FLATE_HIP_TEST_TRANSPORT=<this library> (gather.hip: rccl()).
//
//   build: g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rehearsal_rccl.cpp \
//              -L/opt/rocm/lib -lamdhip64 -lrt -lpthread -o librehearsal_rccl.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x52454852u;  // "REHR"
constexpr int kMaxWorld = 8;

struct Header {
  std::atomic<uint32_t> magic;
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> failed;  // a rank gave up: everybody else does too
  uint32_t world;
  uint64_t slot_bytes;
  uint64_t posted[kMaxWorld][kMaxWorld];  // bytes rank s has put into its slot for rank d
};

struct Op {
  int kind;  // 0 send, 1 recv
  void *buf;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local struct ncclComm *g_group_comm = nullptr;

size_t dt_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct ncclComm {
  Header *h = nullptr;
  uint8_t *slots = nullptr;
  size_t map_bytes = 0;
  int rank = 0, world = 1;
  std::string name;
  double timeout_s = 120.0;
  uint8_t *slot(int src, int dst) { return slots + ((size_t)src * world + dst) * h->slot_bytes; }
  // every rank arrives; false = a peer never came (or gave up)
  bool barrier() {
    const uint32_t gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == (uint32_t)world) {
      h->arrived.store(0);
      h->generation.fetch_add(1);
      return h->failed.load() == 0;
    }
    const double t0 = now_s();
    while (h->generation.load() == gen) {
      if (h->failed.load() || now_s() - t0 > timeout_s) {
        h->failed.store(1);
        return false;
      }
      usleep(50);
    }
    return h->failed.load() == 0;
  }
};

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "rehearsal transport: invalid argument (message larger than a mailbox?)";
    case ncclSystemError: return "rehearsal transport: shared memory / a peer never arrived";
    case ncclUnhandledCudaError: return "rehearsal transport: HIP error";
    default: return "rehearsal transport: error";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/flate_rehearsal_%d_%llx", (int)getpid(),
           (unsigned long long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return ncclInvalidArgument;
  ncclComm *c = new ncclComm();
  c->rank = rank;
  c->world = world;
  c->name.assign(id.internal, strnlen(id.internal, NCCL_UNIQUE_ID_BYTES));
  uint64_t slot_mb = 24;
  if (const char *e = getenv("FLATE_REHEARSAL_SLOT_MB")) slot_mb = strtoull(e, nullptr, 10);
  if (const char *e = getenv("FLATE_REHEARSAL_TIMEOUT_S")) c->timeout_s = atof(e);
  const uint64_t slot_bytes = slot_mb << 20;
  c->map_bytes = 4096 + (size_t)world * world * slot_bytes;
  bool creator = true;
  int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) {
    creator = false;
    fd = shm_open(c->name.c_str(), O_RDWR, 0600);
  }
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->h = reinterpret_cast<Header *>(p);
  c->slots = reinterpret_cast<uint8_t *>(p) + 4096;
  if (creator) {  // (a fresh segment is zero-filled)
    c->h->world = (uint32_t)world;
    c->h->slot_bytes = slot_bytes;
    c->h->magic.store(kMagic);
  }
  const double t0 = now_s();
  while (c->h->magic.load() != kMagic)
    if (now_s() - t0 > c->timeout_s) return ncclSystemError;
  if (c->h->world != (uint32_t)world) return ncclInvalidArgument;
  c->h->attached.fetch_add(1);
  while (c->h->attached.load() < (uint32_t)world)
    if (now_s() - t0 > c->timeout_s) return ncclSystemError;
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  (void)shm_unlink(c->name.c_str());  // (the mapping of the peers lives on until they unmap)
  if (c->h) munmap(c->h, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c,
                           hipStream_t s) {
  if (!c || g_depth) return ncclInvalidArgument;
  const size_t bytes = count * dt_size(dt);
  if (bytes > c->h->slot_bytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (bytes && hipMemcpy(c->slot(c->rank, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  c->h->posted[c->rank][c->rank] = bytes;
  if (!c->barrier()) return ncclSystemError;
  for (int r = 0; r < c->world; ++r) {
    if (c->h->posted[r][r] != bytes) return ncclInvalidArgument;  // every rank gathers the same count
    if (bytes && hipMemcpy((uint8_t *)recv + (size_t)r * bytes, c->slot(r, r), bytes, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  if (!c->barrier()) return ncclSystemError;  // nobody posts again before everybody has read
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

static ncclResult_t run_group(ncclComm *c) {
  ncclResult_t res = ncclSuccess;
  for (const Op &o : g_ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) res = ncclUnhandledCudaError;
  for (int d = 0; d < c->world; ++d) c->h->posted[c->rank][d] = ~0ull;  // "nothing for you"
  for (const Op &o : g_ops) {
    if (o.kind != 0 || res != ncclSuccess) continue;
    if (o.bytes > c->h->slot_bytes || c->h->posted[c->rank][o.peer] != ~0ull) {  // one message per peer and group
      res = ncclInvalidArgument;
      continue;
    }
    if (hipMemcpy(c->slot(c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
      res = ncclUnhandledCudaError;
    c->h->posted[c->rank][o.peer] = o.bytes;
  }
  if (!c->barrier()) return ncclSystemError;
  for (const Op &o : g_ops) {
    if (o.kind != 1 || res != ncclSuccess) continue;
    if (c->h->posted[o.peer][c->rank] != o.bytes) {  // the receive a peer never matched would hang RCCL
      res = ncclInvalidArgument;
      continue;
    }
    if (hipMemcpy(o.buf, c->slot(o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess)
      res = ncclUnhandledCudaError;
  }
  if (!c->barrier()) return ncclSystemError;
  return res;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidArgument;
  if (--g_depth) return ncclSuccess;
  ncclComm *c = g_group_comm;
  g_group_comm = nullptr;
  ncclResult_t res = ncclSuccess;
  if (c) res = run_group(c);
  g_ops.clear();
  return res;
}

static ncclResult_t post(int kind, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c,
                         hipStream_t s) {
  if (!c || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
  if (g_group_comm && g_group_comm != c) return ncclInvalidArgument;
  g_group_comm = c;
  g_ops.push_back({kind, buf, count * dt_size(dt), peer, s});
  if (g_depth) return ncclSuccess;
  ++g_depth;  // outside a group: a group of one (both sides must call, as with RCCL)
  return ncclGroupEnd();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
  return post(0, const_cast<void *>(buf), count, dt, peer, c, s);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
  return post(1, buf, count, dt, peer, c, s);
}

}  // extern "C"
