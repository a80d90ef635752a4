"""Multi-GPU row of the scope table: independent streams shard by contiguous
index range, one process per GPU, no collective in the compress path; the only exchange step is
the concatenation of the compressed shards (RCCL over xGMI, `nccl` backend; `gloo` on CPU for
tests).

Two forms of the exchange (SURVEY section 5 asks for both to be reported):

* mode="allgather": one `all_gather_into_tensor` of every rank's payload padded to a common size
  plus one fused `all_gather_into_tensor` of the metadata (size, stream count, per-stream index).
  Nothing on the issue path waits for the GPU or for a peer: the pad is *sticky* (agreed once, at
  the first call or after an overflow) and the metadata is read on the host only when the result
  is consumed (`wait()`), so the gather of batch k really overlaps the compression of batch k+1.
* mode="sendrecv": grouped `isend/irecv` of the exact sizes to and from every peer, all xGMI
  links at once instead of a ring; the sizes must be known before the receives are posted, so
  this form reads the 16-byte-per-rank size vector on the host first.
"""
import numpy as np


def shard_range(n_streams, rank, world):
    """Contiguous stream range [lo, hi) of `rank` (GPU g takes streams [g*N/G, (g+1)*N/G))."""
    return n_streams * rank // world, n_streams * (rank + 1) // world


def max_shard_streams(n_streams, world):
    """Largest stream count any rank gets from shard_range (known without communication)."""
    return max(shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0]
               for r in range(world)) if world > 0 else 0


class GatherPlan:
    """What all ranks must agree on before a sync-free gather: the padded payload size and the
    largest per-rank stream count.  `pad` only grows; every rank computes the same value because it
    is derived from the gathered metadata of the same batch."""

    def __init__(self, kmax, pad=0, pad_to=1 << 20):
        self.kmax, self.pad, self.pad_to = int(kmax), int(pad), int(pad_to)

    def round(self, nbytes):
        return max((int(nbytes) + self.pad_to - 1) // self.pad_to * self.pad_to, self.pad_to)


class GatheredStreams:
    """All ranks' compressed streams on this rank: rank r's payload sits at buf[r*pad : r*pad+size[r]],
    global stream j (rank-major order) at buf[off[j] : off[j] + length[j]]."""

    def __init__(self, buf, pad, world, kmax, metas=None, works=(), keep=(), plan=None):
        self.buf, self.pad, self.world, self.kmax, self.plan = buf, pad, world, kmax, plan
        self._metas = metas      # device/host int64[world * (kmax + 3)], resolved lazily
        self._works = list(works)
        self._keep = keep        # tensors that must outlive the collectives
        self._resolved = False
        self.sizes = self.counts = self.off = self.length = None
        self.overflow = False    # some rank's payload did not fit `pad` (caller must redo)

    def wait(self):
        """Wait for the collectives and resolve the index on the host; idempotent."""
        for w in self._works:
            w.wait()
        self._works = []
        if not self._resolved:
            m = self._metas.cpu().numpy().reshape(self.world, self.kmax + 3)
            self.sizes, self.counts = m[:, 0].copy(), m[:, 1].copy()
            self.overflow = bool((self.sizes > self.pad).any())
            if self.overflow and self.plan is not None:  # every rank raises it to the same value
                self.plan.pad = max(self.plan.pad, self.plan.round(int(self.sizes.max())))
            off, length = [], []
            for r in range(self.world):
                o = m[r, 2:2 + int(self.counts[r]) + 1].astype(np.uint64)
                off.append(o[:-1] + np.uint64(r * self.pad))
                length.append(o[1:] - o[:-1])
            self.off = np.concatenate(off) if off else np.zeros(0, np.uint64)
            self.length = np.concatenate(length) if length else np.zeros(0, np.uint64)
            self._resolved = True
            self._keep = ()
        return self

    def stream(self, j):
        self.wait()
        o = int(self.off[j])
        return self.buf[o:o + int(self.length[j])]


def _meta_tensor(local_off, kmax, dev):
    import torch
    k = int(local_off.size - 1)
    m = np.zeros(kmax + 3, dtype=np.int64)
    m[0], m[1] = int(local_off[-1]), k
    m[2:2 + k + 1] = local_off.astype(np.int64)
    return torch.from_numpy(m).to(dev)


def gather_compressed(dist, local_buf, local_off, buf=None, plan=None, pad_to=1 << 20, wait=True,
                      mode="allgather"):
    """Concatenate every rank's compressed shard on every rank.

    local_buf: torch uint8 tensor holding this rank's streams back to back (may be larger than the
    payload); local_off: numpy uint64[k+1] offsets.  `plan` (GatherPlan) carries the sticky pad
    between calls; without one a blocking size exchange agrees on it first.  wait=False leaves the
    collectives in flight (call .wait() on the result before reading buf or overwriting
    local_buf).  If result.overflow is set after wait(), a payload exceeded plan.pad: plan.pad has
    been raised and the call must be repeated for that batch."""
    import torch
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local_buf.is_cuda:
        # rehearsal of the multi-GPU path on a box without RCCL peers (bench.py
        # FLATE_BENCH_BACKEND=gloo): gloo moves host memory, so stage through the CPU
        g = gather_compressed(dist, local_buf.cpu(), local_off, buf=None, plan=plan, pad_to=pad_to,
                              wait=True, mode=mode)
        g.buf = g.buf.to(local_buf.device)
        return g
    dev = local_buf.device
    k = int(local_off.size - 1)
    clen = int(local_off[-1])
    if plan is None:
        plan = GatherPlan(kmax=0, pad=0, pad_to=pad_to)
    if plan.pad == 0 or plan.kmax == 0 or mode == "sendrecv":
        # blocking agreement on (largest payload, largest stream count): 16 bytes per rank
        mine = torch.tensor([clen, k], dtype=torch.int64, device=dev)
        every = torch.empty(2 * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(every, mine)
        every = every.cpu().numpy().reshape(world, 2)
        plan.pad = max(plan.pad, plan.round(every[:, 0].max()))
        plan.kmax = max(plan.kmax, int(every[:, 1].max()))
        peer_sizes = every[:, 0]
    if k > plan.kmax:
        raise ValueError("rank holds %d streams, plan allows %d" % (k, plan.kmax))
    pad, kmax = plan.pad, plan.kmax
    meta = _meta_tensor(local_off, kmax, dev)
    metas = torch.empty(world * (kmax + 3), dtype=torch.int64, device=dev)
    works = [dist.all_gather_into_tensor(metas, meta, async_op=True)]
    if buf is None or buf.numel() < pad * world:
        buf = torch.empty(pad * world, dtype=torch.uint8, device=dev)
    keep = [meta]
    if mode == "allgather":
        if local_buf.numel() < pad:
            grown = torch.zeros(pad, dtype=torch.uint8, device=dev)
            grown[:local_buf.numel()] = local_buf
            local_buf = grown
        keep.append(local_buf)
        works.append(dist.all_gather_into_tensor(buf[:pad * world], local_buf[:pad], async_op=True))
    elif mode == "sendrecv":
        rank = dist.get_rank()
        ops = []
        for r in range(world):
            if r == rank:
                continue
            if clen:
                ops.append(dist.P2POp(dist.isend, local_buf[:clen], r))
            if int(peer_sizes[r]):
                ops.append(dist.P2POp(dist.irecv, buf[r * pad:r * pad + int(peer_sizes[r])], r))
        buf[rank * pad:rank * pad + clen].copy_(local_buf[:clen])
        keep.append(local_buf)
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    else:
        raise ValueError("mode must be 'allgather' or 'sendrecv'")
    g = GatheredStreams(buf, pad, world, kmax, metas=metas, works=works, keep=tuple(keep), plan=plan)
    if wait:
        g.wait()
        if g.overflow:  # a payload outgrew the sticky pad (raised by wait()): repeat
            return gather_compressed(dist, local_buf, local_off, buf=None, plan=plan, pad_to=pad_to,
                                     wait=True, mode=mode)
    return g


# ------------------------------------------------------------------------------------------------
# The same exchange through the C ABI (include/flate_hip.h: flate_hip_comm_*, flate_hip_gather_*):
# what a host in the reference's language calls.  The library owns the RCCL communicator and the
# exchange's HIP stream; this class only marshals buffers.  (The torch.distributed form above stays
# for the CPU rehearsal with gloo, where there is no RCCL.)
# ------------------------------------------------------------------------------------------------
MODES = {"allgather": 0, "sendrecv": 1}
E_AGAIN = -9


def gather_layout(rank_bytes, mode="allgather", pad_to=1 << 20):
    """(pad, rank_base[], out_bytes) as flate_hip_gather_layout computes them (host arithmetic)."""
    import ctypes as C
    from . import _lib
    L = _lib.load()
    rb = np.ascontiguousarray(rank_bytes, dtype=np.uint64)
    base = np.zeros(rb.size, dtype=np.uint64)
    pad, need = C.c_uint64(0), C.c_uint64(0)
    rc = L.flate_hip_gather_layout(rb.size, rb.ctypes.data, int(pad_to), MODES[mode], C.byref(pad),
                                   base.ctypes.data, C.byref(need))
    if rc != 0:
        raise ValueError("flate_hip_gather_layout: %d" % rc)
    return int(pad.value), base, int(need.value)


class NativeGathered:
    """Result of a C-ABI exchange: global stream j (rank-major) = buf[off[j] : off[j] + length[j]]."""

    def __init__(self, buf, off, length, pad):
        self.buf, self.off, self.length, self.pad = buf, off, length, pad

    def stream(self, j):
        o = int(self.off[j])
        return self.buf[o:o + int(self.length[j])]


class NativeComm:
    """One flate_hip_comm: an RCCL communicator created by the library from a unique id that rank 0
    makes and `dist` (any torch.distributed backend; None for a one-rank communicator) broadcasts."""

    def __init__(self, eng, rank=0, world=1, dist=None):
        import ctypes as C
        from . import _lib
        self._L, self._eng, self.rank, self.world = _lib.load(), eng, int(rank), int(world)
        uid = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            eng._check(self._L.flate_hip_comm_unique_id(uid.ctypes.data))
        if dist is not None and world > 1:
            import torch
            t = torch.from_numpy(uid)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0)
            uid = t.cpu().numpy()
        self._comm = C.c_void_p()
        eng._check(self._L.flate_hip_comm_init(eng._ctx, uid.ctypes.data, self.rank, self.world,
                                               C.byref(self._comm)))

    @staticmethod
    def probe(eng):
        """Can this process bind RCCL for the C-ABI exchange?  No collective: flate_hip_comm_unique_id binds the
        library at its first use and fails with a message if it cannot (raises FlateError then).  Callers that
        are about to create communicators on several ranks agree on this first, so that no rank is left alone
        in the collective set-up."""
        from . import _lib
        uid = np.zeros(128, dtype=np.uint8)
        eng._check(_lib.load().flate_hip_comm_unique_id(uid.ctypes.data))

    def close(self):
        if self._comm:
            self._L.flate_hip_comm_destroy(self._comm)
            self._comm = None

    def plan(self):
        import ctypes as C
        pad, kmax = C.c_uint64(0), C.c_uint32(0)
        self._L.flate_hip_comm_plan(self._comm, C.byref(pad), C.byref(kmax))
        return int(pad.value), int(kmax.value)

    def set_plan(self, pad, max_streams):
        self._eng._check(self._L.flate_hip_comm_set_plan(self._comm, int(pad), int(max_streams)))

    def _out(self, local_buf, need, out):
        import torch
        if out is None or out.numel() < need:
            out = torch.empty(max(need, 16), dtype=torch.uint8, device=local_buf.device)
        return out

    def gather(self, local_buf, local_off, out=None, mode="allgather", total_streams_cap=None, claim_out_cap=None):
        """Blocking exchange (flate_hip_gather_compressed).  claim_out_cap: the capacity to report for
        `out` instead of its size (tests: FLATE_HIP_E_OUT_TOO_SMALL must come back on EVERY rank)."""
        import ctypes as C
        local_off = np.ascontiguousarray(local_off, dtype=np.uint64)
        k = local_off.size - 1
        cap = int(total_streams_cap or max(k, 1) * self.world * 2 + 16)
        # room for the padded form whatever the peers hold: the caller's buffer or world x bound
        need = (int(local_buf.numel()) + (1 << 20)) * self.world
        out = self._out(local_buf, need, out)
        off, length = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        total = C.c_uint64(0)
        self._eng._check(self._L.flate_hip_gather_compressed(
            self._comm, local_buf.data_ptr(), local_buf.numel(), local_off.ctypes.data, k,
            out.data_ptr(), out.numel() if claim_out_cap is None else int(claim_out_cap),
            off.ctypes.data, length.ctypes.data, cap, C.byref(total), MODES[mode]))
        t = int(total.value)
        return NativeGathered(out, off[:t], length[:t], self.plan()[0])

    def begin(self, local_buf, local_off, out):
        """Overlapped exchange, padded form (flate_hip_gather_begin): returns at once."""
        local_off = np.ascontiguousarray(local_off, dtype=np.uint64)
        self._keep = (local_buf, local_off, out)
        self._eng._check(self._L.flate_hip_gather_begin(
            self._comm, local_buf.data_ptr(), local_buf.numel(), local_off.ctypes.data,
            local_off.size - 1, out.data_ptr(), out.numel()))

    def end(self, total_streams_cap):
        """Waits for the exchange begun last; returns NativeGathered, or None if a shard outgrew the
        pad or a rank holds more streams than the plan allows (FLATE_HIP_E_AGAIN on every rank: the
        plan has been raised, repeat the batch with gather())."""
        import ctypes as C
        cap = int(total_streams_cap)
        off, length = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        total = C.c_uint64(0)
        rc = self._L.flate_hip_gather_end(self._comm, off.ctypes.data, length.ctypes.data, cap, C.byref(total))
        out = self._keep[2]
        self._keep = None
        if rc == E_AGAIN:
            return None
        self._eng._check(rc)
        t = int(total.value)
        return NativeGathered(out, off[:t], length[:t], self.plan()[0])
