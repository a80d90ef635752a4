"""Multi-GPU row of the scope table: independent streams shard by contiguous
index range, one process per GPU, no collective in the compress path; the only exchange step is
the concatenation of the compressed shards (RCCL all-gather over xGMI, `nccl` backend; `gloo` on
CPU for tests)."""
import numpy as np


def shard_range(n_streams, rank, world):
    """Contiguous stream range [lo, hi) of `rank` (GPU g takes streams [g*N/G, (g+1)*N/G))."""
    return n_streams * rank // world, n_streams * (rank + 1) // world


class GatheredStreams:
    """All ranks' compressed streams on this rank: rank r's payload sits at buf[r*pad : r*pad+size[r]],
    global stream j (rank-major order) at buf[off[j] : off[j] + length[j]]."""

    def __init__(self, buf, pad, sizes, counts, off, length, work=None):
        self.buf, self.pad, self.sizes, self.counts, self.off, self.length = \
            buf, pad, sizes, counts, off, length
        self.work = work  # pending payload collective (gather_compressed(..., wait=False)) or None

    def wait(self):
        """Make the current stream (and for gloo the host) wait for the payload; idempotent."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self

    def stream(self, j):
        o = int(self.off[j])
        return self.buf[o:o + int(self.length[j])]


def gather_compressed(dist, local_buf, local_off, buf=None, pad_to=1 << 20, wait=True):
    """Concatenate every rank's compressed shard on every rank.

    wait=False leaves the payload all-gather in flight (call .wait() on the result before reading
    buf or overwriting local_buf): the next batch can then be compressed while xGMI moves this one.

    local_buf: torch uint8 tensor holding this rank's streams back to back (may be larger than the
    payload); local_off: numpy uint64[k+1] offsets.  Uses two all-gathers: the (size, count, index)
    metadata and the payload padded to the largest shard (all_gather needs equal counts)."""
    import torch
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local_buf.is_cuda:
        # rehearsal of the multi-GPU path on a box without RCCL peers (bench.py
        # FLATE_BENCH_BACKEND=gloo): gloo moves host memory, so stage through the CPU
        g = gather_compressed(dist, local_buf.cpu(), local_off, buf=None, pad_to=pad_to, wait=True)
        g.buf = g.buf.to(local_buf.device)
        return g
    dev = local_buf.device
    k = int(local_off.size - 1)
    clen = int(local_off[-1])
    meta = torch.tensor([clen, k], dtype=torch.int64, device=dev)
    metas = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta)
    metas = metas.cpu().numpy().reshape(world, 2)
    sizes, counts = metas[:, 0].copy(), metas[:, 1].copy()
    pad = (int(sizes.max()) + pad_to - 1) // pad_to * pad_to
    pad = max(pad, pad_to)
    kmax = int(counts.max())
    # per-stream index of every rank (padded to kmax+1 entries)
    idx = torch.zeros(kmax + 1, dtype=torch.int64, device=dev)
    idx[:k + 1] = torch.from_numpy(local_off.astype(np.int64)).to(dev)
    idxs = torch.empty(world * (kmax + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(idxs, idx)
    idxs = idxs.cpu().numpy().reshape(world, kmax + 1)
    # payload
    if local_buf.numel() < pad:
        grown = torch.zeros(pad, dtype=torch.uint8, device=dev)
        grown[:local_buf.numel()] = local_buf
        local_buf = grown
    if buf is None or buf.numel() < pad * world:
        buf = torch.empty(pad * world, dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(buf[:pad * world], local_buf[:pad], async_op=True)
    if wait:
        work.wait()
        work = None
    off, length = [], []
    for r in range(world):
        o = idxs[r, :counts[r] + 1].astype(np.uint64)
        off.append(o[:-1] + np.uint64(r * pad))
        length.append(o[1:] - o[:-1])
    return GatheredStreams(buf, pad, sizes, counts, np.concatenate(off), np.concatenate(length), work)
