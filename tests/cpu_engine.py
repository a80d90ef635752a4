"""Test-only stand-in for FlateEngine (FLATE_BENCH_TEST_ENGINE=tests.cpu_engine:OracleEngine):
lets bench.py's N>1 control flow (rank spawn, sharding, gather, verification, JSON line) run on a
box without GPUs.  The compress step is the oracle -- this is test infrastructure, never a
product path, and bench.py labels such a line as not a measurement."""
import numpy as np

from oracle import pyoracle

STAGES = ("lz77_match", "huff_pack", "checksum", "inflate")


class OracleEngine:
    def __init__(self, device=0):
        pyoracle.build()
        self.device = device

    def use_stream(self, ptr):
        pass

    def set_profiling(self, on=True):
        pass

    def set_option(self, name, value):
        pass

    def last_timing(self):
        return {k: (1.0 if k in ("lz77_match", "huff_pack") else 0.0) for k in STAGES}

    def deflate_batch(self, data, in_off, out=None, **kw):
        import torch
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        n = in_off.size - 1
        buf, off, length = pyoracle.deflate_batch(data.numpy(), in_off, nthreads=2)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(length, out=out_off[1:])
        packed = np.concatenate([buf[int(off[i]):int(off[i]) + int(length[i])] for i in range(n)])
        if out is None:
            out = torch.empty(max(packed.size, 16), dtype=torch.uint8)
        out[:packed.size] = torch.from_numpy(packed)
        return out, out_off

    def close(self):
        pass

    def native_comm(self, rank, world, dist):
        """Stand-in for shard.NativeComm (the C-ABI exchange needs a GPU): the same interface over
        torch.distributed, so that bench.py's DEFAULT N > 1 path -- exchange through the C ABI after the timed
        region, compared with the torch.distributed result -- is rehearsed on CPU."""
        return _RehearsalComm(rank, world, dist)


class _Gathered:
    def __init__(self, buf, off, length):
        self.buf, self.off, self.length = buf, off, length

    def stream(self, j):
        o = int(self.off[j])
        return self.buf[o:o + int(self.length[j])]


class _RehearsalComm:
    def __init__(self, rank, world, dist):
        self.rank, self.world, self.dist = rank, world, dist

    def gather(self, local_buf, local_off, mode="allgather"):
        import os
        import time
        import torch
        if os.environ.get("FLATE_TEST_STUCK_RANK") == str(self.rank):  # a rank that never enters the collective
            time.sleep(3600)
        local_off = np.ascontiguousarray(local_off, dtype=np.int64)
        k, nbytes = local_off.size - 1, int(local_off[-1])
        meta = [None] * self.world
        self.dist.all_gather_object(meta, (nbytes, local_off.tolist()))
        pad = (max(m[0] for m in meta) + 4095) // 4096 * 4096
        mine = torch.zeros(pad, dtype=torch.uint8)
        mine[:nbytes] = local_buf[:nbytes]
        parts = [torch.empty(pad, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        offs, lens, chunks, base = [], [], [], 0
        for r, (nb, lo) in enumerate(meta):
            # allgather: rank r's shard at r * pad; sendrecv: the shards back to back (flate_hip.h)
            start = r * pad if mode == "allgather" else base
            chunks.append(parts[r] if mode == "allgather" else parts[r][:nb])
            offs += [start + lo[i] for i in range(len(lo) - 1)]
            lens += [lo[i + 1] - lo[i] for i in range(len(lo) - 1)]
            base += nb
        return _Gathered(torch.cat(chunks), np.array(offs, np.uint64), np.array(lens, np.uint64))

    def close(self):
        pass
