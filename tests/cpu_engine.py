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
