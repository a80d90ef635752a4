"""Kernel time of flate_hip_checksum_batch on the headline batch (16384 x 64 KiB, device pointers) and on the
same GiB as ONE stream.   python3 tools/experiments/checksum_bench.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")
import torch
n, blen = 16384, 65536
eng = flate.FlateEngine(0)
eng.set_profiling(True)
d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
for name, off in (("16384 streams", flate.uniform_offsets(n, blen)), ("one stream", np.array([0, n * blen], np.uint64))):
    for kind in ("adler32", "crc32"):
        ts = []
        for _ in range(5):
            eng.checksum_batch(d, off, kind)
            ts.append(eng.last_timing()["checksum"])
        print("%-14s %-8s kernel ms %s  -> %.0f GB/s" % (name, kind, ["%.3f" % t for t in ts], n * blen / min(ts) / 1e6), flush=True)
import time
off = flate.uniform_offsets(n, blen)
for kind in ("adler32", "crc32"):
    eng.checksum_batch(d, off, kind)
    t0 = time.perf_counter()
    for _ in range(20):
        eng.checksum_batch(d, off, kind)
    print("wall per call (16384 streams, device pointers) %-8s %.3f ms" % (kind, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
