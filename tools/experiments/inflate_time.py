"""Kernel time of the config-5 inflate launch WITHOUT checking the bytes: for timing-only experiment builds
(FLATE_HIP_LIB=build/exp/lib<variant>.so) whose output is wrong on purpose.
    python3 tools/experiments/inflate_time.py [streams]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
flate = importlib.import_module("moonbit-flate_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
blen = 65536
eng = flate.FlateEngine(0)
eng.set_profiling(True)
# compress in pieces of 1 GiB (the encoder's scratch is sized for that)
parts, offs, base = [], [np.zeros(1, np.uint64)], 0
for s0 in range(0, n, 16384):
    k = min(16384, n - s0)
    d = torch.from_numpy(flate.synth("text", k, blen, first_stream=s0)).cuda()
    c, co = eng.deflate_batch(d, flate.uniform_offsets(k, blen))
    parts.append(c[:int(co[-1])].clone())
    offs.append(co[1:] + base)
    base += int(co[-1])
    del d, c
comp = torch.cat(parts)
coff = np.concatenate(offs)
del parts
out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
sizes = [blen] * n
ms = []
for i in range(4):
    eng.inflate_batch(comp, coff, sizes, out=out, check=False)
    ms.append(eng.last_timing()["inflate"])
print(os.environ.get("FLATE_HIP_LIB", "default"), "streams", n, "inflate kernel ms", [round(m, 2) for m in ms],
      "GiB/s", round(n * blen / (min(ms[1:]) * 1e-3) / 2**30, 1))
