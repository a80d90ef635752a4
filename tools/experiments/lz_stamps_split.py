import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ.setdefault("FLATE_HIP_LIB", os.path.abspath("build/exp/libstamps.so"))
flate = importlib.import_module("moonbit-flate_amd")
import torch
n = 16384
eng = flate.FlateEngine(0)
d = torch.from_numpy(flate.synth("text", n, 65536)).cuda()
off = flate.uniform_offsets(n, 65536)
eng.set_profiling(True)
L = importlib.import_module("moonbit-flate_amd._lib").load()
L.flate_hip_debug_lz_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
names = ["dup+issue", "load_wait", "events", "commit", "batches", "matches", "general_cycles", "chase_cycles"]
for label, gb in (("default 4:6.5", 1664), ("LDS blocks only", 0)):
    eng.set_option("guest_blocks", gb)
    for _ in range(2):
        eng.deflate_batch(d, off)
    print(label, eng.last_timing())
    buf = np.zeros((n, 8), dtype=np.uint64)
    k = L.flate_hip_debug_lz_stamps(eng._ctx, buf.ctypes.data, n)
    b = buf[:k].astype(np.float64)
    tot = b[:, 0] + b[:, 1] + b[:, 2] + b[:, 3]
    per = tot / b[:, 4]
    # two populations in the default launch: LDS-table blocks are faster per batch than guests
    order = np.argsort(per)
    for name, idx in (("fast half", order[:k // 2]), ("slow half", order[k // 2:])):
        m = b[idx].mean(axis=0)
        nb = m[4]
        print("  %-9s per batch:" % name, {names[i]: round(m[i] / nb, 1) for i in (0, 1, 2, 3, 6, 7)}, "sum", round((m[0] + m[1] + m[2] + m[3]) / nb), "matches/batch %.2f" % (m[5] / nb))
