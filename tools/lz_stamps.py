"""Dev tool: run the stamped build (build/exp/libstamps.so) and print phase shares of the LZ77 kernel."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FLATE_HIP_LIB", os.path.abspath("build/exp/libstamps.so"))
flate = importlib.import_module("moonbit-flate_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
eng = flate.FlateEngine(0)
d = torch.from_numpy(flate.synth(kind, n, 65536)).cuda()
off = flate.uniform_offsets(n, 65536)
eng.set_profiling(True)
for _ in range(2):
    eng.deflate_batch(d, off)
print(eng.last_timing())
L = importlib.import_module("moonbit-flate_amd._lib").load()
L.flate_hip_debug_lz_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros((n, 8), dtype=np.uint64)
k = L.flate_hip_debug_lz_stamps(eng._ctx, buf.ctypes.data, n)
b = buf[:k].astype(np.float64)
names = ["dup+issue", "load_wait", "events", "commit", "batches", "matches", "general_cycles", "chase_cycles"]
m = b.mean(axis=0)
print({names[i]: round(m[i], 1) for i in range(8)})
nb = m[4]
print("per batch (s_memtime ticks):", {names[i]: round(m[i] / nb, 1) for i in range(4)}, "events/batch", m[5] / nb)
