"""Dev tool: stamped build (build/exp/libstamps.so), pipelined match finder: phase times per batch."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FLATE_HIP_LIB", os.path.abspath("build/exp/libstamps.so"))
flate = importlib.import_module("moonbit-flate_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
guests = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
eng = flate.FlateEngine(0)
eng.set_option("lz_pipe", 1)
eng.set_option("guest_blocks", guests)
d = torch.from_numpy(flate.synth("text", n, 65536)).cuda()
off = flate.uniform_offsets(n, 65536)
eng.set_profiling(True)
for _ in range(2):
    eng.deflate_batch(d, off)
print(eng.last_timing())
L = importlib.import_module("moonbit-flate_amd._lib").load()
L.flate_hip_debug_lz_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.zeros((n, 8), dtype=np.uint64)
k = L.flate_hip_debug_lz_stamps(eng._ctx, buf.ctypes.data, n)
m = buf[:k].astype(np.float64).mean(axis=0)
nb = m[4]
print("batches %.0f matches %.0f | per batch: front+eval %.0f  next-front issue %.0f  events %.0f  commit %.0f | fresh %.2f changed %.2f"
      % (nb, m[5], m[0] / nb, m[1] / nb, m[2] / nb, m[3] / nb, m[6] / nb, m[7] / nb))
