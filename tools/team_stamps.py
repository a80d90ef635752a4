"""Dev tool: run the stamped build (build/exp/libstamps.so) with the team match finder and print its
per-round phase times (s_memtime ticks)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FLATE_HIP_LIB", os.path.abspath("build/exp/libstamps.so"))
flate = importlib.import_module("moonbit-flate_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gst = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
kind = sys.argv[4] if len(sys.argv) > 4 else "text"
eng = flate.FlateEngine(0)
eng.set_option("lz_team", 1)
eng.set_option("team_resident_blocks", res)
eng.set_option("team_guest_blocks", gst)
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
m = buf[:k].astype(np.float64).mean(axis=0)
names = ["barrier_wait(2 waves)", "parse", "front(spec)", "rounds", "usable", "refresh", "events", "ev_eval"]
print({names[i]: round(m[i], 1) for i in range(8)})
r = m[3]
print("per round: parse %.0f = refresh %.0f + ev %.0f + events %.0f + rest %.0f | spec front %.0f  barrier wait per wave %.0f  usable %.2f  rounds/stream %.0f"
      % (m[1] / r, m[5] / r, m[7] / r, m[6] / r, (m[1] - m[5] - m[7] - m[6]) / r, m[2] / r, m[0] / r / 2, m[4] / r, r))
