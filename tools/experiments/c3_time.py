"""Dev tool: config 3 (4096 x 256 KiB) stage times for the loaded library."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
n, blen = 4096, 262144
d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
off = flate.uniform_offsets(n, blen)
out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
for _ in range(2):
    eng.deflate_batch(d, off, out=out)
ts = []
for _ in range(4):
    eng.deflate_batch(d, off, out=out)
    ts.append(eng.last_timing())
print(os.environ.get("FLATE_HIP_LIB", "default"), sys.argv[1:], {k: round(sum(t[k] for t in ts) / len(ts), 2) for k in ("lz77_match", "huff_pack")}, flush=True)
