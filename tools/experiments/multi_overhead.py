"""Dev tool: is the multi-window build of the match finder slower per window?  16384 streams of
65535 B (one window, single-window kernels), of 65536 B (one window + a 1-byte second one: the
multi-window kernels, window units) and 8192 of 131070 B (two full windows)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
for n, blen in ((16384, 65535), (16384, 65536), (8192, 131070), (4096, 262140), (4096, 262144)):
    d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
    off = flate.uniform_offsets(n, blen)
    out = torch.empty(n * blen + (1 << 20), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        eng.deflate_batch(d, off, out=out)
    ts = []
    for _ in range(3):
        eng.deflate_batch(d, off, out=out)
        ts.append(eng.last_timing())
    print(n, blen, {k: round(sum(t[k] for t in ts) / len(ts), 2) for k in ("lz77_match", "huff_pack")}, flush=True)
    del d, out
