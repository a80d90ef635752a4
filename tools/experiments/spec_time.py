"""Dev tool: time the speculative inflater alone at several batch sizes (64 KiB S-text streams)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
blen = 65536
kind = os.environ.get("KIND", "text")
for n in [int(x) for x in sys.argv[1:]] or [64, 256, 1024, 4096, 16384]:
    d = torch.from_numpy(flate.synth(kind, n, blen)).cuda()
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d, off)
    out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
    eng.set_option("inflate_simt_min_streams", 1 << 30)
    eng.set_option("inflate_spec", 2)
    ts = []
    for _ in range(4):
        eng.inflate_batch(comp, coff, [blen] * n, out=out)
        ts.append(eng.last_timing()["inflate"])
    assert torch.equal(out, d)
    t = min(ts[1:])
    print(n, "%.3f ms" % t, "%.1f GiB/s" % (n * blen / 2**30 / (t * 1e-3)), flush=True)
