"""Dev tool: the speculative inflater on every synthetic kind (ms per batch, bit-exact check)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blen = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for kind in flate.SYNTH_KINDS:
    try:
        d = torch.from_numpy(flate.synth(kind, n, blen)).cuda()
    except Exception as e:
        print(kind, "skipped:", e)
        continue
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d, off)
    out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
    res = {}
    for name, simt_min, spec in (("simt", 0, 0), ("spec", 1 << 30, 2)):
        eng.set_option("inflate_simt_min_streams", simt_min)
        eng.set_option("inflate_spec", spec)
        ts = []
        for _ in range(3):
            eng.inflate_batch(comp, coff, [blen] * n, out=out)
            ts.append(eng.last_timing()["inflate"])
        assert torch.equal(out, d), (kind, name)
        res[name] = round(min(ts[1:]), 3)
    print(kind, "ratio %.3f" % (int(coff[-1]) / (n * blen)), res, flush=True)
