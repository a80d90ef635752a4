"""Dev tool: time the size-only pass against the inflate itself (64 KiB S-text streams)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
blen = 65536
for n in [int(x) for x in sys.argv[1:]] or [1024, 16384]:
    d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d, off)
    res = {}
    for spec in (1, 0):
        eng.set_option("inflate_spec", spec)
        ts = []
        for _ in range(3):
            sizes, status, _ = eng.inflate_sizes(comp, coff)
            ts.append(eng.last_timing()["inflate"])
        assert (status == 0).all() and (sizes == blen).all()
        res["sub-block" if spec else "wave per stream"] = round(min(ts[1:]), 3)
    print(n, "size pass ms:", res, flush=True)
