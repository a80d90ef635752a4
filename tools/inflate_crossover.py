"""Dev tool: time the three inflate kernels at several batch sizes (64 KiB S-text streams)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
blen = 65536
for n in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192, 16384]:
    d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d, off)
    out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
    res = {}
    for name, simt_min, lanes, spec in (("wave", 1 << 30, 0, 0), ("simt16", 0, 16, 0), ("simt32", 0, 32, 0),
                                        ("simt64", 0, 64, 0), ("spec", 1 << 30, 0, 2)):
        if name == "wave" and n > 4096:
            continue  # (13 ms per 1024 streams: not worth the minutes)
        eng.set_option("inflate_simt_min_streams", simt_min)
        eng.set_option("inflate_lanes", lanes)
        eng.set_option("inflate_spec", spec)
        for _ in range(2):
            eng.inflate_batch(comp, coff, [blen] * n, out=out)
        assert torch.equal(out, d), name
        res[name] = round(eng.last_timing()["inflate"], 2)
    best = min(res, key=res.get)
    print(n, res, "best", best, "%.1f GiB/s" % (n * blen / 2**30 / (res[best] * 1e-3)), flush=True)
