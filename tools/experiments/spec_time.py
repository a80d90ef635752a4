"""Dev tool: time of the speculative inflater alone at a few batch sizes."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
eng.set_profiling(True)
eng.set_option("inflate_simt_min_streams", 1 << 30)
eng.set_option("inflate_spec", 2)
blen = 65536
for n in [int(x) for x in sys.argv[1:]] or [1024, 4096, 16384]:
    d = torch.from_numpy(flate.synth("text", n, blen)).cuda()
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d, off)
    out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        eng.inflate_batch(comp, coff, [blen] * n, out=out)
    assert torch.equal(out, d)
    ms = eng.last_timing()["inflate"]
    print(os.environ.get("FLATE_HIP_LIB", "default"), n, round(ms, 2), "ms", "%.1f GiB/s" % (n * blen / 2**30 / (ms * 1e-3)), flush=True)
