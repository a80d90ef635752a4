"""Dev tool: run the speculative inflater (a -DFLATE_SPEC_STATS build via FLATE_HIP_LIB) on a few streams; stream 0 prints its counters."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n, blen = int(os.environ.get("N", "64")), 65536
d = torch.from_numpy(flate.synth(kind, n, blen)).cuda()
off = flate.uniform_offsets(n, blen)
comp, coff = eng.deflate_batch(d, off)
eng.set_option("inflate_simt_min_streams", 1 << 30)
eng.set_option("inflate_spec", 2)
eng.set_option("inflate_spec_shape", int(os.environ.get("SHAPE", "0")))
out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
eng.inflate_batch(comp, coff, [blen] * n, out=out)
torch.cuda.synchronize()
assert torch.equal(out, d)
print("compressed bytes of stream 0:", int(coff[1] - coff[0]))
