import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
eng = flate.FlateEngine(0)
n, blen = 131072, 65536
host = flate.synth("text", n, blen)
d_in = torch.from_numpy(host).cuda()
off = flate.uniform_offsets(n, blen)
comp, coff = eng.deflate_batch(d_in, off)
h_comp = comp[:int(coff[-1])].cpu().numpy()
del comp, d_in
torch.cuda.empty_cache()
sizes = [blen] * n
h_out = np.empty(n * blen, dtype=np.uint8)
r1, r2 = eng.host_register(h_comp), eng.host_register(h_out)
eng.set_option("host_pipeline_groups", int(sys.argv[1]))
for i in range(3):
    sys.stderr.write("--- call %d\n" % i)
    t0 = time.perf_counter()
    eng.inflate_batch(h_comp, coff, sizes, out=h_out)
    sys.stderr.write("total %.2f ms\n" % ((time.perf_counter() - t0) * 1e3))
