import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
flate = importlib.import_module("moonbit-flate_amd")
eng = flate.FlateEngine(0)
n, blen = 16384, 65536
host = flate.synth("text", n, blen)
off = flate.uniform_offsets(n, blen)
h_out = np.empty(n * blen + (n * blen >> 3) + 4096, dtype=np.uint8)
r1, r2 = eng.host_register(host), eng.host_register(h_out)
eng.set_option("host_pipeline_groups", int(sys.argv[1]))
eng.set_option("host_pipeline_group_streams", 1024)
for i in range(3):
    sys.stderr.write("--- call %d\n" % i)
    t0 = time.perf_counter()
    eng.deflate_batch(host, off, out=h_out)
    sys.stderr.write("total %.2f ms\n" % ((time.perf_counter() - t0) * 1e3))
