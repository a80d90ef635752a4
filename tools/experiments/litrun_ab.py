"""Kernel time of inflating S-rand (Huffman-only blocks) with the library FLATE_HIP_LIB names; output not
checked (the timing variants of the literal-run path produce wrong bytes on purpose).
    FLATE_HIP_LIB=build/exp/libX.so python3 tools/experiments/litrun_ab.py [streams]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    import torch
    blen = 65536
    eng = flate.FlateEngine(0)
    eng.set_profiling(True)
    host = flate.synth("rand", n, blen)
    off = flate.uniform_offsets(n, blen)
    d_in = torch.from_numpy(host).cuda()
    comp, coff = eng.deflate_batch(d_in, off)
    sizes = np.full(n, blen, dtype=np.uint64)
    out = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(4):
        eng.inflate_batch(comp, coff, sizes, out=out, check=False)
        ts.append(eng.last_timing()["inflate"])
    print(os.environ.get("FLATE_HIP_LIB", "default"), n, "inflate kernel ms:", ["%.3f" % t for t in ts],
          "same bytes:", bool(torch.equal(out, d_in)), flush=True)


if __name__ == "__main__":
    main()
