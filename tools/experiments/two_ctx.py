"""Two contexts on one card, each compressing the whole headline batch from its own host thread: does the
second launch fill the first one's drain (the persistent match finder's tail) and its entropy stage?
    python3 tools/experiments/two_ctx.py [steps]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    import torch
    n, blen = 16384, 65536
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    d_in = torch.from_numpy(host).cuda()
    cap = n * flate.deflate_bound(blen)
    engs = [flate.FlateEngine(0) for _ in range(2)]
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
    for e, o in zip(engs, outs):
        for _ in range(3):
            e.deflate_batch(d_in, off, out=o)
    torch.cuda.synchronize()
    ref = outs[0].cpu().numpy()
    _, ref_off = engs[0].deflate_batch(d_in, off, out=outs[0])
    total = int(ref_off[-1])

    def run(e, o, k):
        for _ in range(k):
            e.deflate_batch(d_in, off, out=o)

    for rep in range(3):
        t0 = time.perf_counter()
        run(engs[0], outs[0], steps)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) * 1e3 / steps
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(engs[i], outs[i], steps // 2)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) * 1e3 / (steps // 2 * 2)
        same = all(np.array_equal(o[:total].cpu().numpy(), ref[:total]) for o in outs)
        print("one context %.3f ms/step (%.2f GiB/s)   two contexts %.3f ms/step (%.2f GiB/s)   same bytes: %s"
              % (one, n * blen / one * 1e3 / 2**30, two, n * blen / two * 1e3 / 2**30, same), flush=True)


if __name__ == "__main__":
    main()
