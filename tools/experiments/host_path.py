"""Host-pointer calls (PCIe inside the call): pageable vs page-locked buffers, group shapes.
    python3 tools/experiments/host_path.py [deflate|inflate|both]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")


def timed(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return r, ts


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    import torch
    eng = flate.FlateEngine(0)
    eng.set_profiling(True)
    n, blen = 16384, 65536
    if what in ("deflate", "both"):
        host = flate.synth("text", n, blen)
        off = flate.uniform_offsets(n, blen)
        h_out = np.empty(n * blen + (n * blen >> 3) + 4096, dtype=np.uint8)
        (_, ref_off), ts = timed(lambda: eng.deflate_batch(host, off, out=h_out), 4)
        ref = h_out[:int(ref_off[-1])].copy()
        print("deflate pageable  groups=4   ms:", ["%.1f" % t for t in ts], {k: round(v, 2) for k, v in eng.last_timing().items() if v}, flush=True)
        # library-allocated page-locked memory (hipHostMalloc) instead of registered pages
        import ctypes as C
        L = eng._L
        pa, pb = C.c_void_p(), C.c_void_p()
        eng._check(L.flate_hip_host_alloc(eng._ctx, host.nbytes, C.byref(pa)))
        eng._check(L.flate_hip_host_alloc(eng._ctx, h_out.nbytes, C.byref(pb)))
        a_in = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_uint8)), shape=(host.nbytes,))
        a_out = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(h_out.nbytes,))
        a_in[:] = host
        for groups, gs in ((4, 4096), (8, 2048)):
            eng.set_option("host_pipeline_groups", groups)
            eng.set_option("host_pipeline_group_streams", gs)
            (_, o), ts = timed(lambda: eng.deflate_batch(a_in, off, out=a_out), 4)
            print("deflate hipHostMalloc groups=%d ms: %s kernels %s" % (groups, ["%.1f" % t for t in ts],
                  {k: round(v, 2) for k, v in eng.last_timing().items() if v}), flush=True)
        del a_in, a_out
        L.flate_hip_host_free(eng._ctx, pa)
        L.flate_hip_host_free(eng._ctx, pb)
        eng.set_option("host_pipeline_groups", 4)
        eng.set_option("host_pipeline_group_streams", 4096)
        t0 = time.perf_counter()
        r1, r2 = eng.host_register(host), eng.host_register(h_out)
        print("register 1 GiB + 1.1 GiB: %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        for lanes, groups, gs in ((1, 4, 4096), (2, 4, 4096), (2, 8, 2048), (2, 16, 1024), (2, 6, 2048), (2, 5, 2048),
                                  (2, 12, 1024), (2, 10, 1024), (1, 8, 2048)):
            eng.set_option("host_pipeline_lanes", lanes)
            eng.set_option("host_pipeline_groups", groups)
            eng.set_option("host_pipeline_group_streams", gs)
            (_, o), ts = timed(lambda: eng.deflate_batch(host, off, out=h_out), 4)
            ok = np.array_equal(o, ref_off) and np.array_equal(h_out[:int(o[-1])], ref)
            print("deflate registered lanes=%d groups=%-2d min-group=%-5d ms: %s  best %.1f GiB/s  same bytes: %s  kernels %s"
                  % (lanes, groups, gs, ["%.1f" % t for t in ts], n * blen / min(ts) * 1e3 / 2**30, ok,
                     {k: round(v, 2) for k, v in eng.last_timing().items() if v}), flush=True)
        eng.set_option("host_pipeline_groups", 4)
        eng.set_option("host_pipeline_group_streams", 4096)
        # raw link rates on this box
        d = torch.empty(n * blen, dtype=torch.uint8, device="cuda")
        hp = torch.from_numpy(host)
        for name, f in (("H2D registered 1 GiB", lambda: (d.copy_(hp, non_blocking=True), torch.cuda.synchronize())),
                        ("D2H registered 1 GiB", lambda: (torch.from_numpy(h_out[:n * blen]).copy_(d, non_blocking=True), torch.cuda.synchronize()))):
            _, ts = timed(f, 3)
            print("%s: %s ms = %.1f GB/s" % (name, ["%.1f" % t for t in ts], n * blen / min(ts) * 1e3 / 1e9), flush=True)
        r1.close()
        r2.close()
        del host, h_out, d
    if what in ("inflate", "both"):
        n = 131072
        host = flate.synth("text", n, blen)
        d_in = torch.from_numpy(host).cuda()
        off = flate.uniform_offsets(n, blen)
        comp, coff = eng.deflate_batch(d_in, off)
        h_comp = comp[:int(coff[-1])].cpu().numpy()
        del comp, d_in
        torch.cuda.empty_cache()
        sizes = [blen] * n
        h_out = np.empty(n * blen, dtype=np.uint8)
        t0 = time.perf_counter()
        h_out[::4096] = 0  # first touch of the 8 GiB, outside every timed call
        print("first touch of 8 GiB: %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        (_, _, olen, st, _), ts = timed(lambda: eng.inflate_batch(h_comp, coff, sizes, out=h_out), 4)
        print("inflate pageable (pre-touched) ms:", ["%.0f" % t for t in ts], "ok", bool((st == 0).all()), flush=True)
        t0 = time.perf_counter()
        r1, r2 = eng.host_register(h_comp), eng.host_register(h_out)
        print("register 3.6 + 8 GiB: %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        for groups in (4, 8, 2):
            eng.set_option("host_pipeline_groups", groups)
            (_, _, olen, st, _), ts = timed(lambda: eng.inflate_batch(h_comp, coff, sizes, out=h_out), 4)
            same = all(np.array_equal(h_out[i * blen:(i + 1) * blen], host[i * blen:(i + 1) * blen]) for i in range(0, n, 4099))
            print("inflate registered groups=%d ms: %s  best %.1f GiB/s out  round trip: %s"
                  % (groups, ["%.0f" % t for t in ts], n * blen / min(ts) * 1e3 / 2**30, same), flush=True)
        r1.close()
        r2.close()
    eng.close()


if __name__ == "__main__":
    main()
