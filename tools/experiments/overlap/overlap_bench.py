#!/usr/bin/env python3
"""EXPERIMENT (round 6): what can run BESIDE the match finder's persistent launch?

The default 16384-stream launch holds 4 LDS-table blocks + 6 guests on every CU = 128 of 128 LDS granules for
about 15.5 of a 17.9 ms step.  bench.py --gpus N overlaps the exchange of batch k with the compression of batch
k+1; an RCCL collective is a kernel that needs a workgroup slot and LDS.  On ONE card this measures, for several
kernels submitted on another stream 4 ms into a match-finder launch:
    start latency   first block running (s_memrealtime in the kernel) - submission (a stamp kernel in front of it)
    where it ran    its span against the match finder's launch [call start, call start + lz77 ms]
    what it cost    the match finder's time in that call against the calls without a neighbour
Kernels: a device-to-device copy of 3.4 GB (what a rank receives per step at 8 GPUs) on 32-64 workgroups of 256-512
threads with 0 / 4 KiB / 40 KiB of LDS each; the runtime's own device-to-device copy; a device-to-host copy of one
shard (SDMA engine); the one-rank flate_hip_gather_begin / _end over real RCCL.
    python tools/experiments/overlap/overlap_bench.py > gpurun_out/r06_overlap.txt
"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")
shard = importlib.import_module("moonbit-flate_amd.shard")
OV = C.CDLL(os.path.join(ROOT, "build", "exp", "liboverlap.so"))
OV.ov_stamp.argtypes = [C.c_void_p, C.c_void_p]
OV.ov_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]

TICK_MS = 1e-5  # s_memrealtime: 100 MHz


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--resident", type=int, default=0, help="resident_blocks option (0 = default 4 per CU)")
    ap.add_argument("--guests", type=int, default=-1, help="guest_blocks option (-1 = default 6 per CU)")
    ap.add_argument("--copies-only", action="store_true", help="only the LDS copy kernels (geometry sweeps)")
    ap.add_argument("--pipeline", action="store_true",
                    help="what bench.py --gpus N does on the CU side: after every batch a stand-in for the exchange of "
                         "that batch (64 WG x 512 threads, --standin-lds bytes of LDS, 3.4 GB, stretched to about "
                         "--standin-ms by sleeping) is queued on another stream while the next batch is compressed")
    ap.add_argument("--standin-lds", type=int, default=37664, help="ncclDevKernel_Generic_1: 37664 B static LDS")
    ap.add_argument("--standin-threads", type=int, default=256, help="ncclDevKernel_Generic_1: 64 workgroups of 256")
    ap.add_argument("--standin-ms", type=float, default=10.0)
    args = ap.parse_args()
    n, blen = 16384, 65536
    dev = torch.device("cuda:0")
    host = flate.synth("text", n, blen)
    d_in = torch.from_numpy(host).to(dev)
    off = flate.uniform_offsets(n, blen)
    eng = flate.FlateEngine(0)
    s_main = torch.cuda.Stream()
    s_side = torch.cuda.Stream()
    eng.use_stream(s_main.cuda_stream)
    eng.set_profiling(True)
    if args.resident:
        eng.set_option("resident_blocks", args.resident)
    if args.guests >= 0:
        eng.set_option("guest_blocks", args.guests)
    print("geometry: resident_blocks %s, guest_blocks %s" % (args.resident or "default (1024)", args.guests if args.guests >= 0 else "default (1536)"))
    out = torch.empty(int(n * blen * 0.6), dtype=torch.uint8, device=dev)
    comp, coff = eng.deflate_batch(d_in, off, out=out)
    clen = int(coff[-1])
    payload = 3_400_000_000 // 16 * 16
    src = torch.empty(payload, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty(payload, dtype=torch.uint8, device=dev)
    stamps = torch.zeros(16, dtype=torch.int64, device=dev)  # [0..2] copy kernel, [4] submit, [5] call start, [6] call end
    pinned = torch.empty(clen, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()

    def reset_stamps():
        stamps.zero_()
        stamps[0] = (1 << 62)
        torch.cuda.synchronize()

    def one_call():
        OV.ov_stamp(s_main.cuda_stream, stamps[5:].data_ptr())
        eng.deflate_batch(d_in, off, out=out)
        OV.ov_stamp(s_main.cuda_stream, stamps[6:].data_ptr())
        t = eng.last_timing()
        return t["lz77_match"], t["huff_pack"]

    base = [one_call() for _ in range(6)][1:]
    lz0 = float(np.median([b[0] for b in base]))
    print("match finder alone: lz77 %.2f ms (min %.2f max %.2f), entropy %.2f ms; one rank's compressed shard %.3f GB"
          % (lz0, min(b[0] for b in base), max(b[0] for b in base), float(np.median([b[1] for b in base])), clen / 1e9))

    if args.pipeline:
        pipeline(args, eng, one_call, s_side, src, dst, payload, stamps, lz0)
        eng.close()
        return
    comm = shard.NativeComm(eng, 0, 1)
    g0 = comm.gather(comp, coff)  # makes the plan
    gout = torch.zeros(comm.plan()[0] + 64, dtype=torch.uint8, device=dev)

    def submit(kind, arg):
        """Queue the neighbour on the side stream (never blocks); returns a finisher."""
        if kind == "copy":
            wgs, thr, lds = arg
            OV.ov_stamp(s_side.cuda_stream, stamps[4:].data_ptr())
            rc = OV.ov_copy(s_side.cuda_stream, dst.data_ptr(), src.data_ptr(), payload, wgs, thr, lds, stamps.data_ptr(), 0)
            assert rc == 0, rc
            return lambda: None
        if kind == "runtime_d2d":
            OV.ov_stamp(s_side.cuda_stream, stamps[4:].data_ptr())
            with torch.cuda.stream(s_side):
                dst.copy_(src, non_blocking=True)
                OV.ov_stamp(s_side.cuda_stream, stamps[1:].data_ptr())
            return lambda: None
        if kind == "d2h":
            OV.ov_stamp(s_side.cuda_stream, stamps[4:].data_ptr())
            with torch.cuda.stream(s_side):
                pinned.copy_(comp[:clen], non_blocking=True)
                OV.ov_stamp(s_side.cuda_stream, stamps[1:].data_ptr())
            return lambda: None
        raise ValueError(kind)

    def measure_rccl(name):
        # as bench.py --gpus N does it, one thread (a ctx is not for two threads): begin (returns at once; the
        # exchange waits on the GPU for what is queued on the ctx's stream, i.e. nothing), compress the next batch,
        # end.  What end still waits for after the call has returned is the part of the exchange that was not hidden.
        alone = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            comm.begin(comp, coff, gout)
            comm.end(n)
            alone.append((time.perf_counter() - t0) * 1e3)
        rows = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            comm.begin(comp, coff, gout)
            t1 = time.perf_counter()
            lz, _ = one_call()
            t2 = time.perf_counter()
            comm.end(n)
            t3 = time.perf_counter()
            rows.append((lz, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        print("%-44s alone begin+end %6.2f ms (%5.0f GB/s) | beside: %s | (alone %.2f)" % (
            name, float(np.median(alone)), clen / np.median(alone) / 1e6,
            "; ".join("lz77 %.2f ms, begin %.3f ms, call %.2f ms, end waited %.3f ms" % r for r in rows), lz0), flush=True)

    def measure(name, kind, arg, payload_bytes):
        # alone
        alone = []
        for _ in range(3):
            reset_stamps()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_side)
            fin = submit(kind, arg)
            r = fin()
            e1.record(s_side)
            torch.cuda.synchronize()
            alone.append(e0.elapsed_time(e1))
        # beside: the neighbour is submitted 4 ms into the second of three back-to-back calls
        reset_stamps()
        started = threading.Event()
        res = []

        def worker():
            for k in range(3):
                if k == 1:
                    started.set()
                res.append(one_call() + (int(stamps[5].item()), int(stamps[6].item())))
        th = threading.Thread(target=worker)
        th.start()
        started.wait()
        time.sleep(0.004)
        t_sub = time.perf_counter()
        fin = submit(kind, arg)
        e1 = torch.cuda.Event()
        e1.record(s_side)
        e1.synchronize()
        t_done = time.perf_counter()
        th.join()
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()
        lz_b, _, c_start, c_end = res[1]
        line = "%-44s alone %6.2f ms (%5.0f GB/s)" % (name, float(np.median(alone)), payload_bytes / np.median(alone) / 1e6)
        line += " | beside: submit -> done %6.2f ms (host clock)" % ((t_done - t_sub) * 1e3)
        if st[4] and c_start:
            sub = (st[4] - c_start) * TICK_MS
            line += ", submitted %.2f ms into the call (lz77 of that call %.2f ms)" % (sub, lz_b)
            if kind == "copy":
                first, last = (st[0] - st[4]) * TICK_MS, (st[1] - st[4]) * TICK_MS
                line += "; first block ran %.3f ms after submission = %.2f ms into the call, last block ended at %.2f ms into the call (%d blocks)" % (
                    first, sub + first, sub + last, int(st[2]))
            elif st[1]:
                line += "; finished %.2f ms into the call" % ((st[1] - c_start) * TICK_MS)
        line += " | match finder in the three calls: %s ms (alone %.2f)" % (", ".join("%.2f" % r[0] for r in res), lz0)
        print(line, flush=True)

    shapes = ((64, 512, 4096), (64, 512, 32768), (64, 512, 40960)) if args.copies_only else \
        ((64, 512, 0), (64, 512, 4096), (64, 512, 40960), (32, 256, 4096), (256, 256, 0), (256, 256, 4096))
    for wgs, thr, lds in shapes:
        measure("copy kernel %3d WG x %3d thr, LDS %5d B" % (wgs, thr, lds), "copy", (wgs, thr, lds), payload)
    if args.copies_only:
        comm.close()
        eng.close()
        return
    measure("runtime device-to-device copy (3.4 GB)", "runtime_d2d", None, payload)
    measure("device-to-host copy of one shard (SDMA)", "d2h", None, clen)
    measure_rccl("flate_hip_gather_begin/_end, one rank, RCCL")
    comm.close()
    eng.close()


def pipeline(args, eng, one_call, s_side, src, dst, payload, stamps, lz0):
    # calibrate the stand-in's nap so that it takes about --standin-ms alone
    def standin(nap):
        rc = OV.ov_copy(s_side.cuda_stream, dst.data_ptr(), src.data_ptr(), payload, 64, args.standin_threads, args.standin_lds, stamps.data_ptr(), nap)
        assert rc == 0

    def alone_ms(nap):
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_side)
            standin(nap)
            e1.record(s_side)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))
    t0 = alone_ms(0)
    nap, t = 0, t0
    if args.standin_ms > t0:
        t1000 = alone_ms(1000)
        nap = int(1000 * (args.standin_ms - t0) / max(t1000 - t0, 1e-3))
        t = alone_ms(nap)
    print("stand-in for one exchange: 64 WG x %d threads, %d B of LDS, 3.4 GB: %.2f ms alone (nap %d; %.2f ms without)" % (args.standin_threads, args.standin_lds, t, nap, t0))
    for with_x in (False, True, False, True):
        torch.cuda.synchronize()
        steps, lzs = 12, []
        t_begin = time.perf_counter()
        for k in range(steps):
            lz, _ = one_call()
            lzs.append(lz)
            if with_x:
                standin(nap)  # the exchange of batch k, queued behind nothing, beside batch k + 1
        t_calls = time.perf_counter()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        print("%-34s %d steps: %.2f ms per step (host clock, calls only), + %.2f ms to drain at the end; match finder median %.2f ms (min %.2f max %.2f; alone %.2f)"
              % ("with the stand-in after every batch" if with_x else "without", steps, (t_calls - t_begin) * 1e3 / steps, (t_end - t_calls) * 1e3,
                 float(np.median(lzs)), min(lzs), max(lzs), lz0), flush=True)


if __name__ == "__main__":
    main()
