#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X deflate-fast engine.

Metric (BASELINE.json): GiB/s of uncompressed input consumed by encode, 64 KiB streams,
deflate-fast level, bit-exact vs the reference restatement.  Workload at N=1 =
BASELINE.json configs[1]: 1 GiB of independent 64 KiB streams (16384 x 65536 B of S-text,
synthetic) on one MI355X.  With N ranks every rank compresses its own 1 GiB shard (weak
scaling, distinct seeds per shard) and, as north_star's exchange step, the compressed shards
are concatenated on every rank by an RCCL all-gather over xGMI (inside the timed region).

One "step" = one pass of the hot path (LZ77 match -> Huffman/pack -> compaction
[-> all-gather]) over the whole batch, input already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def lz_traffic(args, n, blen):
    """HBM bytes per launch of the match finder from the separate rocprofv3 --pmc passes
    (profiles/r01/*pmc_noguests*.json; FETCH_SIZE x2 + WRITE_SIZE, KB -> B).  rocprofv3 serialises
    kernels while counting, so the concurrent resident+guest launch of the default run cannot be
    attributed: the figure is only reported for the --no-guests configuration it was measured on."""
    if not (args.no_guests and args.kind == "text" and n == 16384 and blen == 65536):
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01", "v9_pmc_noguests.json")))
        for k, v in d.items():
            if "lz77_wave_kernel" in k:
                return int(v["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def inflate_traffic(args, n, blen):
    """HBM bytes per launch of inflate_simt_kernel from the rocprofv3 --pmc passes kept in
    profiles/r01/v10_inflate_pmc.json (measured on 65536 x 64 KiB S-text streams; the kernel does the
    same work per stream at any batch size that fills the GPU, so the figure scales with n)."""
    if not (args.kind == "text" and blen == 65536 and n >= 65536 and not args.spliced):
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01", "v10_inflate_pmc.json")))
        h = d["hbm_bytes_per_launch"]
        return int((h["read_raw_FETCH_SIZE"] + h["written_WRITE_SIZE"]) * n / 65536)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=None,
                    help="streams per GPU (default: 16384 = 1 GiB, configs[1]; --mode inflate: 131072 = 8 GiB, configs[4])")
    ap.add_argument("--stream-len", type=int, default=65536)
    ap.add_argument("--kind", default="text", choices=["text", "ramp", "rand", "zero"])
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather (N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=16384,
                    help="streams of the workload the CPU oracle is timed on (about 10 CPU-seconds per GiB)")
    ap.add_argument("--verify", type=int, default=1,
                    help="0 = skip comparing the GPU output with the oracle's in the cpu_baseline leg")
    ap.add_argument("--no-guests", action="store_true",
                    help="match finder with LDS-table blocks only (no L2-table guest blocks); the "
                         "configuration the PMC traffic figure in profiles/ was collected on")
    ap.add_argument("--spliced", action="store_true",
                    help="encode into ONE DEFLATE stream per GPU (flate_hip_deflate_fast_spliced, SURVEY 8f-3)")
    ap.add_argument("--inflate-lanes", type=int, default=0, choices=[0, 16, 32, 64],
                    help="inflate: streams per wavefront (0 = chosen from the batch size)")
    ap.add_argument("--mode", default="deflate", choices=["deflate", "inflate"],
                    help="inflate = BASELINE.json configs[4]: decode the compressed streams (stream index supplied)")
    args = ap.parse_args()
    if args.streams is None:
        args.streams = 131072 if args.mode == "inflate" else 16384

    import torch
    flate = importlib.import_module("moonbit-flate_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # FLATE_BENCH_BACKEND=gloo rehearses the N>1 control flow where RCCL has no peers (several
    # ranks on one GPU); the driver's runs use nccl (= RCCL) with one GPU per rank.
    backend = os.environ.get("FLATE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, blen = args.streams, args.stream_len
    host = flate.synth(args.kind, n, blen, first_stream=rank * n)
    d_in = torch.from_numpy(host).to(dev)
    in_off = flate.uniform_offsets(n, blen)
    in_bytes = n * blen

    eng = flate.FlateEngine(local_rank)
    eng.use_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_profiling(True)
    if args.no_guests:
        eng.set_option("guest_blocks", 0)
    if args.inflate_lanes:
        eng.set_option("inflate_lanes", args.inflate_lanes)
    out = torch.empty(in_bytes + (in_bytes >> 3) + 4096, dtype=torch.uint8, device=dev)

    if args.mode == "inflate":
        return bench_inflate(args, flate, eng, d_in, in_off, n, blen, world, rank, dev, dist)

    gather = (world > 1) and not args.no_gather
    shard = importlib.import_module("moonbit-flate_amd.shard")
    # The exchange step of batch k (all-gather over xGMI) overlaps the compression of batch k+1:
    # two output buffers alternate, a buffer is reused only after the gather that reads it is done,
    # and every gather completes inside the timed region (sync_all waits for all streams).
    outs = [out, torch.empty_like(out)] if gather else [out]
    pending = [None] * len(outs)
    gbuf = None
    nstep = 0

    def step():
        nonlocal gbuf, nstep
        i = nstep % len(outs)
        nstep += 1
        if pending[i] is not None:
            pending[i].wait()
            pending[i] = None
        if args.spliced:  # one DEFLATE stream per GPU; the gather then moves one "stream" per rank
            import numpy as np
            _, nbytes, _ = eng.deflate_spliced(d_in, in_off, out=outs[i])
            out_off = np.array([0, nbytes], dtype=np.uint64)
        else:
            _, out_off = eng.deflate_batch(d_in, in_off, out=outs[i])
        if gather:  # north_star's exchange step: every rank ends up with every compressed shard
            pending[i] = shard.gather_compressed(dist, outs[i], out_off, buf=gbuf, wait=False)
            gbuf = pending[i].buf
        return out_off

    def drain():
        for j, g in enumerate(pending):
            if g is not None:
                g.wait()
                pending[j] = None

    def sync_all():
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_ms = {k: 0.0 for k in flate.STAGES}
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_off = step()
        tm = eng.last_timing()
        for k in stage_ms:
            stage_ms[k] += tm[k]
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    clen = int(out_off[-1])
    ratio = in_bytes / clen

    # cpu_baseline leg (rank 0, N=1, outside the timed region): the oracle compresses the same
    # streams on the host cores; its output is also the checker -- every stream it produced is
    # compared with what the GPU wrote.
    verified = 0
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        ns = min(args.cpu_sample_streams, n)
        cores = min(os.cpu_count() or 1, 16)
        t1 = time.perf_counter()
        o_buf, o_off, o_len = pyoracle.deflate_batch(host[:ns * blen], in_off[:ns + 1], nthreads=cores)
        cdt = time.perf_counter() - t1
        cpu_baseline = {
            "value": round(ns * blen / cdt / 2**30, 4), "unit": "GiB/s", "cores": cores,
            "kind": "port",
            "sample": "first %d of %d streams (%d MiB), oracle C restatement, %d threads, %.1f s wall"
                      % (ns, n, ns * blen >> 20, cores, cdt),
        }
        if not args.spliced and args.verify:  # (spliced output is checked by tests/test_splice.py)
            g_cpu = out[:clen].cpu().numpy()
            for i in range(ns):
                a0, a1 = int(out_off[i]), int(out_off[i + 1])
                b0 = int(o_off[i])
                if a1 - a0 != int(o_len[i]) or not np.array_equal(g_cpu[a0:a1], o_buf[b0:b0 + a1 - a0]):
                    raise SystemExit("PARITY FAILURE at stream %d" % i)
            verified = ns

    if rank == 0:
        steps = args.steps
        value = world * in_bytes * steps / dt / 2**30
        lz_ms = stage_ms["lz77_match"] / steps
        algo_bytes = in_bytes + clen  # B read + C written per stream, all streams
        achieved = algo_bytes / (lz_ms * 1e-3) / 1e9 if lz_ms > 0 else 0.0
        res = {
            "metric": "GiB/s uncompressed throughput (encode), 64 KiB blocks, deflate-fast",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "%d x %d B independent streams per GPU (%.3f GiB/GPU), S-%s, "
                            "deflate-fast, bit-exact vs oracle%s"
                            % (n, blen, in_bytes / 2**30, args.kind,
                               ", spliced into ONE stream per GPU" if args.spliced else ""),
                "streams_per_gpu": n, "stream_len": blen, "kind": args.kind,
                "compressed_bytes_per_gpu": clen, "ratio": round(ratio, 4),
                "gather": "rccl all_gather_into_tensor (padded), overlapped with the next batch" if gather else "none",
                "parity_checked_streams": verified,
                "stage_ms": {k: round(v / steps, 3) for k, v in stage_ms.items()},
            },
            "roofline": {
                "bound": "hbm", "kernel": "lz77_wave_kernel", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": lz_traffic(args, n, blen),
            },
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def bench_inflate(args, flate, eng, d_in, in_off, n, blen, world, rank, dev, dist):
    """Config 5: inflate-only.  Streams compressed once (untimed) by the encoder; one step = one
    inflate_batch over all of them; value = GiB/s of decompressed output."""
    import torch
    sizes = [blen] * n
    out = torch.empty(n * blen, dtype=torch.uint8, device=dev)
    if args.spliced:  # ONE compressed stream + its index, decoded in parallel (config 5 as worded)
        comp, nbytes, bit_off = eng.deflate_spliced(d_in, in_off)
        coff = [0, nbytes]
        run = lambda: eng.inflate_spliced(comp, nbytes, bit_off, sizes, out=out)
    else:
        comp, coff = eng.deflate_batch(d_in, in_off)
        run = lambda: eng.inflate_batch(comp, coff, sizes, out=out)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    ms = 0.0
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, _, olen, status, _ = run()
        ms += eng.last_timing()["inflate"]
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = bool((status == 0).all()) and bool(torch.equal(out, d_in))  # round-trip property, full size
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.spliced:
        import numpy as np
        from oracle import pyoracle
        ns = min(args.cpu_sample_streams * 2, n)  # the decoder is faster than the encoder
        cores = min(os.cpu_count() or 1, 16)
        h_off = np.asarray(coff[:ns + 1], dtype=np.uint64)
        h_comp = comp[:int(h_off[-1])].cpu().numpy()
        t1 = time.perf_counter()
        _, _, o_len, o_st = pyoracle.inflate_batch(h_comp, h_off, [blen] * ns, nthreads=cores)
        cdt = time.perf_counter() - t1
        if int(o_st.any()) or int((o_len != blen).any()):
            raise SystemExit("oracle inflate failed on the sample")
        cpu_baseline = {
            "value": round(ns * blen / cdt / 2**30, 4), "unit": "GiB/s", "cores": cores, "kind": "port",
            "sample": "first %d of %d streams (%d MiB out), oracle C restatement, %d threads, %.1f s wall"
                      % (ns, n, ns * blen >> 20, cores, cdt),
        }
    if rank == 0:
        clen = int(coff[-1])
        k_ms = ms / args.steps
        achieved = (n * blen + clen) / (k_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "GiB/s decompressed output (inflate), 64 KiB streams", "unit": "GiB/s",
            "value": round(world * n * blen * args.steps / dt / 2**30, 3), "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "inflate %s%d x %d B streams per GPU, S-%s, output == input: %s"
                                   % ("ONE spliced stream of " if args.spliced else "", n, blen, args.kind, ok),
                       "stage_ms": {"inflate": round(k_ms, 3)}},
            "roofline": {"bound": "hbm", "kernel": "inflate_simt_kernel" if n > 2048 else "inflate_kernel",
                         "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": inflate_traffic(args, n, blen)},
            "cpu_baseline": cpu_baseline}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
