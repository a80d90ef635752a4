#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X deflate-fast engine.

Metric (BASELINE.json): GiB/s of uncompressed input consumed by encode, 64 KiB streams,
deflate-fast level, bit-exact vs the reference restatement.  Workload at N=1 =
BASELINE.json configs[1]: 1 GiB of independent 64 KiB streams (16384 x 65536 B of S-text,
synthetic) on one MI355X.  With N ranks every rank compresses its own 1 GiB shard (weak
scaling, distinct seeds per shard) and, as north_star's exchange step, the compressed shards
are concatenated on every rank over RCCL/xGMI (inside the timed region, overlapped with the
compression of the next batch).

One "step" = one pass of the hot path (LZ77 match -> Huffman/pack [-> gather]) over the whole
batch, input already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh
child processes, one GPU each, started before this process touches the GPU).

At N = 1 the default run appends, outside the timed region of the headline, the other two
single-GPU configurations of BASELINE.json under "extra": configs[2] (1 GiB of 256 KiB streams)
and configs[4] (inflate-only, 8 GiB of output).
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
XGMI_PEAK_GBS = 7 * 153.0  # 7 links x ~153 GB/s per GPU (task statement)


# ----------------------------------------------------------------------------------------------
# launch
# ----------------------------------------------------------------------------------------------
kExitAbandoned = 3  # a rank left a stuck collective behind (its line is out and says so)


def spawn_ranks(n):
    """Start n worker processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), one GPU each.  The
    parent never initialises the GPU and never execs: it waits and returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def load_engine_module():
    """The product package; FLATE_BENCH_TEST_ENGINE=module:Class (test-only, provided by tests/)
    substitutes the engine so that the N>1 control flow can be rehearsed on a box without GPUs.
    A line produced that way says so in "data" and is not a measurement."""
    flate = importlib.import_module("moonbit-flate_amd")
    spec = os.environ.get("FLATE_BENCH_TEST_ENGINE")
    if not spec:
        return flate, flate.FlateEngine, False
    mod, cls = spec.split(":")
    return flate, getattr(importlib.import_module(mod), cls), True


def cpu_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {"nproc": os.cpu_count() or 1, "model": model}
    # what this process may actually use of them (a GPU box hands out a share of its host)
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cgroup_cpu_limit"] = round(int(txt[0]) / int(txt[1]), 2)
            elif int(txt[0]) > 0:
                info["cgroup_cpu_limit"] = round(int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    # the threads the CPU legs run on: what the process is actually granted, never the host's nproc (256 threads under
    # a 16-CPU cgroup measured oversubscription: 1.24 GiB/s against 1.91 on 16)
    usable = min(info["nproc"], info.get("affinity_cpus", info["nproc"]))
    if "cgroup_cpu_limit" in info:
        usable = min(usable, max(1, int(info["cgroup_cpu_limit"])))
    info["usable_cpus"] = max(1, usable)
    return info


def pmc_traffic(name, key, flate, args, group="lz77", split=None):
    """HBM bytes per launch from the rocprofv3 --pmc passes kept in profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate passes, units/corrections as MI355X_MICROARCH.md prescribes:
    see profiles/r02/README.md) -- but only when that collection describes the code being timed:
    the file carries the source hash of the library it was collected on (tools/traffic_reduce.py,
    flate_hip_build_id) and the run must use the default launch options.
    split = (streams the LDS-table blocks took, streams the guests took) in THIS run: the match finder's
    figure was collected with the queue split fixed (rocprofv3 serialises the two kernels) and the guests'
    share moves it, so each kernel's bytes are scaled from the collection's split to this run's -- the
    figure then describes the timed launch; the collected one and both splits are reported beside it.
    Returns (bytes or None, info dict for the bench line)."""
    tuned = bool(args.option) or args.no_guests
    lib_id = flate.id_component(flate.build_id(), group)
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        try:
            d = json.load(open(path))[key]
            val = int(d["hbm_bytes_per_launch"])
        except Exception:
            continue
        got_id = flate.id_component(d.get("build_id"), group)
        src = {"file": "profiles/%s/%s" % (rnd, name), "kernels": group, "collected_on_build": got_id,
               "collected_at_git_head": d.get("git_head"), "this_build": lib_id,
               "note": "the figure is quoted only when collected_on_build == this_build (hashes of the kernels' sources); "
                       "commits after collected_at_git_head did not touch them"}
        if tuned or got_id is None or got_id != lib_id:
            src["traffic_stale"] = True  # other sources, or non-default launch options: not quoted
            return None, src
        qs = d.get("queue_split") or {}
        if split and qs.get("lds_table_blocks") and qs.get("l2_table_guest_blocks") and split[0] and split[1]:
            lds = sum(v["hbm_bytes_per_launch"] for k, v in d.get("kernels", {}).items() if "lz77_wave_kernel" in k)
            gst = sum(v["hbm_bytes_per_launch"] for k, v in d.get("kernels", {}).items() if "lz77_guest_kernel" in k)
            if lds and gst:
                src.update({"as_collected": val, "collected_at_split": [qs["lds_table_blocks"], qs["l2_table_guest_blocks"]],
                            "this_run_split": [int(split[0]), int(split[1])],
                            "rule": "each kernel's bytes scaled by its streams in this run / in the collection"})
                val = int(lds * split[0] / qs["lds_table_blocks"] + gst * split[1] / qs["l2_table_guest_blocks"])
        return val, src
    return None, {"file": None, "this_build": lib_id}


def summarize(times_s):
    ms = sorted(t * 1e3 for t in times_s)
    return {"mean": round(sum(ms) / len(ms), 3), "median": round(statistics.median(ms), 3),
            "min": round(ms[0], 3), "max": round(ms[-1], 3)}


# ----------------------------------------------------------------------------------------------
# CPU baseline (the oracle on the host cores) -- also the parity checker
# ----------------------------------------------------------------------------------------------
def cpu_leg(host, in_off, n, blen, sample_streams, g_out=None, g_off=None):
    """Time the oracle (16 threads or nproc, and 1 thread on a smaller sample) on the first
    `sample_streams` streams and, when g_out is given, compare every stream it produced with the
    GPU's bytes.  Returns (cpu_baseline dict, verified stream count)."""
    import numpy as np
    from oracle import pyoracle
    ns = min(sample_streams, n)
    info = cpu_info()
    cores = info["usable_cpus"]
    t1 = time.perf_counter()
    o_buf, o_off, o_len = pyoracle.deflate_batch(host[:ns * blen], in_off[:ns + 1], nthreads=cores)
    cdt = time.perf_counter() - t1
    n1 = max(1, min(ns, (64 << 20) // max(blen, 1)))  # 1-thread line: <= 64 MiB (about half a second)
    t1 = time.perf_counter()
    pyoracle.deflate_batch(host[:n1 * blen], in_off[:n1 + 1], nthreads=1)
    cdt1 = time.perf_counter() - t1
    # SURVEY 8(d) context line: libz deflate level 1, raw, the same threads and sample.  A DIFFERENT
    # algorithm (zlib's deflate_fast with a hash chain, not the reference's) -- context, not parity.
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def zl(r):
        tot = 0
        for i in r:
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            tot += len(co.compress(host[int(in_off[i]):int(in_off[i + 1])])) + len(co.flush())
        return tot
    nz = max(1, min(ns, (256 << 20) // max(blen, 1)))
    t1 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        zbytes = sum(ex.map(zl, [range(k, nz, cores) for k in range(cores)]))
    zdt = time.perf_counter() - t1
    base = {
        "value": round(ns * blen / cdt / 2**30, 4), "unit": "GiB/s", "cores": cores, "kind": "port",
        "libz_level1_context": {"value": round(nz * blen / zdt / 2**30, 4), "unit": "GiB/s", "threads": cores,
                                "ratio": round(nz * blen / max(zbytes, 1), 3),
                                "note": "zlib %s deflate level 1, raw: a different algorithm, not a parity target"
                                        % zlib.ZLIB_RUNTIME_VERSION,
                                "sample": "first %d streams (%d MiB), %.2f s wall" % (nz, nz * blen >> 20, zdt)},
        "sample": "first %d of %d streams (%d MiB), oracle C restatement, %d threads, %.1f s wall"
                  % (ns, n, ns * blen >> 20, cores, cdt),
        "single_thread": {"value": round(n1 * blen / cdt1 / 2**30, 4), "unit": "GiB/s",
                          "sample": "first %d streams (%d MiB), %.1f s wall" % (n1, n1 * blen >> 20, cdt1)},
        "host": info,
    }
    verified = 0
    if g_out is not None:
        for i in range(ns):
            a0, a1 = int(g_off[i]), int(g_off[i + 1])
            b0 = int(o_off[i])
            if a1 - a0 != int(o_len[i]) or not np.array_equal(g_out[a0:a1], o_buf[b0:b0 + a1 - a0]):
                raise SystemExit("PARITY FAILURE at stream %d" % i)
        verified = ns
    return base, verified


# ----------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=None,
                    help="streams per GPU (default: 16384 = 1 GiB, configs[1]; --mode inflate: 131072 = 8 GiB, configs[4])")
    ap.add_argument("--stream-len", type=int, default=65536)
    ap.add_argument("--kind", default="text", choices=["text", "ramp", "rand", "zero"])
    ap.add_argument("--no-gather", action="store_true", help="skip the exchange step (N>1)")
    ap.add_argument("--gather-mode", default="allgather", choices=["allgather", "sendrecv"],
                    help="exchange inside the timed region: padded all_gather_into_tensor or grouped isend/irecv")
    ap.add_argument("--native-gather", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--no-native-gather", action="store_true",
                    help="N>1: skip the exchange through the C ABI.  By default, after the timed region, every rank "
                         "also runs flate_hip_gather_compressed on the library's own RCCL communicator, both forms, "
                         "times them against 7 x 153 GB/s of xGMI and compares the result with torch.distributed's "
                         "(a mismatch ends the run with an error)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=16384,
                    help="streams of the workload the CPU oracle is timed on (about 10 CPU-seconds per GiB)")
    ap.add_argument("--verify", type=int, default=1,
                    help="0 = skip comparing the GPU output with the oracle's in the cpu_baseline leg")
    ap.add_argument("--no-guests", action="store_true",
                    help="match finder with LDS-table blocks only (no L2-table guest blocks)")
    ap.add_argument("--no-extra", action="store_true",
                    help="N=1 deflate: skip the configs[2] / configs[4] legs reported under \"extra\"")
    ap.add_argument("--spliced", action="store_true",
                    help="encode into ONE DEFLATE stream per GPU (flate_hip_deflate_fast_spliced, SURVEY 8f-3)")
    ap.add_argument("--inflate-lanes", type=int, default=0, choices=[0, 16, 32, 64],
                    help="inflate: streams per wavefront (0 = chosen from the batch size)")
    ap.add_argument("--mode", default="deflate", choices=["deflate", "inflate"],
                    help="inflate = BASELINE.json configs[4]: decode the compressed streams (stream index supplied)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="flate_hip_set_option knob (developer A/B runs), may repeat")
    args = ap.parse_args()
    if args.streams is None:
        args.streams = 131072 if args.mode == "inflate" else 16384

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]))

    import numpy as np
    import torch
    flate, Engine, test_engine = load_engine_module()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # FLATE_BENCH_BACKEND=gloo rehearses the N>1 control flow where RCCL has no peers (several
    # ranks on one GPU, or the test engine on CPU); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("FLATE_BENCH_BACKEND", "gloo" if test_engine else "nccl")
    cuda = not test_engine
    if cuda and backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    if cuda:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if cuda else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod

    n, blen = args.streams, args.stream_len
    host = flate.synth(args.kind, n, blen, first_stream=rank * n)
    d_in = torch.from_numpy(host).to(dev)
    in_off = flate.uniform_offsets(n, blen)
    in_bytes = n * blen

    eng = Engine(local_rank)
    if cuda:
        eng.use_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_profiling(True)
    if args.no_guests:
        eng.set_option("guest_blocks", 0)
    if args.inflate_lanes:
        eng.set_option("inflate_lanes", args.inflate_lanes)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    env = {"torch": torch, "np": np, "flate": flate, "eng": eng, "dev": dev, "dist": dist,
           "world": world, "rank": rank, "cuda": cuda, "test_engine": test_engine}

    if args.mode == "inflate":
        res = bench_inflate(args, env, d_in, in_off, n, blen, host=host)
    else:
        res = bench_deflate(args, env, host, d_in, in_off, n, blen)
        if rank == 0 and world == 1 and not args.no_extra and not args.spliced and cuda:
            del d_in, host
            torch.cuda.empty_cache()
            res["extra"] = extra_legs(args, env)
    # Teardown.  A collective that never completes cannot be cancelled, only left behind: the line still goes out, it
    # says what was abandoned, and the process then ends with kExitAbandoned instead of 0 so that the launcher's exit
    # code shows what the line's c_abi_error / teardown fields say.
    abandoned = bool(env.get("hard_exit"))
    teardown = {}
    limit = float(os.environ.get("FLATE_BENCH_TEARDOWN_LIMIT_S", "120"))
    if env.get("test_engine") and os.environ.get("FLATE_TEST_LATE_TEARDOWN_RANK") == str(rank):
        time.sleep(float(os.environ.get("FLATE_TEST_LATE_TEARDOWN_S", "10")))  # (tests: a peer that reaches the barrier too late)
    if dist is not None and not abandoned:
        ok, r = run_with_time_limit(lambda: dist.barrier(), env, limit)
        if not ok:
            teardown["barrier"] = "did not finish within %g s; left behind" % limit
        elif isinstance(r, BaseException):  # (a peer that is gone: the collective fails instead of hanging)
            teardown["barrier"] = "raised %s: %s" % (type(r).__name__, str(r)[:200])
        else:
            teardown["barrier"] = "ok"
        abandoned = teardown["barrier"] != "ok"
    if rank == 0:
        if dist is not None:
            res["teardown"] = dict(teardown, abandoned_collective=abandoned,
                                   exit_code=kExitAbandoned if abandoned else 0)
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None and not abandoned:
        ok, r = run_with_time_limit(lambda: dist.destroy_process_group(), env, min(limit, 60.0))
        if not ok or isinstance(r, BaseException):
            sys.stderr.write("bench.py: destroy_process_group %s\n" % ("did not finish in time; left behind" if not ok else "raised %r" % (r,)))
            abandoned = True
    if abandoned:
        # (no eng.close(): the stuck call may still hold the engine; never restart or exec from here)
        sys.stderr.flush()
        os._exit(kExitAbandoned)
    eng.close()


def run_with_time_limit(fn, env, seconds):
    """fn() in a helper thread (the calling thread only waits): (True, its result or the exception it raised), or
    (False, None) when it is still running after `seconds` -- a collective that never completes cannot be cancelled,
    only left behind."""
    import threading
    box = {}

    def work():
        try:
            if env["cuda"]:
                env["torch"].cuda.set_device(env["dev"])
            box["val"] = fn()
        except BaseException as e:  # noqa: BLE001  (SystemExit included: the caller re-raises it)
            box["val"] = e

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(seconds)
    return (not t.is_alive()), box.get("val")


def sync_all(env):
    if env["dist"] is not None:
        env["dist"].barrier()
    if env["cuda"]:
        env["torch"].cuda.synchronize()


def max_over_ranks(env, dt):
    dist, torch = env["dist"], env["torch"]
    if dist is None:
        return dt
    tt = torch.tensor([dt], dtype=torch.float64, device=env["dev"] if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def bench_deflate(args, env, host, d_in, in_off, n, blen):
    torch, np, flate, eng, dev, dist = (env[k] for k in ("torch", "np", "flate", "eng", "dev", "dist"))
    world, rank = env["world"], env["rank"]
    in_bytes = n * blen
    out = torch.empty(in_bytes + (in_bytes >> 3) + 4096, dtype=torch.uint8, device=dev)
    gather = (world > 1) and not args.no_gather
    shard = importlib.import_module("moonbit-flate_amd.shard")
    # The exchange step of batch k overlaps the compression of batch k+1: two output buffers
    # alternate, a buffer is reused only after the gather that reads it is done, and every gather
    # completes inside the timed region (sync_all waits for all of them).  Nothing on the issue
    # path of the gather waits for the GPU (sticky pad, metadata resolved at wait()).
    outs = [out, torch.empty_like(out)] if gather else [out]
    pending = [None] * len(outs)
    plan = shard.GatherPlan(kmax=1 if args.spliced else n)
    gbuf = [None]
    nstep = [0]
    last = {}

    def finish(i):
        g = pending[i]
        if g is None:
            return
        pending[i] = None
        g.wait()
        if g.overflow:  # a shard outgrew the sticky pad (first batches only): repeat, blocking
            g = shard.gather_compressed(dist, last[i][0], last[i][1], plan=plan, mode=args.gather_mode)
        gbuf[0] = g.buf
        last["gathered"] = g

    def step():
        i = nstep[0] % len(outs)
        nstep[0] += 1
        finish(i)
        if args.spliced:  # one DEFLATE stream per GPU; the gather then moves one "stream" per rank
            _, nbytes, _ = eng.deflate_spliced(d_in, in_off, out=outs[i])
            out_off = np.array([0, nbytes], dtype=np.uint64)
        else:
            _, out_off = eng.deflate_batch(d_in, in_off, out=outs[i])
        if gather:  # north_star's exchange step: every rank ends up with every compressed shard
            buf = gbuf[0] if gbuf[0] is not None and gbuf[0].numel() >= plan.pad * world else None
            pending[i] = shard.gather_compressed(dist, outs[i], out_off, buf=buf, plan=plan, wait=False,
                                                 mode=args.gather_mode)
            last[i] = (outs[i], out_off)
        return out_off, i

    def drain():
        for i in range(len(pending)):
            finish(i)

    for _ in range(args.warmup):
        step()
    drain()
    stage_ms = {k: 0.0 for k in flate.STAGES}
    sync_all(env)
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        out_off, last_i = step()
        step_s.append(time.perf_counter() - ts)  # deflate_batch returns after its stream has drained
        tm = eng.last_timing()
        for k in stage_ms:
            stage_ms[k] += tm[k]
    drain()
    sync_all(env)
    dt = max_over_ranks(env, time.perf_counter() - t0)

    clen = int(out_off[-1])
    ratio = in_bytes / clen
    steps = args.steps
    split = None
    if hasattr(eng, "last_resident_share"):
        a, b = eng.last_resident_share()
        if b:
            split = {"lds_table_blocks": a, "l2_table_guest_blocks": b - a, "queued": b}

    # Parity of the exchange step FIRST, on the buffer the timed region's last overlapped
    # (wait=False) gather produced -- before anything else may write a gather buffer -- and then on
    # two more overlapped steps (both output buffers, both orders of reuse), each checked right after
    # its finish().  Rank 0 regenerates a strided sample of every rank's streams from their seeds.
    gathered_verified = 0

    def check_gathered(g, per_rank):
        from oracle import pyoracle
        cnt = 0
        for r in range(world):
            for i in range(0, n, max(1, n // per_rank)):
                src = flate.synth(args.kind, 1, blen, first_stream=r * n + i)
                if bytes(g.stream(r * n + i).cpu().numpy()) != pyoracle.deflate(src):
                    raise SystemExit("PARITY FAILURE in the gathered buffer: rank %d stream %d" % (r, i))
                cnt += 1
        return cnt

    check = gather and args.verify and not args.spliced and not args.no_cpu_baseline
    if check:
        if rank == 0:
            gathered_verified += check_gathered(last["gathered"], 64)
        for _ in range(2):  # untimed: the overlapped path again, every finished gather checked
            step()
            drain()
            if rank == 0:
                gathered_verified += check_gathered(last["gathered"], 8)
        sync_all(env)

    # exchange step alone, both forms (outside the timed region; N>1); its own buffer (buf=None):
    # the buffers of the overlapped path above are not touched any more
    gather_info = "none"
    if gather:
        gather_info = {"mode_in_timed_region": args.gather_mode, "overlapped_with_next_batch": True,
                       "gathered_parity_checked_streams": gathered_verified}
        g = last["gathered"]
        recv = int(g.sizes.sum() - g.sizes[rank])
        gather_info.update({"bytes_received_per_gpu": recv, "padded_bytes_per_rank": int(g.pad)})
        # (the timed region above is what `value` reports; a failure of this side measurement is
        # recorded, it must not cost the run its line)
        modes = (args.gather_mode,) + tuple(m for m in ("allgather", "sendrecv") if m != args.gather_mode)
        side_buf = None
        for mode in modes:
            try:
                ts = []
                for _ in range(3):
                    sync_all(env)
                    t1 = time.perf_counter()
                    sg = shard.gather_compressed(dist, outs[last_i], out_off, buf=side_buf, plan=plan, mode=mode)
                    side_buf = sg.buf
                    sync_all(env)
                    ts.append(time.perf_counter() - t1)
                gather_info[mode + "_ms"] = summarize(ts)
            except Exception as e:  # noqa: BLE001
                gather_info[mode + "_error"] = "%s: %s" % (type(e).__name__, e)
                break
        del side_buf
        # (FLATE_BENCH_FORCE_NATIVE=1: rehearsals on a one-GPU box -- gloo between the ranks, the library's TEST build
        # with the tests' rehearsal transport in place of RCCL -- take the same path)
        native_ok = (env["cuda"] and (dist.get_backend() == "nccl" or os.environ.get("FLATE_BENCH_FORCE_NATIVE") == "1")) \
            or hasattr(eng, "native_comm")
        if not args.no_native_gather and native_ok:
            # the same exchange through include/flate_hip.h (what a MoonBit / C++ host calls): ON by default, so
            # that the first run on a multi-GPU node exercises flate_hip_gather_* over real RCCL / xGMI.
            # It is a side measurement that has never run on more than one real GPU: it may fail, it must never
            # cost the run its line, and it must never HANG it -- (1) every rank first probes, without a
            # collective, that it can bind RCCL, and the ranks agree on that before anyone enters the collective
            # communicator set-up; (2) the whole section runs under a time limit in a helper thread; a rank whose
            # section is still stuck in a collective when the limit passes records that and leaves the process
            # with os._exit once its line is out (main()).
            def native_section():
                ok_local, why = 1, ""
                if not hasattr(eng, "native_comm"):
                    try:
                        shard.NativeComm.probe(eng)
                    except Exception as e:  # noqa: BLE001
                        ok_local, why = 0, "%s: %s" % (type(e).__name__, e)
                flag = torch.tensor([ok_local], dtype=torch.int32, device=env["dev"] if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    raise RuntimeError("a rank cannot bind RCCL for the C-ABI exchange (this rank: %s)" % (why or "ok"))
                comm = eng.native_comm(rank, world, dist) if hasattr(eng, "native_comm") else shard.NativeComm(eng, rank, world, dist)
                nat = {}
                for mode in ("allgather", "sendrecv"):
                    ts = []
                    for _ in range(3):
                        sync_all(env)
                        t1 = time.perf_counter()
                        ng = comm.gather(outs[last_i], out_off, mode=mode)
                        sync_all(env)
                        ts.append(time.perf_counter() - t1)
                    for j in range(0, world * n, max(1, world * n // 64)):
                        if not torch.equal(ng.stream(j), g.stream(j)):
                            raise SystemExit("PARITY FAILURE: C-ABI exchange (%s) differs from torch.distributed at stream %d" % (mode, j))
                    ms = summarize(ts)
                    gbs = recv / (ms["mean"] * 1e-3) / 1e9
                    nat[mode + "_ms"] = ms
                    nat[mode + "_GBs"] = round(gbs, 2)
                    nat[mode + "_frac_of_xgmi"] = round(gbs / XGMI_PEAK_GBS, 4)
                nat["compared_streams"] = len(range(0, world * n, max(1, world * n // 64)))
                comm.close()
                return nat

            done, val = run_with_time_limit(native_section, env, float(os.environ.get("FLATE_BENCH_NATIVE_LIMIT_S", "240")))
            if not done:
                env["hard_exit"] = True
                gather_info["c_abi_error"] = "the C-ABI exchange did not finish within its time limit; left behind, the process ends with exit code %d" % kExitAbandoned
                print("bench: " + gather_info["c_abi_error"], file=sys.stderr)
            elif isinstance(val, SystemExit):
                raise val  # a mismatch ends the run
            elif isinstance(val, BaseException):  # (a missing RCCL must not cost the run its line)
                gather_info["c_abi_error"] = "%s: %s" % (type(val).__name__, val)
                print("bench: C-ABI exchange failed: %s" % gather_info["c_abi_error"], file=sys.stderr)
            else:
                gather_info["c_abi"] = val
        mins = [gather_info[m + "_ms"]["min"] for m in ("allgather", "sendrecv") if m + "_ms" in gather_info]
        if mins:
            best = min(mins)
            gather_info.update({
                "achieved_GBs": round(recv / (best * 1e-3) / 1e9, 2), "xgmi_peak_GBs": XGMI_PEAK_GBS,
                "frac_of_xgmi": round(recv / (best * 1e-3) / 1e9 / XGMI_PEAK_GBS, 4)})

    # cpu_baseline leg (rank 0, outside the timed region), the SAME at every N so that the 1-, 2-,
    # 4- and 8-GPU lines read side by side: the oracle compresses rank 0's own shard (the first
    # --cpu-sample-streams streams of its 1 GiB) on the host cores (16 threads, and 1 thread on a
    # smaller sample); every stream it produced is compared with what this GPU wrote.
    verified = 0
    cpu_baseline = None
    if rank == 0 and not args.no_cpu_baseline:
        g_cpu = outs[last_i][:clen].cpu().numpy() if (args.verify and not args.spliced) else None
        cpu_baseline, verified = cpu_leg(host, in_off, n, blen, args.cpu_sample_streams, g_cpu, out_off)
        if args.spliced and args.verify and world == 1:
            from oracle import pyoracle
            ns = min(n, 2048)
            o1, nb1, b1 = eng.deflate_spliced(d_in[:ns * blen], in_off[:ns + 1])
            ref, ref_off = pyoracle.deflate_spliced(host[:ns * blen], in_off[:ns + 1])
            if bytes(o1[:nb1].cpu().numpy()) != ref or not (b1 == ref_off).all():
                raise SystemExit("PARITY FAILURE: spliced stream differs from the oracle")
            verified = ns
    verified += gathered_verified

    value = world * in_bytes * steps / dt / 2**30
    lz_ms = stage_ms["lz77_match"] / steps
    algo_bytes = in_bytes + clen  # B read + C written per stream, all streams
    achieved = algo_bytes / (lz_ms * 1e-3) / 1e9 if lz_ms > 0 else 0.0
    checked = ("bit-exact vs oracle (%d streams compared)" % verified) if verified else "parity not checked in this run"
    traffic, traffic_src = None, None
    if args.kind == "text" and n == 16384 and blen == 65536:
        traffic, traffic_src = pmc_traffic("lz77_traffic.json", "lz77_default_16384x65536_text", flate, args,
                                           split=(split["lds_table_blocks"], split["l2_table_guest_blocks"]) if split else None)
    return {
        "metric": "GiB/s uncompressed throughput (encode), 64 KiB blocks, deflate-fast",
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic" if not env["test_engine"] else "TEST-ENGINE (control-flow rehearsal, not a measurement)",
        "config": {
            "workload": "%d x %d B independent streams per GPU (%.3f GiB/GPU), S-%s, deflate-fast, %s%s"
                        % (n, blen, in_bytes / 2**30, args.kind, checked,
                           ", spliced into ONE stream per GPU" if args.spliced else ""),
            "streams_per_gpu": n, "stream_len": blen, "kind": args.kind,
            "compressed_bytes_per_gpu": clen, "ratio": round(ratio, 4),
            "gather": gather_info,
            "parity_checked_streams": verified,
            "stage_ms": {k: round(v / steps, 3) for k, v in stage_ms.items()},
            "step_ms": summarize(step_s),
            "lz77_streams_by_kernel": split,
        },
        "roofline": {
            "bound": "hbm", "kernel": "lz77 match finder (resident + guest launch)", "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src,
        },
        "cpu_baseline": cpu_baseline,
    }


def bench_inflate(args, env, d_in, in_off, n, blen, host=None, steps=None, warmup=None):
    """Config 5: inflate-only.  Streams compressed once (untimed) by the encoder; one step = one
    inflate_batch over all of them; value = GiB/s of decompressed output."""
    torch, np, eng, dev, dist = (env[k] for k in ("torch", "np", "eng", "dev", "dist"))
    world, rank = env["world"], env["rank"]
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    sizes = [blen] * n
    out = torch.empty(n * blen, dtype=torch.uint8, device=dev)
    if args.spliced:  # ONE compressed stream + its index, decoded in parallel (config 5 as worded)
        comp, nbytes, bit_off = eng.deflate_spliced(d_in, in_off)
        coff = [0, nbytes]
        run = lambda: eng.inflate_spliced(comp, nbytes, bit_off, sizes, out=out)
    else:
        comp, coff = eng.deflate_batch(d_in, in_off)
        run = lambda: eng.inflate_batch(comp, coff, sizes, out=out)

    for _ in range(warmup):
        run()
    ms = 0.0
    step_s = []
    sync_all(env)
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        _, _, olen, status, _ = run()
        step_s.append(time.perf_counter() - ts)
        ms += eng.last_timing()["inflate"]
    sync_all(env)
    dt = max_over_ranks(env, time.perf_counter() - t0)
    ok = bool((status == 0).all()) and bool(torch.equal(out, d_in))  # round-trip property, full size
    if not ok:
        raise SystemExit("INFLATE ROUND-TRIP FAILURE: output != input")
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.spliced:
        from oracle import pyoracle
        ns = min(args.cpu_sample_streams * 2, n)  # the decoder is faster than the encoder
        info = cpu_info()
        cores = info["usable_cpus"]
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
            "host": info,
        }
    host_leg = None
    if getattr(args, "inflate_host_leg", False) and not args.spliced and world == 1:
        # the same call with pageable host buffers (copies pipelined with the decoding over groups of
        # streams): the PCIe-inclusive rate, never the headline
        h_comp = comp[:int(coff[-1])].cpu().numpy()
        h_out = np.empty(n * blen, dtype=np.uint8)
        h_out[::4096] = 0  # first touch of the output pages (8 GiB: 0.3 s of page faults) is not the call's time

        def host_calls(reps):
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                _, _, h_len, h_st, _ = eng.inflate_batch(h_comp, coff, sizes, out=h_out)
                ts.append(time.perf_counter() - t1)
            if int(h_st.any()) or int((h_len != blen).any()):
                raise SystemExit("host-pointer inflate failed")
            for i in range(0, n, 4099):
                if not np.array_equal(h_out[i * blen:(i + 1) * blen], d_in[i * blen:(i + 1) * blen].cpu().numpy()):
                    raise SystemExit("host-pointer inflate: output != input at stream %d" % i)
            return ts
        ts_page = host_calls(3)
        with eng.host_register(h_comp), eng.host_register(h_out):
            host_calls(1)
            ts = host_calls(4)
        host_leg = {"value": round(n * blen / min(ts) / 2**30, 2), "unit": "GiB/s", "step_ms": summarize(ts),
                    "buffers": "page-locked once by the caller (flate_hip_host_register), reused across calls",
                    "pageable_buffers": {"value": round(n * blen / min(ts_page) / 2**30, 2), "unit": "GiB/s",
                                         "step_ms": summarize(ts_page)},
                    "note": "same workload, input and output in HOST memory: %.2f GiB in and %.2f GiB out over "
                            "PCIe, pipelined with the decoding over groups of streams (not the headline value)"
                            % (int(coff[-1]) / 2**30, n * blen / 2**30)}
        del h_comp, h_out
    clen = int(coff[-1])
    k_ms = ms / steps
    achieved = (n * blen + clen) / (k_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    if args.kind == "text" and blen == 65536 and n in (131072, 16384) and not args.spliced:
        traffic, traffic_src = pmc_traffic("inflate_traffic.json", "inflate_%dx65536_text" % n, env["flate"], args,
                                           group="inflate")
    return {
        "metric": "GiB/s decompressed output (inflate), 64 KiB streams", "unit": "GiB/s",
        "value": round(world * n * blen * steps / dt / 2**30, 3), "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "inflate %s%d x %d B streams per GPU, S-%s, output == input checked on every byte"
                               % ("ONE spliced stream of " if args.spliced else "", n, blen, args.kind),
                   "compressed_bytes_per_gpu": clen,
                   "stage_ms": {"inflate": round(k_ms, 3)}, "step_ms": summarize(step_s)},
        # (the library's choice at its default options: the sub-block decoder below 45056 streams)
        "roofline": {"bound": "hbm", "kernel": "inflate_simt_kernel" if n >= 45056 else "inflate_spec_kernel",
                     "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": traffic, "traffic_source": traffic_src},
        "cpu_baseline": cpu_baseline,
        **({"end_to_end_host_pointers": host_leg} if host_leg else {})}


def extra_legs(args, env):
    """N=1 only, after the headline: BASELINE.json configs[2] (4096 x 262144 B, multi-window match
    finding, parity-checked on a strided sample) and configs[4] (inflate-only, 131072 streams =
    8 GiB of output, output == input on every byte).  Same engine, same measurement rules."""
    torch, np, flate, eng, dev = (env[k] for k in ("torch", "np", "flate", "eng", "dev"))
    extra = {}
    # ---- the headline shape on the other synthetic inputs of BASELINE.md (S-ramp, S-zero, S-rand) and
    # the same call with HOST pointers (one H2D of the input + one D2H of the output inside the call:
    # the PCIe-inclusive rate; never the headline value)
    n, blen = 16384, 65536
    in_off = flate.uniform_offsets(n, blen)
    out = torch.empty(n * blen + (n * blen >> 3) + 4096, dtype=torch.uint8, device=dev)
    others = {}
    for kind in ("ramp", "zero", "rand", "text"):
        host = flate.synth(kind, n, blen)
        d_in = torch.from_numpy(host).to(dev)
        eng.deflate_batch(d_in, in_off, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            _, out_off = eng.deflate_batch(d_in, in_off, out=out)
            ts.append(time.perf_counter() - t1)
        clen = int(out_off[-1])
        checked = 0
        if args.verify and not args.no_cpu_baseline:
            from oracle import pyoracle
            g_cpu = out[:clen].cpu().numpy()
            for i in range(0, n, 256):
                if bytes(g_cpu[int(out_off[i]):int(out_off[i + 1])]) != pyoracle.deflate(host[i * blen:(i + 1) * blen]):
                    raise SystemExit("PARITY FAILURE (S-%s) at stream %d" % (kind, i))
                checked += 1
        if kind != "text":
            others["S-" + kind] = {"value": round(n * blen / min(ts) / 2**30, 2), "unit": "GiB/s", "ratio": round(n * blen / clen, 3),
                                   "step_ms": summarize(ts), "parity_checked_streams": checked}
        else:  # host pointers: what a Writer-style caller of the C ABI sees (PCIe inside the call)
            h_out = np.empty(out.numel(), dtype=np.uint8)
            eng.deflate_batch(host, in_off, out=h_out)  # (first touch of the output pages: not timed)

            def host_calls(reps):
                ts = []
                for _ in range(reps):
                    t1 = time.perf_counter()
                    _, h_off = eng.deflate_batch(host, in_off, out=h_out)
                    ts.append(time.perf_counter() - t1)
                if int(h_off[-1]) != clen or not np.array_equal(h_out[:clen], out[:clen].cpu().numpy()):
                    raise SystemExit("host-pointer call produced different bytes")
                return ts
            ts_page = host_calls(3)
            with eng.host_register(host), eng.host_register(h_out):
                host_calls(1)
                ts = host_calls(4)
            extra["end_to_end_host_pointers"] = {
                "value": round(n * blen / min(ts) / 2**30, 2), "unit": "GiB/s", "step_ms": summarize(ts),
                "buffers": "page-locked once by the caller (flate_hip_host_register), reused across calls",
                "pageable_buffers": {"value": round(n * blen / min(ts_page) / 2**30, 2), "unit": "GiB/s",
                                     "step_ms": summarize(ts_page)},
                "note": "same workload, input and output in HOST memory: the call copies 1 GiB in and %.2f GiB out "
                        "over PCIe, pipelined with the compression over groups of streams on two lanes; bytes "
                        "compared with the device-resident run (not the headline value)" % (clen / 2**30)}
            # the containers' checksums of the same GiB (flate_hip_checksum_batch): one read of
            # the input, the kernels' event time -- a kernel family that CAN be priced against the HBM roofline
            if hasattr(eng, "checksum_batch"):
                import zlib
                sums = {}
                for ck, ref in (("adler32", zlib.adler32), ("crc32", zlib.crc32)):
                    eng.checksum_batch(d_in, in_off, ck)
                    kms = []
                    cts = []  # the whole call as the caller sees it (staging, launches, the read-back of the sums)
                    for _ in range(5):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        got = eng.checksum_batch(d_in, in_off, ck)
                        torch.cuda.synchronize()
                        cts.append(time.perf_counter() - t1)
                        kms.append(eng.last_timing()["checksum"])
                    for i in range(0, n, 512):
                        if int(got[i]) != ref(host[i * blen:(i + 1) * blen].tobytes()):
                            raise SystemExit("checksum %s differs from zlib's at stream %d" % (ck, i))
                    gbs = n * blen / (sum(kms) / len(kms) * 1e-3) / 1e9  # from the MEAN, like every other leg
                    sums[ck] = {"kernel_ms": summarize([k * 1e-3 for k in kms]), "GB_per_s": round(gbs, 1),
                                "call_ms": summarize(cts), "call_GB_per_s": round(n * blen / (sum(cts) / len(cts)) / 1e9, 1),
                                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(gbs / HBM_PEAK_GBS, 4)},
                                "checked_against_zlib_streams": len(range(0, n, 512))}
                extra["container_checksums_16384x65536"] = sums
        del host, d_in
    extra["other_inputs_16384x65536"] = others
    del out
    torch.cuda.empty_cache()
    # ---- configs[2]
    n, blen = 4096, 262144
    host = flate.synth("text", n, blen)
    d_in = torch.from_numpy(host).to(dev)
    in_off = flate.uniform_offsets(n, blen)
    out = torch.empty(n * blen + (n * blen >> 3) + 4096, dtype=torch.uint8, device=dev)
    for _ in range(2):
        eng.deflate_batch(d_in, in_off, out=out)
    stage = {k: 0.0 for k in flate.STAGES}
    step_s = []
    steps = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        _, out_off = eng.deflate_batch(d_in, in_off, out=out)
        step_s.append(time.perf_counter() - ts)
        tm = eng.last_timing()
        for k in stage:
            stage[k] += tm[k]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clen = int(out_off[-1])
    verified = 0
    if args.verify and not args.no_cpu_baseline:
        from oracle import pyoracle
        g_cpu = out[:clen].cpu().numpy()
        for i in range(0, n, 32):  # 128 streams x 256 KiB = 32 MiB through the oracle
            want = pyoracle.deflate(host[i * blen:(i + 1) * blen])
            if bytes(g_cpu[int(out_off[i]):int(out_off[i + 1])]) != want:
                raise SystemExit("PARITY FAILURE (configs[2]) at stream %d" % i)
            verified += 1
    lz_ms = stage["lz77_match"] / steps
    ach = (n * blen + clen) / (lz_ms * 1e-3) / 1e9
    c3_traffic, c3_src = pmc_traffic("lz77_traffic.json", "lz77_default_4096x262144_text", flate, args)
    extra["config3_1GiB_of_256KiB_streams"] = {
        "metric": "GiB/s uncompressed throughput (encode), 256 KiB streams, deflate-fast",
        "value": round(n * blen * steps / dt / 2**30, 3), "unit": "GiB/s", "steps": steps,
        "ms_per_step": round(dt / steps * 1e3, 3), "step_ms": summarize(step_s),
        "workload": "4096 x 262144 B streams (4 chained windows each), S-text, compat=moonbit",
        "ratio": round(n * blen / clen, 4), "parity_checked_streams": verified,
        "stage_ms": {k: round(v / steps, 3) for k, v in stage.items()},
        "roofline": {"bound": "hbm", "kernel": "lz77 match finder, multi-window (resident + guest launch)",
                     "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBS, 5),
                     "traffic": c3_traffic, "traffic_source": c3_src},
    }
    del host, d_in, out
    torch.cuda.empty_cache()
    # ---- configs[4]
    n, blen = 131072, 65536
    host = flate.synth("text", n, blen)
    d_in = torch.from_numpy(host).to(dev)
    del host
    in_off = flate.uniform_offsets(n, blen)
    ia = argparse.Namespace(**vars(args))
    ia.spliced, ia.no_cpu_baseline, ia.kind = False, True, "text"
    ia.inflate_host_leg = True
    r = bench_inflate(ia, env, d_in, in_off, n, blen, steps=3, warmup=1)
    extra["config5_inflate_8GiB"] = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step",
                                                       "config", "roofline", "end_to_end_host_pointers")
                                     if k in r}
    # ---- the headline batch's own output inflated again: 16384 streams = 1 GiB (one wavefront per
    # stream, sub-block decoder)
    n = 16384
    ia.inflate_host_leg = False
    r = bench_inflate(ia, env, d_in[:n * blen], in_off[:n + 1], n, blen, steps=5, warmup=2)
    extra["inflate_1GiB_16384_streams"] = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step",
                                                             "config", "roofline") if k in r}
    return extra


if __name__ == "__main__":
    main()
