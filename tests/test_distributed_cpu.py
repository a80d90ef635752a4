"""World-size-2 gloo test (CPU) of the multi-GPU row: contiguous sharding of streams, the
all-gather concatenation of the compressed shards and the global stream index.  The compress step
itself has no CPU path in the product, so each rank's shard is produced by the oracle here."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flate = importlib.import_module("moonbit-flate_amd")
    shard = importlib.import_module("moonbit-flate_amd.shard")
    lo, hi = shard.shard_range(n_streams, rank, world)
    lens = [3000 + 977 * j for j in range(n_streams)]          # ragged streams
    streams = [flate.synth("text", 1, lens[j], first_stream=j) for j in range(lo, hi)]
    comp = [np.frombuffer(pyoracle.deflate(s), dtype=np.uint8) for s in streams]
    off = np.zeros(len(comp) + 1, dtype=np.uint64)
    np.cumsum([c.size for c in comp], out=off[1:])
    local = torch.from_numpy(np.concatenate(comp).copy())
    g = shard.gather_compressed(dist, local, off, pad_to=4096)
    # the overlapped form used by bench.py: two gathers in flight into the same destination, each
    # from its own source buffer; the result of the later one is what remains
    decoy = torch.zeros_like(local)
    g0 = shard.gather_compressed(dist, decoy, off, pad_to=4096, wait=False)
    g1 = shard.gather_compressed(dist, local.clone(), off, buf=g0.buf, pad_to=4096, wait=False)
    g0.wait()
    g1.wait().wait()
    ok = torch.equal(g1.buf[:g.pad * world], g.buf[:g.pad * world])
    # sticky plan: after the first (blocking) agreement nothing on the issue path syncs; a payload
    # that outgrows the pad is reported as overflow by every rank and repeated with a larger pad
    plan = shard.GatherPlan(kmax=shard.max_shard_streams(n_streams, world), pad=4096, pad_to=4096)
    gp = shard.gather_compressed(dist, local, off, plan=plan, wait=False)
    gp.wait()
    ok = ok and gp.overflow and plan.pad >= int(gp.sizes.max())
    gp = shard.gather_compressed(dist, local, off, plan=plan, wait=False).wait()
    ok = ok and not gp.overflow and all(
        bytes(gp.stream(j).numpy()) == bytes(g.stream(j).numpy()) for j in range(n_streams))
    # grouped isend/irecv of the exact sizes (all peers at once)
    gs = shard.gather_compressed(dist, local, off, plan=plan, mode="sendrecv")
    ok = ok and all(bytes(gs.stream(j).numpy()) == bytes(g.stream(j).numpy()) for j in range(n_streams))
    # every rank must now hold every stream, in global order, bit-exact
    for j in range(n_streams):
        want = pyoracle.deflate(flate.synth("text", 1, lens[j], first_stream=j))
        got = bytes(g.stream(j).numpy())
        ok = ok and (got == want)
    ok = ok and int(g.counts.sum()) == n_streams and g.off.size == n_streams
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, int(g.sizes.sum())))


@pytest.mark.parametrize("n_streams", [7, 8])
def test_two_rank_gather_of_compressed_shards(n_streams):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_streams, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def test_shard_range_partitions_exactly():
    import importlib
    sys.path.insert(0, ROOT)
    shard = importlib.import_module("moonbit-flate_amd.shard")
    for n in (0, 1, 7, 16384, 16385):
        for w in (1, 2, 4, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def test_bench_spawns_its_own_ranks_and_verifies_the_gathered_buffer():
    """`bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself; rank 0 prints one JSON line
    with n_gpus = 2, both exchange forms timed and a sample of the gathered buffer checked against
    the oracle.  Runs on CPU through the test engine (gloo)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["FLATE_BENCH_TEST_ENGINE"] = "tests.cpu_engine:OracleEngine"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--streams", "128", "--stream-len", "4096"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert line["config"]["parity_checked_streams"] >= 64
    g = line["config"]["gather"]
    assert g["allgather_ms"]["min"] > 0 and g["sendrecv_ms"]["min"] > 0 and g["bytes_received_per_gpu"] > 0
    assert line["cpu_baseline"]["value"] > 0 and "TEST-ENGINE" in line["data"]
    # the exchange through the C ABI is part of the DEFAULT N > 1 run (here: the test engine's stand-in for it):
    # both forms timed, priced against xGMI, compared with the torch.distributed result
    c = g["c_abi"]
    assert c["allgather_ms"]["mean"] > 0 and c["sendrecv_ms"]["mean"] > 0 and c["compared_streams"] >= 64
    assert "allgather_frac_of_xgmi" in c and "c_abi_error" not in g


def test_bench_without_the_c_abi_exchange_when_asked():
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["FLATE_BENCH_TEST_ENGINE"] = "tests.cpu_engine:OracleEngine"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "64", "--stream-len", "4096", "--no-native-gather"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])["config"]["gather"]
    assert "c_abi" not in g and "c_abi_error" not in g


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_line_survives_a_c_abi_exchange_that_hangs():
    """The default N > 1 run must never lose its line to the side measurement: with one rank stuck outside the
    collective (the other then waits inside it for ever) every rank gives up at the time limit, rank 0's line is
    out -- with the failure recorded -- and every process ends."""
    import json
    import subprocess
    import time
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["FLATE_BENCH_TEST_ENGINE"] = "tests.cpu_engine:OracleEngine"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["FLATE_TEST_STUCK_RANK"] = "1"
    env["FLATE_BENCH_NATIVE_LIMIT_S"] = "5"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "64", "--stream-len", "4096"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 3, r.stderr[-2000:]  # (bench.kExitAbandoned: the hang shows in the exit code too)
    assert time.time() - t0 < 200
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["teardown"]["abandoned_collective"] is True and line["teardown"]["exit_code"] == 3
    g = line["config"]["gather"]
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "c_abi" not in g and "time limit" in g["c_abi_error"]


def test_bench_teardown_that_hangs_shows_in_the_line_and_in_the_exit_code():
    """A rank whose peer does not reach the final barrier in time gives up at its limit: rank 0's line is out, its
    `teardown` field says what was abandoned, and the launcher's exit code is bench.kExitAbandoned -- a hang at
    teardown no longer reads as rc 0."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["FLATE_BENCH_TEST_ENGINE"] = "tests.cpu_engine:OracleEngine"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["FLATE_TEST_LATE_TEARDOWN_RANK"] = "1"
    env["FLATE_TEST_LATE_TEARDOWN_S"] = "12"
    env["FLATE_BENCH_TEARDOWN_LIMIT_S"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "64", "--stream-len", "4096", "--no-native-gather"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["teardown"]["abandoned_collective"] is True and "did not finish" in line["teardown"]["barrier"]
