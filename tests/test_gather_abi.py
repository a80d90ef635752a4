"""The exchange step behind the C ABI (include/flate_hip.h: flate_hip_gather_*, flate_hip_comm_*).

CPU: the layout arithmetic (flate_hip_gather_layout) against the rules of shard.py's torch form.
GPU (one card): a one-rank RCCL communicator -- the index arithmetic, both exchange forms, the
overlapped begin/end pair (and that begin does not wait for the work queued in front of it), the
overflow of the sticky plan -- and TWO ranks as two processes on the one card over the tests'
rehearsal transport (tests/rehearsal_transport/: RCCL refuses two ranks on one device), which is
where the multi-rank branches of csrc/gather.hip run: peer sizes, rank_base placement, the grouped
send/receive loop, refusals and plan overflows decided alike on every rank."""
import importlib
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from util import flate

shard = importlib.import_module("moonbit-flate_amd.shard")


def test_layout_matches_the_torch_form():
    rb = [5 << 20, (3 << 20) + 17, 0, (7 << 20) - 1]
    pad, base, need = shard.gather_layout(rb, "allgather")
    plan = shard.GatherPlan(kmax=1)
    assert pad == plan.round(max(rb)) == 7 << 20
    assert base.tolist() == [r * pad for r in range(4)] and need == 4 * pad
    pad2, base2, need2 = shard.gather_layout(rb, "sendrecv")
    assert pad2 == pad and need2 == sum(rb)
    assert base2.tolist() == [0, rb[0], rb[0] + rb[1], rb[0] + rb[1] + rb[2]]
    # an empty world / empty shards
    assert shard.gather_layout([0], "allgather") == (1 << 20, np.zeros(1, np.uint64), 1 << 20)
    with pytest.raises(ValueError):
        shard.gather_layout([], "allgather")


@pytest.mark.gpu
def test_one_rank_communicator_both_forms_and_overlap(oracle):
    import torch
    n, blen = 96, 40000
    data = flate.synth("text", n, blen, first_stream=300)
    off = flate.uniform_offsets(n, blen)
    eng = flate.FlateEngine(0)
    comm = None
    try:
        d = torch.from_numpy(data).cuda()
        comp, coff = eng.deflate_batch(d, off)
        clen = int(coff[-1])
        comm = shard.NativeComm(eng, 0, 1)
        for mode in ("allgather", "sendrecv"):
            g = comm.gather(comp, coff, mode=mode)
            assert g.off.size == n and np.array_equal(g.length, coff[1:] - coff[:-1])
            assert np.array_equal(g.off, coff[:-1])  # rank 0 sits at the front in both layouts
            assert torch.equal(g.buf[:clen], comp[:clen])
            for j in (0, 17, n - 1):
                assert bytes(g.stream(j).cpu().numpy()) == oracle.deflate(data[j * blen:(j + 1) * blen])
        pad, kmax = comm.plan()
        assert pad == (clen + (1 << 20) - 1) // (1 << 20) * (1 << 20) and kmax == n
        # overlapped pair: issue, compress the next batch meanwhile, collect
        out = torch.zeros(pad + 64, dtype=torch.uint8, device="cuda")
        comm.begin(comp, coff, out)
        comp2, coff2 = eng.deflate_batch(d, off)
        g = comm.end(n)
        assert g is not None and torch.equal(g.buf[:clen], comp[:clen]) and torch.equal(comp2[:clen], comp[:clen])
        # a batch that outgrows the agreed pad: E_AGAIN, the plan is raised, the blocking form delivers
        big = flate.synth("rand", 40, 65536)
        boff = flate.uniform_offsets(40, 65536)
        bcomp, bcoff = eng.deflate_batch(torch.from_numpy(big).cuda(), boff)
        assert int(bcoff[-1]) > pad
        out2 = torch.zeros(4 * pad, dtype=torch.uint8, device="cuda")
        comm.begin(bcomp, bcoff, out2)
        assert comm.end(n) is None and comm.plan()[0] >= int(bcoff[-1])
        g = comm.gather(bcomp, bcoff)
        assert torch.equal(g.buf[:int(bcoff[-1])], bcomp[:int(bcoff[-1])])
        # more streams than the plan allows: not refused by this rank alone (its peers would wait in
        # the collective) -- the exchange runs, end reports E_AGAIN and the plan holds them afterwards
        nk = comm.plan()[1] + 50
        small = flate.synth("text", nk, 200, first_stream=9)
        kcomp, kcoff = eng.deflate_batch(torch.from_numpy(small).cuda(), flate.uniform_offsets(nk, 200))
        comm.begin(kcomp, kcoff, out2)
        assert comm.end(2 * nk) is None and comm.plan()[1] == nk
        comm.begin(kcomp, kcoff, out2)
        g = comm.end(2 * nk)
        assert g is not None and g.off.size == nk and torch.equal(g.buf[:int(kcoff[-1])], kcomp[:int(kcoff[-1])])
        # without a plan begin refuses (a rank-independent condition)
        comm2 = shard.NativeComm(eng, 0, 1)
        try:
            with pytest.raises(flate.FlateError):
                comm2.begin(comp, coff, out2)
        finally:
            comm2.close()
    finally:
        if comm is not None:
            comm.close()
        eng.close()


@pytest.mark.gpu
def test_begin_returns_before_the_work_in_front_of_it_is_done():
    """flate_hip_gather_begin is documented to return at once: the exchange waits -- on the GPU, behind an
    event -- for what is queued on the ctx's stream.  Its metadata copies therefore use pinned host memory
    (an asynchronous copy to or from pageable memory makes the calling thread wait for the stream)."""
    import torch
    n, blen = 64, 30000
    data = flate.synth("text", n, blen, first_stream=11)
    off = flate.uniform_offsets(n, blen)
    eng = flate.FlateEngine(0)
    comm = None
    try:
        comp, coff = eng.deflate_batch(torch.from_numpy(data).cuda(), off)
        comm = shard.NativeComm(eng, 0, 1)
        g0 = comm.gather(comp, coff)  # makes the plan
        side = torch.cuda.Stream()
        eng.use_stream(side.cuda_stream)
        # how long does the spin kernel take per tick on this box?
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            torch.cuda._sleep(2_000_000)
        side.synchronize()
        per_tick = (time.perf_counter() - t0) / 2_000_000
        ticks = int(min(max(0.5 / max(per_tick, 1e-10), 1e6), 4e9))  # about half a second
        out = torch.zeros(comm.plan()[0] + 64, dtype=torch.uint8, device="cuda")
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            torch.cuda._sleep(ticks)
            done.record(side)
        t0 = time.perf_counter()
        comm.begin(comp, coff, out)
        dt = time.perf_counter() - t0
        still_running = not done.query()
        g = comm.end(n)
        assert g is not None and torch.equal(g.buf[:int(coff[-1])], g0.buf[:int(coff[-1])])
        assert still_running and dt < 0.25, (dt, still_running, per_tick, ticks)
    finally:
        eng.use_stream(None)
        if comm is not None:
            comm.close()
        eng.close()


def _rehearsal_transport():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rehearsal_transport")
    src, lib = os.path.join(here, "rehearsal_rccl.cpp"), os.path.join(here, "librehearsal_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread", "-Wl,-rpath,/opt/rocm/lib",
                               "-o", lib])
    return lib


def test_rehearsal_transport_builds_and_exports_what_gather_binds():
    import ctypes as C
    L = C.CDLL(_rehearsal_transport())
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv",
                 "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(L, name), name


def test_the_product_library_has_no_transport_hook():
    """FLATE_HIP_TEST_TRANSPORT is honoured by the test build only (csrc/gather.hip, -DFLATE_HIP_TEST_BUILD):
    the library that ships must not even contain the variable's name."""
    build = importlib.import_module("moonbit-flate_amd.build")
    assert b"FLATE_HIP_TEST_TRANSPORT" not in open(build.build(), "rb").read()
    assert b"FLATE_HIP_TEST_TRANSPORT" in open(build.build_test(), "rb").read()


@pytest.mark.gpu
def test_two_ranks_through_the_c_abi_exchange_on_one_card():
    """tests/tools/native_gather_ranks.py with two processes: both exchange forms, the overlapped pair,
    a pad overflow and a stream-count overflow (E_AGAIN on both ranks, plans raised alike), an `out` that
    is too small on one rank (E_OUT_TOO_SMALL on both), every gathered stream checked against the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the hook that lets a transport replace RCCL exists in the TEST build of the library only
    test_lib = importlib.import_module("moonbit-flate_amd.build").build_test()
    env = dict(os.environ, FLATE_HIP_TEST_TRANSPORT=_rehearsal_transport(), FLATE_REHEARSAL_TIMEOUT_S="90",
               FLATE_HIP_LIB=test_lib)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "native_gather_ranks.py"), "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "native gather ok: 2 ranks" in out.stdout


@pytest.mark.gpu
def test_bench_with_two_ranks_takes_the_c_abi_exchange_by_default_on_one_card():
    """`bench.py --gpus 2` as the driver will run it on a multi-GPU node -- the exchange through the C ABI after the
    timed region, in its guarded helper thread -- rehearsed with two ranks on ONE card: gloo between the ranks,
    the library's test build with the rehearsal transport in place of RCCL (FLATE_BENCH_FORCE_NATIVE)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    test_lib = importlib.import_module("moonbit-flate_amd.build").build_test()
    env = dict(os.environ, FLATE_BENCH_BACKEND="gloo", FLATE_BENCH_FORCE_NATIVE="1", FLATE_HIP_LIB=test_lib,
               FLATE_HIP_TEST_TRANSPORT=_rehearsal_transport(), FLATE_REHEARSAL_TIMEOUT_S="90",
               FLATE_REHEARSAL_SLOT_MB="64")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--streams", "1536", "--cpu-sample-streams", "128"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    g = line["config"]["gather"]
    assert line["n_gpus"] == 2 and "c_abi_error" not in g, g
    assert g["c_abi"]["compared_streams"] >= 48 and g["c_abi"]["allgather_ms"]["mean"] > 0 and g["c_abi"]["sendrecv_ms"]["mean"] > 0


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """The same script over RCCL itself, one GPU per rank (skipped on the one-GPU boxes of this pool)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "FLATE_HIP_TEST_TRANSPORT", "FLATE_HIP_LIB"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "native_gather_ranks.py"), "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "native gather ok: 2 ranks" in out.stdout
