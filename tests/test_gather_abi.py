"""The exchange step behind the C ABI (include/flate_hip.h: flate_hip_gather_*, flate_hip_comm_*).

CPU: the layout arithmetic (flate_hip_gather_layout) against the rules of shard.py's torch form.
GPU (one card): a one-rank RCCL communicator -- the index arithmetic, both exchange forms, the
overlapped begin/end pair, the overflow of the sticky pad.  More ranks need more GPUs; the
multi-rank control flow is rehearsed with gloo in test_distributed_cpu.py."""
import importlib

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
        # more streams than the plan allows is refused before anything is issued
        with pytest.raises(flate.FlateError):
            comm.begin(comp, np.zeros(n + 50, np.uint64), out2)
    finally:
        if comm is not None:
            comm.close()
        eng.close()
