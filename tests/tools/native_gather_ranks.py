"""The C-ABI exchange (flate_hip_gather_*) between N processes.
    python3 tests/tools/native_gather_ranks.py [N]        (spawns its own ranks, gloo for the unique id)
On a box with N GPUs every rank takes its own card.  On a one-GPU box all ranks share card 0, which
RCCL refuses ("Duplicate GPU detected": exit 3) -- unless FLATE_HIP_TEST_TRANSPORT names the tests'
rehearsal transport (tests/rehearsal_transport/: the same entry points over host shared memory), which
is how tests/test_gather_abi.py runs the multi-rank branches of csrc/gather.hip on one card.
Every rank compresses its own shard (the shards differ in size and stream count); both exchange forms,
the overlapped pair, a shard that outgrows the pad, a rank with more streams than the plan allows
(FLATE_HIP_E_AGAIN on every rank, then the blocking form delivers) and an `out` that is too small on
ONE rank (FLATE_HIP_E_OUT_TOO_SMALL on every rank, nobody left in a collective) are run, and every
rank checks every gathered buffer against all ranks' streams (regenerated from their seeds and
compressed by the oracle)."""
import importlib
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    import numpy as np
    import torch
    import torch.distributed as dist
    flate = importlib.import_module("moonbit-flate_amd")
    shard = importlib.import_module("moonbit-flate_amd.shard")
    from oracle import pyoracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    eng = flate.FlateEngine(dev)
    n, blen = 64 + 8 * rank, 50000
    data = flate.synth("text", n, blen, first_stream=1000 * rank)
    off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(torch.from_numpy(data).cuda(), off)
    try:
        comm = shard.NativeComm(eng, rank, world, dist)
    except flate.FlateError as e:
        print("rank %d: communicator refused: %s" % (rank, e))
        sys.exit(3)
    counts = [64 + 8 * r for r in range(world)]
    total = sum(counts)

    def check(g):
        assert g.off.size == total, (g.off.size, total)
        j = 0
        for r in range(world):
            for i in range(0, counts[r], 13):
                src = flate.synth("text", 1, blen, first_stream=1000 * r + i)
                assert bytes(g.stream(j + i).cpu().numpy()) == pyoracle.deflate(src), (r, i)
            j += counts[r]

    for mode in ("allgather", "sendrecv"):
        check(comm.gather(comp, coff, mode=mode))
    pad, kmax = comm.plan()
    out = torch.zeros(world * pad, dtype=torch.uint8, device="cuda")
    comm.begin(comp, coff, out)
    eng.deflate_batch(torch.from_numpy(data).cuda(), off)  # the next batch, beside the exchange
    g = comm.end(total)
    assert g is not None
    check(g)
    # rank_base placement of the exact-size form: shards back to back in rank order
    g = comm.gather(comp, coff, mode="sendrecv")
    sizes = [None] * world
    dist.all_gather_object(sizes, int(coff[-1]))
    assert int(g.off[0]) == 0 and [int(g.off[sum(counts[:r])]) for r in range(world)] == [sum(sizes[:r]) for r in range(world)]

    # -- an `out` that is too small on ONE rank: every rank gets E_OUT_TOO_SMALL, none hangs --------
    for mode in ("allgather", "sendrecv"):
        try:
            comm.gather(comp, coff, mode=mode, claim_out_cap=(1 << 20) if rank == world - 1 else None)
            raise AssertionError("rank %d: the exchange went through" % rank)
        except flate.FlateError as e:
            assert e.code == -2, e
        check(comm.gather(comp, coff, mode=mode))  # and the communicator still works

    # -- a shard that outgrows the agreed pad, on the LAST rank only: E_AGAIN everywhere ------------
    pad0, kmax0 = comm.plan()
    if rank == world - 1:
        nb = (pad0 >> 16) + 24
        big = flate.synth("rand", nb, 65536, first_stream=77)
        bcomp, bcoff = eng.deflate_batch(torch.from_numpy(big).cuda(), flate.uniform_offsets(nb, 65536))
        assert int(bcoff[-1]) > pad0
    else:
        bcomp, bcoff = comp, coff
    out = torch.zeros(world * pad0, dtype=torch.uint8, device="cuda")
    comm.begin(bcomp, bcoff, out)
    assert comm.end(4 * total) is None, "rank %d did not see the overflow" % rank
    pad1, kmax1 = comm.plan()
    pads = [None] * world
    dist.all_gather_object(pads, (pad1, kmax1))
    assert pad1 > pad0 and all(p == pads[0] for p in pads), pads  # raised, and alike on every rank
    g = comm.gather(bcomp, bcoff)
    mine = [None] * world
    dist.all_gather_object(mine, (int(bcoff[-1]), int(bcoff.size - 1)))
    j = 0
    for r in range(world):  # rank r's shard sits at r * pad; its own bytes are checked by that rank
        if r == rank:
            assert torch.equal(g.buf[r * g.pad:r * g.pad + mine[r][0]], bcomp[:mine[r][0]])
            assert int(g.off[j]) == r * g.pad
        j += mine[r][1]
    assert g.off.size == j

    # -- more streams than the plan allows, on rank 0 only: E_AGAIN everywhere, kmax raised alike --
    pad2, kmax2 = comm.plan()
    if rank == 0:
        nk = kmax2 + 37
        small = flate.synth("text", nk, 300, first_stream=5)
        kcomp, kcoff = eng.deflate_batch(torch.from_numpy(small).cuda(), flate.uniform_offsets(nk, 300))
    else:
        kcomp, kcoff = comp, coff
    out = torch.zeros(world * pad2, dtype=torch.uint8, device="cuda")
    comm.begin(kcomp, kcoff, out)
    assert comm.end(8 * total + kmax2) is None, "rank %d did not see the stream-count overflow" % rank
    plans = [None] * world
    dist.all_gather_object(plans, comm.plan())
    assert comm.plan()[1] == kmax2 + 37 and all(p == plans[0] for p in plans), plans
    comm.begin(kcomp, kcoff, out)  # the raised plan holds it now
    g = comm.end(8 * total + kmax2)
    assert g is not None and g.off.size == total - counts[0] + kmax2 + 37
    if rank == 0:
        assert bytes(g.stream(3).cpu().numpy()) == pyoracle.deflate(flate.synth("text", 1, 300, first_stream=8))
    comm.close()
    eng.close()
    dist.barrier()
    if rank == 0:
        print("native gather ok: %d ranks, %d streams, pad %d" % (world, total, pad))


if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:
        main()
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        rc = 0
        for p in procs:
            try:
                p.wait(timeout=240)
            except subprocess.TimeoutExpired:
                p.kill()
                rc = rc or 4
            rc = rc or p.returncode
        sys.exit(rc)
