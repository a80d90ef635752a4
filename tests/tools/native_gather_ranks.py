"""Dev / evidence tool: the C-ABI exchange (flate_hip_gather_*) between N processes.
    python3 tests/tools/native_gather_ranks.py [N]        (spawns its own ranks, gloo for the unique id)
On a box with N GPUs every rank takes its own card.  On a one-GPU box all ranks share card 0, which
RCCL normally refuses ("Duplicate GPU detected"); the script then reports that and exits 3 -- it is
only a probe there.  Every rank compresses its own shard, both exchange forms and the overlapped pair
are run, and every rank checks the gathered buffer against all ranks' streams (regenerated from
their seeds and compressed by the oracle)."""
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
