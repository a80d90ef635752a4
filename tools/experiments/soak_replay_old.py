# Replay of rounds FROM..UPTO of the soak AS IT WAS on commit 36fd1df (with the overlapped entropy stage's options),
# REPS times from the saved random state, against a library built from that commit (FLATE_HIP_LIB=build/exp/libold.so):
# the hunt for the one red soak run of round 4 (profiles/r04/README.md section 7).  300 replays of rounds 219-220: green.
#   FLATE_HIP_LIB=build/exp/libold.so python3 tools/experiments/soak_replay_old.py [from] [upto] [reps]
"""Dev tool: randomized parity soak on the GPU box -- every path against the oracle.
    python tests/tools/soak.py [seconds] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
flate = importlib.import_module("moonbit-flate_amd")
from oracle import pyoracle as O
from util import make_streams

FROM = int(sys.argv[1]) if len(sys.argv) > 1 else 219   # first round executed for real
UPTO = int(sys.argv[2]) if len(sys.argv) > 2 else 220   # last one
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 100
budget = 1e9
rng = np.random.default_rng(1)
kinds = ["text", "ramp", "zero", "rand", "low", "period", "runs"]
sizes = [0, 1, 2, 15, 16, 17, 18, 100, 127, 128, 129, 500, 4000, 30000, 65534, 65535, 65536, 65537, 65551,
         65552, 66000, 100000, 131070, 131071, 131072, 200000, 262144, 400000]
eng = flate.FlateEngine(0)
t0, rounds, streams, nbytes = time.time(), 0, 0, 0
import copy
def body(rounds):
    global streams, nbytes
    n = int(rng.integers(1, 90))
    specs = [(kinds[int(rng.integers(len(kinds)))], int(rng.choice(sizes)) if rng.random() < 0.7
              else int(rng.integers(0, 300000))) for _ in range(n)]
    seed = int(rng.integers(1 << 30))
    data, off = make_streams(specs, seed=seed)
    go = bool(rng.integers(2))
    guests = bool(rng.integers(2))
    eng.set_option("guest_min_streams", 1 if guests else 1 << 30)
    eng.set_option("guest_blocks", int(rng.choice([8, 64, 256])) if guests else 0)
    eng.set_option("overlap_sub_batches", int(rng.choice([0, 0, 2, 8])))  # (only takes effect on uniform batches)
    eng.set_option("entropy_per_block", int(rng.choice([-1, 0, 1, 1])))   # (1: one wavefront per block where possible)
    cm = O.COMPAT_GO if go else O.COMPAT_MOONBIT
    if rng.random() < 0.3:  # some rounds without empty streams, so that the per-block entropy kernels really run
        specs = [(k, max(sz, 1)) for k, sz in specs]
        data, off = make_streams(specs, seed=seed)
    tag = "seed=%d go=%s guests=%s n=%d" % (seed, go, guests, n)
    out, ooff = eng.deflate_batch(data, off, compat_go=go)
    ref, roff, rlen = O.deflate_batch(data, off, compat=cm, nthreads=8)
    for i in range(n):
        a = bytes(out[int(ooff[i]):int(ooff[i + 1])])
        b = bytes(ref[int(roff[i]):int(roff[i]) + int(rlen[i])])
        assert a == b, "deflate stream %d differs (%s, spec %s)" % (i, tag, specs[i])
    one, nb, bit_off = eng.deflate_spliced(data, off, compat_go=go)
    rs, rbo = O.deflate_spliced(data, off, cm)
    assert bytes(one[:nb]) == rs and np.array_equal(bit_off, rbo), "spliced differs (%s)" % tag
    szs = [s for _, s in specs]
    comp = np.concatenate([out[:int(ooff[-1])], np.zeros(8, np.uint8)])
    for simt_min, spec in ((0, 0), (1 << 30, 0), (1 << 30, 2)):  # lane per stream, wave per stream, speculative wave
        eng.set_option("inflate_simt_min_streams", simt_min)
        eng.set_option("inflate_spec", spec)
        eng.set_option("inflate_spec_shape", int(rng.choice([0, 1, 2])))
        eng.set_option("inflate_lanes", int(rng.choice([0, 16, 32, 64])))
        back, _, olen, status, _ = eng.inflate_batch(comp, ooff, szs)
        assert (status == 0).all() and list(olen) == szs, "inflate status (%s)" % tag
        assert bytes(back[:int(off[-1])]) == data[:int(off[-1])].tobytes(), "inflate bytes (%s)" % tag
    eng.set_option("inflate_spec", 1)
    back, _, olen, status, _ = eng.inflate_spliced(np.concatenate([one[:nb], np.zeros(8, np.uint8)]), nb, bit_off, szs)
    assert (status == 0).all() and bytes(back[:int(off[-1])]) == data[:int(off[-1])].tobytes(), "inflate_spliced (%s)" % tag
    if rounds % 7 == 3:  # a uniform batch: the overlapped entropy stage is eligible
        eng.set_option("guest_min_streams", 1)
        eng.set_option("guest_blocks", 64)
        un = int(rng.integers(64, 400))
        if rng.random() < 0.5:
            eng.set_option("overlap_sub_batches", int(rng.choice([1, 4, 8, 16])))
        else:  # the uneven form: one large first part, gated on the blocks' single counts
            eng.set_option("overlap_sub_batches", 0)
            eng.set_option("overlap_tail_streams", int(rng.integers(1, un // 4 + 1)))
        ulen = int(rng.choice([128, 5000, 65535, 65536, 70000, 140000]))
        ud = flate.synth("text", un, ulen, first_stream=int(rng.integers(1 << 20)))
        uo = flate.uniform_offsets(un, ulen)
        out, ooff = eng.deflate_batch(ud, uo, compat_go=go)
        ref, roff, rlen = O.deflate_batch(ud, uo, compat=cm, nthreads=8)
        for i in range(un):
            assert bytes(out[int(ooff[i]):int(ooff[i + 1])]) == bytes(ref[int(roff[i]):int(roff[i]) + int(rlen[i])]), \
                "overlapped deflate stream %d differs (%s)" % (i, tag)
        streams += un
        nbytes += un * ulen
        eng.set_option("overlap_tail_streams", 0)
    if rounds % 5 == 1:  # one of the round's streams through the piecewise decoder (flate_hip_inflate_stream_*)
        j = int(rng.integers(n))
        cj = out[int(ooff[j]):int(ooff[j + 1])] if rounds % 7 != 3 else None
        if cj is not None:
            r = eng.open_inflate_stream()
            got, pos, rc = [], 0, 0
            piece, room = int(rng.integers(700, 70000)), int(rng.integers(1, 90000))
            for _ in range(100000):
                take = max(0, piece - r.pending_input)
                chunk = cj[pos:pos + take]
                pos += chunk.size
                o, rc = r.feed(chunk, final=pos >= cj.size, room=room)
                got.append(o)
                if rc != 0:
                    break
            r.free()
            want_j = data[int(off[j]):int(off[j + 1])].tobytes()
            assert rc == 1 and b"".join(x.tobytes() for x in got) == want_j, "piecewise inflate of stream %d (%s)" % (j, tag)
    return

def skip(rounds):
    """consume the random draws of a round without any GPU work (must mirror body())"""
    n = int(rng.integers(1, 90))
    for _ in range(n):
        int(rng.integers(len(kinds)))
        if rng.random() < 0.7:
            int(rng.choice(sizes))
        else:
            int(rng.integers(0, 300000))
    int(rng.integers(1 << 30)); bool(rng.integers(2))
    guests = bool(rng.integers(2))
    if guests:
        int(rng.choice([8, 64, 256]))
    int(rng.choice([0, 0, 2, 8])); int(rng.choice([-1, 0, 1, 1])); rng.random()
    for _ in range(3):
        int(rng.choice([0, 1, 2])); int(rng.choice([0, 16, 32, 64]))
    if rounds % 7 == 3:
        un = int(rng.integers(64, 400))
        if rng.random() < 0.5:
            int(rng.choice([1, 4, 8, 16]))
        else:
            int(rng.integers(1, un // 4 + 1))
        int(rng.choice([128, 5000, 65535, 65536, 70000, 140000])); int(rng.integers(1 << 20))
    if rounds % 5 == 1:
        int(rng.integers(n))
        if rounds % 7 != 3:
            int(rng.integers(700, 70000)); int(rng.integers(1, 90000))

for r in range(FROM):
    skip(r)
state = copy.deepcopy(rng.bit_generator.state)
fails = 0
for rep in range(REPS):
    rng.bit_generator.state = copy.deepcopy(state)
    for r in range(FROM, UPTO + 1):
        try:
            body(r)
        except Exception as e:  # noqa
            fails += 1
            print("FAIL rep %d round %d: %s" % (rep, r, str(e)[:200]), flush=True)
            break
    if rep % 20 == 19:
        print("rep %d: %d failures, %.0f s" % (rep + 1, fails, time.time() - t0), flush=True)
print("DONE %d reps of rounds %d..%d: %d failures" % (REPS, FROM, UPTO, fails))

