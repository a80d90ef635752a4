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

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
kinds = ["text", "ramp", "zero", "rand", "low", "period", "runs"]
sizes = [0, 1, 2, 15, 16, 17, 18, 100, 127, 128, 129, 500, 4000, 30000, 65534, 65535, 65536, 65537, 65551,
         65552, 66000, 100000, 131070, 131071, 131072, 200000, 262144, 400000]
eng = flate.FlateEngine(0)
t0, rounds, streams, nbytes = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.integers(1, 90))
    specs = [(kinds[int(rng.integers(len(kinds)))], int(rng.choice(sizes)) if rng.random() < 0.7
              else int(rng.integers(0, 300000))) for _ in range(n)]
    seed = int(rng.integers(1 << 30))
    data, off = make_streams(specs, seed=seed)
    go = bool(rng.integers(2))
    guests = bool(rng.integers(2))
    eng.set_option("guest_min_streams", 1 if guests else 1 << 30)
    eng.set_option("guest_blocks", int(rng.choice([8, 64, 256])) if guests else 0)
    int(rng.choice([0, 0, 2, 8]))  # (a draw that once chose an option since removed: later draws stay what they were)
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
    if rounds % 7 == 3:  # a uniform batch on a fixed launch geometry (one persistent launch, few blocks per stream)
        eng.set_option("guest_min_streams", 1)
        eng.set_option("guest_blocks", 64)
        un = int(rng.integers(64, 400))
        if rng.random() < 0.5:
            eng.set_option("resident_blocks", int(rng.choice([1, 4, 8, 16])) * 16)
        else:
            eng.set_option("resident_blocks", int(rng.integers(1, un // 4 + 1)))
        ulen = int(rng.choice([128, 5000, 65535, 65536, 70000, 140000]))
        ud = flate.synth("text", un, ulen, first_stream=int(rng.integers(1 << 20)))
        uo = flate.uniform_offsets(un, ulen)
        out, ooff = eng.deflate_batch(ud, uo, compat_go=go)
        ref, roff, rlen = O.deflate_batch(ud, uo, compat=cm, nthreads=8)
        for i in range(un):
            assert bytes(out[int(ooff[i]):int(ooff[i + 1])]) == bytes(ref[int(roff[i]):int(roff[i]) + int(rlen[i])]), \
                "uniform batch: deflate stream %d differs (%s)" % (i, tag)
        streams += un
        nbytes += un * ulen
        eng.set_option("resident_blocks", 1024)
    if rounds % 4 == 2:  # the containers' checksums of the round's streams, and a framed round trip
        import zlib as _z
        kind = ("adler32", "crc32")[(rounds >> 2) & 1]
        got = eng.checksum_batch(data, off, kind)
        for i in range(n):
            d_i = data[int(off[i]):int(off[i + 1])].tobytes()
            assert int(got[i]) == (_z.adler32 if kind == "adler32" else _z.crc32)(d_i), "checksum %s of stream %d (%s)" % (kind, i, tag)
        if rounds % 8 == 2:
            wrap = "zlib" if kind == "adler32" else "gzip"
            fr, fo = eng.deflate_batch_framed(data, off, wrap, compat_go=go)
            j = int(rng.integers(n))
            m = fr[int(fo[j]):int(fo[j + 1])].tobytes()
            want_j = data[int(off[j]):int(off[j + 1])].tobytes()
            import gzip as _g
            assert (_z.decompress(m) if wrap == "zlib" else _g.decompress(m)) == want_j, "framed member %d (%s)" % (j, tag)
            bo, boff, bl, bst = eng.inflate_batch_framed(fr, fo, wrap, out_sizes=szs)
            assert (bst == 0).all() and bytes(bo[:int(off[-1])]) == data[:int(off[-1])].tobytes(), "framed round trip (%s)" % tag
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
    rounds += 1
    streams += n
    nbytes += int(off[-1])
    if rounds % 10 == 0:
        print("round %d: %d streams, %.1f MB ok (%.0f s)" % (rounds, streams, nbytes / 1e6, time.time() - t0), flush=True)
print("SOAK OK: %d rounds, %d streams, %.1f MB, %.0f s" % (rounds, streams, nbytes / 1e6, time.time() - t0))
