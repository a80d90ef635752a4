"""GPU parity tests of the batch inflater (inflate.mbt / dict-decoder.mbt) against the oracle."""
import zlib

import numpy as np
import pytest

from util import flate, make_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["wave_per_stream", "lane_per_stream", "lane_per_stream_64_row8",
                                        "lane_per_stream_64_row16", "lane_per_stream_64_norow",
                                        "speculative_wave_small_batch", "speculative_wave_large_batch"])
def eng(request):
    """All three inflater kernels (both builds of the third; the lane-per-stream one by batch size -- small test
    batches take its 16-lane form -- and in its 64-lane form with an output row of 8 and 16 dwords and without
    one) must pass every test: the options force one of them."""
    flate.build()
    e = flate.FlateEngine(0)
    e.set_option("inflate_simt_min_streams", 0 if request.param.startswith("lane_per_stream") else 1 << 30)
    e.set_option("inflate_spec", 2 if request.param.startswith("speculative_wave") else 0)
    e.set_option("inflate_spec_shape", 1 if request.param.endswith("small_batch") else 2)
    if request.param.startswith("lane_per_stream_64"):
        e.set_option("inflate_lanes", 64)
        e.set_option("inflate_row_dwords", {"row8": 8, "row16": 16, "norow": 0}[request.param.rsplit("_", 1)[1]])
    yield e
    e.close()


def _pack(blobs):
    lens = np.array([len(b) for b in blobs], dtype=np.uint64)
    off = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    data = np.frombuffer(b"".join(blobs) + b"\0" * 8, dtype=np.uint8).copy()
    return data, off


def test_inflate_round_trip_of_gpu_and_oracle_streams(eng, oracle):
    specs = [("text", 65536), ("ramp", 65536), ("zero", 70000), ("rand", 65536), ("low", 131072),
             ("period", 200000), ("runs", 65535), ("text", 0), ("text", 1), ("text", 16), ("text", 17),
             ("text", 127), ("text", 128), ("text", 300), ("text", 262144)]
    data, off = make_streams(specs)
    comp, coff = eng.deflate_batch(data, off)
    sizes = (off[1:] - off[:-1])
    out, ooff, olen, status, err = eng.inflate_batch(comp, coff, sizes)
    assert (status == 0).all() and (olen == sizes).all()
    assert np.array_equal(out[:int(ooff[-1])], data[:int(off[-1])])
    # identical behaviour to the oracle's decoder on the same bytes
    for i, (kind, n) in enumerate(specs):
        c = bytes(comp[int(coff[i]):int(coff[i + 1])])
        assert oracle.inflate(c, n) == bytes(data[int(off[i]):int(off[i + 1])])


def test_inflate_accepts_foreign_encoders(eng, oracle):
    # stored / fixed / dynamic blocks, long codes, long distances, from zlib at several levels
    rng = np.random.default_rng(5)
    srcs = [b"the quick brown fox jumps over the lazy dog " * 2000, b"abc", b"",
            bytes(rng.integers(0, 8, 120000, dtype=np.uint8)),
            bytes(rng.integers(0, 256, 70000, dtype=np.uint8)),
            bytes(flate.synth("text", 1, 300000)),
            bytes(np.repeat(rng.integers(0, 256, 3000, dtype=np.uint8), rng.integers(1, 300, 3000)))]
    blobs, want = [], []
    for s in srcs:
        for level in (0, 1, 6, 9):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            blobs.append(co.compress(s) + co.flush())
            want.append(s)
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
        blobs.append(co.compress(s) + co.flush())
        want.append(s)
    data, off = _pack(blobs)
    sizes = [len(w) for w in want]
    out, ooff, olen, status, err = eng.inflate_batch(data, off, sizes)
    assert (status == 0).all()
    for i, w in enumerate(want):
        assert bytes(out[int(ooff[i]):int(ooff[i]) + int(olen[i])]) == w, i


def test_inflate_errors_match_oracle(eng, oracle):
    good = oracle.deflate(bytes(flate.synth("text", 1, 5000)))
    rng = np.random.default_rng(9)
    blobs = [bytes([0x07]), b"", good[:-3], good[:100], bytes([1, 5, 0, 0, 0]), bytes([0x03, 0x02, 0x00]),
             bytes([0x05, 0xff, 0xff, 0xff]), bytes([0x04, 0x00])]
    for _ in range(40):  # random corruptions of a valid stream
        g = bytearray(good)
        for k in rng.integers(0, len(g), int(rng.integers(1, 4))):
            g[int(k)] ^= int(rng.integers(1, 256))
        blobs.append(bytes(g))
    for _ in range(20):
        blobs.append(bytes(rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8)))
    data, off = _pack(blobs)
    cap = 20000
    out, ooff, olen, status, err = eng.inflate_batch(data, off, [cap] * len(blobs), check=False)
    for i, bl in enumerate(blobs):
        rc, res, used, eoff = oracle.inflate(bl, cap, full=True)
        want_status = {0: 0, oracle.E_CORRUPT: -4, oracle.E_UNEXPECTED_EOF: -7,
                       oracle.E_OUT_TOO_SMALL: -2}[rc]
        assert int(status[i]) == want_status, (i, bl[:8].hex(), int(status[i]), rc)
        assert int(err[i]) == eoff, (i, int(err[i]), eoff)
        if rc == 0:
            assert bytes(out[int(ooff[i]):int(ooff[i]) + int(olen[i])]) == res


def test_inflate_errors_in_long_streams_match_oracle(eng, oracle):
    """Corruptions anywhere in multi-block streams of a few hundred KB, and output slots that end
    anywhere inside them: status, error offset and the bytes produced up to a corruption's block
    are the oracle's (the sub-block decoder is many batches into the stream when it meets them)."""
    rng = np.random.default_rng(77)
    blobs, caps = [], []
    for kind, n in (("text", 200000), ("low", 150000), ("period", 100000)):
        one, _ = make_streams([(kind, n)], seed=n)
        raw = bytes(one[:n])
        good = oracle.deflate(raw)
        for _ in range(12):
            g = bytearray(good)
            k = int(rng.integers(len(g) // 8, len(g)))
            g[k] ^= int(rng.integers(1, 256))
            blobs.append(bytes(g))
            caps.append(n + 1000)
        for _ in range(6):  # truncated input
            blobs.append(good[:int(rng.integers(len(good) // 4, len(good) - 1))])
            caps.append(n + 1000)
        for _ in range(6):  # output slot too small
            blobs.append(good)
            caps.append(int(rng.integers(1, n)))
    data, off = _pack(blobs)
    out, ooff, olen, status, err = eng.inflate_batch(data, off, caps, check=False)
    for i, bl in enumerate(blobs):
        rc, res, used, eoff = oracle.inflate(bl, caps[i], full=True)
        want_status = {0: 0, oracle.E_CORRUPT: -4, oracle.E_UNEXPECTED_EOF: -7,
                       oracle.E_OUT_TOO_SMALL: -2}[rc]
        assert int(status[i]) == want_status, (i, int(status[i]), rc)
        assert int(err[i]) == eoff, (i, int(err[i]), eoff)
        if rc == 0:
            assert bytes(out[int(ooff[i]):int(ooff[i]) + int(olen[i])]) == res


def test_inflate_token_density_swings(eng, oracle):
    """Streams whose token length changes abruptly -- text, then a run of zeros (258-byte matches of
    two or three bits), then a four-symbol alphabet, then random bytes, and back: the sub-block
    decoder's lists fill up and its sub-block size has to follow; encoders: ours, the oracle's, zlib."""
    rng = np.random.default_rng(2024)
    def piece(kind, n):
        one, _ = make_streams([(kind, n)], seed=int(rng.integers(1 << 30)))
        return bytes(one[:n])
    plains = []
    for _ in range(6):
        parts = [piece(str(rng.choice(["text", "zero", "low", "rand", "runs", "ramp", "period"])),
                       int(rng.choice([300, 3000, 20000, 70000]))) for _ in range(int(rng.integers(3, 9)))]
        plains.append(b"".join(parts))
    blobs, want = [], []
    for p in plains:
        blobs.append(oracle.deflate(p))
        want.append(p)
        for level in (1, 9):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            blobs.append(co.compress(p) + co.flush())
            want.append(p)
    data, off = _pack(blobs)
    out, ooff, olen, status, _ = eng.inflate_batch(data, off, [len(w) for w in want])
    assert (status == 0).all() and list(olen) == [len(w) for w in want]
    for i, w in enumerate(want):
        assert bytes(out[int(ooff[i]):int(ooff[i]) + len(w)]) == w, i
    # and through our own encoder (multi-window streams, both compat modes)
    raw = np.frombuffer(b"".join(plains), dtype=np.uint8).copy()
    roff = np.zeros(len(plains) + 1, np.uint64)
    np.cumsum([len(p) for p in plains], out=roff[1:])
    for go in (False, True):
        comp, coff = eng.deflate_batch(raw, roff, compat_go=go)
        back, boff, blen, st, _ = eng.inflate_batch(comp, coff, [len(p) for p in plains])
        assert (st == 0).all() and bytes(back[:int(boff[-1])]) == raw.tobytes()


def test_inflate_output_too_small(eng, oracle):
    good = oracle.deflate(bytes(flate.synth("text", 1, 5000)))
    data, off = _pack([good])
    out, ooff, olen, status, err = eng.inflate_batch(data, off, [100], check=False)
    assert int(status[0]) == -2


def test_inflate_device_pointers_full_size(eng, oracle):
    import torch
    n = 512
    host = flate.synth("text", n, 65536)
    off = flate.uniform_offsets(n, 65536)
    d = torch.from_numpy(host).cuda()
    comp, coff = eng.deflate_batch(d, off)
    out, ooff, olen, status, err = eng.inflate_batch(comp, coff, [65536] * n)
    assert (status == 0).all() and (olen == 65536).all()
    assert torch.equal(out[:n * 65536], d)


def test_inflate_fuzz_foreign_block_mixes(eng, oracle):
    """Streams from zlib at several levels/strategies (stored, fixed and dynamic blocks mixed, long
    matches at distance 1, sync-flush stored blocks in the middle), sizes from 0 to 300 KB."""
    rng = np.random.default_rng(1234)
    blobs, plains = [], []
    for k in range(160):
        n = int(rng.choice([0, 1, 2, 5, 17, 100, 1000, 5000, 40000, 70000, 300000]))
        kind = k % 5
        if kind == 0:
            raw = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            raw = bytes(n)
        elif kind == 2:
            raw = (rng.integers(0, 4, n, dtype=np.uint8) + 97).tobytes()
        elif kind == 3:
            raw = (b"the quick brown fox jumps over the lazy dog " * (n // 44 + 1))[:n]
        else:
            raw = rng.integers(0, 256, max(n // 50, 1), dtype=np.uint8).tobytes() * 50
            raw = raw[:n]
        level = int(rng.choice([0, 1, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        half = len(raw) // 2
        blob = c.compress(raw[:half]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(raw[half:]) + c.flush()
        blobs.append(blob)
        plains.append(raw)
    data, off = _pack(blobs)
    sizes = [len(p) for p in plains]
    out, ooff, olen, status, _ = eng.inflate_batch(data, off, sizes)
    assert (status == 0).all() and list(olen) == sizes
    for i, p in enumerate(plains):
        assert bytes(out[int(ooff[i]):int(ooff[i]) + len(p)]) == p, i


def test_host_pointer_pipeline_matches_single_pass(oracle):
    # host-pointer calls on large batches are pipelined over groups of streams (option
    # host_pipeline_groups): outputs, lengths, statuses and error offsets equal those of one pass,
    # also when streams of different groups are corrupt or cut short
    n, blen = 49152 + 11, 2048
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    e = flate.FlateEngine(0)
    try:
        comp, coff = e.deflate_batch(host, off)
        comp = np.array(np.asarray(comp)[:int(coff[-1]) + 8], copy=True)
        bad = [5, 20000, 40000, n - 1]
        for k, i in enumerate(bad):  # corrupt block type / damaged code data
            a, b = int(coff[i]), int(coff[i + 1])
            if k % 2 == 0:
                comp[a] |= 0x06
            else:
                comp[a + (b - a) // 2: b] = 0xff
        sizes = [blen] * n
        e.set_option("host_pipeline_groups", 0)
        ref = e.inflate_batch(comp, coff, sizes, check=False)
        e.set_option("host_pipeline_groups", 3)
        got = e.inflate_batch(comp, coff, sizes, check=False)
        assert (ref[3] != 0).sum() >= 2 and (ref[3][[i for i in range(n) if i not in bad]] == 0).all()
        for r, g in zip(ref[1:], got[1:]):  # out_off, out_len, status, err_off
            assert np.array_equal(np.asarray(r), np.asarray(g))
        ok = np.asarray(ref[3]) == 0
        o_ref = np.asarray(ref[0])[:n * blen].reshape(n, blen)
        o_got = np.asarray(got[0])[:n * blen].reshape(n, blen)
        assert np.array_equal(o_ref[ok], o_got[ok])
        assert np.array_equal(o_got[ok], np.asarray(host)[:n * blen].reshape(n, blen)[ok])
        for i in bad:  # and the oracle's decoder agrees on the failing streams
            rc, _, _, eoff = oracle.inflate(bytes(comp[int(coff[i]):int(coff[i + 1])]), blen, full=True)
            want = {0: 0, oracle.E_CORRUPT: -4, oracle.E_UNEXPECTED_EOF: -7, oracle.E_OUT_TOO_SMALL: -2}[rc]
            assert want == int(got[3][i]) and eoff == int(got[4][i]), (i, rc, eoff, int(got[3][i]), int(got[4][i]))
    finally:
        e.close()


def test_lane_per_stream_decoder_in_several_rounds(oracle):
    # more streams than the chip holds lanes (8 wavefronts x 64 lanes per CU): the lane-per-stream
    # decoder runs in equal rounds, each launch starting at its own first stream
    rng = np.random.default_rng(23)
    kinds = [b"", b"a", b"abcabcabcabc" * 9, bytes(rng.integers(0, 256, 90, dtype=np.uint8)), b"\0" * 300,
             bytes(rng.integers(97, 101, 200, dtype=np.uint8))]
    comp_kinds = [oracle.deflate(np.frombuffer(k, np.uint8)) for k in kinds]
    n = 8 * 256 * 64 + 5000
    pick = rng.integers(0, len(kinds), n)
    blobs = [comp_kinds[int(k)] for k in pick]
    data, off = _pack(blobs)
    sizes = np.array([len(kinds[int(k)]) for k in pick], dtype=np.uint64)
    e = flate.FlateEngine(0)
    try:
        e.set_option("inflate_simt_min_streams", 0)
        e.set_option("inflate_spec", 0)
        e.set_option("inflate_lanes", 64)
        out, ooff, olen, status, _ = e.inflate_batch(data, off, sizes)
        assert (status == 0).all() and (olen == sizes).all()
        want = b"".join(kinds[int(k)] for k in pick)
        assert bytes(out[:int(ooff[-1])]) == want
    finally:
        e.close()


def test_literal_only_blocks_errors_and_long_codes(eng, oracle):
    """Blocks without length codes (the encoder's Huffman-only blocks; zlib's Z_HUFFMAN_ONLY) take their own
    path in the sub-block decoder (pointer jumping over the code lengths): random bytes with 7/8/9-bit
    codes, skewed alphabets whose rare bytes have codes longer than the 9-bit table, a one-symbol
    alphabet (1-bit codes: the shortest window), several blocks per stream, and corruptions, truncated
    inputs and output slots that end anywhere -- statuses, error offsets and bytes are the oracle's."""
    rng = np.random.default_rng(404)

    def huff_only(raw, level=6):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        return co.compress(raw) + co.flush()

    raws = [bytes(rng.integers(0, 256, 70000, dtype=np.uint8)),
            bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),                      # several blocks
            bytes(np.minimum(rng.geometric(0.08, 90000) - 1, 255).astype(np.uint8)),  # codes of 2 .. 15 bits
            bytes(np.minimum(rng.geometric(0.5, 50000) - 1, 255).astype(np.uint8)),
            b"a" * 30000, bytes(rng.integers(0, 2, 40000, dtype=np.uint8)),
            bytes(rng.integers(0, 256, 100, dtype=np.uint8)), bytes(rng.integers(0, 256, 17, dtype=np.uint8))]
    blobs, caps = [], []
    for raw in raws:
        for good in (huff_only(raw), oracle.deflate(np.frombuffer(raw, np.uint8))):
            blobs.append(good)
            caps.append(len(raw))
            blobs.append(good)
            caps.append(len(raw) + 77)
            for _ in range(3):  # a flipped byte
                g = bytearray(good)
                k = int(rng.integers(0, len(g)))
                g[k] ^= int(rng.integers(1, 256))
                blobs.append(bytes(g))
                caps.append(len(raw) + 1000)
            for _ in range(2):  # truncated input, output slot too small
                blobs.append(good[:int(rng.integers(1, len(good)))])
                caps.append(len(raw) + 1000)
                blobs.append(good)
                caps.append(int(rng.integers(0, len(raw))))
    data, off = _pack(blobs)
    out, ooff, olen, status, err = eng.inflate_batch(data, off, caps, check=False)
    for i, bl in enumerate(blobs):
        rc, res, used, eoff = oracle.inflate(bl, caps[i], full=True)
        want_status = {0: 0, oracle.E_CORRUPT: -4, oracle.E_UNEXPECTED_EOF: -7, oracle.E_OUT_TOO_SMALL: -2}[rc]
        assert int(status[i]) == want_status, (i, int(status[i]), rc)
        assert int(err[i]) == eoff, (i, int(err[i]), eoff)
        if rc == 0:
            assert bytes(out[int(ooff[i]):int(ooff[i]) + int(olen[i])]) == res
    # sizes without output (FLATE_HIP_SIZE_ONLY) take the same path
    good = [b for b, c in zip(blobs, caps)][0::10]
    d2, o2 = _pack(good)
    olen2, st2, _ = eng.inflate_sizes(d2, o2)
    for i, bl in enumerate(good):
        rc, res, _, _ = oracle.inflate(bl, 1 << 20, full=True)
        assert (int(st2[i]) == 0) == (rc == 0)
        if rc == 0:
            assert int(olen2[i]) == len(res)
