"""One long stream written in pieces (flate_hip_stream_*) and the size-only inflate pass
(FLATE_HIP_SIZE_ONLY): the streaming behaviour of the reference's Writer / Reader for long single
streams (deflate.mbt:280-294; inflate.mbt:382-407), checked against the oracle."""
import numpy as np
import pytest

from util import flate, make_streams, raw_inflate

pytestmark = pytest.mark.gpu
W = 65535


@pytest.fixture(scope="module")
def eng():
    e = flate.FlateEngine(0)
    yield e
    e.close()


def _pieces(eng, data, cuts, compat_go=False):
    """write() the stream in the given whole-window pieces, close() with the rest."""
    w = eng.open_stream(compat_go=compat_go)
    out, pos, emitted = [], 0, []
    for k in cuts:
        b = w.write(data[pos:pos + k * W])
        emitted.append(len(b))
        out.append(b)
        pos += k * W
    out.append(w.close(data[pos:]))
    return np.concatenate(out).tobytes(), emitted


@pytest.mark.parametrize("kind", ["text", "low", "runs", "period", "rand", "zero", "ramp"])
def test_pieces_equal_the_one_shot_stream(eng, oracle, kind):
    tails = [0, 1, 16, 17, 100, 127, 128, 5000, W - 1]
    for t_i, tail in enumerate(tails):
        nwin = 1 + t_i % 4
        data, _ = make_streams([(kind, nwin * W + tail)], seed=77 + t_i)
        data = data[:nwin * W + tail]
        want = oracle.deflate(data)
        cuts = [1] * nwin if t_i % 2 else ([nwin] if nwin < 3 else [1, nwin - 1])
        got, emitted = _pieces(eng, data, cuts)
        assert got == want, (kind, tail, cuts)
        assert raw_inflate(got) == data.tobytes()
        # output really leaves before close: every piece of compressible data hands bytes out
        assert all(e > 0 for e in emitted), emitted


def test_final_only_and_empty_streams(eng, oracle):
    for n in (0, 1, 16, 17, 127, 128, 4000, W, W + 1):
        data, _ = make_streams([("text", n)], seed=5)
        data = data[:n]
        got, _ = _pieces(eng, data, [])
        assert got == oracle.deflate(data), n


def test_go_compat_pieces(eng, oracle):
    data, _ = make_streams([("text", 3 * W + 777)], seed=9)
    data = data[:3 * W + 777]
    got, _ = _pieces(eng, data, [1, 2], compat_go=True)
    assert got == oracle.deflate(data, compat=oracle.COMPAT_GO)


def test_sticky_rules(eng):
    w = eng.open_stream()
    with pytest.raises(flate.FlateError):
        w.write(np.zeros(1000, np.uint8))  # not whole windows
    data = flate.synth("text", 1, W)
    assert len(w.write(data)) > 0          # an argument error is not sticky
    w.close(b"abc")
    w2 = eng.open_stream()
    w2.close(b"")
    with pytest.raises(flate.FlateError):
        w2._write(np.zeros(W, np.uint8), False)  # after close


@pytest.mark.parametrize("kernel", ["sub_block_decoder", "wave_per_stream"])
def test_size_only_pass_gives_the_exact_sizes(eng, oracle, kernel):
    eng.set_option("inflate_spec", 1 if kernel == "sub_block_decoder" else 0)
    try:
        _size_only_checks(eng, oracle)
    finally:
        eng.set_option("inflate_spec", 1)


def _size_only_checks(eng, oracle):
    specs = [("text", 70000), ("zero", 300000), ("rand", 5000), ("low", 0), ("runs", 131070), ("text", 17)]
    data, off = make_streams(specs, seed=3)
    comp, coff = eng.deflate_batch(data, off)
    sizes, status, err_off = eng.inflate_sizes(comp, coff)
    assert (status == 0).all() and sizes.tolist() == [n for _, n in specs]
    # then the exact-size decode; nothing had to be guessed
    back, boff, olen, st, _ = eng.inflate_batch(comp, coff, sizes)
    assert (st == 0).all() and bytes(back[:int(boff[-1])]) == data[:int(off[-1])].tobytes()
    # a broken stream: the size pass reports what the real pass reports
    bad = np.array(comp[:int(coff[1])], copy=True)
    bad[len(bad) // 2] ^= 0x55
    boff1 = np.array([0, bad.size], np.uint64)
    s1, st1, e1 = eng.inflate_sizes(bad, boff1)
    rc, out, consumed, eoff = oracle.inflate(bad, 200000, full=True)
    _, _, l2, st2, e2 = eng.inflate_batch(bad, boff1, [200000], check=False)
    assert int(st1[0]) == int(st2[0]) and int(s1[0]) == int(l2[0]) == len(out) and int(e1[0]) == int(e2[0])
    # truncated input
    s3, st3, _ = eng.inflate_sizes(comp[:int(coff[1]) - 9], np.array([0, int(coff[1]) - 9], np.uint64))
    assert int(st3[0]) == -7 and 0 < int(s3[0]) <= 70000


def test_long_stream_moves_its_origin(oracle):
    # flate_hip_stream has no length limit: past "stream_rebase_bytes" the origin of the positions the
    # kernels work with moves up (the reference's shift_offsets, deflate-fast.mbt:366-389).  With the
    # threshold lowered to two windows an 11-window stream rebases several times; the bytes stay those
    # of the one-shot stream in both compat modes.
    e = flate.FlateEngine(0)
    try:
        e.set_option("stream_rebase_bytes", 2 * W)
        for kind, go in (("text", False), ("period", False), ("text", True), ("runs", False)):
            n = 11 * W + 3000
            data, _ = make_streams([(kind, n)], seed=41)
            data = data[:n]
            want = oracle.deflate(data, compat=oracle.COMPAT_GO if go else 0)
            got, _ = _pieces(e, data, [1, 2, 1, 3, 2, 2], compat_go=go)
            assert got == want, (kind, go)
    finally:
        e.close()


def test_shift_offsets_as_the_reference_behaves(oracle):
    # deflate-fast.mbt:130-132,366-389: at window 32 766 of a Writer (2.1 GB in) `cur` reaches
    # buffer_reset and shift_offsets runs: MoonBit clears the table there (`prev` is empty),
    # Go keeps every distance.  With buffer_reset lowered on both sides (test hooks) the shifts fall
    # on windows 2, 5 and 8 of an 11-window stream; pieces are cut so that a shift falls on the
    # first, a middle and the last window of a piece, and the origin of the kernels' positions moves
    # (stream_rebase_bytes) in the same stream.
    e = flate.FlateEngine(0)
    try:
        e.set_option("debug_buffer_reset", 3 * W)
        oracle.set_buffer_reset(3 * W)
        for rebase in (1 << 30, 2 * W):
            e.set_option("stream_rebase_bytes", rebase)
            for kind, go in (("text", False), ("text", True), ("period", False), ("runs", False), ("low", True)):
                n = 11 * W + 3000
                data, _ = make_streams([(kind, n)], seed=43)
                data = data[:n]
                want = oracle.deflate(data, compat=oracle.COMPAT_GO if go else 0)
                for cuts in ([2, 1, 3, 2, 3], [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], [11], [3, 5]):
                    got, _ = _pieces(e, data, cuts, compat_go=go)
                    assert got == want, (kind, go, cuts, rebase)
        # and the hook changes the stream in MoonBit mode only
        oracle.set_buffer_reset(0)
        data, _ = make_streams([("text", 4 * W)], seed=43)
        data = data[:4 * W]
        got, _ = _pieces(e, data, [4])
        assert got != oracle.deflate(data)
        got_go, _ = _pieces(e, data, [4], compat_go=True)
        assert got_go == oracle.deflate(data, compat=oracle.COMPAT_GO)
    finally:
        oracle.set_buffer_reset(0)
        e.close()


def test_batch_refuses_a_stream_that_reaches_buffer_reset(eng):
    # a batch stream keeps its table from start to end: one with an LZ77 window at index 32 766 is
    # refused (E_TOO_LARGE) rather than compressed differently from the reference
    big = 32766 * W + 128
    off = np.array([0, big], np.uint64)
    with pytest.raises(flate.FlateError) as ei:
        eng.deflate_batch(np.zeros(16, np.uint8), off, out_cap=64)
    assert ei.value.code == -6


def test_the_real_buffer_reset_is_crossed_as_the_reference_crosses_it(oracle):
    # No lowered threshold here: 32 767 windows + a tail = 2.147 GB through ONE stream, so that `cur`
    # really reaches buffer_reset = 2^31 - 1 - 131070 at window 32 766 (deflate-fast.mbt:55,130-132) and
    # shift_offsets clears the table (MoonBit: `prev` is empty).  The data is mostly zeros with a marker
    # every 4 KiB (one wavefront compresses it at ~60 MB/s and the oracle at ~1 GB/s); the last windows
    # carry text so that a window that keeps its history and one that lost it differ in many places.
    n = 32767 * W + 1000
    data = np.zeros(n, dtype=np.uint8)
    data[::4096] = (np.arange(data[::4096].size) * 37 + 11).astype(np.uint8)
    tail_text = flate.synth("text", 1, 6 * W, first_stream=99)
    data[n - tail_text.size:] = tail_text
    data[n - 2 * W: n - W] = data[n - 3 * W: n - 2 * W]  # window 32 766 repeats window 32 765: matches across the border
    want = oracle.deflate(data)
    try:  # (the shift is visible in this stream: an encoder that never reaches buffer_reset writes other bytes)
        oracle.set_buffer_reset(0x7fffffff)
        assert oracle.deflate(data) != want
    finally:
        oracle.set_buffer_reset(0)
    e = flate.FlateEngine(0)
    try:
        pieces = [4096] * 7 + [4095]
        got, _ = _pieces(e, data, pieces)
        assert len(got) == len(want) and got == want
        # and back through the piecewise decoder (64-bit totals, the window across 34 pieces of 64 MiB)
        r = e.open_inflate_stream()
        comp = np.frombuffer(got, dtype=np.uint8)
        pos = at = 0
        rc = 0
        for _ in range(10000):
            take = max(0, (1 << 20) - r.pending_input)
            chunk = comp[pos:pos + take]
            pos += chunk.size
            o, rc = r.feed(chunk, final=pos >= comp.size, room=64 << 20)
            assert np.array_equal(o, data[at:at + o.size]), at
            at += o.size
            if rc != 0:
                break
        r.free()
        assert rc == 1 and at == n and r.total_in == comp.size
        # the same stream without the clear (Go semantics) is a different stream: the shift was visible
        got_go, _ = _pieces(e, data[n - 3 * W - 1000:], [1, 1], compat_go=True)  # (a fresh stream of the last windows: history kept)
        assert got_go == oracle.deflate(data[n - 3 * W - 1000:], compat=oracle.COMPAT_GO)
    finally:
        e.close()
    del data
