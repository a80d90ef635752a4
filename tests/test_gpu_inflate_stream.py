"""One long DEFLATE stream decoded in pieces (flate_hip_inflate_stream_*): the resumable Decompressor of
the reference (inflate.mbt:252-290,382-407; dict-decoder.mbt:29-60) -- bytes, statuses and error offsets
equal the oracle's whole-stream decode whatever the piece sizes, including corruptions and truncations
that fall on piece borders."""
import zlib

import numpy as np
import pytest

from util import flate, make_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = flate.FlateEngine(0)
    yield e
    e.close()


def drive(eng, comp, in_piece, out_room, rng=None, max_calls=200000, zdict=None, reader=None):
    """Feed `comp` in pieces of in_piece bytes (or random sizes up to it), take out_room bytes at a time.
    Returns (out bytes, final status, err_off, input bytes consumed, calls)."""
    r = reader if reader is not None else eng.open_inflate_stream(zdict)
    comp = np.frombuffer(bytes(comp), dtype=np.uint8)
    pos, out, calls = 0, [], 0
    try:
        while True:
            calls += 1
            assert calls < max_calls, "no progress"
            k = in_piece if rng is None else int(rng.integers(1, in_piece + 1))
            room = out_room if rng is None else int(rng.integers(1, out_room + 1))
            # top the decoder's pending input up to a piece
            take = max(0, k - r.pending_input)
            piece = comp[pos:pos + take]
            pos += piece.size
            final = pos >= comp.size
            o, rc = r.feed(piece, final=final, room=room)
            out.append(o)
            if rc != 0:
                return np.concatenate(out).tobytes(), rc, r.err_off, r.total_in, calls
            if final and o.size == 0 and r.pending_input == 0 and calls > 3 and rc == 0:
                # the decoder wants more although nothing is left: cannot happen with final set
                raise AssertionError("stalled at the end of the input")
    finally:
        if reader is None:
            r.free()


KINDS = [("text", 200000), ("zero", 300000), ("rand", 70000), ("low", 100000), ("runs", 150000), ("period", 180000),
         ("ramp", 66000)]


@pytest.mark.parametrize("kind,n", KINDS)
def test_pieces_give_the_whole_stream(eng, oracle, kind, n):
    data, _ = make_streams([(kind, n)], seed=21)
    data = data[:n]
    comp = oracle.deflate(data)
    want = data.tobytes()
    for in_piece, room in ((1 << 20, 1 << 20), (4096, 10000), (700, 333), (65536, 65536), (1500, 1 << 16)):
        got, rc, _, used, _ = drive(eng, comp, in_piece, room)
        assert rc == 1 and got == want, (kind, in_piece, room, rc, len(got))
        # the reader stops at the end of the final block: roffset = all of the stream
        assert used == len(comp)
    rng = np.random.default_rng(5)
    got, rc, _, used, _ = drive(eng, comp, 3000, 5000, rng=rng)
    assert rc == 1 and got == want and used == len(comp)


def test_foreign_encoders_fixed_and_stored_blocks(eng, oracle):
    data, _ = make_streams([("text", 150000)], seed=4)
    data = data[:150000].tobytes()
    for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        comp = co.compress(data) + co.flush()
        for in_piece, room in ((2048, 4096), (1 << 16, 777), (999, 1 << 17)):
            got, rc, _, used, _ = drive(eng, comp, in_piece, room)
            assert rc == 1 and got == data and used == len(comp), (level, strategy, in_piece, room)
    # trailing bytes behind the final block are not consumed (the reference's reader stops there)
    comp = oracle.deflate(np.frombuffer(data[:5000], np.uint8)) + b"garbage!" * 100
    got, rc, _, used, _ = drive(eng, comp, 1 << 16, 1 << 16)
    assert rc == 1 and got == data[:5000] and used == len(comp) - 800


def test_output_full_exactly_at_the_end_reports_the_end_with_the_last_bytes(eng, oracle):
    data = b"hello world" + b"hello again world"
    comp = oracle.deflate(data)
    r = eng.open_inflate_stream()
    seen = []
    piece, final = np.frombuffer(comp, np.uint8), True
    for _ in range(6):
        o, rc = r.feed(piece, final=final, room=7)
        seen.append((o.tobytes(), rc))
        piece = np.zeros(0, np.uint8)
    r.free()
    # Decompressor::read hands out io.EOF with the last bytes (inflate.mbt:394-397), then (0, EOF) for ever
    assert [len(o) for o, _ in seen] == [7, 7, 7, 7, 0, 0]
    assert [rc for _, rc in seen] == [0, 0, 0, 1, 1, 1]
    assert b"".join(o for o, _ in seen) == data


def test_corruptions_and_truncations_at_piece_borders(eng, oracle):
    data, _ = make_streams([("text", 120000)], seed=8)
    data = data[:120000]
    comp = bytearray(oracle.deflate(data))
    rng = np.random.default_rng(11)
    cases = 0
    for trial in range(40):
        bad = bytearray(comp)
        if trial % 2 == 0:  # flip a byte
            at = int(rng.integers(0, len(bad)))
            bad[at] ^= int(rng.integers(1, 256))
        else:               # truncate
            at = int(rng.integers(1, len(bad)))
            bad = bad[:at]
        rc0, out0, consumed0, eoff0 = oracle.inflate(bytes(bad), 400000, full=True)
        want_rc = {0: 1, oracle.E_CORRUPT: -4, oracle.E_UNEXPECTED_EOF: -7}[rc0]
        # piece borders exactly at, just before and just after the damaged byte, and small pieces throughout
        for in_piece, room in ((max(at, 1), 1 << 20), (max(at - 1, 1), 5000), (at + 1, 1 << 20), (1024, 1 << 20), (611, 4099)):
            got, rc, eoff, used, _ = drive(eng, bytes(bad), in_piece, room)
            assert rc == want_rc, (trial, at, in_piece, rc, want_rc)
            assert got == out0, (trial, at, in_piece, len(got), len(out0))
            if rc == -4:
                assert eoff == eoff0, (trial, at, in_piece, eoff, eoff0)
            cases += 1
    assert cases == 200


def test_sticky_status_and_argument_rules(eng, oracle):
    comp = oracle.deflate(np.frombuffer(b"abc" * 1000, np.uint8))
    r = eng.open_inflate_stream()
    o, rc = r.feed(comp, final=True, room=1 << 16)
    assert rc == 1 and o.tobytes() == b"abc" * 1000
    o, rc = r.feed(b"more", final=True)
    assert rc == 1 and o.size == 0   # after the end: (0, EOF) again
    r.free()
    r = eng.open_inflate_stream()
    o, rc = r.feed(b"\x07\xff\xff", final=True)  # reserved block type
    assert rc == -4 and r.err_off == 1
    o, rc = r.feed(comp, final=True)
    assert rc == -4 and o.size == 0 and r.err_off == 1  # sticky
    r.free()
    r = eng.open_inflate_stream()
    o, rc = r.feed(b"", final=False)
    assert rc == 0 and o.size == 0       # nothing to decode from: call again
    o, rc = r.feed(b"", final=True)
    assert rc == -7                       # an empty stream is an unexpected EOF (inflate.mbt:345-349)
    r.free()


# ---- preset dictionary: &Reader::new_dict / Decompressor::reset(r, dict) (inflate.mbt:315-317,862-884) ----
def zdeflate(data, zdict, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, zdict)
    return co.compress(data) + co.flush()


@pytest.mark.parametrize("dlen", [1, 258, 4096, 32767, 32768, 32769, 50000])
def test_preset_dictionary(eng, oracle, dlen):
    zdict = flate.synth("text", 1, dlen, seed=11).tobytes()
    data = zdict[-min(dlen, 3000):] + flate.synth("text", 1, 90000, seed=12).tobytes() + zdict[:min(dlen, 5000)]
    comp = zdeflate(data, zdict)
    rc0, want, used0, _ = oracle.inflate(comp, len(data) + 8, full=True, zdict=zdict)
    assert rc0 == 0 and want == data
    for in_piece, room in ((1 << 20, 1 << 20), (4096, 10000), (700, 333), (1500, 1 << 16)):
        got, rc, _, used, _ = drive(eng, comp, in_piece, room, zdict=zdict)
        assert rc == 1 and got == data and used == len(comp), (dlen, in_piece, room, rc, len(got))
    got, rc, _, used, _ = drive(eng, comp, 3000, 5000, rng=np.random.default_rng(dlen), zdict=zdict)
    assert rc == 1 and got == data and used == len(comp)


def test_dictionary_too_short_or_missing_is_the_oracles_corrupt(eng, oracle):
    zdict = flate.synth("text", 1, 8192, seed=21).tobytes()
    data = zdict[1000:1400] + b"tail" + zdict[3000:3300] + flate.synth("text", 1, 40000, seed=22).tobytes()
    comp = zdeflate(data, zdict)
    for d in (None, zdict[-1000:], zdict[-5000:], b"x" * 5000 + zdict, zdict):
        rc0, out0, used0, eo0 = oracle.inflate(comp, len(data) + 8, full=True, zdict=d)
        want_rc = {0: 1, oracle.E_CORRUPT: -4}[rc0]
        for in_piece, room in ((1 << 16, 1 << 16), (700, 100), (1 << 16, 3)):
            got, rc, eo, used, _ = drive(eng, comp, in_piece, room, zdict=d)
            assert (rc, got) == (want_rc, out0), (None if d is None else len(d), in_piece, room, rc, len(got), len(out0))
            if rc0:
                assert eo == eo0 and used == used0


def test_copy_from_the_dictionary_running_into_the_output(eng, oracle):
    zdict = b"....abcdeZ"
    data = b"xyz" + b"Zxyz" * 4000
    comp = zdeflate(data, zdict, 9)
    assert oracle.inflate(comp, len(data), zdict=zdict) == data
    for in_piece, room in ((1 << 16, 1 << 16), (640, 5), (1000, 1)):
        got, rc, _, _, _ = drive(eng, comp, in_piece, room, zdict=zdict)
        assert rc == 1 and got == data


def test_reset_reuses_the_handle_with_and_without_a_dictionary(eng, oracle):
    zdict = flate.synth("text", 1, 20000, seed=31).tobytes()
    a = zdict[5000:9000] + flate.synth("text", 1, 50000, seed=32).tobytes()
    ca, cb = zdeflate(a, zdict), oracle.deflate(np.frombuffer(a, np.uint8))
    r = eng.open_inflate_stream()
    try:
        for rounds in range(2):
            r.reset(zdict)
            got, rc, _, _, _ = drive(eng, ca, 2000, 3000, reader=r)
            assert rc == 1 and got == a
            r.reset()  # Decompressor::reset(r, []) -- the dictionary is gone
            got, rc, _, _, _ = drive(eng, cb, 2000, 3000, reader=r)
            assert rc == 1 and got == a
            r.reset()  # a stream that needs the dictionary fails without it, and the error is not kept
            got, rc, _, _, _ = drive(eng, ca, 2000, 3000, reader=r)
            assert rc == -4
        # reset in the middle of a stream: the half-decoded state is dropped
        r.reset(zdict)
        o, rc = r.feed(np.frombuffer(ca[:1000], np.uint8), final=False, room=500)
        assert rc == 0 and o.size > 0
        r.reset(zdict)
        got, rc, _, _, _ = drive(eng, ca, 1 << 16, 1 << 16, reader=r)
        assert rc == 1 and got == a
    finally:
        r.free()
