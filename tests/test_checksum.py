"""Checksums and container formats around the raw streams (optional gzip / zlib wrappers; the
reference has neither).  CPU: the oracle's restatement of RFC 1950 / RFC 1952 pinned against zlib.  GPU:
flate_hip_checksum_batch against the oracle, and the framed output of the host mirrors accepted by zlib."""
import gzip
import zlib

import numpy as np
import pytest

from util import flate, make_streams


def test_oracle_checksums_are_zlibs(oracle):
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 15, 16, 17, 255, 256, 5551, 5552, 5553, 65535, 65536, 65537, 1 << 20, 3_000_001):
        for kind in range(3):
            d = (rng.integers(0, 256, n, dtype=np.uint8) if kind == 0 else
                 np.full(n, 255, np.uint8) if kind == 1 else flate.synth("text", 1, max(n, 1))[:n]).tobytes()
            assert oracle.adler32(d) == zlib.adler32(d), (n, kind)
            assert oracle.crc32(d) == zlib.crc32(d), (n, kind)


def test_oracle_frames_are_accepted_by_zlib_and_gzip(oracle):
    for n in (0, 1, 17, 300, 65536, 200000):
        d = flate.synth("text", 1, max(n, 1))[:n].tobytes()
        raw = oracle.deflate(np.frombuffer(d, np.uint8))
        z = oracle.frame(oracle.FRAME_ZLIB, raw, d)
        g = oracle.frame(oracle.FRAME_GZIP, raw, d)
        assert zlib.decompress(z) == d and gzip.decompress(g) == d
        assert len(z) == len(raw) + 6 and len(g) == len(raw) + 18
        assert oracle.frame(oracle.FRAME_RAW, raw, d) == raw
        # a damaged trailer is what the containers are for
        bad = bytearray(z)
        bad[-1] ^= 1
        with pytest.raises(zlib.error):
            zlib.decompress(bytes(bad))


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def eng():
    e = flate.FlateEngine(0)
    yield e
    e.close()


SIZES = [0, 1, 2, 15, 16, 17, 1023, 1024, 1025, 4095, 5552, 65535, 65536, 65537, 131071, 200000, 1 << 20, (1 << 24) + 3]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adler32", "crc32"])
def test_checksum_batch_is_zlibs(eng, oracle, kind):
    import torch
    rng = np.random.default_rng(17)
    ref = zlib.adler32 if kind == "adler32" else zlib.crc32
    for fill in ("rand", "ff", "text"):
        parts = []
        for n in SIZES:
            parts.append(rng.integers(0, 256, n, dtype=np.uint8) if fill == "rand" else
                         np.full(n, 255, np.uint8) if fill == "ff" else flate.synth("text", 1, max(n, 1))[:n])
        off = np.zeros(len(parts) + 1, np.uint64)
        np.cumsum([p.size for p in parts], out=off[1:])   # (streams start at arbitrary alignments)
        data = np.concatenate(parts + [np.zeros(16, np.uint8)])
        want = [ref(p.tobytes()) for p in parts]
        got = eng.checksum_batch(data, off, kind)
        assert got.tolist() == want, (kind, fill)
        d_data = torch.from_numpy(data).cuda()
        assert eng.checksum_batch(d_data, off, kind).tolist() == want
        assert want[3] == (oracle.adler32 if kind == "adler32" else oracle.crc32)(parts[3])
    # one stream of 1 GiB (16384 pieces folded by one thread) -- and the same bytes as 16384 streams
    n, blen = 16384, 65536
    host = flate.synth("text", n, blen)
    d = torch.from_numpy(host).cuda()
    whole = eng.checksum_batch(d, np.array([0, n * blen], np.uint64), kind)
    assert int(whole[0]) == ref(host.tobytes())
    each = eng.checksum_batch(d, flate.uniform_offsets(n, blen), kind)
    for i in (0, 1, 777, n - 1):
        assert int(each[i]) == ref(host[i * blen:(i + 1) * blen].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("wrap", ["zlib", "gzip"])
def test_framed_batches(eng, oracle, wrap):
    specs = [("text", 65536), ("rand", 70000), ("zero", 100000), ("text", 0), ("text", 1), ("text", 17), ("ramp", 200000)]
    data, off = make_streams(specs, seed=5)
    framed, foff = eng.deflate_batch_framed(data, off, wrap)
    kind = oracle.FRAME_ZLIB if wrap == "zlib" else oracle.FRAME_GZIP
    for i, (_, n) in enumerate(specs):
        plain = data[int(off[i]):int(off[i + 1])].tobytes()
        member = framed[int(foff[i]):int(foff[i + 1])].tobytes()
        assert member == oracle.frame(kind, oracle.deflate(np.frombuffer(plain, np.uint8)), plain), (wrap, i)
        assert (zlib.decompress(member) if wrap == "zlib" else gzip.decompress(member)) == plain
    # and back: sizes from the decoder's size-only pass (zlib) or the members' ISIZE (gzip)
    out, ooff, olen, status = eng.inflate_batch_framed(framed, foff, wrap)
    assert (status == 0).all() and [int(x) for x in olen] == [n for _, n in specs]
    assert out[:int(ooff[-1])].tobytes() == data[:int(off[-1])].tobytes()
    # members written by zlib itself (other levels, a gzip member with a file name)
    plain = data[:65536].tobytes()
    if wrap == "zlib":
        foreign = [zlib.compress(plain, lvl) for lvl in (1, 6, 9)]
    else:
        import io
        buf = io.BytesIO()
        with gzip.GzipFile(filename="some name.txt", mode="wb", fileobj=buf, mtime=12345) as f:
            f.write(plain)
        foreign = [gzip.compress(plain, 1), buf.getvalue()]
    fo = np.zeros(len(foreign) + 1, np.uint64)
    np.cumsum([len(x) for x in foreign], out=fo[1:])
    fdata = np.frombuffer(b"".join(foreign) + b"\0" * 8, np.uint8)
    out, ooff, olen, status = eng.inflate_batch_framed(fdata, fo, wrap, out_sizes=[65536] * len(foreign))
    assert (status == 0).all()
    for i in range(len(foreign)):
        assert out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() == plain
    # a damaged checksum, a damaged payload byte that still decodes, a bad header
    bad = bytearray(framed.tobytes())
    t = int(foff[1]) - (1 if wrap == "zlib" else 5)   # last byte of member 0's Adler-32 / CRC-32
    bad[t] ^= 0x40
    bad[int(foff[1])] ^= 0xFF                          # member 1: first header byte
    out, ooff, olen, status = eng.inflate_batch_framed(np.frombuffer(bytes(bad), np.uint8), foff, wrap,
                                                        out_sizes=[n for _, n in specs])
    assert status[0] == -4 and status[1] == -4 and (status[2:] == 0).all()


@pytest.mark.gpu
def test_whole_batch_as_one_gzip_member_and_one_zlib_stream(eng):
    # every stream compressed on its own, in parallel, spliced into ONE DEFLATE stream -- and framed as one file
    specs = [("text", 65536)] * 40 + [("rand", 70000), ("text", 0), ("zero", 300000), ("text", 17)]
    data, off = make_streams(specs, seed=9)
    plain = data[:int(off[-1])].tobytes()
    assert gzip.decompress(eng.deflate_spliced_framed(data, off, "gzip")) == plain
    assert zlib.decompress(eng.deflate_spliced_framed(data, off, "zlib")) == plain

