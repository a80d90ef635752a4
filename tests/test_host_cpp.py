"""C++ host mirror of the reference API (Writer::new/write/close) over the C ABI, on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from util import flate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _compile():
    flate.build()
    exe = os.path.join(HERE, "host_cpp", "driver")
    src = os.path.join(HERE, "host_cpp", "driver.cpp")
    libdir = os.path.join(ROOT, "moonbit-flate_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "moonbit-flate_amd", "host"), "-L" + libdir,
                           "-lflate_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    return exe


def test_host_mirror_compiles_without_gpu():
    exe = _compile()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_host_mirror_matches_reference_behaviour(oracle):
    exe = _compile()
    import struct
    import tempfile
    import zlib
    zdict = flate.synth("text", 1, 40000, seed=41).tobytes()
    dict_plain = zdict[30000:33000] + flate.synth("text", 1, 30000, seed=42).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, zdict)
    dict_comp = co.compress(dict_plain) + co.flush()
    case = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
    case.write(struct.pack("<I", len(zdict)) + zdict + struct.pack("<I", len(dict_comp)) + dict_comp)
    case.close()
    out = subprocess.run([exe, case.name], capture_output=True, text=True, timeout=120)
    os.unlink(case.name)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict()
    streams, reads, zmem, gmem = [], [], [], []
    for ln in out.stdout.splitlines():
        k, _, v = ln.partition(" ")
        if k == "zlibm":
            zmem.append(bytes.fromhex(v))
        elif k == "gzipm":
            gmem.append(bytes.fromhex(v))
        elif k == "stream":
            streams.append(bytes.fromhex(v))
        elif k == "read":
            reads.append(v)
        else:
            lines[k] = v
    assert lines["write"] == "11 17"
    assert lines["close"] == "none"
    want = oracle.deflate(b"hello world" + b"hello again world")
    assert len(want) == 38 and bytes.fromhex(lines["hello"]) == want   # deflate_test.mbt:23
    assert lines["dictwrite"] == "17 none 1" and bytes.fromhex(lines["dicthello"]) == want  # deflate_test.mbt:25-35
    assert lines["trailing"] == "28 EOF NEXT PAYLOAD"                   # make_reader :857-860; read-ahead handed back
    assert lines["close2"] == "none"                                    # deflate.mbt:158-160
    assert lines["write_after_close"] == "0 writer closed"              # deflate.mbt:281-283
    assert lines["batch"] == "none"
    ramp = (np.arange(65536) & 127).astype(np.uint8)
    assert streams[0] == oracle.deflate(b"") == bytes([1, 0, 0, 0xFF, 0xFF])
    assert streams[1] == oracle.deflate(ramp)
    assert streams[2] == oracle.deflate(bytes(100))
    # the same streams inside zlib / gzip containers: the oracle's frames, and zlib's own decoders accept them
    import gzip as gzip_mod
    assert lines["framed"] == "none none"
    # ... and decoded again: zlib members 0, 1 fine, 2 has a bad header; gzip member 1 has a damaged CRC
    assert lines["unframed"] == "none none 0:0 0:65536 -4:0 0:0 -4:65536 0:100"
    for i, plain in enumerate((b"", ramp.tobytes(), bytes(100))):
        assert zmem[i] == oracle.frame(oracle.FRAME_ZLIB, streams[i], plain) and zlib.decompress(zmem[i]) == plain
        assert gmem[i] == oracle.frame(oracle.FRAME_GZIP, streams[i], plain) and gzip_mod.decompress(gmem[i]) == plain
    # Decompressor::read hands out data first and the error with the last bytes (inflate.mbt:382-405)
    assert reads == ["7 none", "7 none", "7 none", "7 EOF", "0 EOF", "0 EOF", "0 EOF"]
    assert lines["rclose"] == "none"                                    # :410-415
    assert lines["reset"] == "100 EOF 0"                                # Decompressor::reset, :862
    assert bytes.fromhex(lines["plain"]) == b"hello worldhello again world"
    rc, _, _, err_off = oracle.inflate(bytes([7]) + want[1:], 100, full=True)
    assert rc == oracle.E_CORRUPT
    assert lines["badread"] == "0 flate: corrupt input before offset %d" % err_off   # :38-40
    assert lines["badclose"] == "flate: corrupt input before offset %d" % err_off
    rc, part, _, _ = oracle.inflate(oracle.deflate(ramp)[:-9], 70000, full=True)
    assert rc == oracle.E_UNEXPECTED_EOF
    assert lines["cutread"] == "%d unexpected EOF" % len(part)
    assert lines["spliced"] == "none"
    data = np.concatenate([ramp, np.zeros(100, np.uint8)])
    ref, _ = oracle.deflate_spliced(data, np.array([0, 0, 65536, 65636], np.uint64))
    assert bytes.fromhex(lines["one"]) == ref
    assert lines["unspliced"] == "none 65636 0"
    # BatchWriter: same bytes per stream as lone Writers, sticky state per Writer
    assert lines["bw_close"] == "none" and lines["bw_close2"] == "none"
    assert bytes.fromhex(lines["bw0"]) == want
    assert bytes.fromhex(lines["bw1"]) == oracle.deflate(ramp)
    assert bytes.fromhex(lines["bw2"]) == bytes([1, 0, 0, 0xFF, 0xFF])
    assert lines["bw_write_after_close"] == "0 writer closed"
    # chunked Writer: one spliced stream of independent 65536-byte chunks = the oracle's spliced
    # compressor over the same cut; any inflater (zlib here) returns the input
    import zlib
    x, big = 12345, bytearray()
    for _ in range(300000):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        big.append(b"etaoin shrdlu"[(x >> 24) % 13])
    cuts = np.array([0, 65536, 131072, 196608, 262144, 300000], np.uint64)
    ref, _ = oracle.deflate_spliced(np.frombuffer(bytes(big), np.uint8), cuts)
    got = bytes.fromhex(lines["chunked_bytes"])
    assert lines["chunked"] == "none %d" % len(ref) and got == ref
    assert zlib.decompressobj(-15).decompress(got) == bytes(big)
    assert lines["chunked_back"] == "300000 EOF 0"
    # the default Writer on one long stream: bit-exact, and bytes reached the sink before close()
    x, long_ = 777, bytearray()
    for _ in range(5 * 65535 + 4321):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        long_.append(b"the quick brown fox "[(x >> 24) % 20])
    want_long = oracle.deflate(bytes(long_))
    st, before, total = lines["streamed"].split()
    assert st == "none" and int(total) == len(want_long) and 0 < int(before) < int(total)
    assert bytes.fromhex(lines["streamed_bytes"]) == want_long
    assert lines["streamed_back"] == "%d EOF 0" % len(long_)   # Reader without a size hint
    assert lines["wrong_hint"] == "%d EOF" % len(long_)        # ... and with a wrong one
    # 6 MB through a Reader with a 64 KiB input piece and a 100 000-byte output piece: all bytes, ioeof
    # with the last ones, and never more resident than the two pieces
    assert lines["long_reader"] == "none 6000000 EOF 0 1"
    # &Reader::new_dict, then the same handle reset without and with the dictionary (inflate.mbt:315,862)
    assert lines["dict_read"] == "%d EOF" % len(dict_plain)
    assert bytes.fromhex(lines["dict_plain"]) == dict_plain
    rc, part, _, eoff = oracle.inflate(dict_comp, len(dict_plain) + 8, full=True)
    assert rc == oracle.E_CORRUPT
    assert lines["nodict_read"] == "%d flate: corrupt input before offset %d" % (len(part), eoff)
    assert lines["redict_read"] == "%d EOF 1" % len(dict_plain)
