"""the streams of a batch as ONE legal DEFLATE stream.
CPU part: the oracle's spliced compressor against zlib and against its own per-stream output;
GPU part: flate_hip_deflate_fast_spliced against the oracle, bit for bit."""
import zlib

import numpy as np
import pytest

from util import flate, make_streams

# every block kind in every position: stored tails (1..16 B), Huffman-only (17..127 B and
# incompressible windows), dynamic, multi-window, empty streams
SPECS = [("text", 65536), ("text", 0), ("ramp", 70000), ("rand", 300), ("text", 17), ("zero", 1),
         ("text", 131070), ("period", 5000), ("low", 200000), ("text", 65535), ("rand", 65536),
         ("text", 16), ("runs", 40000), ("text", 127), ("text", 128)]


def _inflaters(eng):
    """The two kernels that decode a spliced stream from its index (and both builds of the second):
    every inflate_spliced check runs on each."""
    try:
        for name, spec, shape in (("lane_per_stream", 0, 0), ("sub_block_small_batch", 2, 1), ("sub_block_large_batch", 2, 2)):
            eng.set_option("inflate_spec", spec)
            eng.set_option("inflate_spec_shape", shape)
            yield name
    finally:
        eng.set_option("inflate_spec", 1)
        eng.set_option("inflate_spec_shape", 0)


def _inflate_one(stream):
    d = zlib.decompressobj(-15)
    res = d.decompress(stream) + d.flush()
    assert d.eof and d.unused_data == b""
    return res


def _bits(buf, lo, hi):
    """bits [lo, hi) of a byte string as an int (LSB first, DEFLATE order)"""
    a = np.frombuffer(buf, dtype=np.uint8)[lo >> 3:(hi + 7) >> 3]
    v = int.from_bytes(a.tobytes(), "little") >> (lo & 7)
    return v & ((1 << (hi - lo)) - 1)


@pytest.mark.parametrize("compat", ["moonbit", "go"])
def test_oracle_spliced_is_one_legal_stream(oracle, compat):
    data, off = make_streams(SPECS * 3, seed=11)
    cm = oracle.COMPAT_GO if compat == "go" else oracle.COMPAT_MOONBIT
    spliced, bit_off = oracle.deflate_spliced(data, off, cm)
    assert _inflate_one(spliced) == data[:int(off[-1])].tobytes()
    assert len({int(x) % 8 for x in bit_off}) > 4            # joins at many bit alignments
    assert len(spliced) == (int(bit_off[-1]) + 3 + 7) // 8 + 4


def test_oracle_spliced_blocks_are_the_per_stream_blocks(oracle):
    """Up to its first stored block a stream's bits are those of its own Writer output, shifted."""
    data, off = make_streams(SPECS, seed=4)
    spliced, bit_off = oracle.deflate_spliced(data, off)
    for i, (kind, n) in enumerate(SPECS):
        own = oracle.deflate(data[int(off[i]):int(off[i + 1])])
        nbits = int(bit_off[i + 1] - bit_off[i])
        tail = n % 65535
        if 1 <= tail <= 16:                 # ends with a stored block: compare in front of it
            nbits -= 32 + 8 * tail          # LEN, NLEN, raw bytes (byte aligned in both)
            own_payload = 8 * (len(own) - 5 - 4 - tail)      # up to the byte boundary before LEN
            pre = min(nbits, own_payload) - 10               # stay clear of header bits + padding
            if pre > 0:
                assert _bits(spliced, int(bit_off[i]), int(bit_off[i]) + pre) == _bits(own, 0, pre)
            assert spliced[(int(bit_off[i + 1]) >> 3) - tail:int(bit_off[i + 1]) >> 3] == \
                   data[int(off[i + 1]) - tail:int(off[i + 1])].tobytes()
        elif nbits:
            assert _bits(spliced, int(bit_off[i]), int(bit_off[i + 1])) == _bits(own, 0, nbits)


def test_oracle_spliced_of_nothing_and_of_one(oracle):
    s0, b0 = oracle.deflate_spliced(np.zeros(1, np.uint8), np.zeros(1, np.uint64))
    assert s0 == b"\x01\x00\x00\xff\xff" and list(b0) == [0]
    data, off = make_streams([("text", 3000)], seed=9)
    s1, _ = oracle.deflate_spliced(data, off)
    assert s1 == oracle.deflate(data[:3000])      # one stream: exactly Writer::new/write/close


@pytest.fixture(scope="module")
def eng():
    flate.build()
    e = flate.FlateEngine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("compat", ["moonbit", "go"])
def test_gpu_spliced_equals_oracle_bit_for_bit(eng, oracle, compat):
    data, off = make_streams(SPECS * 3, seed=11)
    go = compat == "go"
    out, n, bit_off = eng.deflate_spliced(data, off, compat_go=go)
    ref, ref_off = oracle.deflate_spliced(data, off, oracle.COMPAT_GO if go else oracle.COMPAT_MOONBIT)
    assert np.array_equal(bit_off, ref_off)
    assert n == len(ref)
    assert bytes(out[:n]) == ref
    assert _inflate_one(bytes(out[:n])) == data[:int(off[-1])].tobytes()


@pytest.mark.gpu
def test_gpu_spliced_device_pointers_and_relation_to_the_batch_output(eng, oracle):
    """64 KiB streams end in a stored tail block, so every join is byte aligned and
    the spliced stream is the batch output minus the closing blocks -- checked at 2048 streams
    with a misaligned destination."""
    import torch
    n, blen = 2048, 65536
    host = flate.synth("text", n, blen, seed=0x5EED0001)
    d_in = torch.from_numpy(host).cuda()
    in_off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d_in, in_off)
    whole = torch.full((int(coff[-1]) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    dst = whole[3:]
    out, nbytes, bit_off = eng.deflate_spliced(d_in, in_off, out=dst)
    assert whole[:3].tolist() == [0xA5] * 3        # nothing in front of the destination is touched
    assert not (bit_off % 8).any()
    assert nbytes == int(coff[-1]) - 5 * (n - 1)
    comp_h, res = comp.cpu().numpy(), out[:nbytes].cpu().numpy()
    want = np.concatenate([comp_h[int(coff[i]):int(coff[i + 1]) - 5] for i in range(n)] +
                          [np.frombuffer(b"\x01\x00\x00\xff\xff", dtype=np.uint8)])
    assert np.array_equal(res, want)
    assert _inflate_one(res.tobytes()) == host.tobytes()


@pytest.mark.gpu
def test_gpu_spliced_edge_cases(eng, oracle):
    out, n, bit_off = eng.deflate_spliced(np.zeros(8, np.uint8), np.zeros(1, np.uint64))
    assert bytes(out[:n]) == b"\x01\x00\x00\xff\xff"
    for specs in ([("text", 0)], [("text", 0), ("text", 0), ("text", 1)], [("rand", 5)] * 70,
                  [("text", 3000)], [("text", 20), ("zero", 3), ("text", 20), ("zero", 16)] * 9):
        data, off = make_streams(specs, seed=2)
        out, n, bit_off = eng.deflate_spliced(data, off)
        ref, ref_off = oracle.deflate_spliced(data, off)
        assert bytes(out[:n]) == ref and np.array_equal(bit_off, ref_off)
    data, off = make_streams([("text", 30000)] * 4, seed=3)
    small = np.zeros(100, np.uint8)
    with pytest.raises(flate.FlateError) as ei:
        eng.deflate_spliced(data, off, out=small)
    assert ei.value.code == -2  # FLATE_HIP_E_OUT_TOO_SMALL


@pytest.mark.gpu
def test_gpu_spliced_full_size_config2(eng):
    """BASELINE config 2 (16384 x 64 KiB S-text, 1 GiB): size and a sample of segments against the
    batch output; the whole result is checked block by block by the 2048-stream test above."""
    import torch
    n, blen = 16384, 65536
    d_in = torch.from_numpy(flate.synth("text", n, blen, seed=0x5EED0001)).cuda()
    in_off = flate.uniform_offsets(n, blen)
    comp, coff = eng.deflate_batch(d_in, in_off)
    out, nbytes, bit_off = eng.deflate_spliced(d_in, in_off)
    assert nbytes == int(coff[-1]) - 5 * (n - 1)
    assert not (bit_off % 8).any() and int(bit_off[-1]) // 8 + 5 == nbytes
    for i in np.linspace(0, n - 1, 97).astype(int):
        a, b = int(coff[i]), int(coff[i + 1]) - 5
        p = int(bit_off[i]) // 8
        assert torch.equal(out[p:p + (b - a)], comp[a:b])
    assert bytes(out[nbytes - 5:nbytes].cpu().numpy()) == b"\x01\x00\x00\xff\xff"


@pytest.mark.gpu
@pytest.mark.parametrize("compat", ["moonbit", "go"])
def test_gpu_inflate_spliced_from_the_index(eng, oracle, compat):
    """One spliced stream decoded in parallel from bit_off == what a single Reader over the whole
    stream produces (the oracle's inflater), piece by piece."""
    data, off = make_streams(SPECS * 3, seed=13)
    go = compat == "go"
    spliced, bit_off = oracle.deflate_spliced(data, off, oracle.COMPAT_GO if go else oracle.COMPAT_MOONBIT)
    sizes = [int(off[i + 1] - off[i]) for i in range(len(off) - 1)]
    comp = np.frombuffer(spliced + b"\0" * 8, dtype=np.uint8).copy()
    whole = oracle.inflate(spliced, int(off[-1]))
    for kernel in _inflaters(eng):
        out, ooff, olen, status, _ = eng.inflate_spliced(comp, len(spliced), bit_off, sizes)
        assert (status == 0).all() and list(olen) == sizes, kernel
        assert bytes(out[:int(off[-1])]) == whole == data[:int(off[-1])].tobytes(), kernel


@pytest.mark.gpu
def test_gpu_inflate_spliced_bad_index_and_full_round_trip(eng, oracle):
    import torch
    n, blen = 4096, 65536
    host = flate.synth("text", n, blen, seed=0x5EED0001)
    d_in = torch.from_numpy(host).cuda()
    in_off = flate.uniform_offsets(n, blen)
    comp, nbytes, bit_off = eng.deflate_spliced(d_in, in_off)
    bad = bit_off.copy()
    bad[7] += 1          # piece 6 now runs into piece 7's first block, piece 7 starts mid-block
    for kernel in _inflaters(eng):
        out, _, olen, status, _ = eng.inflate_spliced(comp, nbytes, bit_off, [blen] * n)
        assert (status == 0).all() and (olen == blen).all() and torch.equal(out[:n * blen], d_in), kernel
        _, _, _, status, _ = eng.inflate_spliced(comp, nbytes, bad, [blen] * n, check=False)
        assert status[6] in (-2, -4) and status[7] != 0, kernel   # output overflow or off-boundary stop
        assert (np.delete(status, [6, 7]) == 0).all(), kernel


@pytest.mark.gpu
def test_gpu_spliced_fuzz_random_sizes(eng, oracle):
    rng = np.random.default_rng(77)
    kinds = ["text", "ramp", "zero", "rand", "low", "period", "runs"]
    for rnd in range(4):
        specs = [(kinds[int(rng.integers(len(kinds)))],
                  int(rng.choice([0, 1, 3, 15, 16, 17, 18, 100, 127, 128, 129, 2000, 65534, 65535, 65536,
                                  65537, 65551, 65552, 70000, 131070, 131071, 140000])))
                 for _ in range(60)]
        data, off = make_streams(specs, seed=100 + rnd)
        go = bool(rnd & 1)
        out, n, bit_off = eng.deflate_spliced(data, off, compat_go=go)
        ref, ref_off = oracle.deflate_spliced(data, off, oracle.COMPAT_GO if go else oracle.COMPAT_MOONBIT)
        assert bytes(out[:n]) == ref and np.array_equal(bit_off, ref_off), rnd
        sizes = [s for _, s in specs]
        for kernel in _inflaters(eng):
            back, _, olen, status, _ = eng.inflate_spliced(out, n, bit_off, sizes)
            assert (status == 0).all() and list(olen) == sizes, kernel
            assert bytes(back[:int(off[-1])]) == data[:int(off[-1])].tobytes(), kernel


@pytest.mark.gpu
def test_config5_full_size_round_trip(eng):
    """BASELINE configs[4] at full size: 131072 x 64 KiB (8 GiB) -> ONE compressed stream ->
    decoded in parallel from its index == the input (encode -> decode round-trip property)."""
    import torch
    n, blen = 131072, 65536
    d_in = torch.from_numpy(flate.synth("text", n, blen, seed=0x5EED0001)).cuda()
    in_off = flate.uniform_offsets(n, blen)
    comp, nbytes, bit_off = eng.deflate_spliced(d_in, in_off)
    assert 3.0e9 < nbytes < 4.5e9 and int(bit_off[-1]) // 8 + 5 == nbytes
    comp = comp[:nbytes + 8].clone()          # drop the bound-sized buffer before the 8 GiB output
    out, _, olen, status, _ = eng.inflate_spliced(comp, nbytes, bit_off, [blen] * n)
    assert (status == 0).all() and (olen == blen).all()
    assert torch.equal(out[:n * blen], d_in)
