"""Pins the CPU oracle against every known answer the reference's own tests hold
for the deflate-fast path and against independent inflaters."""
import os
import re
import zlib

import numpy as np
import pytest

REF = "/root/reference"


def raw_inflate(b):
    d = zlib.decompressobj(-15)
    out = d.decompress(b)
    assert d.eof, "zlib did not reach the final block"
    assert d.unused_data == b""
    return out


def ramp(n):
    return bytes((i & 127) for i in range(n))


# ---- scalar KATs ---------------------------------------------------------------

def test_token_offset_kat(oracle):
    # token.mbt:95-99
    assert oracle.lib().orc_token_offset(2143289471) == 127
    assert oracle.lib().orc_token_length(2143289471) == 255
    assert oracle.lib().orc_match_token(255, 127) == 2143289471


def test_reverse16_kat(oracle):
    # bits.mbt:24-27
    assert oracle.lib().orc_reverse16(32768) == 1


def test_reverse_bits_kat(oracle):
    # huffman-code.mbt:289-292
    assert oracle.lib().orc_reverse_bits(64, 7) == 1


def _parse_mbt_int_array(path, name):
    txt = open(path).read()
    m = re.search(r"let %s\s*:[^=]*=\s*\[(.*?)\n\]" % name, txt, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_code_tables_match_reference_text(oracle):
    # The oracle derives length_codes/offset_codes from RFC 1951 rules; check them
    # against the literal tables in token.mbt:30-61 (text study, not execution).
    L = oracle.lib()
    lc = _parse_mbt_int_array(os.path.join(REF, "token.mbt"), "length_codes")
    oc = _parse_mbt_int_array(os.path.join(REF, "token.mbt"), "offset_codes")
    assert len(lc) == 256 and len(oc) == 256
    assert [L.orc_length_code(i) for i in range(256)] == lc
    assert [L.orc_offset_code(i) for i in range(256)] == oc
    rev = _parse_mbt_int_array(os.path.join(REF, "bits.mbt"), "rev8tab")
    assert [L.orc_reverse16(i << 8) for i in range(256)] == rev


def test_code_tables_spot(oracle):
    L = oracle.lib()
    assert [L.orc_length_code(i) for i in (0, 7, 8, 9, 10, 254, 255)] == [0, 7, 8, 8, 9, 27, 28]
    assert [L.orc_offset_code(i) for i in (0, 3, 4, 5, 6, 255, 256, 32767)] == \
        [0, 3, 4, 4, 5, 15, 16, 29]


# ---- hand-derived vectors --------------------------------------------

def test_ramp_first_tokens(oracle):
    t = oracle.DeflateFast().encode(ramp(65535))
    assert t[:128].tolist() == list(range(128))
    assert t[128] == 0
    assert t[129] == 2143289471  # len 258 dist 128 (token.mbt:96)
    assert t[130] == 2143289727  # len 258 dist 384
    assert t[131] == 2143289983  # len 258 dist 640


def test_hello_world_38_bytes(oracle):
    # deflate_test.mbt:12-23: write(11) + write(17) + close -> 38 bytes
    data = b"hello world" + b"hello again world"
    c = oracle.deflate(data, writes=[11, 17])
    assert len(c) == 38
    assert raw_inflate(c) == data
    # hand-derived literal code lengths for the 17..127-byte write_block_huff path
    freq = np.zeros(286, dtype=np.int32)
    for b in data:
        freq[b] += 1
    freq[256] = 1
    _, lens = oracle.huffman_generate(freq, 15)
    want = {" ": 3, "a": 4, "d": 4, "e": 4, "g": 5, "h": 4, "i": 5, "l": 2, "n": 5,
            "o": 3, "r": 4, "w": 4}
    for ch, ln in want.items():
        assert lens[ord(ch)] == ln, ch
    assert lens[256] == 5
    # SURVEY 8c vector (2): literal bits (incl. EOB) = 101
    assert int((freq * lens.astype(np.int64)).sum()) == 101


def test_64k_stream_layout(oracle):
    # 65536 bytes -> [dynamic block 65535][stored 1][stored 0 BFINAL]
    c, blocks = oracle.deflate(ramp(65536), with_blocks=True)
    assert [(k, n) for k, n, _, _ in blocks] == [(2, 65535), (0, 1), (0, 0)]
    assert c[-5:] == bytes([1, 0, 0, 0xFF, 0xFF])
    assert raw_inflate(c) == ramp(65536)


# ---- TestBestSpeed (deflate-fast_test.mbt:14-100) --------------------------------

TEST_CASES = [
    [65536, 0], [65536, 1], [65536, 1, 256], [65536, 1, 65536], [65536, 14], [65536, 15],
    [65536, 16], [65536, 16, 256], [65536, 16, 65536], [65536, 127], [65536, 128],
    [65536, 128, 256], [65536, 128, 65536], [65536, 129], [65536, 65536, 256],
    [65536, 65536, 65536],
]
FIRST_N = [1, 65534, 65535, 65536, 65537, 131072]


@pytest.mark.parametrize("first_n", FIRST_N)
def test_best_speed_round_trips(oracle, first_n):
    abcabc = ramp(131072)
    for tc in TEST_CASES:
        sizes = [first_n] + tc[1:]
        want = b"".join(abcabc[:n] for n in sizes)
        got_c = oracle.deflate(want, writes=sizes)
        # the write pattern must not influence the stream (window staging, deflate.mbt:222-294)
        assert got_c == oracle.deflate(want)
        assert raw_inflate(got_c) == want
        assert oracle.inflate(got_c, len(want)) == want


# ---- small-size policy (deflate.mbt:238-257) and misc ----------------------------

@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 18, 100, 127, 128, 129, 300, 4096])
def test_small_sizes(oracle, n):
    rng = np.random.default_rng(n)
    data = bytes(rng.integers(97, 105, n, dtype=np.uint8))
    c, blocks = oracle.deflate(data, with_blocks=True)
    assert raw_inflate(c) == data
    assert oracle.inflate(c, n) == data
    kinds = [k for k, _, _, _ in blocks]
    if n == 0:
        assert kinds == [0] and c == bytes([1, 0, 0, 0xFF, 0xFF])
    elif n <= 16:
        assert kinds == [0, 0]
    elif n < 128:
        assert kinds == [1, 0]
    else:
        assert kinds[0] in (1, 2) and kinds[-1] == 0


def test_never_stored_in_moonbit_mode_but_go_stores(oracle):
    # size/8 threshold is unreachable -> random data stays Huffman-coded
    rnd = bytes(np.random.default_rng(7).integers(0, 256, 65535, dtype=np.uint8))
    c_m, b_m = oracle.deflate(rnd, with_blocks=True)
    c_g, b_g = oracle.deflate(rnd, with_blocks=True, compat=oracle.COMPAT_GO)
    assert b_m[0][0] == 1 and b_g[0][0] == 0
    assert raw_inflate(c_m) == rnd and raw_inflate(c_g) == rnd


def test_compat_modes_agree_on_compressible_single_window(oracle):
    rng = np.random.default_rng(3)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(200)]
    data = b" ".join(words[int(i)] for i in rng.integers(0, 200, 14000))[:65536]
    assert oracle.deflate(data) == oracle.deflate(data, compat=oracle.COMPAT_GO)


def test_multi_window_divergence_d1_is_valid_deflate(oracle):
    # cross-window candidates give length-4 matches in MoonBit mode
    rng = np.random.default_rng(5)
    blk = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    data = (blk * 100)[:262144]
    c_m = oracle.deflate(data)
    c_g = oracle.deflate(data, compat=oracle.COMPAT_GO)
    assert raw_inflate(c_m) == data and raw_inflate(c_g) == data
    assert oracle.inflate(c_m, len(data)) == data
    assert len(c_g) <= len(c_m)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_round_trip(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(0, 200000))
    kind = seed % 4
    if kind == 0:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    elif kind == 1:
        data = bytes(rng.integers(0, 4, n, dtype=np.uint8))
    elif kind == 2:
        data = (bytes(rng.integers(0, 256, 97, dtype=np.uint8)) * (n // 97 + 1))[:n]
    else:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        a[rng.integers(0, 2, n) == 0] = 0
        data = bytes(a)
    for compat in (oracle.COMPAT_MOONBIT, oracle.COMPAT_GO):
        c = oracle.deflate(data, compat=compat)
        assert raw_inflate(c) == data
        assert oracle.inflate(c, n) == data


# ---- huffman code construction ----------------------------------------------------

def _kraft_ok(lens):
    nz = lens[lens > 0].astype(np.int64)
    return nz.size <= 2 or int((1 << (15 - nz)).sum()) == (1 << 15)


@pytest.mark.parametrize("seed", range(8))
def test_huffman_generate_properties(oracle, seed):
    rng = np.random.default_rng(seed)
    n = [286, 30, 19][seed % 3]
    max_bits = 7 if n == 19 else 15
    freq = rng.integers(0, 5, n).astype(np.int32) * rng.integers(0, 1000, n).astype(np.int32)
    if seed >= 4:  # skewed, forces the length limit
        freq = (2 ** rng.integers(0, 24, n)).astype(np.int32)
    codes, lens = oracle.huffman_generate(freq, max_bits)
    assert (lens[freq == 0] == 0).all()
    assert (lens[freq > 0] >= 1).all() and lens.max() <= max_bits
    assert _kraft_ok(lens)
    # canonical: within one length, codes (un-reversed) increase with the symbol
    for l in range(1, max_bits + 1):
        idx = np.nonzero(lens == l)[0]
        vals = [oracle.lib().orc_reverse_bits(int(codes[i]), l) for i in idx]
        assert vals == sorted(vals)


# ---- inflate error paths (inflate.mbt:38,377,438-444,677) -------------------------

def test_inflate_errors(oracle):
    rc, out, used, off = oracle.inflate(bytes([0x07]), 10, full=True)  # BTYPE=3
    assert rc == oracle.E_CORRUPT and off == 1
    rc, *_ = oracle.inflate(b"", 10, full=True)
    assert rc == oracle.E_UNEXPECTED_EOF
    good = oracle.deflate(b"abcdefgh" * 100)
    rc, *_ = oracle.inflate(good[:-3], 1000, full=True)
    assert rc == oracle.E_UNEXPECTED_EOF
    rc, *_ = oracle.inflate(bytes([1, 5, 0, 0, 0]), 10, full=True)  # LEN/NLEN mismatch
    assert rc == oracle.E_CORRUPT
    # distance beyond history: fixed block, first symbol is a match
    rc, *_ = oracle.inflate(bytes([0x03, 0x02, 0x00]), 10, full=True)
    assert rc in (oracle.E_CORRUPT, oracle.E_UNEXPECTED_EOF)


def test_inflate_accepts_zlib_streams(oracle):
    # stored / fixed / dynamic blocks produced by an independent encoder
    rng = np.random.default_rng(11)
    text = (b"the quick brown fox jumps over the lazy dog " * 400)
    for level, data in ((0, text), (1, b"abc"), (6, text), (9, bytes(rng.integers(0, 8, 50000, dtype=np.uint8)))):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(data) + co.flush()
        assert oracle.inflate(c, len(data)) == data


def test_oracle_inflate_batch_matches_single_stream_decoder(oracle):
    """orc_inflate_batch (threads) == orc_inflate_stream per stream, including the error codes."""
    import numpy as np
    from util import make_streams
    data, off = make_streams([("text", 20000), ("ramp", 70000), ("zero", 300), ("rand", 5000), ("text", 0)], seed=5)
    blobs = [oracle.deflate(data[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    blobs.append(blobs[0][:-9])           # truncated -> unexpected EOF
    blobs.append(b"\x07" + blobs[1][1:])  # block type 3 -> corrupt
    sizes = [int(off[i + 1] - off[i]) for i in range(len(off) - 1)] + [20000, 70000]
    coff = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum([len(b) for b in blobs], out=coff[1:])
    comp = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    out, ooff, olen, status = oracle.inflate_batch(comp, coff, sizes, nthreads=3)
    for i, blob in enumerate(blobs):
        rc, res, _, _ = oracle.inflate(blob, sizes[i], full=True)
        assert status[i] == rc
        assert olen[i] == len(res)
        assert bytes(out[int(ooff[i]):int(ooff[i]) + len(res)]) == res
    assert list(status[:5]) == [0] * 5 and status[5] == oracle.E_UNEXPECTED_EOF and status[6] == oracle.E_CORRUPT


def test_shift_offsets_branches_with_lowered_buffer_reset(oracle):
    """deflate-fast.mbt:130-132,366-389: when `cur` reaches buffer_reset (window 32 766 of a Writer)
    shift_offsets runs.  MoonBit's `prev` is always empty, so the table is CLEARED
    (:367-374): the window after it starts without history.  In Go's semantics `prev` holds the last
    window and the offsets only move down: no distance changes.  The test hook lowers buffer_reset
    so that both branches run within a few windows."""
    from util import make_streams, raw_inflate
    W = 65535
    n = 8 * W + 2000
    data, _ = make_streams([("text", n)], seed=31)
    data = data[:n]
    plain = {c: oracle.deflate(data, compat=c) for c in (oracle.COMPAT_MOONBIT, oracle.COMPAT_GO)}
    try:
        oracle.set_buffer_reset(3 * W)  # cur = 65535 (k + 1) at window k: shifts at windows 2, 5, 8
        # `cur` after the shift is max_match_offset + 1 (:372,388)
        df = oracle.DeflateFast(oracle.COMPAT_MOONBIT)
        curs = []
        for k in range(7):
            df.encode(data[k * W:(k + 1) * W])
            curs.append(df.cur)
        assert curs == [2 * W, 3 * W, 32769 + W, 32769 + 2 * W, 32769 + 3 * W, 32769 + W, 32769 + 2 * W]
        low = {c: oracle.deflate(data, compat=c) for c in (oracle.COMPAT_MOONBIT, oracle.COMPAT_GO)}
    finally:
        oracle.set_buffer_reset(0)
    for c in low:
        assert raw_inflate(low[c]) == data.tobytes()
    # Go: the shift is invisible.  MoonBit: windows 2 and 5 lose the candidates of the window before
    assert low[oracle.COMPAT_GO] == plain[oracle.COMPAT_GO]
    assert low[oracle.COMPAT_MOONBIT] != plain[oracle.COMPAT_MOONBIT]
    # the first two windows' blocks are untouched by the first shift
    _, blocks_plain = oracle.deflate(data, with_blocks=True)
    try:
        oracle.set_buffer_reset(3 * W)
        _, blocks_low = oracle.deflate(data, with_blocks=True)
    finally:
        oracle.set_buffer_reset(0)
    assert blocks_low[:2] == blocks_plain[:2] and blocks_low[2][2] != blocks_plain[2][2]
