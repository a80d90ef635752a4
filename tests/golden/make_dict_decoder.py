"""Decoder-side known answer of the reference (dict-decoder_wbtest.mbt:9-291) as DEFLATE streams.

tests/golden/dict_decoder.json holds the test's data (poem, the 166 (dist, len) references, the
ABC / fox strings).  This script replays the test's script of insertions and copies into (a) the
expected text, built exactly as the reference test builds `want` (:231-283), and (b) a list of
LZ77 operations, which a tiny fixed-Huffman writer (RFC 1951 3.2.6; test-side code, not the
product's encoder) turns into DEFLATE streams:

  poem        -- the script as it stands (copies with dist < len, dist == everything written so
                 far, dist == window size of the test's 2 KiB DictDecoder)
  poem_wrap   -- the same script behind 32000 bytes of literals, so that its copies straddle the
                 32 KiB history wrap of a DEFLATE decoder (dict-decoder.mbt:114-185)

    python tests/golden/make_dict_decoder.py      # adds "streams" to dict_decoder.json
"""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "dict_decoder.json")

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
            131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
             2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, n):          # LSB-first field
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):         # Huffman code: most significant bit first
        self.put(int(format(c, "0%db" % n)[::-1], 2), n)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def fixed_lit(b, sym):
    if sym < 144:
        b.code(0x30 + sym, 8)
    elif sym < 256:
        b.code(0x190 + sym - 144, 9)
    elif sym < 280:
        b.code(sym - 256, 7)
    else:
        b.code(0xC0 + sym - 280, 8)


def fixed_stream(ops):
    """ops: ('lit', bytes) | ('copy', dist, len with 3 <= len <= 258).  One fixed block, BFINAL=1."""
    b = Bits()
    b.put(1, 1)
    b.put(1, 2)
    for op in ops:
        if op[0] == "lit":
            for ch in op[1]:
                fixed_lit(b, ch)
        else:
            _, dist, ln = op
            lc = max(i for i in range(29) if LEN_BASE[i] <= ln)
            fixed_lit(b, 257 + lc)
            b.put(ln - LEN_BASE[lc], LEN_EXTRA[lc])
            dc = max(i for i in range(30) if DIST_BASE[i] <= dist)
            b.code(dc, 5)
            b.put(dist - DIST_BASE[dc], DIST_EXTRA[dc])
    fixed_lit(b, 256)
    return b.done()


def split_copy(dist, ln):
    """A copy of any length as DEFLATE matches (3..258 each) of the same distance."""
    out = []
    while ln > 0:
        c = min(ln, 258)
        if 0 < ln - c < 3:
            c -= 3 - (ln - c)
        out.append(("copy", dist, c))
        ln -= c
    return out


def script(d, prefix=b""):
    """Returns (ops, want): the reference test's script (:213-283) and its expected text."""
    poem, abc, fox = d["poem"].encode(), d["abc"].encode(), d["fox"].encode()
    window = d["window"]
    ops, want, written = [], bytearray(prefix), [len(prefix)]
    if prefix:
        ops.append(("lit", prefix))

    def lit(s):
        ops.append(("lit", s))
        written[0] += len(s)

    def copy(dist, ln):
        ops.extend(split_copy(dist, ln))
        written[0] += ln

    def hist_size():  # dict-decoder.mbt: bytes in the window
        return min(written[0] - len(prefix), window)

    lit(b".")
    want += b"."
    pos = 0
    for dist, ln in d["poem_refs"]:
        if dist == 0:
            lit(poem[pos:pos + ln])
        else:
            copy(dist, ln)
        pos += ln
    assert pos == len(poem)
    want += poem
    copy(hist_size(), 33)
    want += want[len(prefix):len(prefix) + 33]
    lit(abc)
    copy(len(abc), 59 * len(abc))
    want += abc * 60
    lit(fox)
    copy(len(fox), 9 * len(fox))
    want += fox * 10
    lit(b".")
    copy(1, 9)
    want += b"." * 10
    up = poem.upper()
    lit(up)
    copy(len(poem), 7 * len(poem))
    want += up * 8
    copy(hist_size(), 10)
    to_drop = len(want) - len(prefix) - window
    want += want[len(prefix) + to_drop:len(prefix) + to_drop + 10]
    return ops, bytes(want)


def replay(ops):
    out = bytearray()
    for op in ops:
        if op[0] == "lit":
            out += op[1]
        else:
            _, dist, ln = op
            assert 1 <= dist <= len(out) and dist <= 32768 and 3 <= ln <= 258
            for _ in range(ln):
                out.append(out[-dist])
    return bytes(out)


def filler(n):
    x, out = 0x5EED0003, bytearray()
    while len(out) < n:
        x = (x * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        out.append(97 + (x >> 59) % 26 if (x >> 40) % 7 else 32)
    return bytes(out)


def build(d):
    streams = {}
    for name, prefix in (("poem", b""), ("poem_wrap", filler(32000))):
        ops, want = script(d, prefix)
        assert replay(ops) == want, name  # the script and the reference's `want` agree
        comp = fixed_stream(ops)
        streams[name] = {"deflate_hex": comp.hex(), "out_len": len(want),
                         "out_sha256": hashlib.sha256(want).hexdigest(),
                         "matches": sum(1 for o in ops if o[0] == "copy")}
    return streams


def expected(d, name):
    return script(d, b"" if name == "poem" else filler(32000))[1]


if __name__ == "__main__":
    d = json.load(open(PATH))
    d["streams"] = build(d)
    json.dump(d, open(PATH, "w"), indent=1)
    import zlib
    for k, v in d["streams"].items():  # an independent inflater agrees
        assert zlib.decompressobj(-15).decompress(bytes.fromhex(v["deflate_hex"])) == expected(d, k)
        print(k, v["out_len"], "bytes,", v["matches"], "matches,", len(v["deflate_hex"]) // 2, "compressed")
