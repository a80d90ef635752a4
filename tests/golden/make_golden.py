"""Regenerates tests/golden/vectors.json.  Two kinds of vectors (data only, no source text):

  reference_kat   -- the known answers the reference's own tests hold for this path
                     (file:line of each in the entry), typed in by hand;
  oracle_streams  -- inputs (small, seeded) with the compressed bytes of the CPU oracle in both
                     compat modes, the spliced form, and the match tokens of the first window.

The reference itself cannot run in the build image (MoonBit, no moon/go), so the second kind pins
the *oracle's* output at the commit that passed every reference_kat: a later change of the oracle,
or of the HIP path, that alters a bit shows up against these bytes (tests/test_golden.py) even
where no reference test holds the answer.

    python tests/golden/make_golden.py        # rewrites vectors.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import pyoracle  # noqa: E402
from util import make_streams  # noqa: E402

REFERENCE_KAT = {
    "token": {"where": "token.mbt:95-99", "token": 2143289471, "offset": 127, "length_minus_3": 255},
    "reverse16": {"where": "bits.mbt:24-27", "in": 32768, "out": 1},
    "reverse_bits": {"where": "huffman-code.mbt:289-292", "in": 64, "bits": 7, "out": 1},
    "hello": {"where": "deflate_test.mbt:12-35", "writes": ["hello world", "hello again world"],
              "compressed_len": 38},
    "ramp_tokens": {"where": "deflate-fast_test.mbt:15-24 data; SURVEY 8c trace", "window": 65535,
                    "literals": 129, "tokens_129_131": [2143289471, 2143289727, 2143289983]},
    "hello_code_lengths": {"where": "SURVEY 8c (2)", "lengths": {" ": 3, "a": 4, "d": 4, "e": 4, "g": 5,
                                                              "h": 4, "i": 5, "l": 2, "n": 5, "o": 3,
                                                              "r": 4, "w": 4, "EOB": 5},
                           "literal_bits": 101, "num_codegens": 18},
    "size_matrix": {"where": "deflate-fast_test.mbt:27-49",
                    "sizes": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                              65534, 65535, 65536, 65537, 131070, 131071, 131072, 131073]},
}

SPECS = [("text", 0), ("text", 1), ("text", 16), ("text", 17), ("text", 127), ("text", 128), ("text", 700),
         ("ramp", 1000), ("zero", 300), ("rand", 200), ("period", 900), ("runs", 800), ("low", 1200),
         ("text", 65536), ("ramp", 65600), ("text", 66000)]


def main():
    data, off = make_streams(SPECS, seed=2024)
    streams = []
    for i, (kind, n) in enumerate(SPECS):
        raw = data[int(off[i]):int(off[i + 1])]
        ent = {"kind": kind, "len": n, "moonbit": pyoracle.deflate(raw).hex(),
               "go": pyoracle.deflate(raw, compat=pyoracle.COMPAT_GO).hex()}
        if n <= 2000:
            ent["input"] = raw.tobytes().hex()
        toks = pyoracle.DeflateFast().encode(raw[:65535]) if n >= 128 else np.zeros(0, np.uint32)
        ent["first_window_tokens"] = int(toks.size)
        ent["first_window_matches"] = [int(t) for t in toks if t >> 30][:8]
        streams.append(ent)
    spliced, bit_off = pyoracle.deflate_spliced(data, off)
    out = {"reference_kat": REFERENCE_KAT,
           "oracle_streams": {"generator": "tests/util.make_streams(SPECS, seed=2024)",
                              "specs": [list(s) for s in SPECS], "streams": streams,
                              "spliced": spliced.hex(), "bit_off": [int(x) for x in bit_off]}}
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote vectors.json:", os.path.getsize(os.path.join(HERE, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
