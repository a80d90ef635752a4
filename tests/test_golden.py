"""Committed fixtures (tests/golden/vectors.json, made by tests/golden/make_golden.py):
the reference's own known answers, and oracle-pinned compressed bytes that both the oracle and the
HIP path must keep reproducing."""
import json
import os

import numpy as np
import pytest

from util import flate, make_streams

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")))
KAT, OS = G["reference_kat"], G["oracle_streams"]
SPECS = [tuple(s) for s in OS["specs"]]


def _inputs():
    data, off = make_streams(SPECS, seed=2024)
    for i, ent in enumerate(OS["streams"]):       # the generator is part of the fixture: check it
        if "input" in ent:
            assert data[int(off[i]):int(off[i + 1])].tobytes().hex() == ent["input"]
    return data, off


def test_reference_known_answers_hold_for_the_oracle(oracle):
    L = oracle.lib()
    t = KAT["token"]
    assert L.orc_token_offset(t["token"]) == t["offset"] and L.orc_token_length(t["token"]) == t["length_minus_3"]
    assert L.orc_reverse16(KAT["reverse16"]["in"]) == KAT["reverse16"]["out"]
    assert L.orc_reverse_bits(KAT["reverse_bits"]["in"], KAT["reverse_bits"]["bits"]) == KAT["reverse_bits"]["out"]
    h = KAT["hello"]
    joined = "".join(h["writes"]).encode()
    assert len(oracle.deflate(joined, writes=[len(w) for w in h["writes"]])) == h["compressed_len"]
    r = KAT["ramp_tokens"]
    toks = oracle.DeflateFast().encode(bytes(i & 127 for i in range(r["window"])))
    assert (toks[:r["literals"]] < 256).all() and toks[129:132].tolist() == r["tokens_129_131"]
    for n in KAT["size_matrix"]["sizes"]:          # the reference's round-trip size matrix
        raw = bytes(i & 127 for i in range(n))
        assert oracle.inflate(oracle.deflate(raw), n) == raw


def test_oracle_reproduces_the_pinned_streams(oracle):
    data, off = _inputs()
    for i, ent in enumerate(OS["streams"]):
        raw = data[int(off[i]):int(off[i + 1])]
        assert oracle.deflate(raw).hex() == ent["moonbit"], i
        assert oracle.deflate(raw, compat=oracle.COMPAT_GO).hex() == ent["go"], i
        if ent["len"] >= 128:
            toks = oracle.DeflateFast().encode(raw[:65535])
            assert toks.size == ent["first_window_tokens"]
            assert [int(t) for t in toks if t >> 30][:8] == ent["first_window_matches"]
    spliced, bit_off = oracle.deflate_spliced(data, off)
    assert spliced.hex() == OS["spliced"] and [int(x) for x in bit_off] == OS["bit_off"]


@pytest.fixture(scope="module")
def eng():
    flate.build()
    e = flate.FlateEngine(0)
    yield e
    e.close()


@pytest.mark.gpu
def test_hip_path_reproduces_the_pinned_streams(eng):
    """No oracle here: the HIP path against the committed bytes."""
    data, off = _inputs()
    for key, go in (("moonbit", False), ("go", True)):
        out, ooff = eng.deflate_batch(data, off, compat_go=go)
        for i, ent in enumerate(OS["streams"]):
            assert bytes(out[int(ooff[i]):int(ooff[i + 1])]).hex() == ent[key], (key, i)
    one, n, bit_off = eng.deflate_spliced(data, off)
    assert bytes(one[:n]).hex() == OS["spliced"] and [int(x) for x in bit_off] == OS["bit_off"]
    # ... and back: both decoders on the pinned bytes, and the indexed decode of the pinned splice
    blobs = [bytes.fromhex(ent["moonbit"]) for ent in OS["streams"]]
    coff = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum([len(b) for b in blobs], out=coff[1:])
    comp = np.frombuffer(b"".join(blobs) + b"\0" * 8, dtype=np.uint8).copy()
    sizes = [s[1] for s in SPECS]
    for simt_min in (0, 1 << 30):
        eng.set_option("inflate_simt_min_streams", simt_min)
        back, _, olen, status, _ = eng.inflate_batch(comp, coff, sizes)
        assert (status == 0).all() and list(olen) == sizes
        assert bytes(back[:int(off[-1])]) == data[:int(off[-1])].tobytes()
    sp = np.frombuffer(bytes.fromhex(OS["spliced"]) + b"\0" * 8, dtype=np.uint8).copy()
    back, _, olen, status, _ = eng.inflate_spliced(sp, len(sp) - 8, np.array(OS["bit_off"], np.uint64), sizes)
    assert (status == 0).all() and bytes(back[:int(off[-1])]) == data[:int(off[-1])].tobytes()
