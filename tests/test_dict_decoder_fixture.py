"""The reference's only decoder-side known answer (dict-decoder_wbtest.mbt:9-291, test
"DictDecoder"): its data is committed in tests/golden/dict_decoder.json, its script of insertions
and copies is encoded as fixed-Huffman DEFLATE streams by tests/golden/make_dict_decoder.py, and
every decoder here -- the oracle's inflater on CPU, both GPU inflate kernels -- must reproduce the
text the reference test expects (`want`, :231-283): copies with dist < len (RLE), dist == all the
bytes written so far, dist == the window size, and (poem_wrap) copies across the 32 KiB wrap."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from util import flate

HERE = os.path.dirname(os.path.abspath(__file__))
D = json.load(open(os.path.join(HERE, "golden", "dict_decoder.json")))
_spec = importlib.util.spec_from_file_location("make_dict_decoder",
                                               os.path.join(HERE, "golden", "make_dict_decoder.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)
NAMES = sorted(D["streams"])


def test_fixture_is_what_its_generator_makes():
    assert len(D["poem_refs"]) == 166 and sum(l for _, l in D["poem_refs"]) == len(D["poem"]) == 763
    assert gen.build(D) == D["streams"]
    for name in NAMES:
        want = gen.expected(D, name)
        assert len(want) == D["streams"][name]["out_len"]
        assert hashlib.sha256(want).hexdigest() == D["streams"][name]["out_sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_inflater_reproduces_the_reference_text(oracle, name):
    s = D["streams"][name]
    got = oracle.inflate(bytes.fromhex(s["deflate_hex"]), s["out_len"])
    assert got == gen.expected(D, name)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["wave_per_stream", "lane_per_stream"])
def test_gpu_inflaters_reproduce_the_reference_text(kernel):
    flate.build()
    eng = flate.FlateEngine(0)
    eng.set_option("inflate_simt_min_streams", 0 if kernel == "lane_per_stream" else 1 << 30)
    # each fixture stream several times in one batch (lanes of one wavefront at different phases)
    order = [NAMES[i % len(NAMES)] for i in range(70)]
    blobs = [bytes.fromhex(D["streams"][k]["deflate_hex"]) for k in order]
    off = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum([len(b) for b in blobs], out=off[1:])
    data = np.frombuffer(b"".join(blobs) + b"\0" * 8, dtype=np.uint8).copy()
    sizes = [D["streams"][k]["out_len"] for k in order]
    out, ooff, olen, status, _ = eng.inflate_batch(data, off, sizes)
    assert (status == 0).all() and olen.tolist() == sizes
    for i, k in enumerate(order):
        assert bytes(out[int(ooff[i]):int(ooff[i]) + sizes[i]]) == gen.expected(D, k), (kernel, i, k)
    eng.close()
