"""CPU-side checks of the product library: it loads, exports every symbol the header declares,
and its host-only entry points behave (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import flate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    flate.build()
    from importlib import import_module
    return import_module("moonbit-flate_amd._lib").load()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "flate_hip.h")).read()
    declared = set(re.findall(r"\b(flate_hip_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("flate_hip_ctx")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), name
    listed = set(__import__("importlib").import_module("moonbit-flate_amd._lib").EXPORTS)
    assert declared == listed


def test_strerror_and_bound(lib):
    assert lib.flate_hip_strerror(0) == b"ok"
    assert b"CPU" in lib.flate_hip_strerror(-5)
    assert flate.deflate_bound(0) >= 5
    assert flate.deflate_bound(65536) >= 65536 + 10


def test_checksum_batch_refuses_bad_arguments_before_it_touches_a_device(lib):
    off = (C.c_uint64 * 2)(0, 4)
    out = (C.c_uint32 * 1)()
    buf = (C.c_uint8 * 4)(1, 2, 3, 4)
    assert lib.flate_hip_checksum_batch(None, buf, off, 1, 1, out, 0) == -1      # no ctx
    # (kind, offsets and pointers are checked in front of any HIP call as well; that needs a ctx, i.e. a GPU:
    # tests/test_checksum.py)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(flate.FlateError) as ei:
        flate.FlateEngine(0)
    assert ei.value.code == -5


def test_synth_is_reproducible_and_thread_independent():
    a = flate.synth("text", 8, 4096, nthreads=1)
    b = flate.synth("text", 8, 4096, nthreads=4)
    assert np.array_equal(a, b)
    c = flate.synth("text", 4, 4096, first_stream=4)
    assert np.array_equal(a[4 * 4096:], c)
    r = flate.synth("ramp", 2, 300)
    assert r[:300].tolist() == [i & 127 for i in range(300)] and r[300] == 0
    assert flate.synth("zero", 1, 100).sum() == 0
    rd = flate.synth("rand", 1, 65536)
    assert 120 < rd.mean() < 136


def test_synth_text_ratio_in_band(oracle):
    # headline workload: deflate-fast ratio of S-text should sit between 2 and 3
    a = flate.synth("text", 4, 65536)
    tot = sum(len(oracle.deflate(a[i * 65536:(i + 1) * 65536])) for i in range(4))
    assert 2.0 < 4 * 65536 / tot < 3.0


def test_tokens_from_matches_roundtrip(oracle):
    data = flate.synth("text", 1, 20000)
    toks = oracle.DeflateFast().encode(data)
    # rebuild match records from the oracle tokens and expand again
    pos, tok, p = [], [], 0
    for t in toks:
        if t >= (1 << 30):
            pos.append(p)
            tok.append(t)
            p += ((int(t) >> 22) & 0xFF) + 3
        else:
            p += 1
    got = flate.tokens_from_matches(data, np.array(pos, np.uint32), np.array(tok, np.uint32))
    assert np.array_equal(got, toks)


def test_moonbit_stub_compiles_and_links_against_the_library(lib, tmp_path):
    """integration/moonbit/flate_hip_stub.c (the C side of the MoonBit binding) stays in
    step with include/flate_hip.h: it compiles warning-free and its symbols resolve against the .so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = tmp_path / "libstub.so"
    libdir = os.path.join(root, "moonbit-flate_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC",
                           "-I" + os.path.join(root, "include"),
                           os.path.join(root, "integration", "moonbit", "flate_hip_stub.c"),
                           "-L" + libdir, "-lflate_hip", "-Wl,-rpath," + libdir, "-Wl,--no-undefined",
                           "-o", str(so)])
    import ctypes
    stub = ctypes.CDLL(str(so))
    for name in ("flate_hip_mbt_ctx_new", "flate_hip_mbt_ctx_is_null",
                 "flate_hip_mbt_comm_new", "flate_hip_mbt_comm_is_null", "flate_hip_mbt_stream_new",
                 "flate_hip_mbt_stream_is_null", "flate_hip_mbt_inflate_sizes", "flate_hip_mbt_inflate_stream_new",
                 "flate_hip_mbt_inflate_stream_is_null", "flate_hip_mbt_inflate_stream_read"):
        assert hasattr(stub, name)
    # every symbol the .mbt binding names in an `extern "C" fn ... = "sym"` exists in the stub or
    # in the library, and the binding never copies a buffer on its way to C
    import re
    mbt = open(os.path.join(root, "integration", "moonbit", "flate_hip_native.mbt")).read()
    code = "\n".join(ln for ln in mbt.splitlines() if not ln.lstrip().startswith("//"))
    syms = re.findall(r'\)\s*->\s*\w+\s*=\s*"(\w+)"', code)
    assert len(syms) >= 7
    for name in syms:
        assert hasattr(stub, name) or hasattr(lib, name), name
    assert "from_fixedarray" not in code
