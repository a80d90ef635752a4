"""Fuzzes the lane-accurate host model of the wave64 match finder against the oracle (CPU only).
The HIP kernel is a transcription of this model; GPU parity is checked in test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util import flate, make_streams, oracle_tokens_per_chunk

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_model", "lz77_wave_model.cpp")
LIB = os.path.join(HERE, "host_model", "liblz77_wave_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB])
    L = C.CDLL(LIB)
    L.model_lz77.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.model_lz77.restype = C.c_int
    return L


def run_model(model, sb, compat_go=False, slot_tags=False):
    sb = np.ascontiguousarray(sb, dtype=np.uint8)
    nch = len(flate.lz_chunks(sb.size))
    recs = np.zeros((max(nch, 1) * 16384, 2), dtype=np.uint32)
    nm = np.zeros(max(nch, 1), dtype=np.uint32)
    stats = np.zeros(8, dtype=np.uint64)
    pad = np.concatenate([sb, np.zeros(64, np.uint8)])  # the model may read own16 near the end
    got = model.model_lz77(pad.ctypes.data, sb.size, (1 if compat_go else 0) | (2 if slot_tags else 0), recs.ctypes.data,
                           nm.ctypes.data, stats.ctypes.data)
    assert got == nch
    out = []
    for c in range(nch):
        r = recs[c * 16384:c * 16384 + int(nm[c])]
        out.append((r[:, 0].copy(), r[:, 1].copy()))
    return out, stats


def check(model, oracle, specs, compat_go=False, seed=1234, slot_tags=False):
    data, off = make_streams(specs, seed=seed)
    tot = np.zeros(8, dtype=np.uint64)
    for i, (kind, n) in enumerate(specs):
        sb = data[int(off[i]):int(off[i + 1])]
        chunks, stats = run_model(model, sb, compat_go, slot_tags)
        tot += stats
        want = oracle_tokens_per_chunk(oracle, sb, compat=1 if compat_go else 0)
        for k, ((start, cn), w) in enumerate(zip(flate.lz_chunks(n), want)):
            pos, tok = chunks[k]
            got = flate.tokens_from_matches(sb[start:start + cn], pos, tok)
            assert got.size == w.size, (kind, n, start, got.size, w.size)
            bad = np.nonzero(got != w)[0]
            assert bad.size == 0, (kind, n, start, int(bad[0]), hex(got[bad[0]]), hex(w[bad[0]]))
    return tot


KINDS = ["text", "low", "period", "runs", "rand", "zero", "ramp"]


def test_model_single_window(model, oracle):
    specs = [(k, n) for k in KINDS for n in (128, 129, 300, 5000, 40000, 65535, 65536)]
    st = check(model, oracle, specs)
    assert st[0] > 0 and st[2] > 0


def test_model_multi_window(model, oracle):
    specs = [(k, n) for k in KINDS for n in (65535 + 128, 131072, 200000)]
    check(model, oracle, specs)
    check(model, oracle, specs, compat_go=True)


@pytest.mark.parametrize("seed", range(6))
def test_model_fuzz(model, oracle, seed):
    rng = np.random.default_rng(seed)
    specs = [(KINDS[int(rng.integers(0, len(KINDS)))], int(rng.integers(128, 140000)))
             for _ in range(25)]
    check(model, oracle, specs, seed=seed, compat_go=bool(seed & 1))


def test_model_batches_amortise_events(model, oracle):
    # the point of dense batches: several matches per table/candidate round trip on text
    st = check(model, oracle, [("text", 65536)] * 4)
    dense, sparse, events = int(st[0]), int(st[1]), int(st[2])
    assert events / (dense + sparse) > 3.0, (dense, sparse, events)


@pytest.mark.parametrize("seed", range(3))
def test_model_slot_tags_change_nothing(model, oracle, seed):
    # the guest blocks' 2-bit slot tags (lz77_kernels.hip: tag_of / tag_set) only skip slot reads
    # that could not have produced a match: same tokens as the oracle with the filter on
    rng = np.random.default_rng(100 + seed)
    specs = [(KINDS[int(rng.integers(0, len(KINDS)))], int(rng.integers(128, 140000))) for _ in range(20)]
    specs += [(k, 65536) for k in KINDS]
    check(model, oracle, specs, seed=seed, slot_tags=True)
    check(model, oracle, specs[:8], compat_go=True, seed=seed, slot_tags=True)
