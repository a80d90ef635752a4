"""GPU parity of the pipelined one-wave match finder (lz77_pipe_kernels.hip): same token streams
and bytes as the oracle, for the one-block-per-stream launch and for the persistent resident + guest
launch (shared queue), single- and multi-window streams, both compat modes."""
import numpy as np
import pytest

from util import flate, make_streams
from test_gpu_parity import (SINGLE_WINDOW, MULTI_WINDOW, EDGE_SIZES, _check_tokens, _check_streams)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe():
    flate.build()
    e = flate.FlateEngine(0)
    e.set_option("lz_pipe", 1)
    yield e
    e.close()


def test_pipe_tokens_single_window(pipe, oracle):
    _check_tokens(pipe, oracle, SINGLE_WINDOW, False)


def test_pipe_tokens_multi_window(pipe, oracle):
    _check_tokens(pipe, oracle, MULTI_WINDOW, False)
    _check_tokens(pipe, oracle, MULTI_WINDOW, False, compat_go=True)


def test_pipe_streams_edge_sizes_and_kinds(pipe, oracle):
    _check_streams(pipe, oracle, [("text", n) for n in EDGE_SIZES])
    _check_streams(pipe, oracle, SINGLE_WINDOW + MULTI_WINDOW)
    _check_streams(pipe, oracle, SINGLE_WINDOW + MULTI_WINDOW, compat_go=True)


def test_pipe_fuzz_random_lengths(pipe, oracle):
    rng = np.random.default_rng(777)
    kinds = ["text", "low", "period", "runs", "rand", "zero", "ramp"]
    specs = [(kinds[int(rng.integers(0, len(kinds)))], int(rng.integers(0, 150000))) for _ in range(80)]
    _check_streams(pipe, oracle, specs)


@pytest.mark.parametrize("resident,guests", [(4, 8), (1, 16)])
def test_pipe_persistent_resident_and_guest_blocks(oracle, resident, guests):
    e = flate.FlateEngine(0)
    try:
        e.set_option("lz_pipe", 1)
        e.set_option("guest_min_streams", 1)
        e.set_option("resident_blocks", resident)
        e.set_option("guest_blocks", guests)
        _check_streams(e, oracle, SINGLE_WINDOW + MULTI_WINDOW + [("text", 65536)] * 40)
        _check_streams(e, oracle, [("text", 65536), ("runs", 65536), ("low", 65536), ("period", 65536)] * 16)
    finally:
        e.close()


def test_pipe_batch_2k_streams_full_size(pipe, oracle):
    n = 2048
    data = flate.synth("text", n, 65536)
    off = flate.uniform_offsets(n, 65536)
    out, out_off = pipe.deflate_batch(data, off)
    o_out, o_off, o_len = oracle.deflate_batch(data, off, nthreads=8)
    for i in range(n):
        a = out[int(out_off[i]):int(out_off[i + 1])]
        b = o_out[int(o_off[i]):int(o_off[i]) + int(o_len[i])]
        assert a.size == b.size and np.array_equal(a, b), i
