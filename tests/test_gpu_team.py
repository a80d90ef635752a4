"""GPU parity of the two-wavefronts-per-stream match finder (lz77_team_kernels.hip): same token
streams and bytes as the oracle, for the one-block-per-stream launch and for the persistent
resident + guest team launch (shared queue), single- and multi-window streams, both compat modes."""
import numpy as np
import pytest

from util import flate, make_streams
from test_gpu_parity import (SINGLE_WINDOW, MULTI_WINDOW, EDGE_SIZES, _check_tokens, _check_streams)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def team():
    flate.build()
    e = flate.FlateEngine(0)
    e.set_option("lz_team", 1)
    yield e
    e.close()


def test_team_tokens_single_window(team, oracle):
    _check_tokens(team, oracle, SINGLE_WINDOW, False)


def test_team_tokens_multi_window(team, oracle):
    _check_tokens(team, oracle, MULTI_WINDOW, False)
    _check_tokens(team, oracle, MULTI_WINDOW, False, compat_go=True)


def test_team_streams_edge_sizes_and_kinds(team, oracle):
    _check_streams(team, oracle, [("text", n) for n in EDGE_SIZES])
    _check_streams(team, oracle, SINGLE_WINDOW + MULTI_WINDOW)
    _check_streams(team, oracle, SINGLE_WINDOW + MULTI_WINDOW, compat_go=True)


def test_team_fuzz_random_lengths(team, oracle):
    rng = np.random.default_rng(4242)
    kinds = ["text", "low", "period", "runs", "rand", "zero", "ramp"]
    specs = [(kinds[int(rng.integers(0, len(kinds)))], int(rng.integers(0, 150000))) for _ in range(80)]
    _check_streams(team, oracle, specs)


@pytest.mark.parametrize("resident,guests", [(4, 8), (1, 16), (8, 0)])
def test_team_persistent_resident_and_guest_teams(oracle, resident, guests):
    e = flate.FlateEngine(0)
    try:
        e.set_option("lz_team", 1)
        e.set_option("guest_min_streams", 1)
        e.set_option("team_resident_blocks", resident)
        e.set_option("team_guest_blocks", guests)
        _check_streams(e, oracle, SINGLE_WINDOW + MULTI_WINDOW + [("text", 65536)] * 40)
        _check_streams(e, oracle, [("text", 65536), ("runs", 65536), ("low", 65536), ("period", 65536)] * 16)
    finally:
        e.close()


def test_team_batch_2k_streams_full_size(team, oracle):
    n = 2048
    data = flate.synth("text", n, 65536)
    off = flate.uniform_offsets(n, 65536)
    out, out_off = team.deflate_batch(data, off)
    o_out, o_off, o_len = oracle.deflate_batch(data, off, nthreads=8)
    for i in range(n):
        a = out[int(out_off[i]):int(out_off[i + 1])]
        b = o_out[int(o_off[i]):int(o_off[i]) + int(o_len[i])]
        assert a.size == b.size and np.array_equal(a, b), i
