"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
must reproduce the oracle (CPU restatement of the reference) bit for bit."""
import numpy as np
import pytest

from util import flate, make_streams, oracle_tokens_per_chunk, raw_inflate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    flate.build()
    e = flate.FlateEngine(0)
    yield e
    e.close()


SINGLE_WINDOW = [
    ("text", 65536), ("text", 65535), ("text", 40000), ("ramp", 65536), ("zero", 65536),
    ("rand", 65536), ("low", 65536), ("period", 65536), ("runs", 65536), ("text", 128),
    ("text", 129), ("text", 143), ("text", 1000), ("runs", 300), ("zero", 200), ("low", 5000),
    ("period", 33000), ("rand", 128), ("ramp", 65534),
]

MULTI_WINDOW = [
    ("text", 65537 + 200), ("text", 131072), ("ramp", 131072), ("period", 262144),
    ("runs", 200000), ("zero", 140000), ("low", 131070), ("rand", 70000), ("text", 196605),
    ("text", 65535 * 2 + 127), ("text", 65535 * 2 + 128),
    # long streams: 16-bit table positions wrap many times (periodic sweeps), sparse scans are
    # cut at kSpanMax inside incompressible stretches
    ("text", 1 << 21), ("period", 1500000), ("rand", 300000), ("runs", 700000),
]


def _check_tokens(eng, oracle, specs, lz_serial, compat_go=False):
    data, off = make_streams(specs)
    chunks = eng.lz77_matches(data, off, lz_serial=lz_serial, compat_go=compat_go)
    k = 0
    for i, (kind, n) in enumerate(specs):
        sb = data[int(off[i]):int(off[i + 1])]
        want = oracle_tokens_per_chunk(oracle, sb, compat=1 if compat_go else 0)
        for (start, cn), w in zip(flate.lz_chunks(n), want):
            pos, tok = chunks[k]
            got = flate.tokens_from_matches(sb[start:start + cn], pos, tok)
            assert got.size == w.size, (kind, n, start, got.size, w.size)
            bad = np.nonzero(got != w)[0]
            assert bad.size == 0, (kind, n, start, int(bad[0]), got[bad[0]], w[bad[0]])
            k += 1
    assert k == len(chunks)


@pytest.mark.parametrize("lz_serial", [True, False], ids=["serial", "wave"])
def test_tokens_single_window(eng, oracle, lz_serial):
    _check_tokens(eng, oracle, SINGLE_WINDOW, lz_serial)


@pytest.mark.parametrize("lz_serial", [True, False], ids=["serial", "wave"])
def test_tokens_multi_window(eng, oracle, lz_serial):
    _check_tokens(eng, oracle, MULTI_WINDOW, lz_serial)


@pytest.mark.parametrize("lz_serial", [True, False], ids=["serial", "wave"])
def test_tokens_multi_window_go_compat(eng, oracle, lz_serial):
    _check_tokens(eng, oracle, MULTI_WINDOW, lz_serial, compat_go=True)


EDGE_SIZES = [0, 1, 2, 15, 16, 17, 18, 100, 127, 128, 129, 255, 4096, 65534, 65535, 65536, 65537,
              65535 + 16, 65535 + 17, 65535 + 127, 65535 + 128, 131070, 131071, 131072]


def _check_streams(eng, oracle, specs, compat_go=False, device=False):
    data, off = make_streams(specs)
    if device:
        import torch
        d = torch.from_numpy(data).cuda()
        out, out_off = eng.deflate_batch(d, off, compat_go=compat_go)
        out = out.cpu().numpy()
    else:
        out, out_off = eng.deflate_batch(data, off, compat_go=compat_go)
    for i, (kind, n) in enumerate(specs):
        sb = data[int(off[i]):int(off[i + 1])] if n else np.zeros(0, np.uint8)
        want = oracle.deflate(sb, compat=1 if compat_go else 0)
        got = bytes(out[int(out_off[i]):int(out_off[i + 1])])
        if got != want:
            first = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), None)
            raise AssertionError("stream %d (%s,%d): len got %d want %d first diff at %s" %
                                 (i, kind, n, len(got), len(want), first))
        assert raw_inflate(got) == bytes(sb)


def test_streams_edge_sizes(eng, oracle):
    _check_streams(eng, oracle, [("text", n) for n in EDGE_SIZES])
    _check_streams(eng, oracle, [("ramp", n) for n in EDGE_SIZES])


def test_streams_data_kinds(eng, oracle):
    _check_streams(eng, oracle, SINGLE_WINDOW + MULTI_WINDOW)


def test_streams_device_pointers(eng, oracle):
    _check_streams(eng, oracle, SINGLE_WINDOW[:8] + MULTI_WINDOW[:3], device=True)


def test_streams_go_compat(eng, oracle):
    _check_streams(eng, oracle, SINGLE_WINDOW + MULTI_WINDOW, compat_go=True)


def test_best_speed_matrix(eng, oracle):
    # deflate-fast_test.mbt:14-100: the 96 streams of TestBestSpeed (write sizes only change
    # how the window is staged, not the bytes), deduplicated by total content.
    abc = (np.arange(131072) & 127).astype(np.uint8)
    cases = [[0], [1], [1, 256], [1, 65536], [14], [15], [16], [16, 256], [16, 65536], [127],
             [128], [128, 256], [128, 65536], [129], [65536, 256], [65536, 65536]]
    streams = []
    for tc in cases:
        for first in [1, 65534, 65535, 65536, 65537, 131072]:
            streams.append(np.concatenate([abc[:k] for k in [first] + tc]))
    lens = np.array([s.size for s in streams], dtype=np.uint64)
    off = np.zeros(len(streams) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    data = np.concatenate(streams)
    out, out_off = eng.deflate_batch(data, off)
    for i, s in enumerate(streams):
        got = bytes(out[int(out_off[i]):int(out_off[i + 1])])
        assert got == oracle.deflate(s), i
        assert raw_inflate(got) == bytes(s)


def test_fuzz_random_lengths(eng, oracle):
    rng = np.random.default_rng(99)
    kinds = ["text", "low", "period", "runs", "rand", "zero", "ramp"]
    specs = [(kinds[int(rng.integers(0, len(kinds)))], int(rng.integers(0, 150000)))
             for _ in range(60)]
    _check_streams(eng, oracle, specs)


def test_guest_blocks_give_identical_streams(oracle):
    # resident (LDS-table) and guest (L2-table) match-finder blocks share one stream queue
    e = flate.FlateEngine(0)
    try:
        e.set_option("guest_min_streams", 1)
        e.set_option("guest_blocks", 8)
        e.set_option("resident_blocks", 4)
        _check_streams(e, oracle, SINGLE_WINDOW + MULTI_WINDOW + [("text", 65536)] * 40)
        e.set_option("resident_blocks", 1)
        e.set_option("guest_blocks", 64)
        _check_streams(e, oracle, [("text", 65536), ("runs", 65536), ("low", 65536)] * 20)
    finally:
        e.close()


def test_out_too_small_is_reported(eng):
    data, off = make_streams([("rand", 65536)] * 4)
    with pytest.raises(flate.FlateError) as ei:
        eng.deflate_batch(data, off, out_cap=1000)
    assert ei.value.code == -2


def test_batch_1k_streams_full_size(eng, oracle):
    # 1024 x 64 KiB of S-text, byte-for-byte against the oracle (threads)
    n = 1024
    data = flate.synth("text", n, 65536)
    off = flate.uniform_offsets(n, 65536)
    out, out_off = eng.deflate_batch(data, off)
    o_out, o_off, o_len = oracle.deflate_batch(data, off, nthreads=8)
    for i in range(n):
        a = out[int(out_off[i]):int(out_off[i + 1])]
        b = o_out[int(o_off[i]):int(o_off[i]) + int(o_len[i])]
        assert a.size == b.size and np.array_equal(a, b), i


def test_full_size_config2_round_trip_property(eng, oracle):
    # BASELINE configs[1] at full size: 16384 x 64 KiB.  Size-independent properties: every stream
    # inflates back to its input (GPU inflater, itself oracle-checked), ends with the final empty
    # stored block, and a strided sample is byte-identical to the oracle.
    import torch
    n, blen = 16384, 65536
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    d = torch.from_numpy(host).cuda()
    comp, coff = eng.deflate_batch(d, off, out_cap=n * blen)
    back, _, olen, status, _ = eng.inflate_batch(comp, coff, [blen] * n)
    assert (status == 0).all() and (olen == blen).all()
    assert torch.equal(back[:n * blen], d)
    c = comp[:int(coff[-1])].cpu().numpy()
    ends = coff[1:].astype(np.int64)
    tail = np.stack([c[ends - 5 + k] for k in range(5)], axis=1)
    assert (tail == np.array([1, 0, 0, 0xFF, 0xFF], dtype=np.uint8)).all()
    for i in range(0, n, 257):
        assert bytes(c[int(coff[i]):int(coff[i + 1])]) == oracle.deflate(host[i * blen:(i + 1) * blen]), i


def test_full_size_config3_multi_window(eng, oracle):
    # BASELINE configs[2]: 256 KiB streams (4 chained windows each); 1024 of them byte-exact vs the
    # oracle sample + full round trip.
    import torch
    n, blen = 1024, 262144
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    d = torch.from_numpy(host).cuda()
    comp, coff = eng.deflate_batch(d, off, out_cap=n * blen)
    back, _, olen, status, _ = eng.inflate_batch(comp, coff, [blen] * n)
    assert (status == 0).all() and torch.equal(back[:n * blen], d)
    c = comp[:int(coff[-1])].cpu().numpy()
    for i in range(0, n, 61):
        assert bytes(c[int(coff[i]):int(coff[i + 1])]) == oracle.deflate(host[i * blen:(i + 1) * blen]), i


def test_full_size_config3_default_geometry(oracle):
    # BASELINE configs[2] at FULL size with the DEFAULT launch geometry: 4096 x 256 KiB streams is above
    # guest_min_streams, so the persistent resident + guest MULTI kernels (shared queue, swept 16-bit
    # tables) are what runs.  Both compat modes; byte-exact against the oracle on a strided sample,
    # and every stream round-trips through the GPU inflater.
    import torch
    n, blen = 4096, 262144
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    d = torch.from_numpy(host).cuda()
    e = flate.FlateEngine(0)   # fresh engine: default options
    try:
        for go in (False, True):
            comp, coff = e.deflate_batch(d, off, out_cap=n * blen, compat_go=go)
            back, _, olen, status, _ = e.inflate_batch(comp, coff, [blen] * n)
            assert (status == 0).all() and (olen == blen).all() and torch.equal(back[:n * blen], d)
            c = comp[:int(coff[-1])].cpu().numpy()
            for i in range(0, n, 127):
                want = oracle.deflate(host[i * blen:(i + 1) * blen], compat=oracle.COMPAT_GO if go else 0)
                assert bytes(c[int(coff[i]):int(coff[i + 1])]) == want, (go, i)
            del comp, back
    finally:
        e.close()


def test_launch_geometry_does_not_change_the_bytes(oracle):
    # the persistent match finder with few or many LDS-table and guest blocks (every block then takes many
    # streams from the queue, or hardly any), and multi-window streams with and without window units
    n = 1024
    data = flate.synth("text", n, 65536)
    off = flate.uniform_offsets(n, 65536)
    want, w_off, w_len = oracle.deflate_batch(data, off, nthreads=8)
    e = flate.FlateEngine(0)
    try:
        e.set_option("guest_min_streams", 1)
        for blocks in ((8, 8), (64, 256), (1024, 1664)):
            e.set_option("resident_blocks", blocks[0])
            e.set_option("guest_blocks", blocks[1])
            out, out_off = e.deflate_batch(data, off)
            for i in range(n):
                a = out[int(out_off[i]):int(out_off[i + 1])]
                b = want[int(w_off[i]):int(w_off[i]) + int(w_len[i])]
                assert a.size == b.size and np.array_equal(a, b), (blocks, i)
        # multi-window streams: the match finder hands windows between blocks (window units) or keeps a
        # stream on its block
        n2, blen2 = 192, 150000
        data2 = flate.synth("text", n2, blen2, first_stream=5000)
        off2 = flate.uniform_offsets(n2, blen2)
        want2, w_off2, w_len2 = oracle.deflate_batch(data2, off2, nthreads=8)
        for units in (1, 0):
            e.set_option("window_units", units)
            out, out_off = e.deflate_batch(data2, off2)
            for i in range(n2):
                a = out[int(out_off[i]):int(out_off[i + 1])]
                b = want2[int(w_off2[i]):int(w_off2[i]) + int(w_len2[i])]
                assert a.size == b.size and np.array_equal(a, b), ("multi", units, i)
    finally:
        e.close()


def test_lost_window_handover_is_an_error_not_a_hang(oracle):
    # window-granular scheduling: a block waits (bounded) for the unit with its ticket.  The test
    # hook drops one hand-over; the call must come back with FLATE_HIP_E_INTERNAL and a message that
    # names the wait -- and the engine must work again afterwards.
    n, blen = 24, 200000
    data = flate.synth("text", n, blen, first_stream=7000)
    off = flate.uniform_offsets(n, blen)
    e = flate.FlateEngine(0)
    try:
        e.set_option("guest_min_streams", 1)
        e.set_option("guest_blocks", 8)
        e.set_option("resident_blocks", 8)
        e.set_option("spin_limit_polls", 20000)
        e.set_option("debug_drop_window_push", 3)
        with pytest.raises(flate.FlateError) as ei:
            e.deflate_batch(data, off)
        assert ei.value.code == -8 and "never handed over" in str(ei.value)
        e.set_option("debug_drop_window_push", 0)
        out, out_off = e.deflate_batch(data, off)
        for i in range(0, n, 5):
            assert bytes(out[int(out_off[i]):int(out_off[i + 1])]) == oracle.deflate(data[i * blen:(i + 1) * blen]), i
    finally:
        e.close()


def test_batch_without_progress_is_an_error_not_a_hang(oracle):
    # The parser's progress guard: a batch that leaves s, the sparse state and the event index where it
    # found them would repeat for ever (round 5's kDenseKeep = 62 build did, and hung its box).  The test
    # hook makes the third dense batch of every chunk forget its progress; the call must come back with
    # FLATE_HIP_E_INTERNAL and name the guard -- for the LDS-table blocks, the guests and the one-block-
    # per-stream launch, single- and multi-window -- and the engine must work again afterwards.
    cases = ((24, 65536, ()), (24, 200000, ()),
             (24, 65536, (("guest_min_streams", 1), ("guest_blocks", 8), ("resident_blocks", 8))),
             (24, 200000, (("guest_min_streams", 1), ("guest_blocks", 8), ("resident_blocks", 8))))
    for n, blen, opts in cases:
        data = flate.synth("text", n, blen, first_stream=7100)
        off = flate.uniform_offsets(n, blen)
        e = flate.FlateEngine(0)
        try:
            for k, v in opts:
                e.set_option(k, v)
            e.set_option("debug_stall_batch", 3)
            with pytest.raises(flate.FlateError) as ei:
                e.deflate_batch(data, off)
            assert ei.value.code == -8 and "no progress" in str(ei.value), (blen, opts)
            e.set_option("debug_stall_batch", 0)
            out, out_off = e.deflate_batch(data, off)
            for i in range(0, n, 5):
                assert bytes(out[int(out_off[i]):int(out_off[i + 1])]) == oracle.deflate(data[i * blen:(i + 1) * blen]), i
        finally:
            e.close()


def test_host_pointer_pipeline_matches_single_pass(oracle):
    # host-pointer calls on large batches are pipelined over groups of streams (option
    # host_pipeline_groups): same bytes and the same index as one pass, and as the oracle
    n, blen = 8192 + 37, 65536 - 3
    host = flate.synth("text", n, blen)
    off = flate.uniform_offsets(n, blen)
    e = flate.FlateEngine(0)
    try:
        e.set_option("host_pipeline_groups", 0)
        c0, o0 = e.deflate_batch(host, off)
        total = int(o0[-1])
        for lanes, groups in ((1, 2), (2, 2), (2, 3), (1, 3)):
            e.set_option("host_pipeline_lanes", lanes)
            e.set_option("host_pipeline_groups", groups)
            c1, o1 = e.deflate_batch(host, off)
            assert np.array_equal(np.asarray(o0), np.asarray(o1)), (lanes, groups)
            assert np.array_equal(np.asarray(c0)[:total], np.asarray(c1)[:total]), (lanes, groups)
        # page-locked buffers (flate_hip_host_register): the same calls, the same bytes
        out_reg = np.empty(total + 4096, dtype=np.uint8)
        with e.host_register(host), e.host_register(out_reg):
            for lanes, groups in ((2, 3), (2, 0)):
                e.set_option("host_pipeline_lanes", lanes)
                e.set_option("host_pipeline_groups", groups)
                out_reg[:] = 0
                _, o2 = e.deflate_batch(host, off, out=out_reg)
                assert np.array_equal(np.asarray(o0), np.asarray(o2)), (lanes, groups)
                assert np.array_equal(np.asarray(c0)[:total], out_reg[:total]), (lanes, groups)
        e.set_option("host_pipeline_lanes", 2)
        c = np.asarray(c1)
        for i in range(0, n, 509):
            assert bytes(c[int(o1[i]):int(o1[i + 1])]) == oracle.deflate(host[i * blen:(i + 1) * blen]), i
        # an output buffer that is too small is still reported
        e.set_option("host_pipeline_groups", 2)
        with pytest.raises(flate.FlateError):
            e.deflate_batch(host, off, out_cap=int(o0[-1]) // 2)
    finally:
        e.close()


def test_host_pointer_pipeline_uneven_streams_three_groups(oracle):
    # groups are cut by BYTES: streams of very different sizes, three and four groups, same bytes and
    # the same index as one pass (and as the oracle on a sample)
    rng = np.random.default_rng(17)
    lens = np.where(rng.random(2300) < 0.3, rng.integers(60000, 140000, 2300), rng.integers(0, 9000, 2300))
    lens[:40] = 0
    specs = [("text", int(l)) for l in lens]
    host, off = make_streams(specs, seed=8)
    assert int(off[-1]) >= (64 << 20)
    e = flate.FlateEngine(0)
    try:
        e.set_option("guest_min_streams", 1)
        e.set_option("host_pipeline_group_streams", 250)
        e.set_option("host_pipeline_groups", 0)
        c0, o0 = e.deflate_batch(host, off)
        total = int(o0[-1])
        for groups in (3, 4):
            e.set_option("host_pipeline_groups", groups)
            c1, o1 = e.deflate_batch(host, off)
            assert np.array_equal(np.asarray(o0), np.asarray(o1)), groups
            assert np.array_equal(np.asarray(c0)[:total], np.asarray(c1)[:total]), groups
        c = np.asarray(c1)
        for i in range(0, len(specs), 97):
            assert bytes(c[int(o1[i]):int(o1[i + 1])]) == oracle.deflate(host[int(off[i]):int(off[i + 1])]), i
        # and back, through the pipelined inflate (it cuts by input + output bytes)
        back, boff, olen, st, _ = e.inflate_batch(c[:total], o1, lens)
        assert (st == 0).all() and np.array_equal(np.asarray(back)[:int(boff[-1])], host[:int(off[-1])])
    finally:
        e.close()


def test_entropy_stage_with_one_wavefront_per_block(oracle):
    # option "entropy_per_block": histogram and pack kernels run per BLOCK (multi-window streams); the
    # blocks of a stream then meet at bit granularity inside shared dwords.  Same bytes as the oracle in
    # the batch form (the spliced form keeps the per-stream kernels: same bytes there too), both compat modes, stored / Huffman-only / dynamic blocks,
    # tails of every kind behind full windows.
    specs = [("text", 4 * 65535 + t) for t in (0, 1, 16, 17, 127, 128, 5000)] + \
            [("rand", 2 * 65535 + 9), ("zero", 3 * 65535), ("runs", 65535 + 40), ("low", 65536), ("text", 5),
             ("period", 200000), ("text", 65535), ("rand", 17), ("ramp", 131070)]
    data, off = make_streams(specs, seed=21)
    e = flate.FlateEngine(0)
    try:
        e.set_option("entropy_per_block", 1)
        for go in (False, True):
            out, ooff = e.deflate_batch(data, off, compat_go=go)
            for i in range(len(specs)):
                want = oracle.deflate(data[int(off[i]):int(off[i + 1])], compat=oracle.COMPAT_GO if go else 0)
                assert bytes(out[int(ooff[i]):int(ooff[i + 1])]) == want, (go, i, specs[i])
            one, nb, bit_off = e.deflate_spliced(data, off, compat_go=go)
            ref, ref_off = oracle.deflate_spliced(data, off, oracle.COMPAT_GO if go else 0)
            assert bytes(one[:nb]) == ref and (bit_off == ref_off).all(), go
        # a misaligned output buffer (streams start at every byte offset inside a dword)
        import torch
        d = torch.from_numpy(data).cuda()
        out, ooff = e.deflate_batch(data, off)
        buf = torch.zeros(int(ooff[-1]) + 64, dtype=torch.uint8, device="cuda")
        for mis in (1, 2, 3):
            o2, oo2 = e.deflate_batch(d, off, out=buf[mis:])
            assert np.array_equal(oo2, ooff) and bytes(o2[:int(oo2[-1])].cpu().numpy()) == bytes(out[:int(ooff[-1])]), mis
        # an empty stream in the batch: the per-stream kernels take over, same bytes
        data2, off2 = make_streams(specs[:3] + [("text", 0)] + specs[3:6], seed=22)
        out2, ooff2 = e.deflate_batch(data2, off2)
        for i in range(7):
            assert bytes(out2[int(ooff2[i]):int(ooff2[i + 1])]) == oracle.deflate(data2[int(off2[i]):int(off2[i + 1])]), i
    finally:
        e.close()
