"""The byte-parallel tile walk of huff_hist_kernel (huff_pack_kernels.hip, walk_tile), restated with
numpy uint32 arithmetic and checked against a position-by-position classification (CPU only).
Covers what the kernel's comments claim: no position ever receives two marks (plain byte stores),
two shift-adds are the inclusive prefix sums of four byte marks, the coverage bytes never borrow,
the byte sums / dot products (v_sad_u8, v_dot4_u32_u8) give the lane totals, the packed meta byte
and the rank-th record of the tile is the token of the rank-th lane that holds a match start."""
import numpy as np
import pytest

TILE = 256
M32 = np.uint64(0xffffffff)


def clamp04(v):
    return np.clip(v, 0, 4)


def byte_sum(x):            # v_sad_u8 x, 0, 0
    return sum((x >> np.uint64(8 * k)) & np.uint64(0xff) for k in range(4))


def udot4(x, w, acc=0):     # v_dot4_u32_u8
    r = np.zeros_like(x) + np.uint64(acc) if np.isscalar(acc) else acc.copy()
    for k in range(4):
        r = r + ((x >> np.uint64(8 * k)) & np.uint64(0xff)) * np.uint64((w >> (8 * k)) & 0xff)
    return r & M32


def walk_tile_model(recs, P0, n, cov_until):
    """recs: list of (pos, length) with P0 <= pos < P0 + TILE, sorted, non-overlapping.
    Returns per lane (lit_mask, match_k, record index of the match start) as the kernel computes them."""
    marks = np.zeros(TILE, np.uint8)   # 1: first covered position, 2: last covered position, 4: start
    for pos, ln in recs:
        o = pos - P0
        assert marks[o] == 0
        marks[o] = 4
        if o + 1 < TILE:
            assert marks[o + 1] == 0
            marks[o + 1] = 1
        if o + ln - 1 < TILE:
            assert marks[o + ln - 1] == 0
            marks[o + ln - 1] = 2
    m = marks.view("<u4").astype(np.uint64)
    ONES = np.uint64(0x01010101)
    first, last = m & ONES, (m >> np.uint64(1)) & ONES
    tot = byte_sum(first).astype(np.int64) - byte_sum(last).astype(np.int64)
    base = np.cumsum(tot) - tot          # wave_incl_scan(tot) - tot
    assert ((base == 0) | (base == 1)).all()
    lane_pos = P0 + 4 * np.arange(64)
    t_lo = clamp04(cov_until - lane_pos).astype(np.uint64)
    low = ((ONES << (np.uint64(8) * t_lo)) >> np.uint64(32)) & M32
    e = (first + (np.uint64(1) << np.uint64(32)) - ((last << np.uint64(8)) & M32) + base.astype(np.uint64)) & M32
    e2 = (e + (e << np.uint64(8))) & M32          # v_lshl_add_u32 e, 8, e
    e4 = (e2 + (e2 << np.uint64(16))) & M32       # v_lshl_add_u32 e2, 16, e2
    cov = e4 | low
    assert ((cov & ~ONES) == 0).all()   # bytes are 0 or 1: no borrow, no carry
    t_act = clamp04(n - lane_pos).astype(np.uint64)
    act = ((ONES << (np.uint64(8) * t_act)) >> np.uint64(32)) & M32
    start = (m >> np.uint64(2)) & act
    lit = act & ~(cov | start) & M32
    meta = udot4(lit, 0x08040201, udot4(start, 0x70503010))   # TileTok::pack
    lit_mask = (meta & np.uint64(15)).astype(np.int64)
    match_k = np.where(meta & np.uint64(16), (meta >> np.uint64(5)).astype(np.int64), -1)
    assert (meta < 128).all()
    has = start != 0
    rank = np.cumsum(has) - has          # popcount of the ballot below the lane
    rec_of = np.where(has, rank, -1)
    return lit_mask, match_k, rec_of


def walk_tile_naive(recs, P0, n, cov_until):
    covered = np.zeros(TILE, bool)
    starts = np.zeros(TILE, bool)
    for k in range(TILE):
        if P0 + k < cov_until:
            covered[k] = True
    for pos, ln in recs:
        o = pos - P0
        starts[o] = True
        covered[o + 1:min(o + ln, TILE)] = True
    lit_mask = np.zeros(64, np.int64)
    match_k = np.full(64, -1)
    for L in range(64):
        for k in range(4):
            pk = P0 + 4 * L + k
            if pk >= n:
                continue
            if starts[4 * L + k]:
                match_k[L] = k
            elif not covered[4 * L + k]:
                lit_mask[L] |= 1 << k
    rec_of = np.full(64, -1)
    for j, (pos, ln) in enumerate(recs):
        if pos < n:
            rec_of[(pos - P0) // 4] = j
    return lit_mask, match_k, rec_of


@pytest.mark.parametrize("seed", range(8))
def test_tile_walk_arithmetic(seed):
    rng = np.random.default_rng(seed)
    for _ in range(60):
        P0 = int(rng.integers(0, 200)) * TILE
        n = P0 + int(rng.integers(1, TILE + 1)) if rng.random() < 0.3 else P0 + TILE + 1000
        cov_until = P0 + int(rng.integers(0, 260)) if rng.random() < 0.5 else 0
        recs, pos = [], max(P0, cov_until)
        dense = rng.random() < 0.5
        while True:
            pos += int(rng.integers(0, 3 if dense else 40))
            ln = int(rng.choice([4, 4, 5, 6, 9, 17, 64, 258]))
            if pos >= P0 + TILE or pos + ln > n:
                break
            recs.append((pos, ln))
            pos += ln
        got = walk_tile_model(recs, P0, n, cov_until)
        want = walk_tile_naive(recs, P0, n, cov_until)
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), (seed, P0, n, cov_until, recs[:5])
        # what huff_pack_kernel relies on: a lane's literals all precede its match
        for L in range(64):
            if got[1][L] >= 0:
                assert got[0][L] >> got[1][L] == 0
