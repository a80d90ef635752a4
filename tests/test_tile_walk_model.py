"""The byte-parallel tile walk of huff_hist_kernel (huff_pack_kernels.hip, walk_tile), restated with
numpy uint32 arithmetic and checked against a position-by-position classification (CPU only).
Covers what the kernel's comments claim: a multiply by 0x01010101 is the inclusive prefix sum of
four byte marks, the coverage bytes never borrow, and one more multiply packs the literal mask."""
import numpy as np
import pytest

TILE = 256


def clamp04(v):
    return np.clip(v, 0, 4)


def walk_tile_model(recs, P0, n, cov_until):
    """recs: list of (pos, length) with P0 <= pos < P0 + TILE, sorted, non-overlapping.
    Returns per lane (lit_mask, match_k) exactly as the kernel computes them."""
    p = np.zeros(TILE, np.uint8)   # +1 marks: first covered position of a match
    m = np.zeros(TILE, np.uint8)   # 1 just past a match
    s = np.zeros(TILE, np.uint8)   # match starts
    for pos, ln in recs:
        o = pos - P0
        s[o] = 1
        if o + 1 < TILE:
            p[o + 1] = 1
        if o + ln < TILE:
            m[o + ln] = 1
    as_dw = lambda a: a.view("<u4").astype(np.uint64)
    pd, md, sd = as_dw(p), as_dw(m), as_dw(s)
    M32 = np.uint64(0xffffffff)
    pc = (pd * np.uint64(0x01010101)) & M32
    nc = (md * np.uint64(0x01010101)) & M32
    tot = ((pc >> np.uint64(24)).astype(np.int64) - (nc >> np.uint64(24)).astype(np.int64))
    base = np.cumsum(tot) - tot          # wave_incl_scan(tot) - tot
    assert ((base == 0) | (base == 1)).all()
    lane_pos = P0 + 4 * np.arange(64)
    t_lo = clamp04(cov_until - lane_pos).astype(np.uint64)
    low = ((np.uint64(0x01010101) << (np.uint64(8) * t_lo)) >> np.uint64(32)) & M32
    cov = ((pc + base.astype(np.uint64) * np.uint64(0x01010101) - nc) & M32) | low
    assert ((cov & ~np.uint64(0x01010101)) == 0).all()   # bytes are 0 or 1: no borrow, no carry
    t_act = clamp04(n - lane_pos).astype(np.uint64)
    act = ((np.uint64(0x01010101) << (np.uint64(8) * t_act)) >> np.uint64(32)) & M32
    start = sd & act
    lit = act & ~(cov | start) & M32
    lit_mask = ((lit * np.uint64(0x01020408)) & M32) >> np.uint64(24)
    match_k = np.full(64, -1)
    for L in range(64):
        if start[L]:
            v = int(start[L])
            match_k[L] = ((v & -v).bit_length() - 1) >> 3
    return lit_mask.astype(np.int64), match_k


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
    return lit_mask, match_k


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
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (seed, P0, n, cov_until, recs[:5])
        # what huff_pack_kernel relies on: a lane's literals all precede its match
        for L in range(64):
            if got[1][L] >= 0:
                assert got[0][L] >> got[1][L] == 0
