"""CPU model of how the lane-per-stream inflater (csrc/inflate_kernels.hip: canon_len, PerLen) reads a
per-length array out of REGISTERS without an index: the length of a canonical code is 16 minus the number
of limits above its 15 left-justified bits, and with e[k] = V[k+1] - V[k+2] (V[16] = 0) the sum of e[k]
over exactly those limits telescopes to V[len] (mod 2^16).  The kernel does this with packed 16-bit
compares and multiply-adds; here the same arithmetic in numpy, against a table lookup, for random
length-limited codes -- including delta[] (offs - first, which wraps below zero) and thr[]."""
import numpy as np


def random_lengths(rng, n, max_bits=15):
    """Code lengths of a complete prefix code over n symbols (some unused), all <= max_bits."""
    while True:
        used = int(rng.integers(2, n + 1))
        # split the unit interval: start from one code of length 0 and split random leaves
        lens = [0]
        while len(lens) < used:
            i = int(rng.integers(len(lens)))
            if lens[i] >= max_bits:
                continue
            l = lens.pop(i) + 1
            lens += [l, l]
        out = np.zeros(n, dtype=np.int64)
        out[rng.permutation(n)[:used]] = lens
        return out


def build(lens, lit):
    cnt = np.bincount(lens, minlength=16)[:16].copy()
    cnt[0] = 0
    low = np.bincount(lens[:256], minlength=16)[:16] if lit else np.zeros(16, np.int64)
    lim = np.zeros(16, np.int64)
    val = np.zeros(17, np.int64)
    first = np.zeros(16, np.int64)
    offs = np.zeros(16, np.int64)
    code = off = 0
    for k in range(1, 16):
        code <<= 1
        first[k], offs[k] = code, off
        val[k] = (code + low[k]) if lit else ((off - code) & 0xFFFF)
        code += cnt[k]
        off += cnt[k]
        lim[k - 1] = code << (15 - k)
    lim[15] = 0
    e = np.array([(val[j + 1] - val[j + 2]) & 0xFFFF for j in range(15)] + [0], dtype=np.int64)
    sorted_syms = np.array(sorted(np.flatnonzero(lens), key=lambda s: (lens[s], s)), dtype=np.int64)
    return lim, e, val, first, offs, sorted_syms


def canon_len(c15, lim, e):
    """The kernel's loop: ind = (c15 - lim) is negative as a 16-bit value <=> lim is above c15."""
    ind = (((c15 - lim) & 0xFFFF) >> 15) & 1
    return 16 - int(ind.sum()), int((ind * e).sum() & 0xFFFF)


def test_per_length_array_without_an_index():
    rng = np.random.default_rng(12)
    for trial in range(300):
        lit = trial % 2 == 0
        n = 286 if lit else int(rng.choice([19, 30]))
        lens = random_lengths(rng, n)
        lim, e, val, first, offs, ss = build(lens, lit)
        for sym in np.flatnonzero(lens):
            L = int(lens[sym])
            rank = int(np.count_nonzero((lens == L) & (np.arange(n) < sym)))
            code = int(first[L]) + rank
            tail = int(rng.integers(0, 1 << (15 - L))) if L < 15 else 0
            c15 = (code << (15 - L)) | tail  # the code followed by arbitrary bits
            length, v = canon_len(c15, lim, e)
            assert length == L, (trial, sym, L, length)
            assert v == int(val[L]) & 0xFFFF
            if lit:  # thr: symbols >= 256 are the last of their length
                assert (code >= v) == (sym >= 256)
            else:    # delta: the sorted index
                assert int(ss[(code + v) & 0xFFFF]) == sym


def test_incomplete_code_has_no_length():
    # a single code of length 1 (the reference accepts it, inflate.mbt:161): the other half of the code
    # space decodes to "no code" (length 16) and the per-length sum is 0
    lens = np.zeros(30, dtype=np.int64)
    lens[7] = 1
    lim, e, val, first, offs, ss = build(lens, False)
    assert canon_len(0, lim, e)[0] == 1
    assert canon_len(1 << 14, lim, e) == (16, 0)
