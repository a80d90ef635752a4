"""Dev check: eager (level-parallel) package-merge == oracle bit_counts (huffman-code.mbt:112-244)."""
import sys
import numpy as np
sys.path.insert(0, '.')  # run from the repository root: python tests/tools/eager_pm_check.py
from oracle import pyoracle as o


def eager_lengths(freq, max_bits):
    freq = np.asarray(freq, dtype=np.int64)
    syms = np.nonzero(freq)[0]
    n = syms.size
    lens = np.zeros(freq.size, dtype=np.int64)
    if n <= 2:
        lens[syms] = 1
        return lens
    order = np.lexsort((syms, freq[syms]))       # by (freq, sym)
    leaves = freq[syms][order]                   # ascending
    mb = min(max_bits, n - 1)
    # level lists: merged freq + leaf rank (position of leaf i in the merged list)
    prev = leaves.copy()                         # L_1
    leaf_rank = [None, np.arange(n)]             # level 1: only leaves
    for lvl in range(2, mb + 1):
        npairs = prev.size // 2
        pairs = prev[0:2 * npairs:2] + prev[1:2 * npairs:2]
        # ties take the pair: pair before leaf when equal
        # rank(leaf i) = i + #pairs with sum <= leaf_i ; rank(pair j) = j + #leaves with freq < pair_j
        lr = np.arange(n) + np.searchsorted(pairs, leaves, side='right')
        pr = np.arange(npairs) + np.searchsorted(leaves, pairs, side='left')
        merged = np.empty(n + npairs, dtype=np.int64)
        merged[lr] = leaves
        merged[pr] = pairs
        leaf_rank.append(lr)
        prev = merged
    # top-down: m items needed at level mb = 2n-2
    m = 2 * n - 2
    counts = np.zeros(mb + 2, dtype=np.int64)    # counts[level] = #leaves among first m_level items
    for lvl in range(mb, 0, -1):
        a = int(np.searchsorted(leaf_rank[lvl], m, side='left'))  # leaves with rank < m
        counts[lvl] = a
        m = 2 * (m - a)
    # bit_count[bits] for bits = 1.. : counts[level]-counts[level-1], level = mb down to 1
    bit_count = {}
    bits = 1
    for lvl in range(mb, 0, -1):
        bit_count[bits] = counts[lvl] - counts[lvl - 1]
        bits += 1
    # assign: last bit_count[1] entries (highest freq) get length 1, and so on
    sorted_syms = syms[order]
    end = n
    for b in range(1, mb + 1):
        c = int(bit_count[b])
        lens[sorted_syms[end - c:end]] = b
        end -= c
    assert end == 0, (end, bit_count)
    return lens


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    bad = 0
    for t in range(trials):
        n = [286, 30, 19][t % 3]
        mb = 7 if n == 19 else 15
        mode = rng.integers(0, 6)
        if mode == 0:
            f = rng.integers(0, 3, n) * rng.integers(0, 100, n)
        elif mode == 1:
            f = 2 ** rng.integers(0, 20, n)
        elif mode == 2:
            f = rng.integers(0, 2, n) * (1 + rng.geometric(0.01, n))
        elif mode == 3:
            f = rng.integers(1, 4, n)
        elif mode == 4:
            f = np.zeros(n, dtype=np.int64); k = rng.integers(1, min(n, 12)); f[rng.choice(n, k, replace=False)] = rng.integers(1, 50, k)
        else:
            a = np.array([1, 1]); 
            while a.size < n: a = np.append(a, a[-1] + a[-2])
            f = a[:n] % 60000 + 1
            rng.shuffle(f)
        f = f.astype(np.int32)
        _, want = o.huffman_generate(f, mb)
        got = eager_lengths(f, mb)
        if not np.array_equal(got, want.astype(np.int64)):
            bad += 1
            if bad < 5:
                print("MISMATCH", n, mb, f.tolist(), got.tolist(), want.tolist())
    print("trials", trials, "bad", bad)


main()
