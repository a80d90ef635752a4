"""build_code of huff_code_kernel (huff_pack_kernels.hip), restated step by step with numpy and
checked against the oracle's HuffmanEncoder::generate (huffman-code.mbt:295-343) on the CPU.

The restatement follows the kernel's own formulation, not the textbook one, so that what the kernel's
comments claim is tested where no GPU is needed:
 * the rank of a leaf among the pairs of a level (and of a pair among the leaves) is built bit by
   bit from the top with a wave-uniform trip count -- element c + s - 1 passes the test <=> the
   count is >= c + s;
 * a pair wins a tie against a leaf (`<` at huffman-code.mbt:187);
 * the leaves needed at a level are the set bits of that level's leaf bitmap below m, m halving
   its way down from 2n - 2 (:168, :234-243);
 * bit_count[b] = counts[mb - b + 1] - counts[mb - b] (one level per lane), len_base by a prefix
   sum, the canonical first codes by the running shift-and-add, and a leaf's length is
   1 + #{k in 1 .. mb - 1 : len_base[k] <= its rank from the top};
 * codes are handed out in symbol order within a length and bit-reversed (:250-283).
"""
import numpy as np
import pytest

from oracle import pyoracle as o


def _count_prefix(sorted_vals, m, s0, pred):
    """The kernel's branch-free search: number of leading elements of sorted_vals[0:m] that satisfy
    the (monotone) predicate, built from the top bit down; s0 = first step (a power of two)."""
    c = 0
    s = s0
    while s >= 1:
        a = c + s
        v = sorted_vals[min(a, m) - 1]      # the load is unconditional, clamped into the list
        if a <= m and pred(v):
            c = a
        s >>= 1
    return c


def build_code_model(freq, max_bits):
    """-> (length, bit-reversed code) per symbol, exactly as build_code / build_code_sorted do it."""
    freq = np.asarray(freq, dtype=np.int64)
    nsym = freq.size
    lens = np.zeros(nsym, np.int64)
    codes = np.zeros(nsym, np.int64)
    syms = np.nonzero(freq)[0]
    n = syms.size
    if n <= 2:                                # :326-336
        for r, sy in enumerate(syms):
            lens[sy], codes[sy] = 1, r
        return lens, codes
    key = (freq[syms] << 9) | syms            # distinct keys: (freq, symbol) order
    rank = np.array([(key < k).sum() for k in key])          # rank sort
    sfreq = np.zeros(n, np.int64)
    sfreq[rank] = freq[syms]
    mb = min(max_bits, n - 1)
    s0 = 1 << (int(n).bit_length() - 1)
    prev = sfreq.copy()                       # level 1: the leaves themselves
    leaf_bits = {}
    for lvl in range(2, mb + 1):
        np_ = prev.size // 2
        pairs = prev[0:2 * np_:2] + prev[1:2 * np_:2]
        nxt = np.zeros(n + np_, np.int64)
        bits = np.zeros(n + np_, bool)
        for i in range(n):                    # leaf i -> i + #pairs with sum <= leaf
            r = i + _count_prefix(pairs, np_, s0, lambda v, f=sfreq[i]: v <= f)
            nxt[r] = sfreq[i]
            bits[r] = True
        for j in range(np_):                  # pair j -> j + #leaves with freq < sum
            nxt[j + _count_prefix(sfreq, n, s0, lambda v, p=pairs[j]: v < p)] = pairs[j]
        assert (np.diff(nxt) >= 0).all()      # the merged list is sorted and every slot was written
        leaf_bits[lvl] = bits
        prev = nxt
    counts = np.zeros(64, np.int64)           # "lane L holds counts[L]"
    m = 2 * n - 2
    for lvl in range(mb, 1, -1):
        a = int(leaf_bits[lvl][:m].sum())     # popcount of the bitmap words masked below m
        counts[lvl] = a
        m = 2 * (m - a)
    counts[1] = min(m, n)
    bc = np.zeros(64, np.int64)
    for b in range(1, mb + 1):
        bc[b] = counts[mb - b + 1] - counts[mb - b]
    len_base = np.cumsum(bc)
    first_code = np.zeros(16, np.int64)
    code = 0
    for b in range(1, 16):
        code <<= 1
        first_code[b] = code
        code += bc[b]
    for t in range(n):                        # the item with sort key key[t] is leaf rank[t]
        from_top = n - 1 - rank[t]
        bl = 1 + sum(1 for k in range(1, mb) if from_top >= len_base[k])
        lens[syms[t]] = bl
    running = np.zeros(16, np.int64)
    for i in range(nsym):                     # symbol order within each length
        L = lens[i]
        if L:
            c = first_code[L] + running[L]
            running[L] += 1
            codes[i] = int(format(int(c), "0%db" % L)[::-1], 2)
    return lens, codes


def _histograms(rng, trials):
    for t in range(trials):
        n = [286, 30, 19][t % 3]
        mb = 7 if n == 19 else 15
        mode = int(rng.integers(0, 7))
        if mode == 0:
            f = rng.integers(0, 3, n) * rng.integers(0, 100, n)
        elif mode == 1:
            f = 2 ** rng.integers(0, 16, n)                      # many ties between leaves and pairs
        elif mode == 2:
            f = rng.integers(0, 2, n) * (1 + rng.geometric(0.01, n))
        elif mode == 3:
            f = rng.integers(1, 4, n)
        elif mode == 4:
            f = np.zeros(n, np.int64)
            k = int(rng.integers(1, min(n, 12)))
            f[rng.choice(n, k, replace=False)] = rng.integers(1, 50, k)
        elif mode == 5:
            a = [1, 1]                                            # Fibonacci: needs the length limit
            while len(a) < n:
                a.append(a[-1] + a[-2])
            f = np.array(a[:n]) % 60000 + 1
            rng.shuffle(f)
        else:
            f = np.zeros(n, np.int64)                             # 3 .. 5 symbols: mb = n - 1
            k = int(rng.integers(3, 6))
            f[rng.choice(n, k, replace=False)] = rng.integers(1, 1000, k)
        yield f.astype(np.int64), mb


@pytest.mark.parametrize("seed", range(4))
def test_code_kernel_model_equals_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    for f, mb in _histograms(rng, 90):
        want_codes, want_lens = o.huffman_generate(f.astype(np.int32), mb)
        lens, codes = build_code_model(f, mb)
        assert np.array_equal(lens, want_lens.astype(np.int64)), (f.tolist(), mb)
        used = lens > 0
        assert np.array_equal(codes[used], want_codes.astype(np.int64)[used]), (f.tolist(), mb)


def test_uniform_search_is_the_count():
    rng = np.random.default_rng(7)
    for _ in range(300):
        m = int(rng.integers(1, 300))
        vals = np.sort(rng.integers(0, 50, m))
        x = int(rng.integers(-1, 52))
        s0 = 1 << (int(max(m, int(rng.integers(m, 400)))).bit_length() - 1)   # any power of two with 2 s0 > m
        assert _count_prefix(vals, m, s0, lambda v: v <= x) == int((vals <= x).sum())
        assert _count_prefix(vals, m, s0, lambda v: v < x) == int((vals < x).sum())
