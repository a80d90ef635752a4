"""CPU model of the lane-per-stream inflater's output ROW (csrc/inflate_kernels.hip: row_keep_first, row_join,
row_place and the flush rule of inflate_simt_kernel<LPW, ROWD>): a lane collects its output in ROWD registers and
stores whole aligned pieces.  The kernel does it with statically indexed registers only -- one v_perm_b32 per dword
for the byte shift, a barrel of select stages for the dword shift, OR into a row whose bytes at and behind the write
position are zero.  Here the same dword arithmetic in Python, step by step as the kernel does it, against plain byte
concatenation: random steps (copy chunks of 0..16 valid bytes out of 16 loaded ones, 0..8 literals), for rows of 8
and 16 dwords, including the stores a copy forces when its source reaches into the row."""
import numpy as np

M32 = 0xFFFFFFFF


def perm(s0, s1, sel):
    """v_perm_b32: result byte i = byte sel[i] of the 8 bytes {s0 (4..7), s1 (0..3)} (selectors 0..7 only)."""
    both = s1 | (s0 << 32)
    r = 0
    for i in range(4):
        k = (sel >> (8 * i)) & 0xFF
        assert k < 8
        r |= ((both >> (8 * k)) & 0xFF) << (8 * i)
    return r


def keep_first(d, k):
    out = []
    for j in range(4):
        nb = min(max(k - 4 * j, 0), 4)
        out.append(d[j] & (M32 if nb >= 4 else (1 << (8 * nb)) - 1))
    return out


def join(a, lo, hi, k):
    sel = (0x07060504 - 0x01010101 * (k & 3)) & M32
    l0, l1, l2 = perm(lo, 0, sel), perm(hi, lo, sel), perm(0, hi, sel)
    q = k >> 2
    pick = lambda *opts: next((v for cond, v in opts if cond), 0)  # noqa: E731
    a = list(a)
    a[0] |= pick((q == 0, l0))
    a[1] |= pick((q == 1, l0), (q == 0, l1))
    a[2] |= pick((q == 2, l0), (q == 1, l1), (q == 0, l2))
    a[3] |= pick((q == 3, l0), (q == 2, l1), (q == 1, l2))
    a[4] |= pick((q == 4, l0), (q == 3, l1), (q == 2, l2))
    a[5] |= pick((q == 4, l1), (q == 3, l2))
    return a


def place(a, o, rowd):
    f_n = rowd + 6
    sel = (0x07060504 - 0x01010101 * (o & 3)) & M32
    v = [0] * f_n
    for j in range(7):
        v[j] = perm(a[j] if j < 6 else 0, a[j - 1] if 1 <= j <= 6 else 0, sel)
    q, bit, live = o >> 2, 1, 7
    while bit < rowd:
        on = (q & bit) != 0
        live = min(live + bit, f_n)
        for j in range(f_n - 1, -1, -1):
            if j < live:
                v[j] = (v[j - bit] if j >= bit else 0) if on else v[j]
        bit <<= 1
    return v


def dwords(b, n):
    b = bytes(b) + b"\0" * (4 * n - len(b))
    return [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(n)]


def run_lane(rng, rowd, steps, slot_tail):
    """One lane: random steps through the kernel's phase (2) / early store / final store; returns what memory
    holds and what it should hold."""
    want = bytearray()
    cap_guess = steps * 24 + 64
    mem = bytearray(b"\xEE" * cap_guess)  # the slot (0xEE = never written)
    row, rbase, wpos = [0] * rowd, 0, 0
    plan = []
    for _ in range(steps):
        k = int(rng.choice([0, 0, 3, 4, 5, 6, 7, 8, 11, 15, 16]))
        nl = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 8]))
        plan.append((k, nl, rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), rng.integers(0, 256, 8, dtype=np.uint8).tobytes(),
                     bool(rng.integers(0, 6) == 0)))
    total = sum(k + nl for k, nl, *_ in plan)
    out_cap = total + slot_tail  # the slot ends slot_tail bytes behind the stream's last byte

    def row_store():
        if rbase + 4 * rowd <= out_cap:
            for j in range(rowd):
                mem[rbase + 4 * j:rbase + 4 * j + 4] = row[j].to_bytes(4, "little")
        else:
            for j in range(rowd):
                for t in range(4):
                    p = rbase + 4 * j + t
                    if p < wpos:
                        mem[p] = (row[j] >> (8 * t)) & 0xFF

    for k, nl, loaded, lits, early in plan:
        # the 16 loaded bytes: only the first k are the copy's, the rest is whatever memory held
        a = keep_first(dwords(loaded, 4), k) + [0, 0]
        lo, hi = dwords(lits[:nl], 2)
        a = join(a, lo, hi, k)
        f = place(a, wpos - rbase, rowd)
        row = [row[j] | f[j] for j in range(rowd)]
        want += loaded[:k] + lits[:nl]
        wpos += k + nl
        if wpos - rbase >= 4 * rowd:
            row_store()
            row = [f[rowd + j] if j < 6 else 0 for j in range(rowd)]
            rbase += 4 * rowd
        if early and wpos > rbase:  # a copy whose source reaches into the row: memory must hold the row first
            row_store()
            assert bytes(mem[:wpos]) == bytes(want)
    if wpos > rbase:
        row_store()
    return bytes(mem), bytes(want), out_cap


def test_row_equals_byte_concatenation():
    rng = np.random.default_rng(5)
    for rowd in (8, 16):
        for trial in range(300):
            mem, want, out_cap = run_lane(rng, rowd, int(rng.integers(1, 40)), int(rng.integers(0, 3 * rowd)))
            assert mem[:len(want)] == want, (rowd, trial)
            # nothing behind the slot is touched, whatever the last row's store did inside it
            assert set(mem[out_cap:]) <= {0xEE}, (rowd, trial)


def test_perm_selector_is_a_left_shift_by_bytes():
    for s in range(4):
        sel = (0x07060504 - 0x01010101 * s) & M32
        x, below = 0xA4A3A2A1, 0xB4B3B2B1
        assert perm(x, below, sel) == ((x << (8 * s)) | (below >> (32 - 8 * s) if s else 0)) & M32
