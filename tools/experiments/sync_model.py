"""Host model for the sub-block inflater: how fast does a deflate token decoder started at a wrong bit
offset fall in step with the real token chain?  (decides sub-block size / rounds before any HIP)

For one stream produced by the oracle: parse the first dynamic block's header, list the true token
starts, then for every sub-block boundary k * SB start a decoder at that guess and record the bit
distance until its token starts coincide with the true chain.
usage: sync_model.py [kind] [SB ...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle
flate = importlib.import_module("moonbit-flate_amd.synth") if False else None

ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def canon(lens):
    codes, code, mx = {}, 0, max(lens)
    bl = [0] * (mx + 2)
    for l in lens:
        if l:
            bl[l] += 1
    nxt = [0] * (mx + 2)
    for b in range(1, mx + 1):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    for s, l in enumerate(lens):
        if l:
            codes[(l, nxt[l])] = s
            nxt[l] += 1
    return codes, mx


class Bits:
    def __init__(self, data):
        self.bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8), bitorder="little")
        self.n = len(self.bits)

    def get(self, pos, n):
        v = 0
        for i in range(n):
            v |= int(self.bits[pos + i]) << i
        return v

    def sym(self, pos, table):
        codes, mx = table
        c = 0
        for l in range(1, mx + 1):
            if pos + l > self.n:
                return None, 0
            c = (c << 1) | int(self.bits[pos + l - 1])
            s = codes.get((l, c))
            if s is not None:
                return s, l
        return None, 0


LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]


def token(b, pos, lit, dist):
    """-> bits of the token at pos, or 0 if it is not a literal / match (EOB, invalid)"""
    s, l = b.sym(pos, lit)
    if s is None or s == 256 or s > 285:
        return 0
    if s < 256:
        return l
    t = l + LEXTRA[s - 257]
    d, dl = b.sym(pos + t, dist)
    if d is None or d > 29:
        return 0
    return t + dl + (0 if d < 4 else (d - 2) >> 1)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "text"
    sbs = [int(x) for x in sys.argv[2:]] or [96, 160, 224, 288]
    synth = importlib.import_module("moonbit-flate_amd").synth
    data = synth(kind, 4, 65536)[65536:131072].tobytes()
    comp = bytes(pyoracle.deflate(data))
    b = Bits(comp)
    pos = 0
    hdr = b.get(pos, 3); pos += 3
    assert hdr >> 1 == 2, "first block is not dynamic"
    nlit, ndist, nclen = b.get(pos, 5) + 257, b.get(pos + 5, 5) + 1, b.get(pos + 10, 4) + 4
    pos += 14
    cl = [0] * 19
    for i in range(nclen):
        cl[ORDER[i]] = b.get(pos, 3); pos += 3
    clt = canon(cl)
    lens = []
    while len(lens) < nlit + ndist:
        s, l = b.sym(pos, clt); pos += l
        if s < 16:
            lens.append(s)
        elif s == 16:
            lens += [lens[-1]] * (3 + b.get(pos, 2)); pos += 2
        elif s == 17:
            lens += [0] * (3 + b.get(pos, 3)); pos += 3
        else:
            lens += [0] * (11 + b.get(pos, 7)); pos += 7
    lit, dist = canon(lens[:nlit]), canon(lens[nlit:])
    first = pos
    true = set()
    while True:
        true.add(pos)
        t = token(b, pos, lit, dist)
        if t == 0:
            break
        pos += t
    end = pos
    print(f"{kind}: {len(comp)} bytes, {len(true)} tokens, {(end - first) / len(true):.2f} bits per token; long codes: max lit {lit[1]}, dist {dist[1]}")
    for SB in sbs:
        ds, never = [], 0
        for g in range(first + SB, end - 64, SB):
            p = g
            while p not in true:
                t = token(b, p, lit, dist)
                if t == 0 or p > g + 4096:
                    p = None
                    break
                p += t
            if p is None:
                never += 1
                ds.append(1 << 30)
            else:
                ds.append(p - g)
        ds = np.array(ds)
        ok = ds < (1 << 30)
        q = lambda f: int(np.quantile(ds[ok], f))
        within = [(ds <= k * SB).mean() for k in (1, 2, 3, 4)]
        # the algorithm itself: 64 lanes, lane L decodes from its start until a token starts at or behind
        # the end of its sub-block; rounds until no lane's start changes
        nxt = {}
        def run(p, lim):
            n = 0
            while p < lim:
                t = nxt.get(p)
                if t is None:
                    t = nxt[p] = token(b, p, lit, dist)
                if t == 0:
                    return p, True, n
                p += t
                n += 1
            return p, False, n
        rounds, maxtok = [], 0
        g = first
        while g + 64 * SB + 64 < end:
            st = [g + L * SB for L in range(64)]
            res = [run(st[L], g + (L + 1) * SB) for L in range(64)]
            r = 1
            while True:
                ch = [L for L in range(1, 64) if not res[L - 1][1] and res[L - 1][0] != st[L]]
                if not ch:
                    break
                for L in ch:
                    st[L] = res[L - 1][0]
                new = {L: run(st[L], g + (L + 1) * SB) for L in ch}
                res = [new.get(L, res[L]) for L in range(64)]
                r += 1
            rounds.append(r)
            maxtok = max(maxtok, max(x[2] for x in res))
            stop = [L for L in range(64) if res[L][1]]
            g = res[63][0] if not stop else end
        hist = np.bincount(rounds)
        print(f"SB {SB}: sync distance median {q(.5)} p90 {q(.9)} p99 {q(.99)} max {int(ds[ok].max())}, dead {never}; "
              f"within 1..4 sub-blocks: {' '.join('%.3f' % w for w in within)}; batches {len(rounds)}, rounds mean {np.mean(rounds):.2f} hist {hist.tolist()}, most tokens in a sub-block {maxtok}")


main()
