"""Dev tool: speculative inflater vs input, first mismatch per stream."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from util import flate, make_streams
specs = [("text", 65536), ("ramp", 65536), ("zero", 70000), ("rand", 65536), ("low", 131072),
         ("period", 200000), ("runs", 65535), ("text", 0), ("text", 1), ("text", 16), ("text", 17),
         ("text", 127), ("text", 128), ("text", 300), ("text", 262144)]
data, off = make_streams(specs)
e = flate.FlateEngine(0)
comp, coff = e.deflate_batch(data, off)
sizes = off[1:] - off[:-1]
e.set_option("inflate_simt_min_streams", 1 << 30)
e.set_option("inflate_spec", 2)
out, ooff, olen, status, err = e.inflate_batch(comp, coff, sizes)
for i, (kind, n) in enumerate(specs):
    a = out[int(ooff[i]):int(ooff[i]) + n]
    b = data[int(off[i]):int(off[i]) + n]
    bad = np.nonzero(a != b)[0]
    print(i, kind, n, "status", status[i], "len", olen[i], "mismatches", bad.size, "first", bad[:8].tolist() if bad.size else None)
    if bad.size:
        k = int(bad[0])
        print("   got ", a[max(0, k - 8):k + 24].tolist())
        print("   want", b[max(0, k - 8):k + 24].tolist())

# map the first mismatches of stream 0 onto the encoder's tokens
from oracle import pyoracle
i = 0
n = specs[i][1]
src = data[int(off[i]):int(off[i]) + n]
toks = pyoracle.DeflateFast().encode(src[:65535])
pos, tl = 0, []
for t in toks:
    t = int(t)
    if t >= (1 << 30):
        L, D = ((t >> 22) & 0xff) + 3, (t & 0x3fffff) + 1
        tl.append((pos, "M", L, D)); pos += L
    else:
        tl.append((pos, "L", 1, t)); pos += 1
a = out[int(ooff[i]):int(ooff[i]) + n]
bad = np.nonzero(a != src)[0]
runs = []
for k in bad.tolist():
    if runs and k == runs[-1][1] + 1: runs[-1][1] = k
    else: runs.append([k, k])
import bisect
starts = [x[0] for x in tl]
for r in runs[:6]:
    j = bisect.bisect_right(starts, r[0]) - 1
    print("bad run", r, "got", a[r[0]:r[1] + 1].tolist(), "want", src[r[0]:r[1] + 1].tolist(), "tokens:", tl[max(0, j - 3):j + 2])

