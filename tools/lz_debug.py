"""Dev tool: first divergence between the wave and serial match finders on one stream."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flate = importlib.import_module("moonbit-flate_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
eng = flate.FlateEngine(0)
d = flate.synth(kind, 1, n)
off = flate.uniform_offsets(1, n)
a = eng.lz77_matches(d, off, lz_serial=True)[0]
b = eng.lz77_matches(d, off, lz_serial=False)[0]
print("serial", len(a[0]), "wave", len(b[0]))
k = 0
while k < min(len(a[0]), len(b[0])) and a[0][k] == b[0][k] and a[1][k] == b[1][k]:
    k += 1
print("first diff at record", k)
for j in range(max(0, k - 3), k + 4):
    sa = (int(a[0][j]), ((int(a[1][j]) >> 22) & 255) + 3, (int(a[1][j]) & 0x3fffff) + 1) if j < len(a[0]) else None
    sb = (int(b[0][j]), ((int(b[1][j]) >> 22) & 255) + 3, (int(b[1][j]) & 0x3fffff) + 1) if j < len(b[0]) else None
    print(j, "serial(pos,len,dist)", sa, "wave", sb)
