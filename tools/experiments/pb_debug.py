import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from util import flate, make_streams
from oracle import pyoracle
e = flate.FlateEngine(0)
e.set_option("entropy_per_block", 1)
for spec in [("rand", 2 * 65535 + 9), ("rand", 65535 + 9), ("rand", 65535), ("rand", 65536), ("rand", 2 * 65535), ("text", 2 * 65535 + 9), ("rand", 65535 + 200), ("zero", 2*65535+9)]:
    data, off = make_streams([spec], seed=21)
    try:
        out, ooff = e.deflate_batch(data, off)
        ok = bytes(out[:int(ooff[1])]) == pyoracle.deflate(data[:int(off[1])])
        print(spec, "ok" if ok else "BYTES DIFFER", flush=True)
    except Exception as ex:
        print(spec, "ERR", str(ex)[:120], flush=True)
