"""Many small encode calls on the DEFAULT path (match finder, then the entropy kernels), alternating four
batches of one shape and different data, bytes compared with a first run every time: the counterpart of
the stress runs that looked for the overlapped stage's one red soak run (profiles/r04/README.md section 7).
    python3 tools/experiments/default_stress.py [reps] [un] [ulen] [guest_blocks]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
flate = importlib.import_module("moonbit-flate_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
un = int(sys.argv[2]) if len(sys.argv) > 2 else 87
ulen = int(sys.argv[3]) if len(sys.argv) > 3 else 65535
gb = int(sys.argv[4]) if len(sys.argv) > 4 else 64
eng = flate.FlateEngine(0)
eng.set_option("guest_min_streams", 1)
eng.set_option("guest_blocks", gb)
uo = flate.uniform_offsets(un, ulen)
batches = []
for kind, first in (("text", 598163), ("text", 17), ("ramp", 0), ("text", 90001)):
    ud = flate.synth(kind, un, ulen, first_stream=first)
    ref, roff = eng.deflate_batch(ud, uo, compat_go=True)
    batches.append((ud, ref[:int(roff[-1])].copy(), roff.copy()))
fails, t0 = 0, time.time()
for r in range(reps):
    ud, ref, roff = batches[r % len(batches)]
    try:
        out, ooff = eng.deflate_batch(ud, uo, compat_go=True)
        ok = np.array_equal(ooff, roff) and np.array_equal(out[:int(ooff[-1])], ref)
        why = "bytes differ"
    except flate.FlateError as e:
        ok, why = False, str(e)[:140]
    if not ok:
        fails += 1
        if fails <= 20:
            print("FAIL rep=%d: %s" % (r, why), flush=True)
    if r % 5000 == 4999:
        print("rep %d: %d failures, %.0f s" % (r + 1, fails, time.time() - t0), flush=True)
print("DONE: %d reps, %d failures" % (reps, fails))
