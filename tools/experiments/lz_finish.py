"""How much of the match finder's launch is its TAIL?  A -DFLATE_LZ_FINISH build records when every stream started
and ended and on which block (s_memrealtime, 100 MHz); a block is idle from its last stream's end to the end of the
launch.   FLATE_HIP_LIB=build/exp/libfinish.so python3 tools/experiments/lz_finish.py"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
flate = importlib.import_module("moonbit-flate_amd")
import torch
n = 16384
eng = flate.FlateEngine(0)
d = torch.from_numpy(flate.synth("text", n, 65536)).cuda()
off = flate.uniform_offsets(n, 65536)
eng.set_profiling(True)
L = importlib.import_module("moonbit-flate_amd._lib").load()
L.flate_hip_debug_lz_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
for rep in range(3):
    eng.deflate_batch(d, off)
    tm = eng.last_timing()
    buf = np.zeros((n, 8), dtype=np.uint64)
    k = L.flate_hip_debug_lz_stamps(eng._ctx, buf.ctypes.data, n)
    b = buf[:k]
    t0, t1 = b[:, 0].astype(np.float64), b[:, 1].astype(np.float64)
    kind, blk = (b[:, 2] >> np.uint64(32)).astype(np.int64), (b[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    start, end = t0.min(), t1.max()
    span_us = (end - start) / 100.0
    out = {}
    for kd, name in ((0, "LDS-table"), (1, "guest")):
        m = kind == kd
        ids = blk[m]
        last = {}
        first = {}
        cnt = {}
        for i, e, s in zip(ids, t1[m], t0[m]):
            last[i] = max(last.get(i, 0), e)
            first[i] = min(first.get(i, 1e30), s)
            cnt[i] = cnt.get(i, 0) + 1
        idle_tail = np.array([(end - v) / 100.0 for v in last.values()])
        idle_head = np.array([(v - start) / 100.0 for v in first.values()])
        per = (t1[m] - t0[m]) / 100.0
        out[name] = dict(blocks=len(last), streams=int(m.sum()), stream_us=round(float(per.mean()), 1),
                         tail_idle_us_mean=round(float(idle_tail.mean()), 1), tail_idle_us_max=round(float(idle_tail.max()), 1),
                         head_idle_us_mean=round(float(idle_head.mean()), 1))
    tot_blocks = sum(v["blocks"] for v in out.values())
    lost = sum(v["blocks"] * (v["tail_idle_us_mean"] + v["head_idle_us_mean"]) for v in out.values()) / tot_blocks
    print("launch span %.0f us (events: %.2f ms); block-time lost to head + tail: %.0f us = %.1f %% of the span" % (span_us, tm["lz77_match"], lost, 100 * lost / span_us), out)
