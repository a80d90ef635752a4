"""Reduce the two PMC passes of tools/traffic_collect.sh: per kernel FETCH_SIZE / WRITE_SIZE (KB,
mean over dispatches) -> HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 reports
half of coalesced read bytes in FETCH_SIZE: MI355X_MICROARCH.md, HBM section; confirmed in round 1 on
a kernel that reads exactly what it writes).  The guest match finder is the exception: its fetches
are single 64-B sectors (2-byte table gathers), which FETCH_SIZE reports in full, so its raw value
is used.  Prints one JSON object."""
import collections, csv, glob, json, os, sys

d, n, blen, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])


def per_kernel(path, counter):
    per = collections.defaultdict(float)
    names = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for k, v in per.items():
        agg[names[k]].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


fs, ws = per_kernel(os.path.join(d, "FETCH_SIZE"), "FETCH_SIZE"), per_kernel(os.path.join(d, "WRITE_SIZE"), "WRITE_SIZE")
normal = json.loads(open(os.path.join(d, "normal.json")).read().strip().splitlines()[-1])
clen = normal["config"]["compressed_bytes_per_gpu"]
kern = {}
for k in sorted(set(fs) | set(ws)):
    if "flate::" in k:
        raw = "lz77_guest_kernel" in k
        kern[k] = {"FETCH_SIZE_bytes_raw": int(fs.get(k, 0.0) * 1024), "WRITE_SIZE_bytes": int(ws.get(k, 0.0) * 1024),
                   "read_rule": "raw: single 64-B sector requests (2-byte table gathers)" if raw else
                                "x2: coalesced wide reads (gfx950 FETCH_SIZE reports half)",
                   "hbm_bytes_per_launch": int(((1 if raw else 2) * fs.get(k, 0.0) + ws.get(k, 0.0)) * 1024)}
lz = sum(v["hbm_bytes_per_launch"] for k, v in kern.items() if "lz77" in k)
step = sum(v["hbm_bytes_per_launch"] for v in kern.values())
algo = n * blen + clen
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
build_id = importlib.import_module("moonbit-flate_amd").build_id()
print(json.dumps({
    "workload": "%d x %d B S-text streams" % (n, blen), "algorithmic_bytes": algo,
    # what this collection describes: bench.py quotes it only for a library with the same id
    "build_id": build_id, "git_head": os.environ.get("FLATE_GIT_HEAD", "unknown"),
    "queue_split": {"lds_table_blocks": K, "l2_table_guest_blocks": n - K,
                    "note": "split of the shared queue in an unprofiled run, fixed for the PMC passes "
                            "(option profile_split_streams) because rocprofv3 serialises the two kernels"},
    "hbm_bytes_per_launch": lz, "lz77_over_algorithmic": round(lz / algo, 3),
    "whole_step_hbm_bytes": step, "whole_step_over_algorithmic": round(step / algo, 3),
    "kernels": kern,
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/traffic_collect.sh); "
            "the counters sit at the L2's memory side, Infinity-Cache hits included (MI355X_MICROARCH.md), so for "
            "the guest kernel -- whose tables overflow the 4 MiB L2 of their XCD -- most of these bytes are "
            "L2 <-> Infinity Cache traffic, not HBM"}, indent=1))
