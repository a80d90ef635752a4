"""Reduce rocprofv3 outputs (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into
profiles/<round>/...json.  Units and corrections follow MI355X_MICROARCH.md (HBM section):
FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of coalesced
reads (confirmed here on compact_kernel: it reads exactly what it writes, FETCH reads half)."""
import collections, csv, glob, json, os, sys

def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

def stats(path):
    out = {}
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                              "pct": float(r["Percentage"])}
    return out

def main():
    prof, fetch, write, dst = sys.argv[1:5]
    fs, ws, st = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE"), stats(prof)
    res = {}
    for k in st:
        if "flate::" not in k:
            continue
        f_kb, w_kb = fs.get(k, 0.0), ws.get(k, 0.0)
        res[k] = {"avg_ms": st[k]["avg_ns"] / 1e6, "calls": st[k]["calls"], "pct": st[k]["pct"],
                  "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb,
                  "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024),
                  "note": "FETCH_SIZE x2 (gfx950 coalesced-read correction) + WRITE_SIZE, KB->B"}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))

main()
