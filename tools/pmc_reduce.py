"""Reduce rocprofv3 --pmc passes (csv) into one JSON: per kernel, per counter, the mean over
dispatches, plus per-wave values when SQ_WAVES was collected.
usage: tools/pmc_reduce.py OUT.json DIR [DIR...]   (each DIR = one rocprofv3 -d directory)"""
import collections, csv, glob, json, os, sys


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
            for (disp, ctr), v in per_dispatch.items():
                agg[names[disp]][ctr].append(v)
    res = {}
    for k, ctrs in agg.items():
        if "flate::" not in k:
            continue
        m = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e = {"dispatches": max(len(v) for v in ctrs.values()), "per_launch": m}
        if m.get("SQ_WAVES"):
            e["per_wave"] = {c: v / m["SQ_WAVES"] for c, v in m.items() if c != "SQ_WAVES"}
        res[k] = e
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


main()
