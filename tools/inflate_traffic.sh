#!/bin/bash
# HBM traffic of the config-5 inflate launch (131072 x 64 KiB streams): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes.  usage: tools/inflate_traffic.sh <tag> [streams]
set -e
tag=$1; n=${2:-131072}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/$tag/$ctr -o p --output-format csv -- python3 bench.py --mode inflate --streams $n --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/$ctr.log 2>&1
done
python3 - <<PY
import collections, csv, glob, json, os, sys, importlib
sys.path.insert(0, os.getcwd())
def per_kernel(path, counter):
    per = collections.defaultdict(float); names = {}
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for k, v in per.items(): agg[names[k]].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}
fs, ws = per_kernel("gpurun_out/$tag/FETCH_SIZE", "FETCH_SIZE"), per_kernel("gpurun_out/$tag/WRITE_SIZE", "WRITE_SIZE")
out = {"build_id": importlib.import_module("moonbit-flate_amd").build_id(),
       "git_head": os.environ.get("FLATE_GIT_HEAD", "unknown")}
for k in fs:
    if "inflate" in k:
        out[k] = {"FETCH_SIZE_bytes_raw": int(fs[k] * 1024), "WRITE_SIZE_bytes": int(ws.get(k, 0) * 1024)}
print(json.dumps(out, indent=1))
PY
