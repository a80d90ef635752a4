#!/bin/bash
# usage: tools/huff_ab.sh <tag> "<variant names>"  -- A/B of build/exp/lib<name>.so builds with per-kernel
# times of the entropy stage (rocprofv3 --kernel-trace --stats of a short default bench; one gpurun call)
tag=$1; names=$2
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in $names; do
  export FLATE_HIP_LIB=build/exp/lib$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/$v -o p --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/$tag/$v.json 2> gpurun_out/$tag/$v.err || { echo FAIL $v; tail -5 gpurun_out/$tag/$v.err; }
  python3 - "$v" gpurun_out/$tag <<'PY'
import sys, json, csv, glob
v, d = sys.argv[1], sys.argv[2]
try:
    line = json.loads(open(f"{d}/{v}.json").read().strip().splitlines()[-1])
    print(v, "value", line["value"], line["config"]["stage_ms"], "parity", line["config"]["parity_checked_streams"])
except Exception as e:
    print(v, "no bench line", e)
for f in glob.glob(f"{d}/{v}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "huff" in r["Name"] or "scan" in r["Name"] or "lz77" in r["Name"]:
            print("   %-60s %3s calls  avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
