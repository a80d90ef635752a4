#!/bin/bash
# usage: tools/pmc_multi.sh <tag> "<bench args>" "<pass1 counters>" "<pass2 counters>" ...
# one rocprofv3 --pmc pass per counter group (kernel-trace only), reduced into gpurun_out/<tag>/pmc.json
tag=$1; shift; bargs=$1; shift
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0; dirs=""
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/$tag/p$i -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $bargs > gpurun_out/$tag/p$i.log 2>&1 || echo "pass $i failed"
  dirs="$dirs gpurun_out/$tag/p$i"
done
python3 tools/pmc_reduce.py gpurun_out/$tag/pmc.json $dirs > /dev/null
python3 - <<PY
import json
d=json.load(open("gpurun_out/$tag/pmc.json"))
for k,v in d.items():
    if "lz77" in k or "inflate" in k:
        print(k)
        for c,x in sorted(v["per_launch"].items()): print("   %-40s %16.0f"%(c,x))
PY
