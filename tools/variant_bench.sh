#!/bin/bash
# usage: tools/variant_bench.sh <tag> "<opt=val,opt=val> ..."  -- bench lines for sets of engine options
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for set in $1; do
  i=$((i+1))
  opts=""
  for kv in ${set//,/ }; do opts="$opts --option $kv"; done
  python3 bench.py --steps 5 --warmup 2 --no-extra --cpu-sample-streams 2048 $opts > gpurun_out/$tag/b$i.json 2> gpurun_out/$tag/b$i.err || echo FAIL $set
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/$tag/b$i.json").read().strip().splitlines()[-1])
print("$set", d["value"], d["config"]["stage_ms"], d["config"]["parity_checked_streams"])
PY
done
