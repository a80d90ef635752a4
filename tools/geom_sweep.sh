#!/bin/bash
# usage: tools/geom_sweep.sh <tag> "<res:guests> ..."  -- one-wave match finder by launch geometry
tag=$1; shift
mkdir -p gpurun_out/$tag
for rg in $1; do
  r=${rg%%:*}; g=${rg##*:}
  python3 bench.py --steps 5 --warmup 2 --no-extra --cpu-sample-streams 1024 --option resident_blocks=$r --option guest_blocks=$g > gpurun_out/$tag/b_${r}_${g}.json 2> gpurun_out/$tag/b_${r}_${g}.err || echo FAIL $rg
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/$tag/b_${r}_${g}.json").read().strip().splitlines()[-1])
print("$rg", d["value"], d["config"]["stage_ms"])
PY
done
