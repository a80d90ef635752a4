#!/bin/bash
# usage: tools/team_sweep.sh <tag> "<res:guests> ..."   -- bench lines of the team match finder by geometry
tag=$1; shift
mkdir -p gpurun_out/$tag
for rg in $1; do
  r=${rg%%:*}; g=${rg##*:}
  python3 bench.py --steps 5 --warmup 2 --no-extra --cpu-sample-streams 2048 --option lz_team=1 --option team_resident_blocks=$r --option team_guest_blocks=$g > gpurun_out/$tag/b_${r}_${g}.json 2> gpurun_out/$tag/b_${r}_${g}.err || echo FAIL $rg
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/$tag/b_${r}_${g}.json").read().strip().splitlines()[-1])
print("$rg", d["value"], d["config"]["stage_ms"], d["config"]["parity_checked_streams"])
PY
done
