#!/bin/bash
# usage: tools/inf_bench.sh "<variant names>" [bench args]  -- inflate A/B of build/exp/lib<name>.so builds
names=$1; shift
for v in $names; do
  FLATE_HIP_LIB=build/exp/lib$v.so python3 bench.py --mode inflate --steps 3 --warmup 1 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['config'].get('stage_ms'))"
done
