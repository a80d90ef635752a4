#!/bin/bash
# usage: tools/experiments/inflate_lib_ab.sh "<variant names>" [streams]   A/B of build/exp/lib<name>.so on the inflate bench
n=${2:-131072}
for v in $1; do
  FLATE_HIP_LIB=build/exp/lib$v.so python3 bench.py --mode inflate --streams $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v streams $n', d['value'], d['ms_per_step'])"
done
