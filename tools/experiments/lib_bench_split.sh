#!/bin/bash
names=$1; shift
for v in $names; do
  FLATE_HIP_LIB=build/exp/lib$v.so python3 bench.py --steps 6 --warmup 2 --no-extra --cpu-sample-streams 1024 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['config']['stage_ms']['lz77_match'], d['config'].get('lz77_streams_by_kernel'), d['config']['parity_checked_streams'])"
done
