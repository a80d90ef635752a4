#!/bin/bash
# usage: tools/experiments/guest_geometry.sh "<variant>:<guest_blocks> ..."   A/B of build/exp/lib<variant>.so at a number of guest blocks
for vg in $1; do
  v=${vg%%:*}; g=${vg##*:}
  FLATE_HIP_LIB=build/exp/lib$v.so python3 bench.py --steps 6 --warmup 3 --no-extra --cpu-sample-streams 1024 --option guest_blocks=$g 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v guests $g', d['value'], d['config']['stage_ms']['lz77_match'], d['config']['lz77_streams_by_kernel'], d['config']['parity_checked_streams'])"
done
