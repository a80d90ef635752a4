#!/bin/bash
# usage: tools/experiments/guest_geometry2.sh "<guest_blocks> ..." <streams> <stream_len>   the same launch at several numbers of guest blocks, fresh process each
for g in $1; do
  python3 bench.py --steps 6 --warmup 3 --no-extra --cpu-sample-streams 256 --streams $2 --stream-len $3 --option guest_blocks=$g 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $2 x $3 guests $g', d['value'], d['config']['stage_ms']['lz77_match'], d['config']['lz77_streams_by_kernel'])"
done
