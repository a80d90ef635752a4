#!/bin/bash
for rg in $1; do
  r=${rg%%:*}; g=${rg##*:}
  python3 bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --streams 4096 --stream-len 262144 --option resident_blocks=$r --option guest_blocks=$g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rg', d['value'], d['config']['stage_ms']['lz77_match'])"
done
