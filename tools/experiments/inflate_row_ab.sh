#!/bin/bash
# A/B of the lane-per-stream inflater's output row (option inflate_row_dwords = 0 / 8 / 16), config 5 and 65536 streams
for streams in 131072 65536; do
  for r in 0 8 16 0 8 16; do
    python3 bench.py --mode inflate --streams $streams --steps 3 --warmup 1 --no-cpu-baseline --option inflate_row_dwords=$r 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $streams row $r', d['value'], d['ms_per_step'], d['config'].get('workload','')[:60])"
  done
done
