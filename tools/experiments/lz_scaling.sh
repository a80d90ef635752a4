#!/bin/bash
# usage: tools/experiments/lz_scaling.sh <tag>  -- match-finder time against the number of 64 KiB streams in the
# batch (what a batch pays once, ramp + tail of the persistent launch, against what it pays per stream)
tag=$1; mkdir -p gpurun_out/$tag
for n in 4096 8192 16384 32768 65536; do
  python3 bench.py --streams $n --steps 6 --warmup 2 --no-extra --no-cpu-baseline --verify 0 > gpurun_out/$tag/n$n.json 2> gpurun_out/$tag/n$n.err || { echo FAIL $n; tail -3 gpurun_out/$tag/n$n.err; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/$tag/n$n.json').read().strip().splitlines()[-1])
print($n, d['value'], d['ms_per_step'], d['config']['stage_ms'])"
done
