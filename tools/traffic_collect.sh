#!/bin/bash
# HBM traffic of the default (concurrent resident + guest) match-finder launch and of the whole
# step.  rocprofv3 --pmc serialises kernels, so (1) a normal run reports how the shared queue was
# split, (2) the PMC passes run with that split FIXED (option profile_split_streams): each kernel
# does its share although they no longer overlap.  FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: they do not fit one pass); reduced by tools/traffic_reduce.py.
# usage: tools/traffic_collect.sh <tag> <streams> <stream_len> ["extra bench args"]
set -e
tag=$1; n=$2; blen=$3; extra="${4:-}"
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="--streams $n --stream-len $blen --no-extra --no-cpu-baseline $extra"
python3 bench.py $B --steps 5 --warmup 2 > gpurun_out/$tag/normal.json 2> gpurun_out/$tag/normal.err
K=$(python3 -c "import json;d=json.loads(open('gpurun_out/$tag/normal.json').read().strip().splitlines()[-1]);print(d['config']['lz77_streams_by_kernel']['lds_table_blocks'])")
# FLATE_TRAFFIC_SPLIT: use the split of another (the timed) run instead of this call's own normal run
if [ -n "$FLATE_TRAFFIC_SPLIT" ]; then echo "resident share of this call's run: $K of $n; collecting at the given split $FLATE_TRAFFIC_SPLIT"; K=$FLATE_TRAFFIC_SPLIT; fi
echo "resident share: $K of $n"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/$tag/$ctr -o p --output-format csv -- python3 bench.py $B --steps 3 --warmup 1 --option profile_split_streams=$K > gpurun_out/$tag/$ctr.log 2>&1
done
python3 tools/traffic_reduce.py gpurun_out/$tag $n $blen $K
