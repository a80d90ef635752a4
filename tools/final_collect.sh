#!/bin/bash
# End-of-round evidence run (one gpurun call): gpu tests, soak, default bench, the same bench under
# rocprofv3 --kernel-trace --stats, fabric traffic of the match finder (configs 2 and 3) and of the
# config-5 inflate launch.   usage: FLATE_GIT_HEAD=<commit> [FLATE_SOAK_S=300] [FLATE_COLLECT_LITE=1] tools/final_collect.sh <tag>
# FLATE_COLLECT_LITE=1 stops after the bench under rocprofv3: for changes that leave the match finder and the
# inflaters alone (their build ids, hence the committed traffic figures and inflate profiles, stay valid).
# FLATE_COLLECT_PART=2 FLATE_TRAFFIC_SPLIT=<K> runs only what follows the lite part (a gpurun call is at most 20 minutes:
# part 1 = lite in one call, part 2 in the next, K = config.lz77_streams_by_kernel.lds_table_blocks of part 1's bench.json).
set -e
tag=$1
mkdir -p gpurun_out/$tag
if [ "$FLATE_COLLECT_PART" != "2" ]; then
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/$tag/gpu_tests.txt 2>&1 || { tail -20 gpurun_out/$tag/gpu_tests.txt; exit 1; }
tail -1 gpurun_out/$tag/gpu_tests.txt
timeout -k 10 900 python3 tests/tools/soak.py ${FLATE_SOAK_S:-300} > gpurun_out/$tag/soak.txt 2>&1 || { tail -20 gpurun_out/$tag/soak.txt; exit 1; }
tail -2 gpurun_out/$tag/soak.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1 || { tail -5 gpurun_out/$tag/smoke.txt; exit 1; }
tail -1 gpurun_out/$tag/smoke.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-extra > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.err
if [ -n "$FLATE_COLLECT_LITE" ]; then echo collected-lite; exit 0; fi
fi
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats_inf16k -o p --output-format csv -- python3 bench.py --mode inflate --streams 16384 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$tag/inflate16k_under_rocprof.json 2> gpurun_out/$tag/rocprof_inf16k.err
# the same for config 5 (inflate_simt_kernel) and config 3 (lz77_*_kernel<true>), the program directly behind `--`
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats_c5 -o p --output-format csv -- python3 bench.py --mode inflate --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/config5_under_rocprof.json 2> gpurun_out/$tag/rocprof_c5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats_c3 -o p --output-format csv -- python3 bench.py --streams 4096 --stream-len 262144 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/$tag/config3_under_rocprof.json 2> gpurun_out/$tag/rocprof_c3.err
# fabric traffic of the match finder AT THE QUEUE SPLIT OF THE HEADLINE RUN ABOVE (the guests' share moves the figure)
K=${FLATE_TRAFFIC_SPLIT:-$(python3 -c "import json;d=json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1]);print(d['config']['lz77_streams_by_kernel']['lds_table_blocks'])")}
FLATE_TRAFFIC_SPLIT=$K tools/traffic_collect.sh ${tag}_c2 16384 65536 > gpurun_out/$tag/traffic_c2.json 2> gpurun_out/$tag/traffic_c2.err
env -u FLATE_TRAFFIC_SPLIT tools/traffic_collect.sh ${tag}_c3 4096 262144 "--option window_units=0" > gpurun_out/$tag/traffic_c3.json 2> gpurun_out/$tag/traffic_c3.err
tools/inflate_traffic.sh ${tag}_inf > gpurun_out/$tag/traffic_inflate.json 2> gpurun_out/$tag/traffic_inflate.err
tools/inflate_traffic.sh ${tag}_inf16k 16384 > gpurun_out/$tag/traffic_inflate_16k.json 2> gpurun_out/$tag/traffic_inflate_16k.err
# (the crossover sweep takes minutes of its own: FLATE_COLLECT_CROSSOVER=1, or run tools/inflate_crossover.py in a call of its own)
if [ -n "$FLATE_COLLECT_CROSSOVER" ]; then
  timeout -k 10 400 python3 tools/inflate_crossover.py 256 1024 4096 8192 16384 32768 65536 > gpurun_out/$tag/inflate_crossover.txt 2>&1 || true
fi
echo collected
