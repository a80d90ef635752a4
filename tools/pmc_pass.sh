#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> "<bench args>" COUNTERS...   (one small rocprofv3 --pmc pass)
set -e
out=$1; shift; bargs=$1; shift
timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/$out -o p --output-format csv -- python3 bench.py $bargs > gpurun_out/$out.log 2>&1
