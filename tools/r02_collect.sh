#!/bin/bash
# Round-2 evidence run on the GPU box: bench line, phase stamps, SQ counters of the match finder.
# usage: tools/r02_collect.sh <tag>
set -e
tag=$1
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py --steps 10 --warmup 3 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
cat gpurun_out/$tag/bench.json
if [ -f build/exp/libstamps.so ]; then
  python3 tools/lz_stamps.py 16384 text > gpurun_out/$tag/stamps.log 2>&1 || true
  tail -3 gpurun_out/$tag/stamps.log
fi
B="--steps 2 --warmup 1 --no-cpu-baseline"
pass() { # name benchargs counters...
  local name=$1; shift; local bargs=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/$tag/$name -o p --output-format csv -- python3 bench.py $bargs > gpurun_out/$tag/$name.log 2>&1
}
pass sqA_res "$B --no-guests" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass sqB_res "$B --no-guests" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass sqC_res "$B --no-guests" SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python3 tools/pmc_reduce.py gpurun_out/$tag/sq_resident.json gpurun_out/$tag/sqA_res gpurun_out/$tag/sqB_res gpurun_out/$tag/sqC_res > /dev/null
pass sqA_all "$B" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass sqB_all "$B" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
python3 tools/pmc_reduce.py gpurun_out/$tag/sq_default.json gpurun_out/$tag/sqA_all gpurun_out/$tag/sqB_all > /dev/null
echo done
