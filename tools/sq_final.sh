#!/bin/bash
# SQ counters of the two match-finder kernels of the default launch, with the queue split fixed
# (rocprofv3 serialises them): usage tools/sq_final.sh <tag> <lds_table_share>
tag=$1; K=$2
tools/pmc_multi.sh $tag "--option profile_split_streams=$K" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
