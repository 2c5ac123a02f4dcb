#!/bin/bash
# SQ counter passes for one config-5 step (diagnosis): tools/pmc_sq.sh <tag>  -> gpurun_out/<tag>/sq{1,2}/
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq1 -o a -- python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -o b -- python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/sq2.log 2>&1
cd $ROOT
find $OUT -name "*counter_collection.csv" | head
