#!/bin/bash
# MFMA utilisation / LDS bank conflicts of the config-5 GEMMs (SQ counters, own passes): tools/pmc_mfma.sh <tag>
set -u
TAG=${1:-mfma}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/mfma.log 2>&1
cd $ROOT
find $OUT -name "*counter_collection.csv" | head
