#!/bin/bash
# kernel stats + HBM counters of one bench.py command (run through gpurun from the repo root):
#   tools/gpu_profile_any.sh <tag> <bench.py arguments...>  -> gpurun_out/<tag>/{bench.json, stats/, pmc_fetch/, pmc_write/}
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --no-roofline > $OUT/pmc_write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -12
