#!/bin/bash
# Standard measurement set on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile.sh <tag>      -> gpurun_out/<tag>/{bench_cfg2.json, kernel_stats.csv, pmc_*.csv, bench_cfg{3,4,5}.json}
# rocprofv3 gets the program itself after `--` (python3 ...), counters in their own passes (no trace domains besides
# --kernel-trace), as the pool requires.
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 200 python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/pmc_write.log 2>&1
cd $ROOT
for c in 3 4 5; do
  timeout -k 10 300 python3 bench.py --config $c > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
find $OUT -name "*.csv" | head -20
