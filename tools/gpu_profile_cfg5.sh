# config-5 measurement set (run through gpurun from the repo root): tools/gpu_profile_cfg5.sh <tag> -> gpurun_out/<tag>/{stats5,pmc5_fetch,pmc5_write}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run5}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -o s -- python3 $ROOT/bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats5.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc5_fetch -o f -- python3 $ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc5_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc5_write -o w -- python3 $ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc5_write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head
tail -2 $OUT/stats5.log
