#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, and
# the two separate PMC passes (FETCH_SIZE, WRITE_SIZE) the microarch guide prescribes.
# Outputs land under gpurun_out/$TAG; tools/pmc_to_traffic.py turns them into profiles/.
TAG=${1:-final}
STEPS=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps $STEPS --warmup 5"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $BENCH --no-cpu --no-extras > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $BENCH --no-cpu --no-extras > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $BENCH --no-cpu --no-extras > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.log
# keep only the small summaries (kernel_trace of a PMC pass can be large)
find $OUT -name '*kernel_trace.csv' -size +8M -delete
ls -laR $OUT | tail -30
