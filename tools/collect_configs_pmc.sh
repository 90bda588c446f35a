#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) and
# kernel durations of the C3 (relinearise, rotations; batch 512) and C5 (level-0 multiply + relin + mod switch;
# batch 16) workloads of tools/bench_configs.py.  Runs on the GPU box via gpurun; tools/pmc_configs_summary.py
# turns the counter CSVs into profiles/<round>_configs_pmc.json.
TAG=${1:-cfgpmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
  RUN="python $ROOT/tools/bench_configs.py $cfg"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_stats -o run -- $RUN > $OUT/${cfg}_stats.jsonl 2> $OUT/${cfg}_stats.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${cfg}_fetch -o run -- $RUN > $OUT/${cfg}_fetch.jsonl 2> $OUT/${cfg}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${cfg}_write -o run -- $RUN > $OUT/${cfg}_write.jsonl 2> $OUT/${cfg}_write.log
done
find $OUT -name '*kernel_trace.csv' -size +8M -delete
python $ROOT/tools/pmc_configs_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json | head -80
