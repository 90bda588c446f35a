#!/bin/bash
# Copies the summaries of a final lease (tools/runs/r06_lease12.sh layout) from gpurun_out/<tag>/ into profiles/<tag>_*.
# usage: tools/copy_evidence.sh r06_final4
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
P=$ROOT/profiles
tail -1 $SRC/bench_default.out > $P/${TAG}_bench_record_default_flags.json
grep '^DETAIL ' $SRC/bench_default.out | tail -1 | sed 's/^DETAIL //' > $P/${TAG}_bench_detail_default_flags.json
tail -1 $SRC/prof/bench.json > $P/${TAG}_bench_record_steps20.json
grep '^DETAIL ' $SRC/prof/bench.json | tail -1 | sed 's/^DETAIL //' > $P/${TAG}_bench_detail_steps20.json
python $ROOT/tools/pmc_to_traffic.py $SRC/prof $TAG > /dev/null
cp $SRC/pytest_gpu.log $P/${TAG}_pytest_gpu.log
cp $SRC/smoke.log $P/${TAG}_smoke.log
cp $SRC/f64_ab_v3.jsonl $P/${TAG}_f64_ab_v3.jsonl
cp $SRC/f64_rates.json $P/${TAG}_f64_rates.json
cp $SRC/soak_f64.json $P/${TAG}_soak_f64_vs_integer.json
cat $SRC/sweep_*.json > $P/${TAG}_random_sweeps.jsonl
cp $SRC/latency_breakdown.json $P/${TAG}_latency_breakdown.json
cp $SRC/stock_sets_profile.json $P/${TAG}_stock_sets_profile.json
ls -la $P/${TAG}_*
