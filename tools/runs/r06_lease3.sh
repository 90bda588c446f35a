#!/bin/bash
# Round 6, lease 3: strategy sweep for the F64 key switch, latency breakdown on head, bench + rocprofv3 + PMC passes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_c
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64_multiply_uses" > $OUT/pytest_one.log 2>&1; tail -2 $OUT/pytest_one.log
timeout 900 python tools/f64_ks_modes.py > $OUT/f64_ks_modes.jsonl 2> $OUT/f64_ks_modes.err; tail -2 $OUT/f64_ks_modes.err
cat $OUT/f64_ks_modes.jsonl | cut -c1-330
timeout 300 python tools/latency_breakdown.py > $OUT/latency_breakdown.json 2> $OUT/latency_breakdown.err
bash tools/collect_profiles.sh r06_c/prof 20 > $OUT/collect.log 2>&1
tail -1 $ROOT/gpurun_out/r06_c/prof/bench.json | cut -c1-1900
bash tools/collect_configs_pmc.sh r06_c/cfgpmc > $OUT/cfgpmc.log 2>&1
tail -30 $OUT/cfgpmc.log | cut -c1-200
