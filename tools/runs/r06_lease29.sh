#!/bin/bash
# Round 6, lease 29: long random-shape sweep + soak on the LAST build (parked streams), every family and key-switch strategy, fresh index ranges.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_z
mkdir -p $OUT
cd $ROOT
run() { python tests/random_sweep_gpu.py "$@" 2>/dev/null | tee -a $OUT/random_sweeps_long.jsonl | cut -c1-220; }
run 300 20000000 21000000 0 -
run 180 21000000 22000000 1 -
run 180 22000000 23000000 2 -
run 180 23000000 24000000 0 big
run 120 24000000 25000000 4 big
run 180 25000000 26000000 0 f64
run 120 26000000 27000000 1 f64
run 120 27000000 28000000 2 f64
run 120 28000000 29000000 1 f64wide
run 120 29000000 30000000 0 f64wide
python tools/soak_f64.py 300 > $OUT/soak_f64_long.json 2>/dev/null; cat $OUT/soak_f64_long.json
