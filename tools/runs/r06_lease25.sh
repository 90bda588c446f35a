#!/bin/bash
# Round 6, lease 25: a last long random-shape sweep on the HEAD build, every family and key-switch strategy, fresh index ranges.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_u
mkdir -p $OUT
cd $ROOT
run() { python tests/random_sweep_gpu.py "$@" 2>/dev/null | tee -a $OUT/random_sweeps_long.jsonl | cut -c1-220; }
run 420 3000000 4000000 0 -
run 240 4000000 5000000 1 -
run 240 5000000 6000000 2 -
run 240 6000000 7000000 0 big
run 120 7000000 8000000 4 big
run 240 8000000 9000000 0 f64
run 180 9000000 10000000 1 f64
run 120 10000000 11000000 2 f64
run 180 11000000 12000000 1 f64wide
run 120 12000000 13000000 0 f64wide
python tools/soak_f64.py 600 > $OUT/soak_f64_long.json 2>/dev/null; cat $OUT/soak_f64_long.json
