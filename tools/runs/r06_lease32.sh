#!/bin/bash
# Round 6, lease 32 (LAST build): what is left of the round's GPU budget on random shapes and the F64-vs-integer soak, fresh index ranges.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_zzzz
mkdir -p $OUT
cd $ROOT
run() { python tests/random_sweep_gpu.py "$@" 2>/dev/null | tee -a $OUT/random_sweeps.jsonl | cut -c1-220; }
run 200 60000000 61000000 0 -
run 200 61000000 62000000 2 -
run 150 62000000 63000000 0 f64
run 150 63000000 64000000 1 f64wide
python tools/soak_f64.py 300 > $OUT/soak_f64.json 2>/dev/null; cat $OUT/soak_f64.json
