#!/bin/bash
# Round 6, lease 31 (LAST build): the driver's own N = 1 command line, verbatim, then random shapes with the strategies the earlier
# leases of this build ran least (forced fused / unfused on the F64 families, rows larger than LDS in every strategy).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_zzz
mkdir -p $OUT
cd $ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.out 2> $OUT/bench_driver_line.err; tail -1 $OUT/bench_driver_line.out | cut -c1-400; tail -1 $OUT/bench_driver_line.out | wc -c
run() { python tests/random_sweep_gpu.py "$@" 2>/dev/null | tee -a $OUT/random_sweeps.jsonl | cut -c1-220; }
run 180 50000000 51000000 1 f64
run 180 51000000 52000000 2 f64
run 180 52000000 53000000 0 f64wide
run 150 53000000 54000000 0 big
run 150 54000000 55000000 2 big
run 150 55000000 56000000 4 big
run 180 56000000 57000000 1 -
