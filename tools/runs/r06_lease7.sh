#!/bin/bash
# Round 6, final lease: the whole GPU suite, smoke, the driver-shaped bench with rocprofv3 + PMC passes, soaks, on the HEAD build.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_final
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.out | cut -c1-1900
bash tools/collect_profiles.sh r06_final/prof 20 > $OUT/collect.log 2>&1
tail -1 $OUT/prof/bench.json | cut -c1-400
python tools/soak_f64.py 150 > $OUT/soak_f64.json 2> $OUT/soak_f64.err; cat $OUT/soak_f64.json
python tools/soak.py 150 2 > $OUT/soak_c2_streams2.json 2>/dev/null; cat $OUT/soak_c2_streams2.json
python tests/random_sweep_gpu.py 200 1000 100000 0 > $OUT/random_sweep_auto.json 2>/dev/null; cat $OUT/random_sweep_auto.json | cut -c1-200
python tests/random_sweep_gpu.py 200 100000 200000 0 f64 > $OUT/random_sweep_f64_auto.json 2>/dev/null; cat $OUT/random_sweep_f64_auto.json | cut -c1-200
python tests/random_sweep_gpu.py 120 0 1000 0 big > $OUT/random_sweep_big.json 2>/dev/null; cat $OUT/random_sweep_big.json | cut -c1-200
python bench.py --gpus 2 --batch 2048 --steps 5 --warmup 2 > $OUT/bench_gpus2_shared.out 2>/dev/null; tail -1 $OUT/bench_gpus2_shared.out | cut -c1-1900
