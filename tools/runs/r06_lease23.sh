#!/bin/bash
# Round 6, final lease (5: exact-fit scaler instances on top of lease 4): the whole GPU suite, smoke, bench + rocprofv3 + PMC, the F64 A/B and the rates, soaks, sweeps -- HEAD build.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_final5
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.out | cut -c1-1900
bash tools/collect_profiles.sh r06_final5/prof 20 > $OUT/collect.log 2>&1
tail -1 $OUT/prof/bench.json | cut -c1-300
AB_REPS=3 timeout 900 python tools/f64_ab.py > $OUT/f64_ab_v3.jsonl 2> $OUT/f64_ab_v3.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_final5/f64_ab_v3.jsonl"):
    d = json.loads(l)
    if d["id"] in ("relinearize", "rotate_columns", "inner_sum", "mul_and_relin", "ntt"):
        print(json.dumps(d)[:330])
PY
python - > $OUT/f64_rates.json <<'PY'
import json, sys
sys.path.insert(0, ".")
import fhe_rs_amd as fhe
print(json.dumps({k: max(fhe.ubench_int(k, 0.2) for _ in range(3)) for k in fhe.UBENCH_KINDS}))
PY
cat $OUT/f64_rates.json | cut -c1-900
python tools/soak_f64.py 300 > $OUT/soak_f64.json 2>/dev/null; cat $OUT/soak_f64.json
python tests/random_sweep_gpu.py 240 600000 700000 0 f64 > $OUT/sweep_f64_auto.json 2>/dev/null; cat $OUT/sweep_f64_auto.json | cut -c1-200
python tests/random_sweep_gpu.py 180 700000 800000 1 f64 > $OUT/sweep_f64_fused.json 2>/dev/null; cat $OUT/sweep_f64_fused.json | cut -c1-200
python tests/random_sweep_gpu.py 120 800000 900000 2 f64 > $OUT/sweep_f64_unfused.json 2>/dev/null; cat $OUT/sweep_f64_unfused.json | cut -c1-200
python tests/random_sweep_gpu.py 120 2000 100000 0 > $OUT/sweep_auto.json 2>/dev/null; cat $OUT/sweep_auto.json | cut -c1-200
timeout 300 python tools/latency_breakdown.py > $OUT/latency_breakdown.json 2>/dev/null
timeout 300 python tools/stock_sets_profile.py > $OUT/stock_sets_profile.json 2>/dev/null
python tests/random_sweep_gpu.py 150 900000 1000000 1 f64wide > $OUT/sweep_f64wide_fused.json 2>/dev/null; cat $OUT/sweep_f64wide_fused.json | cut -c1-200
python tests/random_sweep_gpu.py 90 1000000 1100000 0 f64wide > $OUT/sweep_f64wide_auto.json 2>/dev/null; cat $OUT/sweep_f64wide_auto.json | cut -c1-200
