#!/bin/bash
# Round 6, final lease (6: parked internal streams on top of lease 5's build): the whole GPU suite, smoke, bench (default flags and the
# driver's flags) + rocprofv3 + PMC, latency breakdown, the F64 A/B, then random sweeps and the F64-vs-integer soak -- HEAD build.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_final6
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.out | cut -c1-1900
bash tools/collect_profiles.sh r06_final6/prof 20 > $OUT/collect.log 2>&1
tail -1 $OUT/prof/bench.json | cut -c1-1900
timeout 300 python tools/latency_breakdown.py > $OUT/latency_breakdown.json 2>/dev/null
AB_REPS=2 timeout 600 python tools/f64_ab.py > $OUT/f64_ab_v3.jsonl 2> $OUT/f64_ab_v3.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_final6/f64_ab_v3.jsonl"):
    d = json.loads(l)
    if d["id"] in ("relinearize", "rotate_columns", "inner_sum", "mul_and_relin", "ntt"):
        print(json.dumps(d)[:330])
PY
python tools/soak_f64.py 120 > $OUT/soak_f64.json 2>/dev/null; cat $OUT/soak_f64.json
python tests/random_sweep_gpu.py ${SWEEP_S:-120} 3000000 3100000 0 > $OUT/sweep_auto.json 2>/dev/null; cat $OUT/sweep_auto.json | cut -c1-200
python tests/random_sweep_gpu.py ${SWEEP_S:-120} 3100000 3200000 0 f64 > $OUT/sweep_f64_auto.json 2>/dev/null; cat $OUT/sweep_f64_auto.json | cut -c1-200
python tests/random_sweep_gpu.py ${SWEEP_S:-120} 3200000 3300000 1 f64wide > $OUT/sweep_f64wide_fused.json 2>/dev/null; cat $OUT/sweep_f64wide_fused.json | cut -c1-200
