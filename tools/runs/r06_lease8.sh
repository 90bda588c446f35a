#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_g
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64_callers" > $OUT/pytest_callers.log 2>&1; tail -3 $OUT/pytest_callers.log
python tests/random_sweep_gpu.py 300 300000 400000 1 f64 > $OUT/sweep_f64_fused.json 2>/dev/null; cat $OUT/sweep_f64_fused.json | cut -c1-200
python tests/random_sweep_gpu.py 300 500000 600000 2 f64 > $OUT/sweep_f64_unfused.json 2>/dev/null; cat $OUT/sweep_f64_unfused.json | cut -c1-200
python tools/chunk_sweep_sets.py > $OUT/chunk_sweep_sets.jsonl 2>/dev/null; python - <<'PY'
import json, collections
best = collections.defaultdict(lambda: (0, None)); dflt = {}
for l in open("gpurun_out/r06_g/chunk_sweep_sets.jsonl"):
    d = json.loads(l)
    k = d.get("set"); v = d.get("ops_per_s", 0)
    if d.get("chunk") == 0 and d.get("streams") == 2: dflt[k] = v
    if v > best[k][0]: best[k] = (v, (d.get("streams"), d.get("chunk")))
for k in best: print(k, "default", dflt.get(k), "best", best[k])
PY
