#!/bin/bash
# Round 6, lease 2: the F64 kernels on the GPU -- parity first, then the same-process A/B on the stock sets, then the suite.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or stock_sets_on_the_integer" > $OUT/pytest_f64.log 2>&1
tail -5 $OUT/pytest_f64.log
AB_REPS=3 timeout 900 python tools/f64_ab.py > $OUT/f64_ab.jsonl 2> $OUT/f64_ab.err
tail -3 $OUT/f64_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_b/f64_ab.jsonl"):
    d = json.loads(l)
    if d["id"] in ("relinearize", "rotate_columns", "inner_sum", "mul_and_relin", "mul", "ntt", "mul_and_relin_2"):
        print(json.dumps(d)[:400])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
