#!/bin/bash
# Round 6, lease 9: one-word F64 twiddles / key words + radix-8 passes at N = 16384 (new release build) against the previous release.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_j
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_mulmod2.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/lds_pad_ab.py 2>/dev/null)}"
  done
done | tee $OUT/f64_one_word_ab.jsonl | cut -c1-1000
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or default or stock" > $OUT/pytest_f64.log 2>&1; tail -2 $OUT/pytest_f64.log
python - > $OUT/f64_rates.json <<'PY'
import json, sys
sys.path.insert(0, ".")
import fhe_rs_amd as fhe
print(json.dumps({k: max(fhe.ubench_int(k, 0.2) for _ in range(3)) for k in fhe.UBENCH_KINDS}))
PY
cat $OUT/f64_rates.json | cut -c1-600
