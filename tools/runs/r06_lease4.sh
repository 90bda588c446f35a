#!/bin/bash
# Round 6, lease 4: A/B of the forward transform's direct last-pass store (lab variant) against the release build.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_d
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
for round in 1 2 3; do
  for v in lds direct; do
    if [ $v = direct ]; then cp tools/_variants/libfhe_hip_fwd_direct.so fhe.rs_amd/libfhe_hip.so; else cp tools/_variants/libfhe_hip_fwd_lds.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/fwd_direct_ab.py 2>/dev/null)}"
  done
done | tee $OUT/fwd_direct_store_ab.jsonl | cut -c1-420
cp tools/_variants/libfhe_hip_fwd_direct.so fhe.rs_amd/libfhe_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or c2 or f64" > $OUT/pytest_direct.log 2>&1; tail -2 $OUT/pytest_direct.log
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
# the release build with the trimmed F64 epilogue: parity, the A/B again, where a stock-set multiply spends its time
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or default or stock" > $OUT/pytest_f64_v2.log 2>&1; tail -2 $OUT/pytest_f64_v2.log
AB_REPS=3 AB_SETS=8192,16384 timeout 900 python tools/f64_ab.py > $OUT/f64_ab_v2.jsonl 2> $OUT/f64_ab_v2.err
python - <<'PY'
import json
for l in open("gpurun_out/r06_d/f64_ab_v2.jsonl"):
    d = json.loads(l)
    if d["id"] in ("relinearize", "rotate_columns", "inner_sum", "mul_and_relin", "ntt"):
        print(json.dumps(d)[:330])
PY
timeout 300 python tools/stock_sets_profile.py > $OUT/stock_sets_profile.json 2> $OUT/stock_sets_profile.err; tail -c 1500 $OUT/stock_sets_profile.json
