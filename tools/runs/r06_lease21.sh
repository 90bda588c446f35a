#!/bin/bash
# Round 6, lease 21: exact-fit scaler instances (NF = 3 / 5 / 8 / 10 / 16 / 18) against the previous release, libraries alternating.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_r
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_nf.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/scaler_nf_ab.py 2>/dev/null)}"
  done
done > $OUT/scaler_nf_ab.jsonl
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r06_r/scaler_nf_ab.jsonl")]
for k in rows[0]["t"]:
    if k.endswith("_ms"):
        a = statistics.median(r["t"][k] for r in rows if r["build"] == "before"); b = statistics.median(r["t"][k] for r in rows if r["build"] == "new")
        print(k.ljust(24), a, b, "new/before %.3f" % (b / a))
print({k: {r["t"][k] for r in rows} for k in rows[0]["t"] if k.endswith("digest")})
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scal or default or stock or config or extend or f64_multiply" > $OUT/pytest_scaler.log 2>&1; tail -3 $OUT/pytest_scaler.log
