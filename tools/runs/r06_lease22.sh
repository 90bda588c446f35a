#!/bin/bash
# Round 6, lease 22: scaler instances for the levels of the C5 chain (NF = 14 / 23 / 27 / 31) against the previous release, libraries alternating.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_s
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_nf2.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/scaler_nf_ab.py 2>/dev/null)}"
  done
done > $OUT/scaler_nf_ab.jsonl
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r06_s/scaler_nf_ab.jsonl")]
for k in rows[0]["t"]:
    if k.endswith("_ms") and not k.endswith("per_level_ms"):
        a = statistics.median(r["t"][k] for r in rows if r["build"] == "before"); b = statistics.median(r["t"][k] for r in rows if r["build"] == "new")
        print(k.ljust(24), a, b, "new/before %.3f" % (b / a))
print({k: {r["t"][k] for r in rows} for k in rows[0]["t"] if k.endswith("digest")})
for b in ("before", "new"):
    print(b, [round(statistics.median(r["t"]["c5_chain_per_level_ms"][i] for r in rows if r["build"] == b), 3) for i in range(15)])
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scal or config or chain or c5 or extend" > $OUT/pytest_scaler.log 2>&1; tail -3 $OUT/pytest_scaler.log
