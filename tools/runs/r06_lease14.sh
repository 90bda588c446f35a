#!/bin/bash
# Round 6, lease 14: F64 key switch in the 16-coefficients-per-thread geometry at N = 4096 / 8192 (new release) vs the previous release.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_l
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_t512.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/f64_geometry_ab.py 2>/dev/null)}"
  done
done > $OUT/f64_geometry_ab.jsonl
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r06_l/f64_geometry_ab.jsonl")]
for k in rows[0]["t"]:
    if k.endswith("_ms"):
        a = statistics.median(r["t"][k] for r in rows if r["build"] == "before"); b = statistics.median(r["t"][k] for r in rows if r["build"] == "new")
        print(k.ljust(32), a, b, "new/before %.3f" % (b / a))
print({k: {r["t"][k] for r in rows} for k in rows[0]["t"] if k.endswith("digest")})
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or default or stock" > $OUT/pytest_f64.log 2>&1; tail -2 $OUT/pytest_f64.log
timeout 600 python tools/f64_ks_modes.py > $OUT/f64_ks_modes.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r06_l/f64_ks_modes.jsonl"):
    d = json.loads(l)
    print(d["n"], d["batch"], "auto %.4f fused %.4f unfused %.4f best %s auto/best %.3f" % (d["f64_auto_ms"], d["f64_fused_ms"], d["f64_unfused_ms"], d["f64_best"], d["f64_auto_over_best"]))
PY
