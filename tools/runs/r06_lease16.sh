#!/bin/bash
# Round 6, lease 16: the refit FHE_KS_AUTO rule for the F64 instances (auto vs best over the grid), its pin test, F64 / stock parity.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_n
mkdir -p $OUT
cd $ROOT
FHE_MODES_F64_ONLY=1 FHE_MODES_GRID="4096:16,32,64,128,256,1024;8192:8,16,24,32,40,48,56,64,72,80,96,112,128,160,192,224,256,512,1024;16384:4,8,12,16,20,24,28,32,36,40,48,56,64,72,80,96,128,256" \
  timeout 900 python tools/f64_ks_modes.py > $OUT/f64_ks_modes_grid_after.jsonl 2>$OUT/err.log
python - <<'PY'
import json
worst = 0
for l in open("gpurun_out/r06_n/f64_ks_modes_grid_after.jsonl"):
    d = json.loads(l)
    worst = max(worst, d["f64_auto_over_best"])
    print(d["n"], d["batch"], "auto %.4f fused %.4f unfused %.4f best %s auto/best %.3f" % (d["f64_auto_ms"], d["f64_fused_ms"], d["f64_unfused_ms"], d["f64_best"], d["f64_auto_over_best"]))
print("worst auto/best", worst)
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or default or stock or auto_picks" > $OUT/pytest_f64.log 2>&1; tail -3 $OUT/pytest_f64.log
