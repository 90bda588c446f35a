#!/bin/bash
# Round 6, lease 19: the F64 key switch at N = 8192 picks its geometry per launch (1024 x 8 up to one workgroup per CU, 512 x 16 above);
# where FHE_KS_AUTO's unfused window ends for it.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_p
mkdir -p $OUT
cd $ROOT
FHE_MODES_F64_ONLY=1 FHE_MODES_GRID="8192:8,16,24,28,32,36,40,44,48,51,52,56,64,72,80,96,128,256,1024" \
  timeout 900 python tools/f64_ks_modes.py > $OUT/f64_ks_modes_grid_two_geometries.jsonl 2>$OUT/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_p/f64_ks_modes_grid_two_geometries.jsonl"):
    d = json.loads(l)
    print(d["n"], d["batch"], "auto %.4f fused %.4f unfused %.4f best %s auto/best %.3f" % (d["f64_auto_ms"], d["f64_fused_ms"], d["f64_unfused_ms"], d["f64_best"], d["f64_auto_over_best"]))
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or default or stock or auto_picks" > $OUT/pytest_f64.log 2>&1; tail -3 $OUT/pytest_f64.log
