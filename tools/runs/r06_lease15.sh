#!/bin/bash
# Round 6, lease 15: FHE_KS_AUTO's crossover for the F64 instances (N = 8192 in the 512-thread geometry, N = 16384 with one-word keys).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_m
mkdir -p $OUT
cd $ROOT
FHE_MODES_F64_ONLY=1 FHE_MODES_GRID="4096:32,48,64,80,96,128,160,192,256,384;8192:24,32,40,48,64,80,96,112,128,160,192,224,256,320,384,512;16384:8,12,16,20,24,28,32,40,48,56,64,72,80,96,112,128" \
  timeout 900 python tools/f64_ks_modes.py > $OUT/f64_ks_modes_grid.jsonl 2>$OUT/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_m/f64_ks_modes_grid.jsonl"):
    d = json.loads(l)
    print(d["n"], d["batch"], "auto %.4f fused %.4f unfused %.4f best %s auto/best %.3f" % (d["f64_auto_ms"], d["f64_fused_ms"], d["f64_unfused_ms"], d["f64_best"], d["f64_auto_over_best"]))
PY
tail -3 $OUT/err.log
