#!/bin/bash
# Round 6, lease 1: the FP64 gate, the compact bench record with default flags, the GPU suite.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_a
mkdir -p $OUT
cd $ROOT
python - > $OUT/f64_gate.json 2> $OUT/f64_gate.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import fhe_rs_amd as fhe
out = {}
for rep in range(3):
    for k in fhe.UBENCH_KINDS:
        out.setdefault(k, []).append(fhe.ubench_int(k, 0.2))
res = {k: max(v) for k, v in out.items()}
res["_all"] = out
res["_ratio_fwd_f64_over_narrow"] = res["f64_fwd_butterfly"] / res["fwd_butterfly_narrow"]
res["_ratio_inv_f64_over_int"] = res["f64_inv_butterfly"] / res["inv_butterfly"]
res["_ratio_mac_f64_over_int"] = res["f64_mac"] / res["shoup_mac"]
res["_ratio_mulmod_f64_over_shoup"] = res["f64_mulmod"] / res["shoup_lazy"]
print(json.dumps(res, indent=1))
PY
python bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err
cp bench_detail.json $OUT/bench_detail_default.json 2>/dev/null
tail -c 2000 $OUT/bench_default.out > $OUT/bench_default_tail2000.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
cat $OUT/f64_gate.json | head -30
tail -1 $OUT/bench_default.out
