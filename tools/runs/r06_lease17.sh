#!/bin/bash
# Round 6, lease 17: chunk x streams sweep of the stock n = 8192 multiply with the 512-thread F64 key switch, new build vs previous release.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_o
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for v in new before new before; do
  if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_t512.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  timeout 600 python tools/chunk_sweep_sets.py stock8192 0,48,64,96,102,128,160,192,256,512 2>/dev/null | sed "s/^{/{\"build\": \"$v\", /"
done > $OUT/chunk_sweep_stock8192.jsonl
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
python - <<'PY'
import json, collections
t = collections.defaultdict(list)
for l in open("gpurun_out/r06_o/chunk_sweep_stock8192.jsonl"):
    d = json.loads(l)
    t[(d["streams"], d["chunk"], d["build"])].append(d["ms"])
for k in sorted(t):
    print(k, t[k])
PY
