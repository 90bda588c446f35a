#!/bin/bash
# Round 6, lease 24: does the multiply's chunk want to be round-aware at N = 16384 (one workgroup per CU in every kernel)?  stock n = 16384,
# batch 256: chunks around the multiples of 256 / 9 workgroups; C3-shaped (8 x 60-bit) batch 256: chunks around multiples of 256 / 8.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_t
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
timeout 600 python tools/chunk_sweep_sets.py stock16384 0,28,56,57,84,85,86,113,114,128,256 2>/dev/null
done > $OUT/chunk_sweep_stock16384_rounds.jsonl
python - <<'PY'
import json, collections
t = collections.defaultdict(list)
for l in open("gpurun_out/r06_t/chunk_sweep_stock16384_rounds.jsonl"):
    d = json.loads(l)
    t[(d["set"], d["streams"], d["chunk"])].append(d["ms"])
for k in sorted(t):
    print(k, t[k])
PY
