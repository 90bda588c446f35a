#!/bin/bash
# Round 4, GPU call 8: non-temporal last-use loads in the multiply pipeline (FHE_PIPE_NT bits: 1 scalers, 2 forward NTT,
# 4 inverse NTT) -- lab variants against the release build (none), alternating; the variants pass the multiply subset first.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04g
timeout 1500 python tools/ab_mul.py 3 > gpurun_out/r04g/ab_nt.jsonl 2> gpurun_out/r04g/ab_nt.err
python - <<'PY'
import json
for l in open("gpurun_out/r04g/ab_nt.jsonl"):
    d = json.loads(l)
    if "error" in d:
        print(d["build"], d["error"][-200:]); continue
    print(f'{d["build"]:24s} r{d["round"]} b1024 {d["c2_b1024_s1_ms"]}/{d["c2_b1024_s2_ms"]} b64 {d["c2_b64_s1_ms"]} b16 {d["c2_b16_s1_ms"]} c5 {d["c5_b16_ms"]} {d["c2_kernels_ms_per_10"]}')
PY
tail -2 gpurun_out/r04g/ab_nt.err
