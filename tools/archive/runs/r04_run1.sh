#!/bin/bash
# Round 4, GPU call 1: parity of the unfused key switch (forced through fhe_ksk_set_mode) + same-process A/B of the
# strategies on C2 / C3 / C5.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "unfused or key_switch_modes" > gpurun_out/r04a/pytest_unfused.log 2>&1
tail -3 gpurun_out/r04a/pytest_unfused.log
timeout 600 python tools/ks_modes_ab.py c5 c3 c2 --rounds 3 > gpurun_out/r04a/ks_modes_ab.jsonl 2> gpurun_out/r04a/ks_modes_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r04a/ks_modes_ab.jsonl"):
    d = json.loads(l)
    ks = {k: v for k, v in d["kernels"].items() if k.startswith("k")}
    print(d["shape"], d["op"], d["batch"], d["variant"], d["ms"], ks)
PY
tail -3 gpurun_out/r04a/ks_modes_ab.err
