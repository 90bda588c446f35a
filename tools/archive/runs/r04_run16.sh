#!/bin/bash
# Round 4, GPU call 17: every key-switch strategy at SMALL batches (launches that do not fill the device with fused
# workgroups): does the unfused form, whose first stage has digits x more tiles, win there?
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04p
mkdir -p $O
timeout 900 python tools/ks_small_launch_ab.py 3 all > $O/ks_small_batches_all_modes.jsonl 2> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04p/ks_small_batches_all_modes.jsonl"):
    d = json.loads(l)
    print(d["n"], d["moduli"], d["batch"], d["fused_workgroups"], {k: min(v) for k, v in d["ms"].items()})
PY
tail -3 $O/err.log
