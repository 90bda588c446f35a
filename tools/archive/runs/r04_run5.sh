#!/bin/bash
# Round 4, GPU call 6: parity of the merged extension / direction flags (multiply cases incl. C1, C2, C5 batches,
# two streams, graph capture), then the A/B of the builds: default (both on), nodir, nomerge, neither.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04e
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "multiply or config_c1 or config_c2 or c5_bench or random or graph or two_streams or square or custom" > gpurun_out/r04e/pytest_mul.log 2>&1
tail -3 gpurun_out/r04e/pytest_mul.log
timeout 1200 python tools/ab_mul.py 3 > gpurun_out/r04e/ab_mul.jsonl 2> gpurun_out/r04e/ab_mul.err
python - <<'PY'
import json
for l in open("gpurun_out/r04e/ab_mul.jsonl"):
    d = json.loads(l)
    if "error" in d:
        print(d["build"], d["error"][-200:]); continue
    print(f'{d["build"]:28s} r{d["round"]} b1024 s1 {d["c2_b1024_s1_ms"]} s2 {d["c2_b1024_s2_ms"]} | b64 {d["c2_b64_s1_ms"]}/{d["c2_b64_s2_ms"]} b16 {d["c2_b16_s1_ms"]}/{d["c2_b16_s2_ms"]} | c5 {d["c5_b16_ms"]} | {d["c2_kernels_ms_per_10"]}')
PY
tail -2 gpurun_out/r04e/ab_mul.err
