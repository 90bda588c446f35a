#!/bin/bash
# Round 4, GPU call 11: narrow passes in the LDS halves of rows larger than LDS (N >= 32768): parity, then A/B against a
# lab build that reads FHE_LAB_NO_NARROW_SUB=1 (the general passes, as rounds 1-3) on the C5 level-0 step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04j
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or c5 or split or large_rows or n65536 or 65536" > gpurun_out/r04j/pytest_ntt.log 2>&1
tail -3 gpurun_out/r04j/pytest_ntt.log
export FHE_LAB_NO_NARROW_SUB=1
timeout 900 python tools/ab_mul.py 3 > gpurun_out/r04j/ab.jsonl 2> gpurun_out/r04j/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r04j/ab.jsonl"):
    d = json.loads(l)
    if "error" in d:
        print(d["build"], d["error"][-300:]); continue
    print(f'{d["build"]:28s} r{d["round"]} c2 b1024 {d["c2_b1024_s1_ms"]}/{d["c2_b1024_s2_ms"]} c5 b16 {d["c5_b16_ms"]}')
PY
