#!/bin/bash
# Round 4, GPU call 22: FHE_KS_AUTO with the short-digit-loop refinement at N >= 32768 -- both small-launch tables again,
# GPU suite, smoke, bench (last-commit confirmation).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04u
mkdir -p $O
timeout 300 python tools/ks_small_launch_ab.py 3 > $O/ks_small_launch_ab.jsonl 2> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04u/ks_small_launch_ab.jsonl"):
    d = json.loads(l)
    print(d["n"], d["moduli"], d["batch"], d["sub_blocks_8192"], {k: min(v) for k, v in d["ms"].items()})
PY
timeout 600 python tools/ks_small_launch_ab.py 3 all > $O/ks_small_batches_all_modes.jsonl 2>> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04u/ks_small_batches_all_modes.jsonl"):
    d = json.loads(l)
    m = {k: min(v) for k, v in d["ms"].items()}
    best = min(v for k, v in m.items() if k != "auto")
    print(d["n"], d["moduli"], d["batch"], d["fused_workgroups"], m, "auto/best", round(m["auto"] / best, 3))
PY
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04u/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
PY
