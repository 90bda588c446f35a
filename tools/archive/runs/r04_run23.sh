#!/bin/bash
# Round 4, GPU call 24: last-commit confirmation (FHE_KS_AUTO: sub-block tiles for tiny unfused launches at N = 16384) --
# GPU suite, smoke, random sweep of a fresh index range (auto), the all-strategies table, default bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04w
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python tests/random_sweep_gpu.py 240 200000 240000 > $O/random_sweep_auto.json 2> $O/sweep.err; cut -c1-200 $O/random_sweep_auto.json
timeout 600 python tools/ks_small_launch_ab.py 3 all > $O/ks_small_batches_all_modes.jsonl 2>> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04w/ks_small_batches_all_modes.jsonl"):
    d = json.loads(l)
    m = {k: min(v) for k, v in d["ms"].items()}
    best = min(v for k, v in m.items() if k != "auto")
    print(d["n"], d["moduli"], d["batch"], d["fused_workgroups"], m, "auto/best", round(m["auto"] / best, 3))
PY
timeout 200 python tools/soak.py 10000 1 auto_small > $O/soak_auto_small.json 2>> $O/err.log; cat $O/soak_auto_small.json
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04w/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
print(d["other_configs"]["single_ciphertext_latency"])
PY
