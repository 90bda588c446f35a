#!/bin/bash
# Round 4, GPU call 21: last-commit confirmation (GPU suite incl. the test that pins FHE_KS_AUTO's choices, smoke, bench).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04t
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04t/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
oc = d["other_configs"]
print({k: (v.get("ops_per_s"), v.get("frac"), v.get("total_ms")) for k, v in oc.items() if k.startswith(("C3", "C5"))})
print(oc["single_ciphertext_latency"], oc["single_ciphertext_latency_c2"])
PY
