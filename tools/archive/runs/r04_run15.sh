#!/bin/bash
# Round 4, GPU call 16: FHE_KS_AUTO's per-launch choice between the two fused forms at N >= 32768 -- A/B over batch sizes,
# then the GPU suite (incl. the new FUSED_SUB / auto cases), smoke and the default-flag bench of this build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04o
mkdir -p $O
timeout 600 python tools/ks_small_launch_ab.py 3 > $O/ks_small_launch_ab.jsonl 2> $O/ks_small_launch_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r04o/ks_small_launch_ab.jsonl"):
    d = json.loads(l)
    print(d["n"], d["moduli"], d["batch"], d["sub_blocks_8192"], d["auto_takes"], {k: min(v) for k, v in d["ms"].items()})
PY
tail -3 $O/ks_small_launch_ab.err
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04o/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
oc = d["other_configs"]
print({k: (v.get("ops_per_s"), v.get("frac"), v.get("total_ms")) for k, v in oc.items() if k.startswith(("C3", "C5"))})
PY
