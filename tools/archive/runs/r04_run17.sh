#!/bin/bash
# Round 4, GPU call 18: FHE_KS_AUTO with the small-launch rule (unfused while 2 x fused workgroups <= compute units):
# GPU suite, smoke, the all-strategies small-batch table again (auto should track the best column), the bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04q
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python tools/ks_small_launch_ab.py 3 all > $O/ks_small_batches_all_modes.jsonl 2> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04q/ks_small_batches_all_modes.jsonl"):
    d = json.loads(l)
    print(d["n"], d["moduli"], d["batch"], d["fused_workgroups"], {k: min(v) for k, v in d["ms"].items()})
PY
timeout 300 python tools/ks_small_launch_ab.py 3 > $O/ks_small_launch_ab.jsonl 2>> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r04q/ks_small_launch_ab.jsonl"):
    d = json.loads(l)
    print(d["n"], d["moduli"], d["batch"], d["sub_blocks_8192"], {k: min(v) for k, v in d["ms"].items()})
PY
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04q/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
oc = d["other_configs"]
for k, v in oc.items():
    print(k, {a: b for a, b in v.items() if a in ("ops_per_s", "frac", "total_ms", "ms", "polys_per_s", "mac_per_s")} if isinstance(v, dict) else v)
PY
timeout 300 python tools/bench_latency.py > $O/latency.jsonl 2>> $O/err.log; cat $O/latency.jsonl | cut -c1-400
