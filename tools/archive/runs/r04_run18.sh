#!/bin/bash
# Round 4, GPU call 19: last-commit confirmation of the build with the per-launch key-switch choice -- GPU suite, smoke,
# random-shape sweeps of fresh index ranges (auto: small batches reach the unfused kernels; every key forced fused; every
# key forced unfused), determinism soaks, the default-flag bench line, rocprofv3 of the same.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04r
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python tests/random_sweep_gpu.py 240 100000 140000 > $O/random_sweep_auto.json 2> $O/sweep.err
timeout 300 python tests/random_sweep_gpu.py 150 140000 160000 1 > $O/random_sweep_fused.json 2>> $O/sweep.err
timeout 300 python tests/random_sweep_gpu.py 150 160000 180000 2 > $O/random_sweep_unfused.json 2>> $O/sweep.err
cat $O/random_sweep_auto.json $O/random_sweep_fused.json $O/random_sweep_unfused.json | cut -c1-200
timeout 300 python tools/soak.py 3000 1 > $O/soak_streams1.json 2>> $O/soak.err
timeout 300 python tools/soak.py 3000 2 > $O/soak_streams2.json 2>> $O/soak.err
cat $O/soak_streams1.json $O/soak_streams2.json | cut -c1-300
( time timeout 900 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04r/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
PY
