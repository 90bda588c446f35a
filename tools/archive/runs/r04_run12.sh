#!/bin/bash
# Round 4, GPU call 13: last-commit confirmation -- GPU suite, smoke, random sweeps (auto: fresh index range; every key in
# unfused mode), long determinism soaks, the default-flag bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04l
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04l/pytest_gpu.log 2>&1
tail -5 gpurun_out/r04l/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04l/smoke.log 2>&1; tail -1 gpurun_out/r04l/smoke.log
timeout 500 python tests/random_sweep_gpu.py 300 40000 80000 > gpurun_out/r04l/random_sweep_auto.json 2> gpurun_out/r04l/sweep.err
timeout 400 python tests/random_sweep_gpu.py 200 80000 99000 2 > gpurun_out/r04l/random_sweep_unfused.json 2>> gpurun_out/r04l/sweep.err
cat gpurun_out/r04l/random_sweep_auto.json gpurun_out/r04l/random_sweep_unfused.json | cut -c1-200
timeout 400 python tools/soak.py 4000 1 > gpurun_out/r04l/soak_streams1.json 2>> gpurun_out/r04l/soak.err
timeout 400 python tools/soak.py 4000 2 > gpurun_out/r04l/soak_streams2.json 2>> gpurun_out/r04l/soak.err
cat gpurun_out/r04l/soak_streams1.json gpurun_out/r04l/soak_streams2.json
( time timeout 900 python bench.py ) > gpurun_out/r04l/bench_default_flags.json 2> gpurun_out/r04l/bench.err
tail -4 gpurun_out/r04l/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04l/bench_default_flags.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
PY
