#!/bin/bash
# Round 4, GPU call 9: the build as committed -- whole GPU suite, random-shape sweep (fresh index range), determinism soaks
# (one and two streams), the driver-shaped bench line, rocprofv3 kernel stats + the two PMC passes of the same command.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out/r04h
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04h/pytest_gpu.log 2>&1
tail -5 gpurun_out/r04h/pytest_gpu.log
timeout 400 python tests/random_sweep_gpu.py 200 20000 40000 > gpurun_out/r04h/random_sweep.json 2> gpurun_out/r04h/random_sweep.err
cat gpurun_out/r04h/random_sweep.json | cut -c1-300
timeout 300 python tools/soak.py 1000 1 > gpurun_out/r04h/soak_streams1.json 2>> gpurun_out/r04h/soak.err
timeout 300 python tools/soak.py 1000 2 > gpurun_out/r04h/soak_streams2.json 2>> gpurun_out/r04h/soak.err
cat gpurun_out/r04h/soak_streams1.json gpurun_out/r04h/soak_streams2.json
bash tools/collect_profiles.sh r04h/prof 20 > gpurun_out/r04h/collect.log 2>&1
tail -3 gpurun_out/r04h/collect.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04h/prof/bench.json"))
print(d["value"], d["value_all"], d["ms_per_step"], d.get("default_mode", {}).get("value"), d["roofline"]["frac"], d["roofline"]["kernel_sum_ms_per_step"])
print({k: v["ms"] for k, v in d["roofline"]["kernels"].items()})
PY
