#!/bin/bash
# Round 4, GPU call 2: the whole GPU suite on the build with the private memory pool / bounded workspace / unfused
# key-switch option, then the driver-shaped bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04b
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r04b/pytest_gpu.log 2>&1
tail -6 gpurun_out/r04b/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04b/bench.json"))
print(d["value"], d["ms_per_step"], d.get("default_mode", {}).get("value"), d["roofline"]["frac"], {k: v["ms"] for k, v in d["roofline"]["kernels"].items()})
print({k: v.get("ops_per_s") for k, v in d.get("other_configs", {}).items()})
print(d.get("host_api"))
PY
tail -3 gpurun_out/r04b/bench.err
