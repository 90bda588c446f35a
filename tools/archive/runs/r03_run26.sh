#!/bin/bash
# Round 3, GPU call 26: last commits (pool trim in fhe_workspace_trim, header notes) -- GPU suite, smoke, bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['steps'], d['event_free']['value'], d['default_mode']['value'], d['host_api']['abi_device_buffers']['ops_per_s'])"
tail -3 $O/bench.err
