#!/bin/bash
# Round 3, GPU call 19: profiler events carried by the launches themselves (hipExtLaunchKernelGGL), new two-stream
# chunk plan -- GPU suite, the driver-shaped bench line, and the default plan at 8,192 pairs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03s; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03s/bench.json'))
print('value',d['value'],'event_free',d['event_free']['value'],'default',d['default_mode']['value'])
print({k:(v['ms'],v['frac']) for k,v in d['roofline']['kernels'].items()})
print('sum kernel ms/step', sum(v['ms'] for v in d['roofline']['kernels'].values())/d['steps'], 'ms_per_step', d['ms_per_step'])
print({k:v.get('ops_per_s') for k,v in d['other_configs'].items()})
PY
tail -3 $O/bench.err
timeout 300 python - <<'PY'
import sys,os,json
sys.path.insert(0,'tools')
sys.argv=['x','8192']
src=open('tools/chunk_sweep.py').read().replace("for rep in range(2):","for rep in range(1):").replace("(0, 32, 64, 96, 128, 192, 256, 384, 512, 1024)","(0, 128)")
exec(compile(src,'chunk_sweep','exec'))
PY
