#!/bin/bash
# Round 3, GPU call 13 / 20: the final build -- full GPU suite, smoke, random-shape sweep, determinism soaks, the driver-shaped
# bench line, and rocprofv3 kernel stats + the two PMC passes of the same command.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03w; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 500 python tests/random_sweep_gpu.py 400 > $O/random_sweep.json 2>$O/random_sweep.err; tail -c 200 $O/random_sweep.json
timeout 600 python tools/soak.py 1500 2 > $O/soak_streams2.json 2>&1; tail -1 $O/soak_streams2.json
timeout 600 python tools/soak.py 1500 1 c3 > $O/soak_c3.json 2>&1; tail -1 $O/soak_c3.json
timeout 600 python tools/soak.py 30000 2 small > $O/soak_small.json 2>&1; tail -1 $O/soak_small.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; tail -4 $O/bench.err
bash tools/collect_profiles.sh r03w_prof 5 > $O/collect.log 2>&1; tail -5 $O/collect.log
