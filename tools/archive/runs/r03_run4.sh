#!/bin/bash
# Round 3, GPU call 4: what the driver runs at round end, on the final code -- full GPU suite, smoke(), the bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03d; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json; tail -4 $O/bench.err
