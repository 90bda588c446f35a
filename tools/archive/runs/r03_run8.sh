#!/bin/bash
# Round 3, GPU call 8: the split-by-family build -- full GPU suite, two-rank launch, bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03h; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
( time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 ) > $O/gpus2.json 2> $O/gpus2.err; tail -c 700 $O/gpus2.json; tail -3 $O/gpus2.err
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; tail -4 $O/bench.err
