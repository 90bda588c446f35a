#!/bin/bash
# Round 3, GPU call 7: final state -- full GPU suite, smoke, determinism soaks (default two-stream mode, the C3 shape,
# one-chunk batches with the split extension + squaring), and the driver-shaped bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/soak.py 1500 2 > $O/soak_streams2.json 2>&1; cat $O/soak_streams2.json | tail -1
timeout 600 python tools/soak.py 1500 1 c3 > $O/soak_c3.json 2>&1; cat $O/soak_c3.json | tail -1
timeout 600 python tools/soak.py 30000 2 small > $O/soak_small.json 2>&1; cat $O/soak_small.json | tail -1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; tail -4 $O/bench.err
