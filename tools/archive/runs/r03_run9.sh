#!/bin/bash
# Round 3, GPU call 9: random-shape parity sweep on the final build (shapes 48...; default two-stream handle options, so
# small batches take the split-extension path), 240 s.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03i
timeout 600 python tests/random_sweep_gpu.py 240 > gpurun_out/r03i/random_sweep.json 2> gpurun_out/r03i/random_sweep.err
tail -1 gpurun_out/r03i/random_sweep.json | cut -c1-300; tail -2 gpurun_out/r03i/random_sweep.err
