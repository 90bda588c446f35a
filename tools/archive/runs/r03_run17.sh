#!/bin/bash
# Round 3, GPU call 17: chunk size x streams sweep of the C2 multiply on the final kernels (the round-1 choice of 512
# pairs per chunk predates the faster scalers; a chunk of 64 pairs keeps a chunk's intermediates inside the 256 MiB
# Infinity Cache).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03q; mkdir -p $O
for rep in 1 2; do
for streams in 1 2; do
for chunk in 32 64 128 256 512 1024; do
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 --chunk $chunk --streams $streams 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chunk': $chunk, 'streams': $streams, 'ops_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernels_ms': {k:v['ms'] for k,v in d['roofline']['kernels'].items()}}))"
done
done
done > $O/chunk_sweep.jsonl 2>&1
cat $O/chunk_sweep.jsonl | cut -c1-260
