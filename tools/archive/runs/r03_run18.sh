#!/bin/bash
# Round 3, GPU call 18: event-free chunk x streams sweep (tools/chunk_sweep.py), batch 1024 and 8192.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python tools/chunk_sweep.py 1024 > $O/chunk_sweep_1024.jsonl 2>$O/err.log
timeout 600 python tools/chunk_sweep.py 8192 > $O/chunk_sweep_8192.jsonl 2>>$O/err.log
cat $O/chunk_sweep_1024.jsonl $O/chunk_sweep_8192.jsonl; tail -3 $O/err.log
