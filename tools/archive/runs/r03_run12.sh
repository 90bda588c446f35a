#!/bin/bash
# Round 3, GPU call 12: per-instruction issue cost (tools/ubench_issue.cpp) at 1, 2, 4 and 8 waves per SIMD.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03l; mkdir -p $O
for w in 4 1 2 8; do timeout 120 tools/_variants/ubench_issue $w 2380 2>>$O/ubench_issue.err; done > $O/ubench_issue.jsonl
(rocm-smi --showclocks | grep -i sclk) >> $O/ubench_issue.err 2>&1
cat $O/ubench_issue.jsonl | cut -c1-200
