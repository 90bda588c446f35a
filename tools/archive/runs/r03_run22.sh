#!/bin/bash
# Round 3, GPU call 22: split key switch with the uniform +/- branch of the folded stages hoisted out of the element
# loops -- C5 / N = 65536 parity tests, then C5 A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c5 or 65536 or 32768 or split" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python tools/bench_configs.py c5 2>/dev/null | cut -c1-400
done
done > $O/ks_split_hoist_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cut -c1-150 $O/ks_split_hoist_ab.txt
