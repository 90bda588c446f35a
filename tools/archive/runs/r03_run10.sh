#!/bin/bash
# Round 3, GPU call 10: forward butterflies with x folded into the product chain -- parity subset, then same-box A/B
# against the previous release build (C2 per-kernel ms, C3 relinearise / rotations).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "ntt or key_switch or multiply or galois or c2_ or c3_ or c5_bench or random or product_extremes or many_digits or digest or small_traces" > $O/pytest_subset.log 2>&1
tail -2 $O/pytest_subset.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
  if [ $round != 3 ]; then timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120; fi
done
done > $O/fwd_fold_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cat $O/fwd_fold_ab.txt
