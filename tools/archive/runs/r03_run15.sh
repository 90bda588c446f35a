#!/bin/bash
# Round 3, GPU call 15: scaler glue on 32-bit limb chains, PLAIN instances, on top of call 14 (carry-free third column,
# first term -- scaler parity tests, then same-box A/B against the previous release build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scaler or extender or multiply or c2_ or decrypt or switcher or random" > $O/pytest_subset.log 2>&1
tail -2 $O/pytest_subset.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3 4; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done > $O/scaler_limbs_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cat $O/scaler_limbs_ab.txt
