#!/bin/bash
# Round 3, GPU call 5: same-box A/B of the scaler's one-accumulator w (new in-tree library) against the previous
# release build (tools/_variants/libfhe_hip_prev.so), parity of the scaler / multiply / decrypt cases first.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "scaler or multiply or decrypt or c2_ or c1_ or c5_bench or random or extender" > $O/pytest_subset.log 2>&1
tail -2 $O/pytest_subset.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done > $O/scaler_one_acc_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cat $O/scaler_one_acc_ab.txt
python tools/bench_configs.py c5 2>/dev/null | cut -c1-400 > $O/c5_new.txt; cat $O/c5_new.txt
