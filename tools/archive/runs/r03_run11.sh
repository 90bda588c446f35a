#!/bin/bash
# Round 3, GPU call 11: cross products of the lazy Shoup low word summed by multiply-adds (shoup_lo), accumulators as
# the addend of the key-switch MACs -- full GPU suite on the new build, then same-box A/B against the previous release
# build (C2 per-kernel ms, C3), and the lab build's radix-4 plan at N = 16384 (the mixed plan spills 28 B now).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03k; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
  if [ $round != 3 ]; then timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120; fi
done
done > $O/shoup_lo_ab.txt 2>&1
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
for plan in 0 4 0 4; do
  echo "== lab build, FHE_LAB_KS14_PLAN=$plan"
  FHE_LAB_KS14_PLAN=$plan timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120
done >> $O/shoup_lo_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cat $O/shoup_lo_ab.txt
