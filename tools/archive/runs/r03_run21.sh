#!/bin/bash
# Round 3, GPU call 21: RNS instances of the key switch (one conditional subtraction per lifted coefficient, no per-
# element branches), stream-ordered allocation in the ABI -- GPU suite, then same-box A/B (C2 kernels, C5, C3).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03u; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then cp tools/_variants/libfhe_hip_prev.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
  if [ $round != 3 ]; then timeout 300 python tools/bench_configs.py c5 2>/dev/null | cut -c1-150; timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120 | head -1; fi
done
done > $O/ks_rns_ab.txt 2>&1
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
cat $O/ks_rns_ab.txt
timeout 300 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['host_api'])"
