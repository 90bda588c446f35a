#!/bin/bash
# Round 3, GPU call 28: N = 8192 key switch, two workgroups per CU (512 x 16, FHE_LAB_KS13_T512 = 1 mixed / 2 radix-4 /
# 3 radix-8) now with the RNS loader, against the shipped 1024 x 8 RNS kernel (=0) -- lab build, C2 per-kernel ms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04f; mkdir -p $O
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_ship.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
for round in 1 2; do
for v in 0 2 1 3; do
  echo "== lab build, FHE_LAB_KS13_T512=$v (round $round)"
  FHE_LAB_KS13_T512=$v timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done > $O/ks13_t512_rns_ab.txt 2>&1
cp /tmp/lib_ship.so fhe.rs_amd/libfhe_hip.so
cat $O/ks13_t512_rns_ab.txt
