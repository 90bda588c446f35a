#!/bin/bash
# Round 3, GPU call 27: N = 16384 key switch as 512 threads x 32 coefficients with a 256-VGPR budget (two waves per SIMD),
# radix-16 (=1) / radix-8 (=2) passes, RNS loader -- lab build, C3, against the shipped kernel (=0).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04c; mkdir -p $O
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_ship.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
timeout 600 env FHE_LAB_KS14_T512=2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3" > $O/pytest_t512.log 2>&1; tail -1 $O/pytest_t512.log
for round in 1 2; do
for v in 0 1 2; do
  echo "== lab build, FHE_LAB_KS14_T512=$v (round $round)"
  FHE_LAB_KS14_T512=$v timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120
done
done > $O/ks14_t512_ab.txt 2>&1
cp /tmp/lib_ship.so fhe.rs_amd/libfhe_hip.so
cat $O/ks14_t512_ab.txt
