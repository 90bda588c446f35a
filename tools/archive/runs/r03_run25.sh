#!/bin/bash
# Round 3, GPU call 25: C3 key switch on two 8192-point sub-blocks (split kernel, RNS instance, hoisted branches) against
# the whole-row kernel -- lab build, FHE_LAB_KS_SPLIT14.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03z; mkdir -p $O
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_ship.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
for round in 1 2 3; do
for v in 0 1; do
  echo "== lab build, FHE_LAB_KS_SPLIT14=$v (round $round)"
  FHE_LAB_KS_SPLIT14=$v timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120
done
done > $O/c3_split14_ab.txt 2>&1
cp /tmp/lib_ship.so fhe.rs_amd/libfhe_hip.so
cat $O/c3_split14_ab.txt
