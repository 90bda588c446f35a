#!/bin/bash
# Round 3, GPU call 24: RNS instance of the N = 16384 key switch (spills 52 B) against the shipped generic one, C3.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03y; mkdir -p $O
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_ship.so
for round in 1 2 3; do
for v in ship ks14rns; do
  if [ $v = ship ]; then cp /tmp/lib_ship.so fhe.rs_amd/libfhe_hip.so; else cp tools/_variants/libfhe_hip_ks14rns.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  timeout 300 python tools/bench_configs.py c3 2>/dev/null | cut -c1-120
done
done > $O/ks14_rns_ab.txt 2>&1
cp /tmp/lib_ship.so fhe.rs_amd/libfhe_hip.so
cat $O/ks14_rns_ab.txt
