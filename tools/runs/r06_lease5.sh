#!/bin/bash
# Round 6, lease 5: the 512-thread F64 key switch (lab switch, one lab build, env var read per process) + F64 random sweep.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_k
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_lab7.so fhe.rs_amd/libfhe_hip.so
for round in 1 2 3; do
  for v in 0 1; do
    FHE_LAB_KS13_F64_T512=$v python tools/f64_t512_ab.py 2>/dev/null
  done
done | tee $OUT/ks13_f64_t512_ab.jsonl | cut -c1-400
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
