#!/bin/bash
# Round 6, lease 4: A/B of the forward transform's direct last-pass store (lab variant) against the release build.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_d
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
for round in 1 2 3; do
  for v in release direct; do
    if [ $v = direct ]; then cp tools/_variants/libfhe_hip_fwd_direct.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/fwd_direct_ab.py 2>/dev/null)}"
  done
done | tee $OUT/fwd_direct_store_ab.jsonl | cut -c1-420
cp tools/_variants/libfhe_hip_fwd_direct.so fhe.rs_amd/libfhe_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or c2 or f64" > $OUT/pytest_direct.log 2>&1; tail -2 $OUT/pytest_direct.log
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
