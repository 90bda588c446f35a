#!/bin/bash
# Round 6, lease 6: LDS padding A/B (two lab builds, alternating), then parity of the variant.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_f
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
for round in 1 2 3; do
  for v in lab6 ldspad; do
    cp tools/_variants/libfhe_hip_$v.so fhe.rs_amd/libfhe_hip.so
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/lds_pad_ab.py 2>/dev/null)}"
  done
done | tee $OUT/lds_pad_ab.jsonl | cut -c1-700
cp tools/_variants/libfhe_hip_ldspad.so fhe.rs_amd/libfhe_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or c2 or f64 or c3 or key_switch" > $OUT/pytest_ldspad.log 2>&1; tail -2 $OUT/pytest_ldspad.log
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
