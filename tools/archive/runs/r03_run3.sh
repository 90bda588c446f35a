#!/bin/bash
# Round 3, GPU call 3: the new C5 bench-batch tests + graph / stream tests on the release build, then the same-box A/B
# of the one-chunk operand-extension split (lab library: FHE_LAB_MUL_SPLIT_EXT=0 turns it off).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c5_bench or test_multiply or graph or two_streams or concurrent or abi_owned" > $O/pytest_subset.log 2>&1
tail -2 $O/pytest_subset.log
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
for v in 1 0 1 0; do
  echo "== FHE_LAB_MUL_SPLIT_EXT=$v"
  FHE_LAB_MUL_SPLIT_EXT=$v timeout 300 python tools/bench_latency.py 2>/dev/null
  FHE_LAB_MUL_SPLIT_EXT=$v timeout 300 python tools/bench_configs.py c5 2>/dev/null | cut -c1-200
done > $O/split_ext_ab.txt 2>&1
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
cat $O/split_ext_ab.txt
