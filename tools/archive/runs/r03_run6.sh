cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03f
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
for v in 0 1 0 1; do echo "== FHE_LAB_KS_SPLIT14=$v"; FHE_LAB_KS_SPLIT14=$v python tools/bench_configs.py c3 2>/dev/null | cut -c1-330; done > gpurun_out/r03f/c3_split14.txt 2>&1
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
cat gpurun_out/r03f/c3_split14.txt
