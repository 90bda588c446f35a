#!/bin/bash
# Round 3, GPU call 2: the committed evidence -- bench line + rocprofv3 kernel stats + PMC traffic of the release
# build, SQ counters of the shipped key switch and of the two-workgroups-per-CU lab variant, C3 / C5 stats + PMC,
# and the two-rank launch (C4's per-GPU shard on a shared device).
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh r03_final 5 > gpurun_out/r03_final_collect.log 2>&1
tail -3 gpurun_out/r03_final_collect.log
bash tools/collect_sq.sh r03_sq > gpurun_out/r03_sq_collect.log 2>&1
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
FHE_LAB_KS13_T512=2 bash tools/collect_sq.sh r03_sq_t512 > gpurun_out/r03_sq_t512_collect.log 2>&1
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
bash tools/collect_configs_pmc.sh r03_cfgpmc > gpurun_out/r03_cfgpmc_collect.log 2>&1
( time timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/r03_gpus2.json 2> gpurun_out/r03_gpus2.err
tail -c 1500 gpurun_out/r03_gpus2.json; tail -3 gpurun_out/r03_gpus2.err
ls gpurun_out/r03_final gpurun_out/r03_sq gpurun_out/r03_sq_t512 gpurun_out/r03_cfgpmc | head -60
