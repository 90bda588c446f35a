#!/bin/bash
# Round 4, GPU call 20: residue-row (RNS) loader for the whole-row key switch at N = 16384 (the instance no longer spills):
# parity of the variant on the C3 cases, then relinearise at C3 (batch 512) against the same tree without it, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04s
mkdir -p $O
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_rns14.so fhe.rs_amd/libfhe_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c3 or 16384" > $O/pytest_rns14.log 2>&1; tail -2 $O/pytest_rns14.log
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
for rnd in 0 1 2; do
  for lib in base rns14; do
    timeout 200 python tools/ks_relin_time.py tools/_variants/libfhe_hip_$lib.so 16384 8 512 64 >> $O/c3_rns14_ab.jsonl 2>> $O/err.log
  done
done
cat $O/c3_rns14_ab.jsonl
( time timeout 600 python bench.py ) > $O/bench_default_flags.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04s/bench_default_flags.json"))
print(d["value"], d["value_all"], d["other_configs"]["single_ciphertext_latency"], d["other_configs"]["single_ciphertext_latency_c2"])
PY
