#!/bin/bash
# Round 3, GPU call 1: full GPU suite on the release build, the driver-shaped bench line, then same-box A/B of the
# key-switch lab variants (lab library swapped in for the duration, release library restored afterwards).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; tail -5 $O/bench.err
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
cp tools/_variants/libfhe_hip_lab.so fhe.rs_amd/libfhe_hip.so
run_variant() {  # name, env assignments...
  local name=$1; shift
  echo "== $name"
  env "$@" timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "key_switch or galois or test_multiply or c2_mul or many_digits" 2>&1 | tail -1
  env "$@" timeout 300 python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
}
{
run_variant base X=1
run_variant t512_mixed FHE_LAB_KS13_T512=1
run_variant t512_radix4 FHE_LAB_KS13_T512=2
run_variant t512_radix8 FHE_LAB_KS13_T512=3
run_variant half13 FHE_LAB_KS_HALF13=1
run_variant base2 X=1
run_variant t512_mixed2 FHE_LAB_KS13_T512=1
run_variant t512_radix4_2 FHE_LAB_KS13_T512=2
} > $O/ks_variants.txt 2>&1
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
cat $O/ks_variants.txt
timeout 600 python tools/c5_graph_ab.py > $O/c5_graph_ab.txt 2>&1
cat $O/c5_graph_ab.txt | tail -8
