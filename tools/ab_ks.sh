mkdir -p gpurun_out/r02c
for v in ${VARIANTS:-0 1 2}; do
  FHE_LAB_KS_VARIANT=$v python -m pytest tests/test_gpu_parity.py -x -q -k "many_digits or key_switch or galois or test_multiply or c2_mul or c3_relin" 2>&1 | tail -1
  FHE_LAB_KS_VARIANT=$v BK_TAG=ks$v python tools/bench_kernels.py 2>/dev/null | grep -i "key_switch\|multiply"
  echo "variant $v"; FHE_LAB_KS_VARIANT=$v python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done 2>&1 | tee gpurun_out/r02c/ab_ks.txt
