mkdir -p gpurun_out/r02d
for v in 0 1; do
  echo "FHE_LAB_NTT_SWAP=$v"
  FHE_LAB_NTT_SWAP=$v python tests/swap_ntt_check.py hip 2>&1 | tail -1
  FHE_LAB_NTT_SWAP=$v python -m pytest tests/test_gpu_parity.py -x -q -k "c2_mul or c2_full or ntt_full or test_multiply" 2>&1 | tail -1
  FHE_LAB_NTT_SWAP=$v BK_TAG=swap$v python tools/bench_kernels.py 2>/dev/null | grep -i "ntt_forward\|multiply"
  FHE_LAB_NTT_SWAP=$v python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done 2>&1 | tee gpurun_out/r02d/ab_ntt_swap.txt
