# forward NTT at N = 8192: 512 threads x 16 coefficients (4 waves per SIMD) against 1024 x 8 (8 waves per SIMD),
# radix-8 passes (FHE_LAB_NTT_CPT8=3) or radix 8 then 4 (=32).  Same box, alternating.
mkdir -p gpurun_out/r02e
for round in 1 2; do
for v in 0 3 32; do
  echo "== FHE_LAB_NTT_CPT8=$v (round $round)"
  if [ $round = 1 ]; then FHE_LAB_NTT_CPT8=$v python -m pytest tests/test_gpu_parity.py -x -q -k "ntt or c2_mul or test_multiply" 2>&1 | tail -1; fi
  FHE_LAB_NTT_CPT8=$v BK_TAG=cpt$v python tools/bench_kernels.py 2>/dev/null | grep -i "ntt"
  FHE_LAB_NTT_CPT8=$v python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done 2>&1 | tee gpurun_out/r02e/ab_ntt_cpt8.txt
