mkdir -p gpurun_out/r02c
python -m pytest tests/test_gpu_parity.py -x -q -k "key_switch or galois or rgsw or expand or relin or c3 or c5 or random" 2>&1 | tail -1
for v in 1 0; do
  if [ $v = 1 ]; then export FHE_NO_KS_XHAT=1; echo "without xhat"; else unset FHE_NO_KS_XHAT; echo "with xhat"; fi
  python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3()
" 2>/dev/null | cut -c1-200
done 2>&1 | tee gpurun_out/r02c/ab_c3_xhat.txt
