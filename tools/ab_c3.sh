mkdir -p gpurun_out/r02c
for v in 1 0; do
  echo "FHE_KS14_RADIX8=$v"
  FHE_KS14_RADIX8=$v python -m pytest tests/test_gpu_parity.py -x -q -k "c3_relin or c3_bench or random" 2>&1 | tail -1
  FHE_KS14_RADIX8=$v python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3()
" 2>/dev/null | cut -c1-220
done 2>&1 | tee gpurun_out/r02c/ab_c3_radix4.txt
