mkdir -p gpurun_out/r02c
for v in 4 0 4 0; do
  echo "FHE_LAB_KS14_PLAN=$v (0: mixed)"
  FHE_LAB_KS14_PLAN=$v python -m pytest tests/test_gpu_parity.py -x -q -k "c3_relin or c3_bench" 2>&1 | tail -1
  FHE_LAB_KS14_PLAN=$v python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3()
" 2>/dev/null | cut -c1-230
done 2>&1 | tee gpurun_out/r02c/ab_c3_mixed.txt
