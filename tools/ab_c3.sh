mkdir -p gpurun_out/r02c
for v in 0 1; do
  echo "FHE_KS_SPLIT14=$v"
  FHE_KS_SPLIT14=$v python -m pytest tests/test_gpu_parity.py -x -q -k "c3_relin or c3_bench" 2>&1 | tail -1
  FHE_KS_SPLIT14=$v python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3()
" 2>/dev/null
done 2>&1 | tee gpurun_out/r02c/ab_c3.txt
echo "C2 after csub removal"; python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})" | tee -a gpurun_out/r02c/ab_c3.txt
