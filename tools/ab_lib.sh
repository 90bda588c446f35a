# A/B of two builds of the library on one box: tools/_variants/$1 against the in-tree build.
# usage: bash tools/ab_lib.sh libfhe_hip_exact.so "<label A>" "<label B>"
ALT=tools/_variants/$1
mkdir -p gpurun_out/r02e
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_main.so
for round in 1 2; do
for v in alt main; do
  if [ $v = alt ]; then cp $ALT fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  if [ $round = 1 ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "c2_mul or scaler or extender or test_multiply or decrypt or custom or c1_ct or random" 2>&1 | tail -1; fi
  BK_TAG=$v python tools/bench_kernels.py 2>/dev/null | grep -i "scale\|multiply"
  python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
  if [ $round = 1 ]; then python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3()
" 2>/dev/null | cut -c1-120; fi
done
done 2>&1 | tee gpurun_out/r02e/ab_lib_$1.txt
cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so
