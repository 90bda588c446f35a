# A/B of two library builds on the C3 / C5 shapes: tools/_variants/$1 against the in-tree build, alternating
ALT=tools/_variants/$1
mkdir -p gpurun_out/r02e
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_main.so
for round in 1 2; do
for v in alt main; do
  if [ $v = alt ]; then cp $ALT fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  if [ $round = 1 ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "c3 or c5 or galois or key_switch" 2>&1 | tail -1; fi
  python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_configs as b
b.c3(); b.c5()
" 2>/dev/null | cut -c1-150
done
done 2>&1 | tee gpurun_out/r02e/ab_lib_c3_$1.txt
cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so
