# Timing sensitivity (NOT a parity A/B: the FHE_SENS builds compute wrong residues on purpose): how much of a kernel's
# time follows the multiplies of its butterflies.  tools/_variants/libfhe_sens{1,3,4,7}.so against the in-tree build.
mkdir -p gpurun_out/r02e
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_main.so
for round in 1 2; do
for v in main ${SENS:-1 3 4 7}; do
  if [ $v = main ]; then cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so; else cp tools/_variants/libfhe_sens$v.so fhe.rs_amd/libfhe_hip.so; fi
  echo "== $v (round $round)"
  python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
done 2>&1 | tee gpurun_out/r02e/ab_sens.txt
cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so
