# How much of an NTT launch is memory phase?  FHE_DEBUG_NTT_NOMEM=1 points every polynomial of a launch at the first
# one's rows (wrong results, timing only): the kernel then runs out of L2 and what is left is arithmetic + LDS time.
mkdir -p gpurun_out/r02e
for round in 1 2; do
for v in 0 1; do
  echo "== NOMEM=$v (round $round)"
  if [ $v = 1 ]; then export FHE_DEBUG_NTT_NOMEM=1; else unset FHE_DEBUG_NTT_NOMEM; fi
  BK_TAG=nomem$v python tools/bench_kernels.py 2>/dev/null | grep -i "ntt\|ceiling"
done
done 2>&1 | tee gpurun_out/r02e/ab_nomem.txt
