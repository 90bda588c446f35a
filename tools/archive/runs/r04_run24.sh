#!/bin/bash
# Round 4, GPU call 27: random shapes with rows larger than LDS (N = 32768 / 65536, mixed modulus widths, 1-5 moduli, 1-3
# ciphertexts): multiply (+relinearise, +modulus switch), relinearise, rotations against the C oracle, with FHE_KS_AUTO
# and with every key forced to each strategy (1 fused on 16384-point parts, 4 fused on 8192-point sub-blocks, 2 unfused).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04y
mkdir -p $O
i=0
for mode in 0 1 4 2; do
  timeout 200 python tests/random_sweep_gpu.py 110 $((300000 + i * 10000)) $((309999 + i * 10000)) $mode big > $O/random_sweep_big_mode$mode.json 2>> $O/err.log
  cut -c1-260 $O/random_sweep_big_mode$mode.json
  i=$((i + 1))
done
tail -3 $O/err.log
