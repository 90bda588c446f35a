#!/usr/bin/env python3
"""The `next_rows` and `C5_chain_15_levels` legs of bench.py on their own (what tools/runs/r04_run3.sh wraps in
rocprofv3 --kernel-trace --stats for profiles/r04_next_rows_*).  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import fhe_rs_amd as fhe
from fhe_rs_amd import _lib

for a in sys.argv[1:]:
    if a.startswith("--lib="):          # A/B: another build of the library (tools/_variants/...)
        _lib._load_for_tests(a[6:])

n = bench.N_DEGREE
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=bench.MODULI_SIZES)
out = bench.next_rows(fhe, torch, par, bench.make_timeit(torch))
fhe.workspace_trim()
torch.cuda.empty_cache()
if "--no-chain" not in sys.argv:
    out["C5_chain_15_levels"] = bench.c5_chain(fhe, torch)
print(json.dumps(out))
