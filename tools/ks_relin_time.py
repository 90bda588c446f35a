#!/usr/bin/env python3
"""Relinearise (= one key switch + the add) timed on one shape with the in-tree library or a lab variant:
usage: python tools/ks_relin_time.py <lib|default> <n> <nmoduli> <batch> [<batch> ...]   -- prints {batch: ms per call}"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe
from fhe_rs_amd import _lib

if sys.argv[1] != "default":
    _lib._load_for_tests(sys.argv[1])
n, L = int(sys.argv[2]), int(sys.argv[3])
ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
out = {}
for batch in map(int, sys.argv[4:]):
    ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
    for _ in range(3):
        rk.relinearizes(ct3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rk.relinearizes(ct3)
    e1.record()
    torch.cuda.synchronize()
    out[batch] = round(e0.elapsed_time(e1) / 5, 4)
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "n": n, "moduli": L, "ms_per_call": out}))
