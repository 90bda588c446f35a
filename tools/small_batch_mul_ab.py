#!/usr/bin/env python3
"""C2 and stock n = 8192 / n = 16384 multiply + relinearise at small batches (ms per call, default handle options)."""
import json, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import fhe_rs_amd as fhe
from bench import key_for, make_timeit
timeit = make_timeit(torch, 20)
out = {}
for name, n, kw in (("C2", 8192, dict(moduli_sizes=[60] * 4)),
                    ("stock8192", 8192, dict(moduli=[0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001])),
                    ("stock16384", 16384, dict(moduli=[0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001,
                                                       0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]))):
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), **kw)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0)
    d = {}
    for batch in (1, 2, 4, 6, 8, 12, 16, 18, 24, 32):
        a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
        d[batch] = round(timeit(lambda: mul.multiply(a, b)), 4)
    out[name] = d
    del mul, par, ctx
    fhe.workspace_trim()
print(json.dumps(out))
