#!/usr/bin/env python3
"""Default-plan multiply + relinearise throughput at N = 16384 (five bases) and, as controls, C2 / stock n = 8192 / C5."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_rs_amd as fhe
from bench import key_for, make_timeit
timeit = make_timeit(torch, 4)
STOCK16 = [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]
STOCK8 = [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]
out = {}
for name, n, kw, batch, ms_ in (("n16384_L4", 16384, dict(moduli_sizes=[60] * 4), 512, False), ("n16384_L8", 16384, dict(moduli_sizes=[60] * 8), 256, False),
                                ("stock16384", 16384, dict(moduli=STOCK16), 256, False), ("stock16384_b1024", 16384, dict(moduli=STOCK16), 1024, False),
                                ("n16384_L12", 16384, dict(moduli_sizes=[60] * 12), 192, False),
                                ("C2", 8192, dict(moduli_sizes=[60] * 4), 1024, False), ("stock8192", 8192, dict(moduli=STOCK8), 1024, False),
                                ("C5_b64", 32768, dict(moduli_sizes=[60] * 16), 64, True)):
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), **kw)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0, ms_)
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
    out[name] = round(batch / timeit(lambda: mul.multiply(a, b)) * 1e3, 1)
    del a, b, mul, par, ctx
    fhe.workspace_trim(); torch.cuda.empty_cache()
print(json.dumps(out))
