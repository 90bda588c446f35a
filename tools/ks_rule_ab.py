import json, os, sys
sys.path[:0] = [os.getcwd()]
import torch
import fhe_rs_amd as fhe
from bench import key_for, make_timeit, c5_chain
timeit = make_timeit(torch, 5)
out = {}
n = 16384
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=[60] * 8)
ctx = par.context_at_level(0)
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0)
for batch in (256, 36, 72):
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
    out[f"c3shape_mul_batch{batch}_default"] = round(batch / timeit(lambda: mul.multiply(a, b)) * 1e3, 1)
del mul, par, ctx
fhe.workspace_trim()
n = 32768
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=[60] * 16)
ctx = par.context_at_level(0)
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 5)), 0, True)
for batch in (10, 12, 16, 18):
    a, b = ctx.synth_uniform(5, 0, 0, 2, batch), ctx.synth_uniform(5, 0, 2, 2, batch)
    out[f"c5_mul_batch{batch}_default"] = round(batch / timeit(lambda: mul.multiply(a, b)) * 1e3, 1)
del mul, par, ctx
fhe.workspace_trim(); torch.cuda.empty_cache()
ch = c5_chain(fhe, torch)
out["c5_chain_total_ms"] = ch["total_ms"]
out["c5_chain_per_level_ms"] = ch["per_level_ms"]
print(json.dumps(out))
