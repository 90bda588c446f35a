#!/usr/bin/env python3
"""Key-switch-bound calls for the same-box A/B of the batched epilogue loads: C2 multiply (batch 1024, one stream and default),
C2 relinearise / rotation (1024), C3 relinearise / rotation (512), C5 relinearise (16) and level-0 multiply (16), stock
n = 8192 / 16384 multiply.  ms per call (lower is better)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_rs_amd as fhe
from bench import key_for, make_timeit
timeit = make_timeit(torch, 6)
out = {}
n = 8192
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
ksk = key_for(fhe, ctx, 2)
rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
mul = fhe.Multiplicator.default(par, rk, 0)
a, b = ctx.synth_uniform(2, 0, 0, 2, 1024), ctx.synth_uniform(2, 0, 2, 2, 1024)
out["C2_mul_default_ms"] = round(timeit(lambda: mul.multiply(a, b)), 4)
mul.set_streams(1)
out["C2_mul_one_stream_ms"] = round(timeit(lambda: mul.multiply(a, b)), 4)
c3 = fhe.Multiplicator.default(par, None, 0).multiply(a, b)
out["C2_relinearize_ms"] = round(timeit(lambda: rk.relinearizes(c3)), 4)
out["C2_rotate_ms"] = round(timeit(lambda: gk.relinearize(a)), 4)
out["C2_key_switch_ms"] = round(timeit(lambda: ksk.key_switch(a[:, 0].contiguous())), 4)
del a, b, c3, mul, rk, gk, ksk, par, ctx
fhe.workspace_trim(); torch.cuda.empty_cache()
n = 16384
ctx = fhe.Context(fhe.generate_moduli([60] * 8, n), n)
ksk = key_for(fhe, ctx, 3)
rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
ct3 = ctx.synth_uniform(3, 0, 0, 3, 512)
ct2 = ct3[:, :2].contiguous()
out["C3_relinearize_ms"] = round(timeit(lambda: rk.relinearizes(ct3)), 4)
out["C3_rotate_ms"] = round(timeit(lambda: gk.relinearize(ct2)), 4)
del ct3, ct2, rk, gk, ksk, ctx
fhe.workspace_trim(); torch.cuda.empty_cache()
n = 32768
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=[60] * 16)
ctx = par.context_at_level(0)
ksk = key_for(fhe, ctx, 5)
rk = fhe.RelinearizationKey(ksk)
ct3 = ctx.synth_uniform(5, 0, 0, 3, 16)
out["C5_relinearize_ms"] = round(timeit(lambda: rk.relinearizes(ct3)), 4)
mul = fhe.Multiplicator.default(par, rk, 0, True)
a, b = ctx.synth_uniform(5, 0, 0, 2, 16), ctx.synth_uniform(5, 0, 2, 2, 16)
out["C5_mul_ms"] = round(timeit(lambda: mul.multiply(a, b)), 4)
del a, b, ct3, mul, rk, ksk, par, ctx
fhe.workspace_trim(); torch.cuda.empty_cache()
for n, q, batch in ((8192, [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001], 1024),
                    (16384, [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001], 256)):
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0)
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
    out[f"stock{n}_mul_ms"] = round(timeit(lambda: mul.multiply(a, b)), 4)
    del a, b, mul, par, ctx
    fhe.workspace_trim(); torch.cuda.empty_cache()
print(json.dumps(out))
