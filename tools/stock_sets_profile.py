#!/usr/bin/env python3
"""Per-kernel time of Multiplicator::multiply on the reference's stock sets (library profiler, single stream)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import fhe_rs_amd as fhe
from bench import key_for
sets = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
        8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
        16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}
out = {}
for n, q in sets.items():
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, 7))
    batch = 1024 if n <= 8192 else 256
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
    mul = fhe.Multiplicator.default(par, rk, 0).set_streams(1)
    mul.multiply(a, b); torch.cuda.synchronize()
    fhe.prof_reset(); fhe.prof_enable(True)
    for _ in range(5):
        mul.multiply(a, b)
    torch.cuda.synchronize()
    fhe.prof_enable(False)
    rep = fhe.prof_report(); fhe.prof_reset()
    tot = sum(v[1] for v in rep.values())
    out[f"n={n}"] = dict(batch=batch, ms_per_step=round(tot / 5, 3), ops_per_s=round(batch * 5 / tot * 1e3, 1),
                         kernels={k: dict(ms=round(v[1] / 5, 3), share=round(v[1] / tot, 3), launches=v[0] // 5) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])})
    del a, b, mul, rk, par, ctx
    fhe.workspace_trim(); torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
