#!/usr/bin/env python3
"""mul_and_relin_2 (two different extenders: the operand extensions cannot be merged) at small batches, handle option
streams = 1 vs 2 (2: the rhs extension runs on the internal stream, fork / join through events).  ms per call."""
import json, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import fhe_rs_amd as fhe
from bench import key_for, make_timeit
timeit = make_timeit(torch, 30)
out = {}
for name, n, kw in (("C2", 8192, dict(moduli_sizes=[60] * 4)),
                    ("stock8192", 8192, dict(moduli=[0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]))):
    t = fhe.generate_prime(20, 2 * n, (1 << 20) - 1)
    par = fhe.BfvParameters(n, t, **kw)
    ctx = par.context_at_level(0)
    q = par.moduli
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, 7))
    nm = (sum(int(m).bit_length() for m in q) + 61) // 62
    ext, upper = [], (1 << 64) - 1 >> 2
    while len(ext) < nm:
        upper = fhe.generate_prime(62, 2 * n, upper)
        if upper not in q:
            ext.append(upper)
    Q = P = 1
    for m in q:
        Q *= int(m)
    for m in ext:
        P *= int(m)
    mctx = fhe.Context(list(q) + ext, n)
    mul2 = fhe.Multiplicator(fhe.Scaler(ctx, mctx, 1, 1), fhe.Scaler(ctx, mctx, P, Q), fhe.Scaler(mctx, ctx, t, P), rk)
    d = {}
    for batch in (1, 2, 4, 8, 16, 32, 64):
        a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
        r = {}
        for rep in range(2):
            for streams in (1, 2):
                mul2.set_streams(streams)
                r.setdefault(streams, []).append(round(timeit(lambda: mul2.multiply(a, b)), 4))
        d[batch] = {k: min(v) for k, v in r.items()}
    out[name] = d
print(json.dumps(out))
