#!/usr/bin/env python3
"""Exploration behind fhe_mul_set_streams(2): the C2 batch of 1024 pairs split into `parts` calls issued round-robin
on `ns` torch streams, against the single call.  (The library-internal version is what ships; this stays as the
measurement recipe.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, fhe_rs_amd as fhe
n = 8192
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
L = 4
kk = ctx.synth_uniform(1, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
mul = fhe.Multiplicator.default(par, rk, 0)
a = ctx.synth_uniform(1, 0, 0, 2, 1024)
b = ctx.synth_uniform(1, 0, 2, 2, 1024)
pool = [torch.cuda.Stream() for _ in range(4)]
def serial():
    return mul.multiply(a, b)
def multi(parts, ns):
    cur = torch.cuda.current_stream()
    step = 1024 // parts
    for st in pool[:ns]:
        st.wait_stream(cur)
    for i in range(parts):
        with torch.cuda.stream(pool[i % ns]):
            mul.multiply(a[i * step:(i + 1) * step], b[i * step:(i + 1) * step])
    for st in pool[:ns]:
        cur.wait_stream(st)
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    res = {"serial_ms": round(timeit(serial), 3)}
    for parts in (8, 16, 32):
        for ns in (2, 3, 4):
            res[f"p{parts}s{ns}"] = round(timeit(lambda: multi(parts, ns)), 3)
    print(json.dumps(res), flush=True)
