#!/usr/bin/env python3
"""One JSON object per process: the workloads of the LDS-padding A/B (round 6: FHE_LDS_PAD, two lab builds) -- batch transforms
on 60-bit / 62-bit / F64 rows, the stock n = 8192 set's relinearise and multiply, the C2 multiply, C3's relinearise."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

timeit = bench.make_timeit(torch, 8)
med = lambda fn: round(statistics.median(timeit(fn) for _ in range(5)), 4)
out = {}
n = 8192
stock = [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]
for name, q, nn in (("narrow60", fhe.generate_moduli([60] * 4, n), n),
                    ("wide62", [fhe.generate_prime(62, 2 * n, (1 << 62) - k * (1 << 40)) for k in range(1, 5)], n),
                    ("stock_f64", stock, n),
                    ("f64_n16384", [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001], 16384)):
    ctx = fhe.Context(q, nn)
    x = ctx.synth_uniform(3, 0, 0, 1, 2048 if nn == n else 1024).view(-1, len(q), nn)
    out[name + "_fwd_ms"] = med(lambda: ctx.ntt_forward(x))
    out[name + "_inv_ms"] = med(lambda: ctx.ntt_backward(x))
    del x
t = fhe.generate_prime(20, 2 * n, (1 << 20) - 1)
par = fhe.BfvParameters(n, t, moduli=stock)
ctx = par.context_at_level(0)
rk = fhe.RelinearizationKey(bench.key_for(fhe, ctx, 11))
mul = fhe.Multiplicator.default(par, rk, 0)
ct3 = ctx.synth_uniform(11, 0, 0, 3, 1024)
a, b = ctx.synth_uniform(11, 0, 0, 2, 1024), ctx.synth_uniform(11, 0, 2, 2, 1024)
out["stock8192_relinearize_1024_ms"] = med(lambda: rk.relinearizes(ct3))
out["stock8192_mul_and_relin_1024_ms"] = med(lambda: mul.multiply(a, b))
del ct3, a, b, mul, rk
n16 = 16384
stock16 = [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001,
           0x1ffffffe88001, 0x1ffffffe48001]
par = fhe.BfvParameters(n16, fhe.generate_prime(20, 2 * n16, (1 << 20) - 1), moduli=stock16)
ctx = par.context_at_level(0)
rk = fhe.RelinearizationKey(bench.key_for(fhe, ctx, 12))
mul = fhe.Multiplicator.default(par, rk, 0)
ct3 = ctx.synth_uniform(12, 0, 0, 3, 256)
a, b = ctx.synth_uniform(12, 0, 0, 2, 256), ctx.synth_uniform(12, 0, 2, 2, 256)
out["stock16384_relinearize_256_ms"] = med(lambda: rk.relinearizes(ct3))
out["stock16384_mul_and_relin_256_ms"] = med(lambda: mul.multiply(a, b))
one = ct3[:1].contiguous()
out["stock16384_relinearize_single_ms"] = med(lambda: rk.relinearizes(one))
import hashlib
out["digest_stock16384"] = hashlib.sha256(rk.relinearizes(ct3)[:4].cpu().numpy().tobytes()).hexdigest()[:16]
del ct3, a, b, mul, rk, one
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(bench.key_for(fhe, ctx, bench.SEED)), 0).set_streams(1)
a, b = ctx.synth_uniform(bench.SEED, 0, 0, 2, 1024), ctx.synth_uniform(bench.SEED, 0, 2, 2, 1024)
out["c2_mul_and_relin_1024_ms"] = med(lambda: mul.multiply(a, b))
del a, b, mul
n3 = 16384
ctx = fhe.Context(fhe.generate_moduli([60] * 8, n3), n3)
rk = fhe.RelinearizationKey(bench.key_for(fhe, ctx, 0xF4E50003))
ct3 = ctx.synth_uniform(0xF4E50003, 0, 0, 3, 512)
out["c3_relinearize_512_ms"] = med(lambda: rk.relinearizes(ct3))
out["digest"] = hashlib.sha256(rk.relinearizes(ct3)[:4].cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(out))
