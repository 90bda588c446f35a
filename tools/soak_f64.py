#!/usr/bin/env python3
"""Determinism + cross-path soak of the F64 kernels (round 6): on the reference's stock n = 8192 and n = 16384 sets, a batch's
multiply + relinearise, relinearise and rotation are computed ONCE on the integer kernels (fhe_engine_set_f64(0)) and then
`reps` times on the F64 kernels, default two-stream mode; every F64 result must be bit-identical to the integer kernels'.
usage: python tools/soak_f64.py [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SETS = {8192: ([0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001], 512),
        16384: ([0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001], 128)}
out = {}
for n, (q, batch) in SETS.items():
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
    ctx = par.context_at_level(0)
    ksk = bench.key_for(fhe, ctx, 21)
    rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
    mul = fhe.Multiplicator.default(par, rk, 0)
    a, b = ctx.synth_uniform(21, 0, 0, 2, batch), ctx.synth_uniform(21, 0, 2, 2, batch)
    ct3 = ctx.synth_uniform(21, 0, 0, 3, batch)
    fhe.set_f64(False)
    ref = (mul.multiply(a, b).clone(), rk.relinearizes(ct3).clone(), gk.relinearize(a).clone())
    torch.cuda.synchronize()
    fhe.set_f64(True)
    bad, t0 = 0, time.time()
    for _ in range(reps):
        got = (mul.multiply(a, b), rk.relinearizes(ct3), gk.relinearize(a))
        if not all(torch.equal(g, r) for g, r in zip(got, ref)):
            bad += 1
    torch.cuda.synchronize()
    out[f"n={n}"] = dict(batch=batch, repetitions=reps, ops=reps * 3 * batch, mismatches_vs_integer_kernels=bad,
                         seconds=round(time.time() - t0, 1))
    del a, b, ct3, ref, mul, rk, gk, ksk
    fhe.workspace_trim()
    torch.cuda.empty_cache()
print(json.dumps(out))
sys.exit(1 if any(v["mismatches_vs_integer_kernels"] for v in out.values()) else 0)
