#!/usr/bin/env python3
"""One JSON object per process: relinearise / rotate / multiply on the reference's stock sets at several batches, FHE_KS_AUTO
(the product's behaviour), with output digests -- the workload of the A/B of the F64 key switch's thread geometry (round 6)."""
import hashlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

SETS = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
        8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
        16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}
timeit = bench.make_timeit(torch, 8)
med = lambda fn: round(statistics.median(timeit(fn) for _ in range(5)), 4)
out = {}
for n, q in SETS.items():
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
    ctx = par.context_at_level(0)
    ksk = bench.key_for(fhe, ctx, 11)
    rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
    mul = fhe.Multiplicator.default(par, rk, 0)
    for batch in ((1, 16, 64, 256, 1024) if n <= 8192 else (1, 16, 64, 256)):
        ct3 = ctx.synth_uniform(11, 0, 0, 3, batch)
        a, b = ctx.synth_uniform(11, 0, 0, 2, batch), ctx.synth_uniform(11, 0, 2, 2, batch)
        out[f"n{n}_relinearize_{batch}_ms"] = med(lambda: rk.relinearizes(ct3))
        out[f"n{n}_rotate_{batch}_ms"] = med(lambda: gk.relinearize(a))
        if batch in (1, 64, 1024, 256):
            out[f"n{n}_mul_and_relin_{batch}_ms"] = med(lambda: mul.multiply(a, b))
        if batch == 16:
            h = hashlib.sha256()
            for t in (rk.relinearizes(ct3), gk.relinearize(a), mul.multiply(a, b)):
                h.update(t.cpu().numpy().tobytes())
            out[f"n{n}_digest"] = h.hexdigest()[:16]
        del ct3, a, b
    del mul, rk, gk, ksk
    fhe.workspace_trim()
    torch.cuda.empty_cache()
print(json.dumps(out))
