#!/usr/bin/env python3
"""One JSON object: relinearize / rotate / mul_and_relin on the stock n = 8192 set with every key FORCED to the fused key switch,
for the A/B of the 512-thread F64 instance (lab builds read FHE_LAB_KS13_F64_T512 once per process: run this twice)."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

n = 8192
q = [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
ctx = par.context_at_level(0)
ksk = bench.key_for(fhe, ctx, 11).set_mode(1)
rk = fhe.RelinearizationKey(ksk)
mul = fhe.Multiplicator.default(par, rk, 0)
timeit = bench.make_timeit(torch, 5)
out = dict(t512=os.environ.get("FHE_LAB_KS13_F64_T512", "0"))
for batch in (64, 256, 1024):
    ct3 = ctx.synth_uniform(11, 0, 0, 3, batch)
    a, b = ctx.synth_uniform(11, 0, 0, 2, batch), ctx.synth_uniform(11, 0, 2, 2, batch)
    out[f"relinearize_{batch}_ms"] = round(statistics.median(timeit(lambda: rk.relinearizes(ct3)) for _ in range(5)), 4)
    out[f"mul_and_relin_{batch}_ms"] = round(statistics.median(timeit(lambda: mul.multiply(a, b)) for _ in range(5)), 4)
# parity of the variant against the default build's own result is checked by the caller (digest of one output)
import hashlib
ct3 = ctx.synth_uniform(11, 0, 0, 3, 8)
out["digest"] = hashlib.sha256(rk.relinearizes(ct3).cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(out))
