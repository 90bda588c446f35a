#!/usr/bin/env python3
"""Why does bench.py's reference_default_128 leg report 164 k ops/s for the stock n = 8192 mul_and_relin when a fresh process reports
182 k?  The same leg after each piece of bench.main's set-up, in main's order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402
from fhe_rs_amd import _lib  # noqa: E402


def leg(tag):
    fhe.workspace_trim()
    torch.cuda.empty_cache()
    r = bench.reference_default_128(fhe, torch, None, (8192,))
    ids = next(v for v in r.values() if isinstance(v, dict) and "ids" in v)["ids"]
    print(json.dumps({"when": tag, "mul_and_relin": ids["mul_and_relin"]["batch_ops_per_s"], "relinearize": ids["relinearize"]["batch_ops_per_s"],
                      "mul": ids["mul"]["batch_ops_per_s"], "affinity": len(os.sched_getaffinity(0))}), flush=True)


leg("fresh")
n, batch = bench.N_DEGREE, 1024
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=bench.MODULI_SIZES, device=0)
ctx = par.context_at_level(0)
rk = fhe.RelinearizationKey(bench.key_for(fhe, ctx, bench.SEED))
mul = fhe.Multiplicator.default(par, rk, 0)
a, b = ctx.synth_uniform(1, 0, 0, 2, batch), ctx.synth_uniform(1, 0, 2, 2, batch)
mul.set_streams(1)
for _ in range(5):
    mul.multiply(a, b)
torch.cuda.synchronize()
leg("after_c2_steps_streams1")
fhe.prof_reset()
fhe.prof_enable(True)
for _ in range(5):
    mul.multiply(a, b)
torch.cuda.synchronize()
rep = fhe.prof_report()
fhe.prof_enable(False)
leg("after_profiled_steps")
fhe.prof_reset()
leg("after_prof_reset")
mul.set_streams(2)
for _ in range(5):
    mul.multiply(a, b)
torch.cuda.synchronize()
mul.set_streams(1)
leg("after_c2_steps_streams2")
for k in fhe.UBENCH_KINDS:
    fhe.ubench_int(k, 0.05)
fhe.ubench_scaler(par.extender(0), 0.05)
leg("after_ubench")
