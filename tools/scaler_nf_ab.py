#!/usr/bin/env python3
"""One JSON object per process: Multiplicator::multiply on the reference's stock sets and on BASELINE-shaped bases, with output
digests -- the workload of the A/B of the scaler's exact-fit instances (round 6: NF = 3 / 5 / 8 / 10 / 16 / 18)."""
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

SETS = {"stock4096": (4096, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 1024),
        "stock8192": (8192, [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001], 1024),
        "stock16384": (16384, [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                               0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001], 256),
        "c2": (8192, [60] * 4, 1024), "c3shape": (16384, [60] * 8, 128), "c5level0": (32768, [60] * 16, 16),
        "n8192_7x50": (8192, [50] * 7, 256)}
timeit = bench.make_timeit(torch, 6)
out = {}
for name, (n, q, batch) in SETS.items():
    t = fhe.generate_prime(20, 2 * n, (1 << 20) - 1)
    par = fhe.BfvParameters(n, t, moduli=q) if q[0] > 64 else fhe.BfvParameters(n, t, moduli_sizes=q)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(bench.key_for(fhe, ctx, 7)), 0)
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
    out[name + "_ms"] = round(statistics.median(timeit(lambda: mul.multiply(a, b)) for _ in range(5)), 4)
    out[name + "_digest"] = hashlib.sha256(mul.multiply(a[:4], b[:4]).cpu().numpy().tobytes()).hexdigest()[:16]
    del a, b, mul, par, ctx
    fhe.workspace_trim()
    torch.cuda.empty_cache()
if os.environ.get("FHE_AB_CHAIN", "1") == "1":
    ch = bench.c5_chain(fhe, torch)
    out["c5_chain_ms"] = ch["total_ms"]
    out["c5_chain_per_level_ms"] = ch["per_level_ms"]
print(json.dumps(out))
