#!/usr/bin/env python3
"""One JSON object: forward-NTT launch times on three kinds of rows (60-bit narrow, 62-bit wide, stock 43/44-bit F64) and the
C2 multiply's step time with its per-kernel split -- the workload of an A/B of ntt_kernel's forward epilogue
(FHE_FWD_DIRECT_STORE, tools/ab_two_libs.sh)."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

timeit = bench.make_timeit(torch, 10)
out = {}
n = 8192
for name, q in (("narrow60", fhe.generate_moduli([60] * 4, n)),
                ("wide62", [fhe.generate_prime(62, 2 * n, (1 << 62) - k * (1 << 40)) for k in range(1, 5)]),
                ("stock_f64", [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001])):
    ctx = fhe.Context(q, n)
    x = ctx.synth_uniform(3, 0, 0, 1, 2048).view(2048, len(q), n)
    ms = statistics.median(timeit(lambda: ctx.ntt_forward(x)) for _ in range(5))
    out[name + "_fwd_ms"] = round(ms, 4)
    out[name + "_row_ntt_per_s"] = round(2048 * len(q) / ms * 1e3, 0)
    del x
# the C2 multiply (bench.py's step, single stream, library events on)
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(bench.key_for(fhe, ctx, bench.SEED)), 0).set_streams(1)
a, b = ctx.synth_uniform(bench.SEED, 0, 0, 2, 1024), ctx.synth_uniform(bench.SEED, 0, 2, 2, 1024)
ms = statistics.median(timeit(lambda: mul.multiply(a, b)) for _ in range(5))
out["c2_ms_per_1024"] = round(ms, 4)
out["c2_ops_per_s"] = round(1024 / ms * 1e3, 1)
fhe.prof_reset()
fhe.prof_enable(True)
for _ in range(10):
    mul.multiply(a, b)
torch.cuda.synchronize()
fhe.prof_enable(False)
out["c2_kernel_ms"] = {k: round(v[1] / 10, 4) for k, v in sorted(fhe.prof_report().items())}
print(json.dumps(out))
