#!/usr/bin/env python3
"""Round 6 A/B: the FP64-FMA kernels (fhe_engine_set_f64(1), default) against the integer narrow kernels (0) on the
reference's stock parameter sets, SAME process, SAME handles' shapes, alternating -- every hot-path Criterion ID of
crates/fhe/benches/bfv.rs through bench.py's own reference_default_128(), plus the forward / inverse NTT of a batch of
stock-set polynomials.  One JSON line per (set, id): on / off medians and the ratio.  VERDICT r05 #3's targets: stock
n = 8192 relinearize / rotate_columns / inner_sum >= +25 %, mul_and_relin >= +8 %, n = 16384 relinearize >= +25 %."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

REPS = int(os.environ.get("AB_REPS", "3"))
sets = tuple(int(x) for x in os.environ.get("AB_SETS", "4096,8192,16384").split(","))
runs = {True: [], False: []}
for rep in range(REPS):
    for on in ((True, False) if rep % 2 == 0 else (False, True)):
        fhe.set_f64(on)
        runs[on].append(bench.reference_default_128(fhe, torch, None, sets))
        fhe.workspace_trim()
        torch.cuda.empty_cache()
fhe.set_f64(True)
med = statistics.median
for key in runs[True][0]:
    if key == "note":
        continue
    ids = runs[True][0][key]["ids"]
    for idn in ids:
        row = dict(set=key, id=idn)
        for field in ("single_ms", "batch_ops_per_s"):
            if field not in ids[idn]:
                continue
            a = med([r[key]["ids"][idn][field] for r in runs[True]])
            b = med([r[key]["ids"][idn][field] for r in runs[False]])
            row[field] = dict(f64=a, integer=b, f64_over_integer=round(a / b, 4))
        print(json.dumps(row))
# NTT/s on the stock sets' own moduli (fhe-math benches/ntt.rs shape), batch of polynomials in place
timeit = bench.make_timeit(torch, 5)
for n in sets:
    q = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
         8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
         16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}[n]
    ctx = fhe.Context(q, n)
    batch = 2048 if n <= 8192 else 512
    x = ctx.synth_uniform(7, 0, 0, 1, batch).view(batch, len(q), n)
    row = dict(set=f"n={n}", id="ntt", batch_polys=batch, moduli=len(q))
    for name, fn in (("forward", lambda: ctx.ntt_forward(x)), ("backward", lambda: ctx.ntt_backward(x))):
        vals = {True: [], False: []}
        for rep in range(REPS):
            for on in (True, False):
                fhe.set_f64(on)
                vals[on].append(timeit(fn))
        fhe.set_f64(True)
        a, b = med(vals[True]), med(vals[False])
        rows_s = lambda ms: round(batch * len(q) / ms * 1e3, 0)
        row[name] = dict(f64_row_ntt_per_s=rows_s(a), integer_row_ntt_per_s=rows_s(b), f64_over_integer=round(b / a, 4),
                         f64_frac_of_8TBps=round(batch * len(q) * 2 * 8 * n / a / 1e6 / 8000.0, 4))
    print(json.dumps(row))
