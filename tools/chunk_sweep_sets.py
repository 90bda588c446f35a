#!/usr/bin/env python3
"""Chunk size x streams sweep of Multiplicator::multiply (library profiler off) on parameter sets other than C2: the
reference's stock sets (n = 4096 / 8192 / 16384) and a C3-shaped multiply -- where does the handle's default plan
(plan_chunks, tuned on C2 / C5 in rounds 2-3) stand?  One JSON line per (set, streams, chunk); chunk 0 = the default plan."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402
from bench import key_for  # noqa: E402

SETS = {"stock4096": (4096, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 1024),
        "stock8192": (8192, [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001], 1024),
        "stock16384": (16384, [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                               0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001], 256),
        "c3shape": (16384, None, 256), "n4096_4x60": (4096, "4x60", 2048)}
ONLY = sys.argv[1].split(",") if len(sys.argv) > 1 else None      # e.g. "stock8192,stock4096"
CHUNKS = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 16, 32, 64, 128, 256, 512, 1024, 2048]
for name, (n, q, batch) in SETS.items():
    if ONLY and name not in ONLY:
        continue
    t = fhe.generate_prime(20, 2 * n, (1 << 20) - 1)
    if q is None:
        par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 8)
    elif q == "4x60":
        par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
    else:
        par = fhe.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0)
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)

    def run(chunk, streams, steps=6):
        mul.set_chunk(chunk).set_streams(streams)
        for _ in range(2):
            mul.multiply(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            mul.multiply(a, b)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps
    for streams in (1, 2):
        for chunk in CHUNKS:
            if chunk > batch:
                continue
            ms = min(run(chunk, streams), run(chunk, streams))
            print(json.dumps({"set": name, "batch": batch, "chunk": chunk, "streams": streams, "ms": round(ms, 3),
                              "ops_per_s": round(batch / ms * 1e3)}), flush=True)
    del a, b, mul, par, ctx
    fhe.workspace_trim()
    torch.cuda.empty_cache()
