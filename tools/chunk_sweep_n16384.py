#!/usr/bin/env python3
"""Chunk x streams sweep of multiply + relinearise at N = 16384 for several basis sizes (is the two-stream plan's 768 MB
chunk budget, tuned on C2 / C5, right for rows of 128 KiB?).  One JSON line per (shape, streams, chunk); chunk 0 = default."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_rs_amd as fhe
from bench import key_for
n = 16384
STOCK = [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
         0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]
for name, kw, batch in (("L4", dict(moduli_sizes=[60] * 4), 512), ("L6", dict(moduli_sizes=[60] * 6), 384), ("L8", dict(moduli_sizes=[60] * 8), 256),
                        ("stock9", dict(moduli=STOCK), 256), ("L12", dict(moduli_sizes=[60] * 12), 192)):
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), **kw)
    ctx = par.context_at_level(0)
    L, K = ctx.nmoduli, par.mul_context_at_level(0).nmoduli
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 7)), 0)
    a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)

    def run(chunk, streams, steps=5):
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
        for chunk in (0, 16, 32, 48, 64, 96, 128, 192, 256):
            if chunk > batch:
                continue
            ms = min(run(chunk, streams), run(chunk, streams))
            print(json.dumps({"set": name, "L": L, "K": K, "batch": batch, "chunk": chunk, "streams": streams, "ms": round(ms, 3),
                              "ops_per_s": round(batch / ms * 1e3)}), flush=True)
    del a, b, mul, par, ctx
    fhe.workspace_trim(); torch.cuda.empty_cache()
