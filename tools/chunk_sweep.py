#!/usr/bin/env python3
"""Chunk size x streams sweep of the C2 multiply (batch 1024), library profiler OFF (its two events per launch would
charge small chunks for their launch count).  Usage: python tools/chunk_sweep.py [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, L = 8192, 4
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
ctx = par.context_at_level(0)
kk = ctx.synth_uniform(1, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
mul = fhe.Multiplicator.default(par, rk, 0)
a = ctx.synth_uniform(1, 0, 0, 2, batch)
b = ctx.synth_uniform(1, 0, 2, 2, batch)


def run(chunk, streams, steps=10):
    mul.set_chunk(chunk).set_streams(streams)
    for _ in range(3):
        mul.multiply(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        mul.multiply(a, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for rep in range(2):
    for streams in (1, 2):
        for chunk in (0, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
            if chunk > batch:
                continue
            ms = run(chunk, streams)
            print(json.dumps({"batch": batch, "chunk": chunk, "streams": streams, "ms": round(ms, 3),
                              "ops_per_s": round(batch / ms * 1e3)}), flush=True)
