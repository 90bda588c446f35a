#!/usr/bin/env python3
"""Chunk x streams sweep of the C5 level-0 step (N = 32768, 16 x 60-bit, multiply + relinearise + modulus switch) at batch 32 /
64 / 128 and of an N = 32768, 8-moduli multiply at batch 128.  One JSON line per cell; chunk 0 = the default plan."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_rs_amd as fhe
from bench import key_for
n = 32768
for name, L, batches in (("C5", 16, (32, 64, 128)), ("n32768_L8", 8, (128,))):
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=[60] * L)
    ctx = par.context_at_level(0)
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 5)), 0, True)
    for batch in batches:
        a, b = ctx.synth_uniform(5, 0, 0, 2, batch), ctx.synth_uniform(5, 0, 2, 2, batch)

        def run(chunk, streams, steps=3):
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
            for chunk in (0, 8, 16, 32, 64, 128):
                if chunk > batch:
                    continue
                ms = min(run(chunk, streams), run(chunk, streams))
                print(json.dumps({"set": name, "batch": batch, "chunk": chunk, "streams": streams, "ms": round(ms, 3),
                                  "ops_per_s": round(batch / ms * 1e3)}), flush=True)
        del a, b
    del mul, par, ctx
    fhe.workspace_trim(); torch.cuda.empty_cache()
