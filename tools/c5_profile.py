#!/usr/bin/env python3
"""Per-kernel time of the C5 level-0 step (N = 32768, 16 x 60-bit, multiply + relinearise + modulus switch), batch 16 and 64,
single stream, library profiler.  Prints JSON."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import fhe_rs_amd as fhe
from bench import key_for
n, L = 32768, 16
par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * L)
ctx = par.context_at_level(0)
K = par.mul_context_at_level(0).nmoduli
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 5)), 0, True)
out = {}
for streams in (1, 2):
    mul.set_streams(streams)
    for batch in (16, 64):
        a, b = ctx.synth_uniform(5, 0, 0, 2, batch), ctx.synth_uniform(5, 0, 2, 2, batch)
        mul.multiply(a, b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            mul.multiply(a, b)
        e1.record(); torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / 3
        d = dict(wall_ms=round(wall, 3), ops_per_s=round(batch / wall * 1e3, 1))
        if streams == 1:
            fhe.prof_reset(); fhe.prof_enable(True)
            for _ in range(3):
                mul.multiply(a, b)
            torch.cuda.synchronize(); fhe.prof_enable(False)
            rep = fhe.prof_report(); fhe.prof_reset()
            tot = sum(v[1] for v in rep.values())
            d["kernel_sum_ms"] = round(tot / 3, 3)
            d["kernels"] = {k: dict(ms=round(v[1] / 3, 3), share=round(v[1] / tot, 3), launches=v[0] // 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])}
        out[f"streams{streams}_batch{batch}"] = d
        del a, b
print(json.dumps(out, indent=1))
