"""C5 (N = 32768, 16 x 60-bit, multiply + relinearise + modulus switch at level 0): eager launches against a captured
graph replay of the same call, batch 16 and 64 -- how much of the step is launch gaps."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_rs_amd as fhe
import bench

n, L = 32768, 16
t = fhe.generate_prime(20, 2 * n, 1 << 20)
q = fhe.generate_moduli([60] * L, n)
K = L + (60 * L + 60 + 61) // 62
ext, upper = [], 1 << 62
while len(ext) < K - L:
    upper = fhe.generate_prime(62, 2 * n, upper)
    if upper not in q:
        ext.append(upper)
ctx, mctx = fhe.Context(q, n), fhe.Context(q + ext, n)
Q = 1
for m in q:
    Q *= m
extender, down = fhe.Scaler(ctx, mctx, 1, 1), fhe.Scaler(mctx, ctx, t, Q)
mul = fhe.Multiplicator(extender, extender, down, fhe.RelinearizationKey(bench.key_for(fhe, ctx, 0xF4E50005)), True)
timeit = bench.make_timeit(torch, reps=10)
side = torch.cuda.Stream()
for batch in (16, 64):
    a, b = ctx.synth_uniform(0xF4E50005, 0, 0, 2, batch), ctx.synth_uniform(0xF4E50005, 0, 2, 2, batch)
    for streams in (1, 2):
        mul.set_streams(streams)
        eager = timeit(lambda: mul.multiply(a, b))
        with torch.cuda.stream(side):
            mul.multiply(a, b)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                out = mul.multiply(a, b)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            graph_ms = e0.elapsed_time(e1) / 10
        print(json.dumps(dict(batch=batch, streams=streams, eager_ms=round(eager, 3), graph_ms=round(graph_ms, 3),
                              eager_ops_per_s=round(batch / eager * 1e3, 1), graph_ops_per_s=round(batch / graph_ms * 1e3, 1))))
        del g, out
    del a, b
