#!/usr/bin/env python3
"""Small-batch latency of ct x ct + relinearise (C2 parameters) through the `_dev` entry point:
eager launches vs the same call captured once into a hipGraph and replayed (torch.cuda.CUDAGraph
drives hipStreamBeginCapture on the stream the library launches on)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = 8192
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    L = ctx.nmoduli
    kk = ctx.synth_uniform(1, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
    mul = fhe.Multiplicator.default(par, rk, 0)
    side = torch.cuda.Stream()
    for batch in (1, 4, 16, 64):
        a, b = ctx.synth_uniform(1, 0, 0, 2, batch), ctx.synth_uniform(1, 0, 2, 2, batch)
        for _ in range(3):
            mul.multiply(a, b)
        eager = timed(lambda: mul.multiply(a, b), 50)
        with torch.cuda.stream(side):
            mul.multiply(a, b)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = mul.multiply(a, b)
        g.replay()
        graph = timed(g.replay, 50)
        want = mul.multiply(a, b)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        print(json.dumps({"batch": batch, "eager_ms": round(eager, 4), "graph_ms": round(graph, 4),
                          "eager_ops_per_s": round(batch / eager * 1e3), "graph_ops_per_s": round(batch / graph * 1e3)}),
              flush=True)


if __name__ == "__main__":
    main()
