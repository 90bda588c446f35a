#!/usr/bin/env python3
"""Single-ciphertext latency, eager `_dev` calls vs the same calls replayed from a captured hipGraph (torch.cuda.CUDAGraph
around the library's launches) -- the shape the reference's Criterion IDs time (crates/fhe/benches/bfv.rs:247-255: one
`multiply` per iteration).  Prints one JSON object.  Usage: python tools/latency_graph_ab.py [--reps 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import torch
    import fhe_rs_amd as fhe
    sys.path.insert(0, ROOT)
    from bench import key_for

    def wall_ms(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def sync_ms(fn, reps):
        """call + wait per iteration: what a host that needs the result before its next step sees"""
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def ab(name, fn, out):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        eager_q = wall_ms(fn, args.reps)
        graph_q = wall_ms(g.replay, args.reps)
        eager_s = sync_ms(fn, args.reps)
        graph_s = sync_ms(g.replay, args.reps)
        out[name] = dict(eager_ms=round(eager_q, 4), graph_ms=round(graph_q, 4), eager_sync_ms=round(eager_s, 4),
                         graph_sync_ms=round(graph_s, 4))
        print(name, out[name], file=sys.stderr, flush=True)

    out = {}
    # C2
    n = 8192
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, 1))
    m = fhe.Multiplicator.default(par, rk, 0)
    a, b = ctx.synth_uniform(1, 0, 0, 2, 1), ctx.synth_uniform(1, 0, 2, 2, 1)
    ab("C2_mul_and_relin", lambda: m.multiply(a, b), out)
    c3 = fhe.Multiplicator.default(par, None, 0).multiply(a, b)
    ab("C2_relinearize", lambda: rk.relinearizes(c3), out)
    # the reference's stock set n = 8192 / log q = 218
    q = [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli=q)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, 2))
    m2 = fhe.Multiplicator.default(par, rk, 0)
    a2, b2 = ctx.synth_uniform(1, 0, 0, 2, 1), ctx.synth_uniform(1, 0, 2, 2, 1)
    ab("default128_n8192_mul_and_relin", lambda: m2.multiply(a2, b2), out)
    gk = fhe.GaloisKey(key_for(fhe, ctx, 3), 3)
    ab("default128_n8192_rotate_columns", lambda: gk.relinearize(a2), out)
    # C3
    n = 16384
    ctx = fhe.Context(fhe.generate_moduli([60] * 8, n), n)
    ksk = key_for(fhe, ctx, 3)
    rk3 = fhe.RelinearizationKey(ksk)
    ct3 = ctx.synth_uniform(3, 0, 0, 3, 1)
    ab("C3_relinearize", lambda: rk3.relinearizes(ct3), out)
    # C5
    n = 32768
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * 16)
    ctx = par.context_at_level(0)
    m5 = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 5)), 0, True)
    a5, b5 = ctx.synth_uniform(5, 0, 0, 2, 1), ctx.synth_uniform(5, 0, 2, 2, 1)
    ab("C5_level0_mul_relin_modswitch", lambda: m5.multiply(a5, b5), out)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
