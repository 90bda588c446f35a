#!/usr/bin/env python3
"""mul_and_relin / mul / relinearize on the reference's stock parameter sets (default_parameters_128: n = 4096, 8192, 16384)
at the bench's batches, for same-box A/Bs of changes that touch them (round 5: scaler instances for K = 6 / 10 / 18).
Prints one JSON line: ops/s per set and ID."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    import torch
    import fhe_rs_amd as fhe
    from bench import key_for, make_timeit
    timeit = make_timeit(torch, 5)
    sets = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
            8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
            16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                    0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}
    out = {}
    for n, q in sets.items():
        par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli=q)
        ctx = par.context_at_level(0)
        rk = fhe.RelinearizationKey(key_for(fhe, ctx, 7))
        batch = 1024 if n <= 8192 else 256
        a, b = ctx.synth_uniform(7, 0, 0, 2, batch), ctx.synth_uniform(7, 0, 2, 2, batch)
        mul, plain = fhe.Multiplicator.default(par, rk, 0), fhe.Multiplicator.default(par, None, 0)
        d = {"mul_and_relin": round(batch / timeit(lambda: mul.multiply(a, b)) * 1e3, 1),
             "mul": round(batch / timeit(lambda: plain.multiply(a, b)) * 1e3, 1)}
        mul.set_streams(1)
        d["mul_and_relin_one_stream"] = round(batch / timeit(lambda: mul.multiply(a, b)) * 1e3, 1)
        mul.set_streams(2)
        a1, b1 = a[:1].contiguous(), b[:1].contiguous()
        d["mul_and_relin_single_ms"] = round(timeit(lambda: mul.multiply(a1, b1)), 4)
        out[f"n={n}"] = d
        del a, b, mul, plain, rk, par, ctx
        fhe.workspace_trim()
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
