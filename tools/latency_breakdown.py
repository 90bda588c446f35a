#!/usr/bin/env python3
"""Where one single-ciphertext call spends its time: per-launch kernel durations (the library's dispatch-stamped HIP
events) against the call's wall time on the stream, for C2 / the stock n = 8192 set / C3 / C5.  The difference is what the
dependent launches cost between kernels.  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    import torch
    import fhe_rs_amd as fhe
    from bench import key_for

    def measure(fn, reps=50):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        fhe.prof_reset()
        fhe.prof_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        fhe.prof_enable(False)
        rep = fhe.prof_report()
        fhe.prof_reset()
        kern = {k: dict(launches_per_call=v[0] / reps, us_per_call=round(v[1] / reps * 1e3, 2)) for k, v in rep.items()}
        ksum = sum(v[1] for v in rep.values()) / reps
        nl = sum(v[0] for v in rep.values()) / reps
        return dict(wall_ms=round(wall, 4), kernel_sum_ms=round(ksum, 4), launches=nl,
                    gap_us_per_launch=round((wall - ksum) / max(nl, 1) * 1e3, 2), kernels=kern)

    out = {}
    n = 8192
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, 1))
    for streams in (1, 2):
        m = fhe.Multiplicator.default(par, rk, 0).set_streams(streams)
        a, b = ctx.synth_uniform(1, 0, 0, 2, 1), ctx.synth_uniform(1, 0, 2, 2, 1)
        out[f"C2_mul_and_relin_streams{streams}"] = measure(lambda: m.multiply(a, b))
    gk = fhe.GaloisKey(key_for(fhe, ctx, 3), 3)
    out["C2_rotate_columns"] = measure(lambda: gk.relinearize(a))
    q = [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli=q)
    ctx = par.context_at_level(0)
    m2 = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 2)), 0)
    a2, b2 = ctx.synth_uniform(1, 0, 0, 2, 1), ctx.synth_uniform(1, 0, 2, 2, 1)
    out["default128_n8192_mul_and_relin"] = measure(lambda: m2.multiply(a2, b2))
    n = 32768
    par = fhe.BfvParameters(n, fhe.generate_prime(20, 2 * n, 1 << 20), moduli_sizes=[60] * 16)
    ctx = par.context_at_level(0)
    m5 = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(fhe, ctx, 5)), 0, True)
    a5, b5 = ctx.synth_uniform(5, 0, 0, 2, 1), ctx.synth_uniform(5, 0, 2, 2, 1)
    out["C5_level0_mul_relin_modswitch"] = measure(lambda: m5.multiply(a5, b5))
    # one workgroup's latency by tile size: 64 rows (a quarter of the CUs) of n points, forward and inverse, 60- and 62-bit
    lat = {}
    for n in (512, 1024, 2048, 4096, 8192, 16384):
        for bits in (60, 62):
            c = fhe.Context([fhe.generate_prime(bits, 2 * n, 1 << bits)], n)
            x = c.synth_uniform(7, 0, 0, 1, 64)
            f = measure(lambda: c.ntt_forward(x), 100)
            b = measure(lambda: c.ntt_backward(x), 100)
            lat[f"n={n}/{bits}bit"] = dict(fwd_us=round(f["kernel_sum_ms"] * 1e3, 2), inv_us=round(b["kernel_sum_ms"] * 1e3, 2))
    out["ntt_workgroup_latency_64_rows"] = lat
    print(json.dumps(out))


if __name__ == "__main__":
    main()
