#!/usr/bin/env python3
"""Same-process, alternating A/B of the key-switch evaluation strategies (fhe_ksk_set_mode) on BASELINE.json's shapes:
   C2 (N=8192, 4 moduli, batch 1024): relinearise, and the whole ct x ct + relinearise
   C3 (N=16384, 8 moduli, batch 512): relinearise, rotate columns
   C5 (N=32768, 16 moduli, batch 16 / 64): relinearise, and the level-0 multiply + relinearise + modulus switch
For each (shape, op) every variant (mode, W budget) is timed `rounds` times in alternation (ABAB...), torch events on
the current stream, 3 calls per timing.  One JSON line per (shape, op, variant) with all rounds, plus the library's
per-kernel HIP-event breakdown of one call.  usage: python tools/ks_modes_ab.py [c2] [c3] [c5] [--rounds N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe

MiB = 1 << 20
K = fhe.KeySwitchingKey


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def breakdown(fn):
    fn()
    torch.cuda.synchronize()
    fhe.prof_reset()
    fhe.prof_enable(True)
    fn()
    torch.cuda.synchronize()
    fhe.prof_enable(False)
    return {k: [v[0], round(v[1], 3)] for k, v in sorted(fhe.prof_report().items())}


def key_for(ctx, seed):
    L = ctx.nmoduli
    kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
    return K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())


def run(shape, op, fn, ksk, variants, batch, rounds):
    times = {name: [] for name, _, _ in variants}
    for _ in range(rounds):
        for name, mode, budget in variants:
            ksk.set_mode(mode, budget)
            times[name].append(round(timeit(fn), 4))
    for name, mode, budget in variants:
        ksk.set_mode(mode, budget)
        t = times[name]
        print(json.dumps(dict(shape=shape, op=op, variant=name, mode=mode, w_budget=budget, batch=batch, ms=t,
                              best_ms=min(t), ops_per_s=round(batch / min(t) * 1e3, 1), kernels=breakdown(fn))), flush=True)
    ksk.set_mode(0, 0)


def c2(rounds):
    n, batch = 8192, 1024
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    ksk = key_for(ctx, 0xF4E50002)
    rk = fhe.RelinearizationKey(ksk)
    mul = fhe.Multiplicator.default(par, rk, 0).set_streams(1)
    a, b = ctx.synth_uniform(2, 0, 0, 2, batch), ctx.synth_uniform(2, 0, 2, 2, batch)
    ct3 = ctx.synth_uniform(2, 0, 0, 3, batch)
    V = [("fused", 1, 0), ("unfused_128M", 2, 0), ("unfused_64M", 2, 64 * MiB), ("unfused_512M", 2, 512 * MiB)]
    run("C2", "relinearize", lambda: rk.relinearizes(ct3), ksk, V, batch, rounds)
    run("C2", "mul_relin", lambda: mul.multiply(a, b), ksk, V[:2] + V[3:], batch, rounds)


def c3(rounds):
    n, L, batch = 16384, 8, 512
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    ksk = key_for(ctx, 0xF4E50003)
    rk, gk3 = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
    ct3 = ctx.synth_uniform(3, 0, 0, 3, batch)
    ct2 = ct3[:, :2].contiguous()
    V = [("fused", 1, 0), ("unfused_128M", 2, 0), ("unfused_sub_128M", 3, 0), ("unfused_sub_64M", 3, 64 * MiB),
         ("unfused_sub_256M", 3, 256 * MiB), ("unfused_sub_1G", 3, 1024 * MiB), ("unfused_1G", 2, 1024 * MiB)]
    run("C3", "relinearize", lambda: rk.relinearizes(ct3), ksk, V, batch, rounds)
    run("C3", "rotate_columns", lambda: gk3.relinearize(ct2), ksk, V[:3], batch, rounds)


def c5(rounds):
    n, L = 32768, 16
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
    ctx = par.context_at_level(0)
    ksk = key_for(ctx, 0xF4E50005)
    rk = fhe.RelinearizationKey(ksk)
    mul = fhe.Multiplicator.default(par, rk, 0, mod_switch=True)
    V = [("fused", 1, 0), ("unfused_128M", 2, 0), ("unfused_64M", 2, 64 * MiB), ("unfused_256M", 2, 256 * MiB),
         ("unfused_1G", 2, 1024 * MiB), ("unfused_4G", 2, 4096 * MiB)]
    for batch in (16, 64):
        a, b = ctx.synth_uniform(5, 0, 0, 2, batch), ctx.synth_uniform(5, 0, 2, 2, batch)
        ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
        run("C5", "relinearize", lambda: rk.relinearizes(ct3), ksk, V, batch, rounds)
        run("C5", "mul_relin_modswitch", lambda: mul.multiply(a, b), ksk, V[:2] + V[3:5], batch, rounds)
        del a, b, ct3
    fhe.workspace_trim()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a in ("c2", "c3", "c5")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    for name in (args or ["c2", "c3", "c5"]):
        {"c2": c2, "c3": c3, "c5": c5}[name](rounds)
        fhe.workspace_trim()
        torch.cuda.empty_cache()
