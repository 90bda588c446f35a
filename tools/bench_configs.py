#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one MI355X (torch events, inputs resident):
C3: N=16384, 8x60-bit, relinearise (3->2 parts) and rotation (Galois key switch);
C5: N=32768, 16x60-bit, multiply + relinearise + modulus switch at the first chain levels.
Prints one JSON line per measurement."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe


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


def key_for(ctx, seed):
    L = ctx.nmoduli
    kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
    return fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())


def c3(batch=512):
    n, L = 16384, 8
    q = fhe.generate_moduli([60] * L, n)
    ctx = fhe.Context(q, n)
    ksk = key_for(ctx, 0xF4E50003)
    rk, gk3, gkr = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3), fhe.GaloisKey(ksk, 2 * n - 1)
    ct3 = ctx.synth_uniform(0xF4E50003, 0, 0, 3, batch)
    ct2 = ct3[:, :2].contiguous()
    R = 8 * n
    for name, fn, rows in (("C3 relinearize 3->2 (n=16384, 8x60b)", lambda: rk.relinearizes(ct3), 2 * L + L * L + 4 * L),
                           ("C3 rotate columns e=3", lambda: gk3.relinearize(ct2), 2 * L + L * L + 3 * L),
                           ("C3 rotate rows e=2N-1", lambda: gkr.relinearize(ct2), 2 * L + L * L + 3 * L)):
        ms = timeit(fn)
        print(json.dumps(dict(name=name, batch=batch, ms=round(ms, 3), ops_per_s=round(batch / ms * 1e3, 1),
                              stage_model_GBps=round(batch * rows * R / ms / 1e6, 1), kernels=breakdown(fn))))


def c5(batch=16, levels=2):
    n, L = 32768, 16
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
    for level in range(levels):
        ctx = par.context_at_level(level)
        Ll, K = ctx.nmoduli, par.mul_context_at_level(level).nmoduli
        rk = fhe.RelinearizationKey(key_for(ctx, 0xF4E50005 + level))
        m = fhe.Multiplicator.default(par, rk, level, mod_switch=True)
        a = ctx.synth_uniform(0xF4E50005, 0, 0, 2, batch)
        b = ctx.synth_uniform(0xF4E50005, 0, 2, 2, batch)
        ms = timeit(lambda: m.multiply(a, b))
        rows = 22 * K + 7 * Ll + Ll * Ll + 4 * Ll + 12 * Ll - 6
        print(json.dumps(dict(name=f"C5 multiply+relin+modswitch level {level} (n=32768, L={Ll}, K={K})", batch=batch,
                              ms=round(ms, 3), ops_per_s=round(batch / ms * 1e3, 1),
                              stage_model_GBps=round(batch * rows * 8 * n / ms / 1e6, 1),
                              kernels=breakdown(lambda: m.multiply(a, b)) if level == 0 else None)))


def breakdown(fn):
    """Per-kernel HIP-event times (ms) of one call (the library's own profiler)."""
    fn()
    torch.cuda.synchronize()
    fhe.prof_reset()
    fhe.prof_enable(True)
    fn()
    torch.cuda.synchronize()
    fhe.prof_enable(False)
    return {k: dict(launches=v[0], ms=round(v[1], 3)) for k, v in sorted(fhe.prof_report().items())}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "c3"):
        c3()
    if which in ("all", "c5"):
        c5()
