#!/usr/bin/env python3
"""A/B of library builds on the ct x ct + relinearise pipeline, one box, alternating (ABC ABC ...): the in-tree build
("default") and every tools/_variants/*.so.  Per build and round: C2 (N=8192, 4x60-bit) at batch 1024 -- event-free
ms per step on one stream and in the handle's default two-stream mode, the per-kernel HIP-event totals of 10
single-stream steps -- plus batch 16 / 64 and the C5 level-0 step at batch 16.  One JSON line per (build, round).
usage: python tools/ab_mul.py [rounds=3]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, json, time
sys.path.insert(0, %r)
import torch, fhe_rs_amd as fhe
from fhe_rs_amd import _lib
if sys.argv[1] != 'default':
    _lib._load_for_tests(sys.argv[1])

def key_for(ctx, seed):
    L = ctx.nmoduli
    kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
    return fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())

def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

res = {}
n = 8192
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
rk = fhe.RelinearizationKey(key_for(ctx, 1))
mul = fhe.Multiplicator.default(par, rk, 0)
for batch in (1024, 64, 16):
    a, b = ctx.synth_uniform(1, 0, 0, 2, batch), ctx.synth_uniform(1, 0, 2, 2, batch)
    mul.set_streams(1)
    res[f"c2_b{batch}_s1_ms"] = round(timeit(lambda: mul.multiply(a, b), 20), 4)
    mul.set_streams(2)
    res[f"c2_b{batch}_s2_ms"] = round(timeit(lambda: mul.multiply(a, b), 20), 4)
    if batch == 1024:
        mul.set_streams(1)
        fhe.prof_reset(); fhe.prof_enable(True)
        for _ in range(10):
            mul.multiply(a, b)
        torch.cuda.synchronize()
        fhe.prof_enable(False)
        res["c2_kernels_ms_per_10"] = {k: round(v[1], 2) for k, v in sorted(fhe.prof_report().items())}
    del a, b
del mul, rk, par
fhe.workspace_trim(); torch.cuda.empty_cache()
n, L = 32768, 16
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
ctx = par.context_at_level(0)
mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(key_for(ctx, 5)), 0, mod_switch=True)
a, b = ctx.synth_uniform(5, 0, 0, 2, 16), ctx.synth_uniform(5, 0, 2, 2, 16)
res["c5_b16_ms"] = round(timeit(lambda: mul.multiply(a, b), 5), 4)
print(json.dumps(res))
""" % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so")))
    for rnd in range(rounds):
        for lib in ["default"] + variants:
            r = subprocess.run([sys.executable, "-c", CODE, lib], capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            try:
                d = json.loads(line)
            except Exception:
                d = dict(error=(r.stderr or r.stdout)[-400:])
            d.update(build=os.path.basename(lib), round=rnd)
            print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
