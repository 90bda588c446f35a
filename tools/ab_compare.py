#!/usr/bin/env python3
"""A/B harness used for every kernel experiment of the round: runs the C2 ct x ct + relinearise
pipeline (batch 1024) three times per library and prints the per-kernel HIP-event totals of 5
steps, alternating between the tree's build ("default") and any other builds dropped into
tools/_variants/*.so (e.g. the previous build, or one compiled with an experiment macro).
Alternation within ONE gpurun call matters: boxes of the pool differ by several percent.

    cp fhe.rs_amd/libfhe_hip.so tools/_variants/libB.so   # keep the old build
    ... edit kernels, python __graft_entry__.py ...
    gpurun -- 'python tools/ab_compare.py'
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, json
sys.path.insert(0, %r)
import torch, fhe_rs_amd as fhe
from fhe_rs_amd import _lib
if sys.argv[1] != 'default':
    _lib._load_for_tests(sys.argv[1])
n = 8192
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
L = 4
kk = ctx.synth_uniform(1, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
mul = fhe.Multiplicator.default(par, rk, 0)
a = ctx.synth_uniform(1, 0, 0, 2, 1024)
b = ctx.synth_uniform(1, 0, 2, 2, 1024)
for _ in range(3):
    mul.multiply(a, b)
torch.cuda.synchronize()
res = []
for rep in range(3):
    fhe.prof_reset()
    fhe.prof_enable(True)
    for _ in range(5):
        mul.multiply(a, b)
    torch.cuda.synchronize()
    fhe.prof_enable(False)
    res.append({k: round(v[1], 2) for k, v in fhe.prof_report().items()})
print(json.dumps(res))
""" % ROOT


def main():
    variants = sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so")))
    order = (["default"] + variants) * 2
    for lib in order:
        r = subprocess.run([sys.executable, "-c", CODE, lib], capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip()[-1200:], r.stderr.strip()[-300:] if r.returncode else "")


if __name__ == "__main__":
    main()
