#!/usr/bin/env python3
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import fhe_rs_amd as fhe
import full_size
from full_size import u64
from fhe_oracle import bfv as obfv, coracle, synth
n, nmod, batch = 8192, 4, 1
q = obfv.generate_moduli([60] * nmod, n); t = full_size.plaintext_modulus(n); seed = synth.seed_for_config(2)
par = fhe.BfvParameters(n, t, moduli=q); ctx = par.context_at_level(0)
o = full_size.oracle_level(n, q, t, 0); cb, cm_ = o["cb"], o["cm"]
lhs = ctx.synth_uniform(seed, 0, 0, 2, batch); rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
m = fhe.Multiplicator.default(par, None, 0)
out = m.multiply(lhs, rhs); torch.cuda.synchronize()
flat = u64(out).reshape(-1)
stage = os.environ.get("FHE_DEBUG_STAGE")
K = 9
if stage in ("1", "2"):
    part0 = 0 if stage == "1" else 2
    w0 = o["cel"].scale(cb.synth_poly(seed, 0, part0), True)      # [K][N] slot 0
    g = flat[: K * n].reshape(K, n)
    print("stage", stage, "slot0 rows equal:", [bool(np.array_equal(g[r], w0[r])) for r in range(K)], "(rows<4 not written when copy is skipped)")
    w1 = o["cel"].scale(cb.synth_poly(seed, 0, part0 + 1), True)
    g1 = flat[K * n: 12 * n].reshape(3, n)
    print("stage", stage, "slot1 rows0-2 equal:", [bool(np.array_equal(g1[r], w1[r])) for r in range(3)])
    for r in range(4, K):
        bad = np.nonzero(g[r] != w0[r])[0]
        print(" row", r, "mismatches", len(bad), "first idx", bad[:8], "last", bad[-3:] if len(bad) else None)
else:
    e = [o["cel"].scale(cb.synth_poly(seed, 0, pp), True) for pp in range(4)]
    c0 = cm_.poly_mul(e[0], e[2])
    g = flat[: K * n].reshape(K, n)
    print("stage 3 ten slot0 rows equal:", [bool(np.array_equal(g[r], c0[r])) for r in range(K)])
