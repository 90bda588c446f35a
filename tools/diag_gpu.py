#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-C-oracle diagnosis at full sizes (developer tool)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import fhe_rs_amd as fhe
import full_size
from full_size import u64
from fhe_oracle import bfv as obfv, coracle, synth

def run(n, nmod, batch):
    q = obfv.generate_moduli([60] * nmod, n); t = full_size.plaintext_modulus(n); seed = synth.seed_for_config(2)
    par = fhe.BfvParameters(n, t, moduli=q); ctx = par.context_at_level(0); mctx = par.mul_context_at_level(0)
    o = full_size.oracle_level(n, q, t, 0); cb, cm_ = o["cb"], o["cm"]
    c0, c1 = full_size.device_key(ctx, seed, nmod)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1); crk = full_size.host_key(cb, seed, nmod)
    idx = [0, batch - 1]
    # 1. key switch
    p = ctx.synth_uniform(seed, 0, 5, 1, batch)[:, 0].contiguous()
    g0, g1 = ksk.key_switch(p); torch.cuda.synchronize()
    for i in idx:
        w0, w1 = crk.key_switch(cb.synth_poly(seed, i, 5))
        print(n, "key_switch", i, np.array_equal(u64(g0[i]), w0), np.array_equal(u64(g1[i]), w1))
    # 2. relinearize
    ct3 = ctx.synth_uniform(seed, 0, 0, 3, batch)
    got = fhe.RelinearizationKey(ksk).relinearizes(ct3); torch.cuda.synchronize()
    for i in idx:
        parts = [cb.synth_poly(seed, i, pp) for pp in range(3)]
        k0, k1 = crk.key_switch(cb.poly_ntt_backward(parts[2]))
        print(n, "relinearize", i, np.array_equal(u64(got[i]), np.stack([cb.poly_add(parts[0], k0), cb.poly_add(parts[1], k1)])))
    # 3. extend / down scale
    x = ctx.synth_uniform(seed, 0, 0, 2, batch)
    e = par.extender(0).scale(x, ntt=True); torch.cuda.synchronize()
    for i in idx:
        print(n, "extend", i, np.array_equal(u64(e[i, 1]), o["cel"].scale(cb.synth_poly(seed, i, 1), True)))
    xm = mctx.synth_uniform(seed, 0, 0, 3, batch)
    dd = par.down_scaler(0).scale(xm, ntt=True); torch.cuda.synchronize()
    for i in idx:
        print(n, "down", i, np.array_equal(u64(dd[i, 2]), o["cdn"].scale(cm_.synth_poly(seed, i, 2), True)))
    # 4. multiply without / with relin
    lhs = ctx.synth_uniform(seed, 0, 0, 2, batch); rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
    for rk, crk_ in ((None, None), (fhe.RelinearizationKey(ksk), crk)):
        m = fhe.Multiplicator.default(par, rk, 0)
        out = m.multiply(lhs, rhs); torch.cuda.synchronize()
        cmul = coracle.CMul(cb, cm_, o["cel"], o["cel"], o["cdn"], crk_, False)
        for i in idx:
            l = np.stack([cb.synth_poly(seed, i, 0), cb.synth_poly(seed, i, 1)]); r = np.stack([cb.synth_poly(seed, i, 2), cb.synth_poly(seed, i, 3)])
            want = cmul.multiply(l, r); g = u64(out[i])
            print(n, "multiply relin=%s" % (rk is not None), i, [np.array_equal(g[pp], want[pp]) for pp in range(want.shape[0])])

for n, nmod, batch in ((1024, 4, 6), (8192, 4, 6), (8192, 4, 70)):
    run(n, nmod, batch)
