#!/usr/bin/env python3
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import fhe_rs_amd as fhe
import full_size
from full_size import u64
from fhe_oracle import bfv as obfv, coracle, synth

n, nmod, batch = 8192, 4, 6
q = obfv.generate_moduli([60] * nmod, n); t = full_size.plaintext_modulus(n); seed = synth.seed_for_config(2)
par = fhe.BfvParameters(n, t, moduli=q); ctx = par.context_at_level(0); mctx = par.mul_context_at_level(0)
o = full_size.oracle_level(n, q, t, 0); cb, cm_ = o["cb"], o["cm"]
x = ctx.synth_uniform(seed, 0, 0, 2, batch)
e = par.extender(0).scale(x, ntt=True); torch.cuda.synchronize()
for i in range(batch):
    print("extend ct", i, [np.array_equal(u64(e[i, s]), o["cel"].scale(cb.synth_poly(seed, i, s), True)) for s in range(2)])
# row-level detail for ct 0 slot 0
w = o["cel"].scale(cb.synth_poly(seed, 0, 0), True); g = u64(e[0, 0])
print("extend ct0 slot0 rows", [bool(np.array_equal(g[r], w[r])) for r in range(9)])
xm = mctx.synth_uniform(seed, 0, 0, 3, batch)
dd = par.down_scaler(0).scale(xm, ntt=True); torch.cuda.synchronize()
for i in range(batch):
    print("down ct", i, [np.array_equal(u64(dd[i, s]), o["cdn"].scale(cm_.synth_poly(seed, i, s), True)) for s in range(3)])
lhs = ctx.synth_uniform(seed, 0, 0, 2, batch); rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
m = fhe.Multiplicator.default(par, None, 0)
out1 = m.multiply(lhs, rhs); torch.cuda.synchronize()
out2 = m.multiply(lhs, rhs); torch.cuda.synchronize()
print("deterministic:", torch.equal(out1, out2))
cmul = coracle.CMul(cb, cm_, o["cel"], o["cel"], o["cdn"], None, False)
for i in range(batch):
    l = np.stack([cb.synth_poly(seed, i, 0), cb.synth_poly(seed, i, 1)]); r = np.stack([cb.synth_poly(seed, i, 2), cb.synth_poly(seed, i, 3)])
    want = cmul.multiply(l, r); g = u64(out1[i])
    print("mul ct", i, [[bool(np.array_equal(g[pp][rr], want[pp][rr])) for rr in range(4)] for pp in range(3)])
# composition through the API: extend, element-wise tensor, down
el = par.extender(0).scale(lhs, ntt=True); er = par.extender(0).scale(rhs, ntt=True)
c0 = mctx.mul(el[:, 0].contiguous().clone(), er[:, 0].contiguous())
c2 = mctx.mul(el[:, 1].contiguous().clone(), er[:, 1].contiguous())
c1 = mctx.add(mctx.mul(el[:, 0].contiguous().clone(), er[:, 1].contiguous()), mctx.mul(el[:, 1].contiguous().clone(), er[:, 0].contiguous()))
ten = torch.stack([c0, c1, c2], dim=1).contiguous()
comp = par.down_scaler(0).scale(ten, ntt=True); torch.cuda.synchronize()
print("composition == multiply:", torch.equal(comp, out1))
for i in (0, batch - 1):
    l = np.stack([cb.synth_poly(seed, i, 0), cb.synth_poly(seed, i, 1)]); r = np.stack([cb.synth_poly(seed, i, 2), cb.synth_poly(seed, i, 3)])
    print("composition vs oracle ct", i, np.array_equal(u64(comp[i]), cmul.multiply(l, r)))
