#!/usr/bin/env python3
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import fhe_rs_amd as fhe
import full_size
from full_size import u64
from fhe_oracle import bfv as obfv, coracle, synth
n, nmod, batch = int(os.environ.get("DN", "8192")), 4, 1
q = obfv.generate_moduli([60] * nmod, n); t = full_size.plaintext_modulus(n); seed = synth.seed_for_config(2)
par = fhe.BfvParameters(n, t, moduli=q); ctx = par.context_at_level(0)
o = full_size.oracle_level(n, q, t, 0); cb, cm_ = o["cb"], o["cm"]
lhs = ctx.synth_uniform(seed, 0, 0, 2, batch); rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
m = fhe.Multiplicator.default(par, None, 0)
out = m.multiply(lhs, rhs); torch.cuda.synchronize()
flat = u64(out).reshape(12, n)
K = 9
e = [o["cel"].scale(cb.synth_poly(seed, 0, pp), True) for pp in range(4)]
c0 = cm_.poly_mul(e[0], e[2]); c2 = cm_.poly_mul(e[1], e[3])
c1 = cm_.poly_add(cm_.poly_mul(e[0], e[3]), cm_.poly_mul(e[1], e[2]))
want = np.concatenate([c0, c1, c2])   # 27 rows slot-major for nb=1
off = int(os.environ.get("FHE_DEBUG_OFFSET", "0"))
for r in range(12):
    w = want[off + r]; g = flat[r]
    bad = np.nonzero(g != w)[0]
    info = ""
    if len(bad):
        mod = (q + [4611686018427322369, 4611686018427289601, 4611686018426454017, 4611686018426257409, 4611686018425815041])[(off + r) % K]
        info = f" first {bad[:4]} last {bad[-2:]} g={int(g[bad[0]])} w={int(w[bad[0]])} diff%p={(int(g[bad[0]])-int(w[bad[0]]))%mod} mod={mod}"
    print("N", n, "ten row", off + r, "(slot %d row %d)" % ((off + r) // K, (off + r) % K), "mismatches", len(bad), info)
