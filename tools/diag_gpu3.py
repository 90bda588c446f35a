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
par = fhe.BfvParameters(n, t, moduli=q); ctx = par.context_at_level(0)
o = full_size.oracle_level(n, q, t, 0); cb, cm_ = o["cb"], o["cm"]
lhs = ctx.synth_uniform(seed, 0, 0, 2, batch); rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
m = fhe.Multiplicator.default(par, None, 0)
cmul = coracle.CMul(cb, cm_, o["cel"], o["cel"], o["cdn"], None, False)
want = [cmul.multiply(np.stack([cb.synth_poly(seed, i, 0), cb.synth_poly(seed, i, 1)]), np.stack([cb.synth_poly(seed, i, 2), cb.synth_poly(seed, i, 3)])) for i in range(batch)]
for chunk in (0, 1, 2):
    fhe.set_chunk(chunk)
    out = m.multiply(lhs, rhs); torch.cuda.synchronize()
    print(os.environ.get("TAG"), "chunk", chunk, [[bool(np.array_equal(u64(out[i])[p], want[i][p])) for p in range(3)] for i in range(batch)])
