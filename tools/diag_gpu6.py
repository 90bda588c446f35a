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
flat = u64(out).reshape(12, n)
e = [o["cel"].scale(cb.synth_poly(seed, 0, pp), True) for pp in range(4)]
pbl = [cb.poly_ntt_backward(cb.synth_poly(seed, 0, pp)) for pp in range(4)]
# raw mode, offset 0: ten rows 0..8 = c00 rows 0..8 (as read by the tensor kernel), rows 9..11 = c10 rows 0..2
g = flat[8]; w = e[0][8]
bad = np.nonzero(g != w)[0]
print("c00 row 8 as read: mismatches", len(bad))
if len(bad):
    i = int(bad[0])
    print(" idx", bad[:6], "read", int(g[i]), "expected(NTT)", int(w[i]))
    # is the wrong value the pre-NTT (PowerBasis scaled) value?  compute PowerBasis extension row 8
    ext_pb = o["cel"].scale(pbl[0], False)   # PowerBasis in -> PowerBasis out rows
    print(" equals pre-NTT value:", int(g[i]) == int(ext_pb[8][i]), " count equal to pre-NTT among bad:", int(np.sum(g[bad] == ext_pb[8][bad])))
    print(" zero count among bad:", int(np.sum(g[bad] == 0)))
if os.environ.get("FHE_DEBUG_OFFSET") == "9":
    g = flat[8]; w = e[2][8]   # dump rows 9..20 -> index 8 is ten row 17 = c10 row 8
    bad = np.nonzero(g != w)[0]
    print("c10 row 8 as read: mismatches", len(bad))
    if len(bad):
        i = int(bad[0])
        ext_pb = o["cel"].scale(pbl[2], False)
        print(" idx", bad[:6], "read", int(g[i]), "expected(NTT)", int(w[i]), "pre-NTT", int(ext_pb[8][i]))
        print(" count equal to pre-NTT among bad:", int(np.sum(g[bad] == ext_pb[8][bad])), " zeros:", int(np.sum(g[bad] == 0)))
        for r in range(9):
            print("  c10 row", r, "mismatches", int(np.sum(flat[r] != e[2][r])))
