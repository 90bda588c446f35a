#!/usr/bin/env python3
"""Where one wave of the fused key switch spends its cycles (diagnostic build -DFHE_PHASE_TIMING, library copied over
the in-tree one by tools/ab_lib-style scripts): C2 shape, 512 polynomials per launch; thread 0 of workgroup 777 stamps
the shader clock at phase boundaries.  Prints cycles per digit per phase."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe
from fhe_rs_amd import _lib

N, B = 8192, 512
t = fhe.generate_prime(20, 2 * N, 1 << 20)
par = fhe.BfvParameters(N, t, moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
L = ctx.nmoduli
kk = ctx.synth_uniform(2, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, N)
ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
p1 = ctx.synth_uniform(2, 0, 0, 1, B)[:, 0].contiguous()
lib = _lib.lib()
fn = lib.fhe_debug_phase_timing
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_size_t]
buf = (C.c_uint64 * 64)()
ksk.key_switch(p1); torch.cuda.synchronize(); fn(buf, 64)          # warm-up, clear
reps = 20
for _ in range(reps):
    ksk.key_switch(p1)
torch.cuda.synchronize()
fn(buf, 64)
names = {0: "loop top / previous end barrier exit", 1: "lift + tile write", 2: "barrier after the tile write",
         8: "pass 1 (2 stages)", 9: "barrier", 10: "pass 2 (2 stages)", 11: "barrier", 12: "pass 3 (3 stages)",
         13: "wave-local sync", 14: "pass 4 (3 stages)", 15: "wave-local sync", 16: "pass 5 (3 stages)",
         3: "key prefetch issue (after pass 5)", 4: "barrier before the MAC", 5: "Shoup MAC (4 chunks, key loads)",
         6: "barrier after the MAC"}
per = {names.get(k, str(k)): round(buf[k] / (reps * L), 1) for k in range(64) if buf[k]}
print(json.dumps(dict(cycles_per_digit=per, total=round(sum(per.values()), 1)), indent=1))
