#!/usr/bin/env python3
"""C5 relinearise (N = 32768, 16 x 60-bit, 16 ciphertexts) with the fused and with the unfused key switch, three calls
each: the workload tools/runs/r04_run9.sh wraps in rocprofv3 (kernel stats, FETCH_SIZE, WRITE_SIZE) for the traffic of the
two strategies (profiles/r04_ks_c5_pmc.json)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe

n, L, batch = 32768, 16, 16
ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
rk = fhe.RelinearizationKey(ksk)
ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
for mode in (fhe.KeySwitchingKey.FUSED, fhe.KeySwitchingKey.UNFUSED):
    ksk.set_mode(mode)
    for _ in range(3):
        rk.relinearizes(ct3)
    torch.cuda.synchronize()
print("ok")
