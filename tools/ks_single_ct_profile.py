#!/usr/bin/env python3
"""One ciphertext through relinearise at the C2 / C3 / C5 shapes, 20 calls with the fused strategy forced and 20 with
FHE_KS_AUTO (which takes the unfused kernels for launches this small): the workload tools/runs/r04_run25.sh wraps in
rocprofv3 --kernel-trace --stats, so that the per-kernel durations behind profiles/r04_ks_small_batches_all_modes.txt are on
file (a fused launch with a handful of workgroups costs ONE workgroup's walk over all digits)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe

K = fhe.KeySwitchingKey
for n, L in ((8192, 4), (16384, 8), (32768, 16)):
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    rk = fhe.RelinearizationKey(ksk)
    ct3 = ctx.synth_uniform(5, 0, 0, 3, 1)
    for mode in (K.FUSED, K.AUTO):
        ksk.set_mode(mode)
        for _ in range(20):
            rk.relinearizes(ct3)
        torch.cuda.synchronize()
print("ok")
