#!/usr/bin/env python3
"""Key switch on rows larger than LDS: 16384-point parts (FHE_KS_FUSED) against 8192-point sub-blocks (FHE_KS_FUSED_SUB)
and the engine's per-launch choice (FHE_KS_AUTO) over the batch sizes either side of "the launch fills the device".
Relinearise (one key switch + the add), same process, alternating, `rounds` rounds, 5 calls per timing.
With `all`: every strategy (also FHE_KS_UNFUSED / _SUB, whose first stage has batch x digits x key moduli tiles and fills
the device where the fused kernels' batch x key moduli workgroups do not) on the C2 / C3 / C5 shapes at small batches.
usage: python tools/ks_small_launch_ab.py [rounds] [all]   -- one JSON line per (shape, batch)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe

K = fhe.KeySwitchingKey
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


cus = torch.cuda.get_device_properties(0).multi_processor_count
if "stock" in sys.argv:
    # round 5: the reference's stock sets (default_parameters_128: 3 / 5 / 9 digits of 36-49 bits) -- does the per-launch
    # rule derived on BASELINE's shapes (4 / 8 / 16 digits of 60 bits) pick the faster strategy there too?
    SETS = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
            8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
            16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                    0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}
    for n, q in SETS.items():
        L = len(q)
        ctx = fhe.Context(q, n)
        kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
        rk = fhe.RelinearizationKey(ksk)
        modes = [("fused", K.FUSED), ("unfused", K.UNFUSED), ("auto", K.AUTO)]
        for batch in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128):
            ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
            ms = {name: [] for name, _ in modes}
            for _ in range(rounds):
                for name, mode in modes:
                    ksk.set_mode(mode)
                    ms[name].append(timeit(lambda: rk.relinearizes(ct3), reps=10))
            best = min(("fused", "unfused"), key=lambda k_: min(ms[k_]))
            print(json.dumps({"n": n, "moduli": L, "batch": batch, "fused_workgroups": batch * L, "compute_units": cus,
                              "ms": {k_: min(v) for k_, v in ms.items()}, "best": best,
                              "auto_within_pct_of_best": round((min(ms["auto"]) / min(ms[best]) - 1) * 100, 1)}), flush=True)
    sys.exit(0)
if "rounds12" in sys.argv:
    # round 5: fused launches of one to two and a half rounds of workgroups (a partly filled last round) on BASELINE's shapes
    for n, L, wgs in ((8192, 4, (192, 256, 288, 320, 384, 448, 512, 576, 640)), (16384, 8, (192, 256, 288, 320, 384, 448, 512, 576, 640)),
                      (32768, 16, (256, 320, 384, 512, 576, 640))):
        ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
        kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
        rk = fhe.RelinearizationKey(ksk)
        per = L * max(1, n // 16384)
        for wg in wgs:
            batch = wg // per
            ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
            ms = {"fused": [], "unfused": [], "auto": []}
            for _ in range(rounds):
                for name, mode in (("fused", K.FUSED), ("unfused", K.UNFUSED), ("auto", K.AUTO)):
                    ksk.set_mode(mode)
                    ms[name].append(timeit(lambda: rk.relinearizes(ct3), reps=10))
            best = min(("fused", "unfused"), key=lambda k_: min(ms[k_]))
            print(json.dumps({"n": n, "moduli": L, "batch": batch, "fused_workgroups": batch * per, "compute_units": cus,
                              "ms": {k_: min(v) for k_, v in ms.items()}, "best": best,
                              "auto_within_pct_of_best": round((min(ms["auto"]) / min(ms[best]) - 1) * 100, 1)}), flush=True)
    sys.exit(0)
if "all" in sys.argv:
    for n, L, batches in ((8192, 4, (1, 2, 4, 8, 16, 32, 64, 128, 256)), (16384, 8, (1, 2, 4, 8, 16, 32, 64)), (32768, 16, (1, 2, 4, 8, 16)),
                          (4096, 3, (1, 4, 16, 64, 256))):
        ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
        kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
        rk = fhe.RelinearizationKey(ksk)
        modes = [("fused", K.FUSED), ("unfused", K.UNFUSED), ("auto", K.AUTO)]
        if n == 16384:
            modes.insert(2, ("unfused_sub", K.UNFUSED_SUB))
        if n > 16384:
            modes.insert(1, ("fused_sub", K.FUSED_SUB))
        for batch in batches:
            ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
            ms = {name: [] for name, _ in modes}
            for _ in range(rounds):
                for name, mode in modes:
                    ksk.set_mode(mode)
                    ms[name].append(timeit(lambda: rk.relinearizes(ct3), reps=10))
            print(json.dumps({"n": n, "moduli": L, "batch": batch, "fused_workgroups": batch * L * max(1, n // 16384),
                              "compute_units": cus, "ms": ms}), flush=True)
    sys.exit(0)
for n, L, batches in ((32768, 16, (1, 2, 3, 4, 6, 8, 16)), (32768, 4, (1, 4, 8, 12, 16, 24, 32, 64)), (65536, 4, (1, 4, 8, 12, 16, 32))):
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    rk = fhe.RelinearizationKey(ksk)
    for batch in batches:
        ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
        ms = {"parts_16384": [], "sub_blocks_8192": [], "auto": []}
        for _ in range(rounds):
            for name, mode in (("parts_16384", K.FUSED), ("sub_blocks_8192", K.FUSED_SUB), ("auto", K.AUTO)):
                ksk.set_mode(mode)
                ms[name].append(timeit(lambda: rk.relinearizes(ct3)))
        sub_blocks = batch * L * (n // 8192)
        print(json.dumps({"n": n, "moduli": L, "batch": batch, "sub_blocks_8192": sub_blocks, "compute_units": cus,
                          "auto_takes": "sub_blocks_8192" if sub_blocks <= cus else "parts_16384", "ms": ms}), flush=True)
