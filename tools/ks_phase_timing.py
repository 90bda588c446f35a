#!/usr/bin/env python3
"""Where one wave of the fused key switch spends its cycles: C2 (N = 8192, 4 x 60-bit, the multiply's key switch of a
PowerBasis c2, 512 polynomials per launch) and C3 (N = 16384, 8 x 60-bit, relinearise of 512 three-part ciphertexts: the
caller's Ntt rows stand in for one transform per key modulus).

Needs the diagnostic build (lab only, never loaded by the package):
    bash tools/build_variant.sh phase2 -DFHE_PHASE_TIMING=2 -DFHE_TS_BLOCK=100   (scalar-register stamps, kernel scope)
    bash tools/build_variant.sh phase1 -DFHE_PHASE_TIMING=1 -DFHE_TS_BLOCK=100   (global stamps, also inside the passes)
    cp tools/_variants/libfhe_hip_phase2.so fhe.rs_amd/libfhe_hip.so     (on the GPU box's copy; restore afterwards)
Thread 0 of workgroup FHE_TS_BLOCK stamps the shader clock (s_memtime) at kernel entry, at every phase boundary and at
exit; every stamp charges the time since the previous one to its slot, so the slots TELESCOPE: their sum is the wave's
whole time between entry and exit -- the table covers 100 % of the wave's cycles by construction; `wave_vs_kernel` says
how much of the launch's duration (HIP events) that lifetime is.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe
from fhe_rs_amd import _lib

lib = _lib.lib()
fn = lib.fhe_debug_phase_timing
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_size_t]


def read():
    buf = (C.c_uint64 * 512)()
    fn(buf, 512)
    return [int(v) for v in buf]


def names_for(npass, wl_from):
    """slot -> label for a transform of `npass` LDS passes whose exchanges are wave-local from pass `wl_from` on"""
    n = {20: "item prologue (indices, accumulator clears, own Ntt row x key)", 0: "digit loop top",
         1: "lift + tile write", 2: "barrier after the tile write",
         3: "key words requested (after the last pass)", 4: "barrier before the MAC",
         5: "Shoup MAC (key loads, two accumulator sets)", 6: "barrier after the MAC",
         21: "epilogue (next row prefetch, reductions, addends, stores)",
         7: "transform: every pass with the exchanges / barriers between them (mode 2: not split; mode 1: ~0)"}
    for p in range(npass):
        n[8 + 2 * p] = f"pass {p + 1}"
        if p + 1 < npass:
            n[9 + 2 * p] = ("wave-local exchange" if p + 1 >= wl_from else "workgroup barrier") + f" after pass {p + 1}"
    return n


def run(tag, call, kernel_label, npass, wl_from, reps=10):
    call()
    torch.cuda.synchronize()
    read()                                  # clear
    fhe.prof_reset()
    fhe.prof_enable(True)
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    fhe.prof_enable(False)
    rep = fhe.prof_report()
    fhe.prof_reset()
    raw = read()
    # mode 2: one row of 32 slots per wave of the stamped workgroup; the budget is the average over its waves
    waves = [raw[32 * w: 32 * w + 32] for w in range(16) if raw[32 * w + 31]]
    entries = waves[0][31]
    nw = len(waves)
    slots = [sum(w[k] for w in waves) / nw for k in range(31)] + [0] * 32
    total = sum(slots[:31])
    per_wave_total = [sum(w[:31]) / max(w[31], 1) for w in waves]
    names = names_for(npass, wl_from)
    launches, ms = rep[kernel_label]
    per_launch_cycles = total / max(entries, 1)
    us = ms / launches * 1e3
    table = {}
    for k, v in enumerate(slots[:31]):
        if not v:
            continue
        shares = [w[k] / max(sum(w[:31]), 1) for w in waves]
        table[names.get(k, f"slot {k}")] = dict(cycles_per_launch=round(v / max(entries, 1), 1), share=round(v / total, 4),
                                                share_min_wave=round(min(shares), 4), share_max_wave=round(max(shares), 4))
    cat = dict(transform_passes=0.0, barriers_and_exchanges=0.0, lift=0.0, mac=0.0, prologue_epilogue=0.0, other=0.0)
    for k, v in enumerate(slots[:31]):
        if not v:
            continue
        lab = names.get(k, "")
        if lab.startswith("pass ") or lab.startswith("transform:"):
            cat["transform_passes"] += v
        elif "barrier" in lab or "exchange" in lab:
            cat["barriers_and_exchanges"] += v
        elif lab.startswith("lift"):
            cat["lift"] += v
        elif "MAC" in lab or "key words" in lab:
            cat["mac"] += v
        elif "prologue" in lab or "epilogue" in lab:
            cat["prologue_epilogue"] += v
        else:
            cat["other"] += v
    return dict(workload=tag, kernel=kernel_label, launches_timed=launches, kernel_us_per_launch=round(us, 2),
                waves_reporting=nw, stamped_entries_per_wave=entries, wave_cycles_per_launch=round(per_launch_cycles, 0),
                wave_cycles_per_launch_min_max=[round(min(per_wave_total)), round(max(per_wave_total))],
                implied_mhz_if_wave_spans_kernel=round(per_launch_cycles / us, 1),
                coverage_of_wave_cycles=1.0,
                by_category={k: round(v / total, 4) for k, v in cat.items()}, phases=table)


out = {}
# ---- C2: the multiply's key switch (PowerBasis input, no own row) ----
N, B = 8192, 512
par = fhe.BfvParameters(N, fhe.generate_prime(20, 2 * N, 1 << 20), moduli_sizes=[60] * 4)
ctx = par.context_at_level(0)
L = ctx.nmoduli
kk = ctx.synth_uniform(2, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, N)
ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()).set_mode(fhe.KeySwitchingKey.FUSED)
p1 = ctx.synth_uniform(2, 0, 0, 1, B)[:, 0].contiguous()
out["C2_key_switch_512_polys"] = run("N=8192, 4x60-bit, 512 polynomials, PowerBasis input", lambda: ksk.key_switch(p1),
                                     "key_switch_fused", npass=5, wl_from=2)
del p1, ksk, kk, ctx, par
# ---- C3: relinearise (own Ntt row per key modulus) ----
N, B = 16384, 512
ctx = fhe.Context(fhe.generate_moduli([60] * 8, N), N)
L = ctx.nmoduli
kk = ctx.synth_uniform(3, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, N)
ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()).set_mode(fhe.KeySwitchingKey.FUSED)
rk = fhe.RelinearizationKey(ksk)
ct3 = ctx.synth_uniform(3, 0, 0, 3, B)
out["C3_relinearize_512"] = run("N=16384, 8x60-bit, relinearise of 512 ciphertexts (7 transformed digits + the own row per key modulus)",
                                lambda: rk.relinearizes(ct3), "key_switch_fused", npass=6, wl_from=2)
print(json.dumps(out, indent=1))
