#!/usr/bin/env python3
"""Per-entry-point timings on the C2 shapes (torch events on the current stream).
Prints one JSON line per measurement; used for A/B experiments and roofline bookkeeping."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe

N, SIZES, B = 8192, [60] * 4, int(os.environ.get("BK_BATCH", "512"))
SEED = 0xF4E50002


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tag = os.environ.get("BK_TAG", "")
    t = fhe.generate_prime(20, 2 * N, 1 << 20)
    par = fhe.BfvParameters(N, t, moduli_sizes=SIZES)
    ctx, mctx = par.context_at_level(0), par.mul_context_at_level(0)
    L, K = ctx.nmoduli, mctx.nmoduli
    R = 8 * N
    x = ctx.synth_uniform(SEED, 0, 0, 2, B)            # [B,2,L,N]
    xm = mctx.synth_uniform(SEED, 0, 0, 3, B // 2)     # [B/2,3,K,N]
    kk = ctx.synth_uniform(SEED, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, N)
    ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    rk = fhe.RelinearizationKey(ksk)
    mul = fhe.Multiplicator.default(par, rk, 0)
    ext, down = par.extender(0), par.down_scaler(0)
    rows = B * 2 * L
    res = []

    def rec(name, ms, alg_bytes, units, unit_name):
        res.append(dict(tag=tag, name=name, ms=round(ms, 4), GBps=round(alg_bytes / ms / 1e6, 1),
                        rate=round(units / ms * 1e3, 1), unit=unit_name))

    # measured HBM ceiling on this box (report fractions against nominal AND measured):
    # device-to-device copy of 2 GiB, read + write bytes counted
    src = torch.empty(1 << 28, dtype=torch.int64, device="cuda")
    dst = torch.empty_like(src)
    ms = timeit(lambda: dst.copy_(src))
    rec("hbm ceiling: D2D copy 2 GiB (read+write)", ms, 2 * src.numel() * 8, src.numel() * 8, "B copied/s")
    del src, dst
    ms = timeit(lambda: ctx.ntt_forward(x))
    rec("ntt_forward [B,2,4,8192]", ms, rows * 2 * R, rows, "row-NTT/s")
    ms = timeit(lambda: ctx.ntt_backward(x))
    rec("ntt_backward [B,2,4,8192]", ms, rows * 2 * R, rows, "row-NTT/s")
    ms = timeit(lambda: mctx.ntt_forward(xm))
    rec("ntt_forward 62-bit [B/2,3,9,8192]", ms, (B // 2) * 3 * K * 2 * R, (B // 2) * 3 * K, "row-NTT/s")
    pb = x.clone()
    ms = timeit(lambda: ext.scale(pb, ntt=False))
    rec("extend scale PowerBasis 4->9 (5 new rows)", ms, B * 2 * (L + (K - L)) * R, B * 2, "poly/s")
    ms = timeit(lambda: down.scale(xm, ntt=False))
    rec("down scale PowerBasis 9->4", ms, (B // 2) * 3 * (K + L) * R, (B // 2) * 3, "poly/s")
    p1 = x[:, 0].contiguous()
    ms = timeit(lambda: ksk.key_switch(p1))
    rec("key_switch [B,4,8192]", ms, B * (L * L + 2 * L) * R, B, "poly/s")
    ms = timeit(lambda: mul.multiply(x, x))
    rec("multiply+relin", ms, B * (22 * K + 7 * L + L * L + 4 * L) * R, B, "ops/s")
    # PIR server inner loop (dot_product_scalar): 256 query ciphertexts shared by 32 database rows
    count, rowsdb = 256, 32
    q = ctx.synth_uniform(SEED, 0, 0, 2, count)                       # [count,2,L,N]
    db = ctx.synth_uniform(SEED, 1000, 0, count, rowsdb).reshape(rowsdb, count, L, N)
    ms = timeit(lambda: ctx.dot_product_scalar(q, db))
    rec("dot_product_scalar 256 cts x 32 db rows", ms, (rowsdb * count * L + count * 2 * L + rowsdb * 2 * L) * R,
        rowsdb * count, "ct*pt MAC/s")
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
