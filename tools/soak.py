#!/usr/bin/env python3
"""Determinism soak on the GPU: the C2 ct x ct + relinearise batch is recomputed many times and every
result must be bit-identical to the first (which bench.py / the parity tests tie to the oracle).  A
missing wait state in hand-placed asm, a race on LDS or on the workspace pool would show up here as
a sporadic mismatch rather than as a reproducible wrong answer."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fhe_rs_amd as fhe


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n = 8192
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    L = ctx.nmoduli
    kk = ctx.synth_uniform(7, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(ksk), 0).set_streams(1)
    gk = fhe.GaloisKey(ksk, 3)
    a, b = ctx.synth_uniform(7, 0, 0, 2, 1024), ctx.synth_uniform(7, 0, 2, 2, 1024)
    ref_m, ref_r = mul.multiply(a, b), gk.relinearize(a)   # single-stream reference
    torch.cuda.synchronize()
    streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mul.set_streams(streams)                               # 2 (the default): chunks alternate between two streams
    bad, t0 = 0, time.time()
    for i in range(reps):
        m, r = mul.multiply(a, b), gk.relinearize(a)
        if not (torch.equal(m, ref_m) and torch.equal(r, ref_r)):
            bad += 1
    torch.cuda.synchronize()
    print(json.dumps(dict(repetitions=reps, streams=streams, ops=reps * 2048, mismatches=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


def c3(reps):
    """The same for config C3's shape: relinearise + rotation at N = 16384, 8 moduli (radix-4 key-switch passes, the
    Ntt-form digit shortcut), 256 ciphertexts per call."""
    n, L = 16384, 8
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    kk = ctx.synth_uniform(9, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
    ct3 = ctx.synth_uniform(9, 0, 0, 3, 256)
    ct2 = ct3[:, :2].contiguous()
    ref_r, ref_g = rk.relinearizes(ct3), gk.relinearize(ct2)
    torch.cuda.synchronize()
    bad, t0 = 0, time.time()
    for _ in range(reps):
        r, g = rk.relinearizes(ct3), gk.relinearize(ct2)
        if not (torch.equal(r, ref_r) and torch.equal(g, ref_g)):
            bad += 1
    torch.cuda.synchronize()
    print(json.dumps(dict(config="C3 shape", repetitions=reps, ops=reps * 512, mismatches=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


def small(reps):
    """Round 3: one-chunk batches in the handle's default mode -- the lhs / rhs operand extensions run side by side on
    the caller's and the internal stream (per-call fork / join events from the pool) -- and the squaring shortcut:
    batch 16 at C2, every result against the single-stream reference."""
    n = 8192
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4)
    ctx = par.context_at_level(0)
    L = ctx.nmoduli
    kk = ctx.synth_uniform(7, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(ksk), 0).set_streams(1)
    a, b = ctx.synth_uniform(7, 0, 0, 2, 16), ctx.synth_uniform(7, 0, 2, 2, 16)
    ref_m, ref_s = mul.multiply(a, b), mul.multiply(a, a.clone())   # single stream, general pipeline
    torch.cuda.synchronize()
    mul.set_streams(2)
    bad, t0 = 0, time.time()
    for i in range(reps):
        m, sq = mul.multiply(a, b), mul.multiply(a, a)
        if i % 8 == 0 and not (torch.equal(m, ref_m) and torch.equal(sq, ref_s)):
            bad += 1
    torch.cuda.synchronize()
    print(json.dumps(dict(config="C2 batch 16, two streams (split extension) + squaring shortcut", repetitions=reps,
                          ops=reps * 32, checked=(reps + 7) // 8, mismatches=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


def auto_small(reps):
    """Round 4: launches that FHE_KS_AUTO sends to the unfused kernels (few fused workgroups) -- relinearise and a
    rotation at the C2 / C3 / C5 shapes with 8 / 4 / 2 ciphertexts, every result against the SAME call with the fused
    strategy forced on the handle (cross-strategy equality and determinism in one check)."""
    K = fhe.KeySwitchingKey
    bad, total, t0 = 0, 0, time.time()
    for n, L, batch in ((8192, 4, 8), (16384, 8, 4), (32768, 16, 2)):
        ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
        kk = ctx.synth_uniform(11, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        ksk = K(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
        rk, gk = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3)
        ct3 = ctx.synth_uniform(11, 0, 0, 3, batch)
        ct2 = ct3[:, :2].contiguous()
        ksk.set_mode(K.FUSED)
        ref_r, ref_g = rk.relinearizes(ct3), gk.relinearize(ct2)
        torch.cuda.synchronize()
        ksk.set_mode(K.AUTO)
        for _ in range(reps):
            r, g = rk.relinearizes(ct3), gk.relinearize(ct2)
            if not (torch.equal(r, ref_r) and torch.equal(g, ref_g)):
                bad += 1
            total += 2 * batch
    torch.cuda.synchronize()
    print(json.dumps(dict(config="C2 / C3 / C5 shapes at 8 / 4 / 2 ciphertexts, FHE_KS_AUTO (unfused) against the forced fused result",
                          repetitions=reps, ops=total, mismatches=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


def single(reps):
    """Round 5: one ciphertext pair per call (the reference's Criterion shape).  Such a launch runs the tensor + iNTT of
    ALL rows on the general passes in one launch, where a batch runs the 60-bit rows on the narrow passes: the pair's
    result must equal its slice of the batch-1024 result (cross-path equality) every time (determinism) -- at C2 and on
    the reference's stock n = 8192 / log q = 218 set, whose rotation (folded substitution, unfused key switch) is checked
    the same way."""
    bad, total, t0 = 0, 0, time.time()
    n = 8192
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    for q in (None, [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]):
        par = fhe.BfvParameters(n, t, moduli_sizes=[60] * 4) if q is None else fhe.BfvParameters(n, t, moduli=q)
        ctx = par.context_at_level(0)
        L = ctx.nmoduli
        kk = ctx.synth_uniform(7, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        ksk = fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
        mul = fhe.Multiplicator.default(par, fhe.RelinearizationKey(ksk), 0)
        gk = fhe.GaloisKey(ksk, 3)
        a, b = ctx.synth_uniform(7, 0, 0, 2, 1024), ctx.synth_uniform(7, 0, 2, 2, 1024)
        ref_m, ref_r = mul.multiply(a, b), gk.relinearize(a)
        torch.cuda.synchronize()
        for i in range(reps):
            k = i % 1024
            a1, b1 = a[k:k + 1].contiguous(), b[k:k + 1].contiguous()
            m, r = mul.multiply(a1, b1), gk.relinearize(a1)
            if not (torch.equal(m[0], ref_m[k]) and torch.equal(r[0], ref_r[k])):
                bad += 1
            total += 2
    torch.cuda.synchronize()
    print(json.dumps(dict(config="one pair per call (C2 and stock n=8192/log q=218) against its slice of the batch-1024 result",
                          repetitions=reps, ops=total, mismatches=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "single":
        single(int(sys.argv[1]))
    if len(sys.argv) > 3 and sys.argv[3] == "auto_small":
        auto_small(int(sys.argv[1]))
    if len(sys.argv) > 3 and sys.argv[3] == "c3":
        c3(int(sys.argv[1]))
    if len(sys.argv) > 3 and sys.argv[3] == "small":
        small(int(sys.argv[1]))
    main()
