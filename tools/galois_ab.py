#!/usr/bin/env python3
"""Rotation-family timings for the same-box A/B of the folded Galois substitution (round 5): C3 relinearise / rotations
(batch 512), C2 inner sum (batch 256), C2 oblivious expansion of one ciphertext to N outputs, C2 RGSW external product,
stock n = 8192 / log q = 218 rotate_columns (batch 1024 and 1).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    import torch
    import fhe_rs_amd as fhe
    from bench import key_for, make_timeit
    timeit = make_timeit(torch, 5)
    out = {}
    n, L, batch = 16384, 8, 512
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    ksk = key_for(fhe, ctx, 3)
    rk, gk3, gkr = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3), fhe.GaloisKey(ksk, 2 * n - 1)
    ct3 = ctx.synth_uniform(3, 0, 0, 3, batch)
    ct2 = ct3[:, :2].contiguous()
    out["C3_relinearize_ms"] = round(timeit(lambda: rk.relinearizes(ct3)), 4)
    out["C3_rotate_columns_ms"] = round(timeit(lambda: gk3.relinearize(ct2)), 4)
    out["C3_rotate_rows_ms"] = round(timeit(lambda: gkr.relinearize(ct2)), 4)
    one = ct2[:1].contiguous()
    out["C3_rotate_columns_single_ms"] = round(timeit(lambda: gk3.relinearize(one)), 4)
    del ct3, ct2, one, rk, gk3, gkr, ksk, ctx
    fhe.workspace_trim()
    n, L = 8192, 4
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    ksk = key_for(fhe, ctx, 2)
    seq, i = [], 1
    while i < n // 2:
        seq.append(pow(3, i, 2 * n))
        i *= 2
    seq.append(2 * n - 1)
    ek = fhe.EvaluationKey(n, [fhe.GaloisKey(ksk, e) for e in set(seq + [(n >> l) + 1 for l in range(13)])])
    ct = ctx.synth_uniform(2, 0, 0, 2, 256)
    out["C2_inner_sum_256_ms"] = round(timeit(lambda: ek.computes_inner_sum(ct)), 4)
    one = ct[:1].contiguous()
    out["C2_expand_1_to_8192_ms"] = round(timeit(lambda: ek.expands(one, n)), 4)
    big = ctx.synth_uniform(2, 0, 0, 2, 1024)
    out["C2_rotate_columns_1024_ms"] = round(timeit(lambda: ek.rotates_columns_by(big, 1)), 4)
    out["C2_rotate_columns_single_ms"] = round(timeit(lambda: ek.rotates_columns_by(one, 1)), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
