#!/usr/bin/env python3
"""Generates the committed golden fixtures of tests/golden/ (SURVEY.md 8c "fixtures to commit").

The reference (Rust) cannot be built or run in this image and its own tests hold no golden
ciphertext / NTT vectors (they draw from an unseeded RNG), so these vectors come from the
Python oracle (oracle/fhe_oracle, the line-by-line restatement pinned by the reference's
closed-form assertions and prime-list KATs in tests/test_oracle_*.py) and, at N = 8192, from
the plain-C oracle (itself checked against the Python oracle in tests/test_oracle_c.py).
They freeze today's oracle outputs: any later drift of the oracle, the emulated kernels or the
HIP kernels shows up as a diff against data, not only as a diff against a moving oracle.

    python tests/golden/make_golden.py        # rewrites the fixtures in place

Files:
  bfv_small_traces.json   N=16, L in {1,2,3,6}: inputs, relin/Galois keys and every stage of
                          Multiplicator::multiply (extended, tensor, scaled, relinearised,
                          mod-switched) + one rotation, full coefficient arrays.
  c2_digest.json          BASELINE config C2 (N=8192, 4x60-bit): SHA-256 of the NTT tables and
                          of ct x ct + relinearize on the seeded synthetic inputs (+ first/last
                          64 coefficients of every output row).
  scaler_constants.json   RnsScaler constant blocks (extender, down-scaler, decrypt scaler) of
                          configs C1-C3: SHA-256 + leading values.
  default128_digest.json  The reference's stock sets (default_parameters_128, parameters.rs:218-251), n = 4096 / log q = 109
                          and n = 8192 / log q = 218: SHA-256 of the NTT tables of the multiplication basis and of the
                          C oracle's output ciphertext 0 for every hot-path Criterion ID of benches/bfv.rs (mul, square,
                          mul_and_relin, relinearize, rotate_rows, rotate_columns, inner_sum, expand_4, mul_and_relin_2)
                          and the multiply + modulus-switch chain down to one modulus (tests/ref_params.py; the emulated
                          kernels are checked against the oracle on the way).
  decrypt_n1024.npz/json  N=1024, 2x62-bit: secret-key encryptions, relin key, product
                          ciphertext; the product decrypts to the plaintext product.
"""
import hashlib
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

from fhe_oracle import bfv as obfv, coracle, synth  # noqa: E402
from fhe_oracle.rns import ScalingFactor  # noqa: E402
from fhe_oracle.rq import Context as OCtx, Scaler as OScaler, NTT  # noqa: E402


def rows(polys):
    return [[[int(v) for v in r] for r in p.coefficients] for p in polys]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).tobytes()).hexdigest()


def ksk_rows(k):
    return dict(c0=rows(k.c0), c1=rows(k.c1))


def small_trace(nmod, n=16, seed=2024):
    rng = random.Random(seed + nmod)
    opar = obfv.BfvParameters.default_arc(nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    va, vb = [rng.randrange(t) for _ in range(n)], [rng.randrange(t) for _ in range(n)]
    A, B = sk.encrypt(va, rng, 0), sk.encrypt(vb, rng, 0)
    out = dict(n=n, moduli=opar.moduli, plaintext=t, psi=[op.psi for op in opar.ctx[0].ops],
               mul_moduli=opar.mul_params[0].to.moduli, mul_psi=[op.psi for op in opar.mul_params[0].to.ops],
               lhs=rows(A.c), rhs=rows(B.c), plain_lhs=va, plain_rhs=vb)
    tensor3 = A.mul(B)
    out["tensor_product_3part"] = rows(tensor3.c)           # &ct * &ct, ops/mod.rs:259-358
    if nmod >= 2:
        ork = obfv.RelinearizationKey(sk, rng, 0, 0)
        out["relin_key"] = ksk_rows(ork.ksk)
        trace = {}
        m = obfv.Multiplicator.default(ork)
        res = m.multiply(A, B, trace=trace)
        out["extended"] = rows(trace["extended"])
        out["tensor"] = rows(trace["tensor"])
        out["scaled"] = rows(trace["scaled"])
        out["multiply_relin"] = rows(res.c)
        m2 = obfv.Multiplicator.default(ork)
        m2.enable_mod_switching()
        out["multiply_relin_modswitch"] = rows(m2.multiply(A, B).c)
        ogk = obfv.GaloisKey(sk, 3, 0, 0, rng)
        out["galois_key_e3"] = ksk_rows(ogk.ksk)
        out["rotate_e3"] = rows(ogk.relinearize(A).c)
        # the product decrypts correctly (ops/mul.rs:263-367 asserts the same)
        want = obfv_negacyclic(va, vb, t, n)
        assert sk.decrypt(res) == want and sk.decrypt(m2.multiply(A, B)) == want
    return out


def obfv_negacyclic(a, b, t, n):
    """Plaintext product a(x) * b(x) mod (x^n + 1, t) for the polynomial encoding of encrypt(values)."""
    out = [0] * n
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                k = i + j
                if k < n:
                    out[k] = (out[k] + x * y) % t
                else:
                    out[k - n] = (out[k - n] - x * y) % t
    return out


def c2_digest():
    import full_size
    n, sizes, cfg = 8192, [60] * 4, 2
    q = obfv.generate_moduli(sizes, n)
    t = full_size.plaintext_modulus(n)
    seed = synth.seed_for_config(cfg)
    o = full_size.oracle_level(n, q, t, 0)
    cb = o["cb"]
    ops = o["mul"].ops
    tables = dict(omegas=sha([op.omegas for op in ops]), omegas_shoup=sha([op.omegas_shoup for op in ops]),
                  zetas_inv=sha([op.zetas_inv for op in ops]), zetas_inv_shoup=sha([op.zetas_inv_shoup for op in ops]))
    crk = full_size.host_key(cb, seed, len(q))
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], crk, False)
    outs = []
    for i in (0, 1, 1023):
        lhs = np.stack([cb.synth_poly(seed, i, p) for p in (0, 1)])
        rhs = np.stack([cb.synth_poly(seed, i, p) for p in (2, 3)])
        r = cm.multiply(lhs, rhs)
        outs.append(dict(ct=i, input_sha256=sha(np.stack([lhs, rhs])), output_sha256=sha(r),
                         head=[[[int(v) for v in row[:64]] for row in part] for part in r],
                         tail=[[[int(v) for v in row[-64:]] for row in part] for part in r]))
    return dict(n=n, moduli=q, mul_moduli=o["mul"].moduli, plaintext=t, seed=seed, psi=[op.psi for op in ops],
                tables_sha256=tables, key_sha256=sha(np.stack([np.stack([cb.synth_poly(seed, 0, 8 + 2 * i) for i in range(4)]),
                                                                 np.stack([cb.synth_poly(seed, 0, 9 + 2 * i) for i in range(4)])])),
                outputs=outs)


def scaler_block(s):
    r = s.scaler
    block = [r.gamma, r.gamma_shoup, [v for row in r.omega for v in row], [v for row in r.omega_shoup for v in row],
             r.theta_omega_lo, r.theta_omega_hi, [1 if b else 0 for b in r.theta_omega_sign],
             r.theta_garner_lo, r.theta_garner_hi,
             [r.theta_gamma_lo, r.theta_gamma_hi, 1 if r.theta_gamma_sign else 0, r.theta_garner_shift,
              1 if r.scaling_factor.is_one else 0]]
    flat = [int(v) for part in block for v in part]
    return dict(nfrom=len(s.frm.moduli), nto=len(s.to.moduli), number_common_moduli=s.number_common_moduli,
                sha256=sha(flat), gamma=[int(v) for v in r.gamma], theta_garner_shift=r.theta_garner_shift,
                theta_gamma=[r.theta_gamma_lo, r.theta_gamma_hi, 1 if r.theta_gamma_sign else 0],
                omega_row0=[int(v) for v in r.omega[0]])


def scaler_constants():
    import full_size
    out = {}
    for name, n, sizes in (("C1", 4096, [50]), ("C2", 8192, [60] * 4), ("C3", 16384, [60] * 8)):
        q = obfv.generate_moduli(sizes, n)
        t = full_size.plaintext_modulus(n)
        nbits = sum(m.bit_length() for m in q)
        ext = obfv.extended_basis_primes(n, q, -(-(nbits + 60) // 62))
        base, mul, pt = OCtx(q, n), OCtx(q + ext, n), OCtx([t], n)
        out[name] = dict(n=n, moduli=q, ext=ext, plaintext=t,
                         extender=scaler_block(OScaler(base, mul, ScalingFactor.one())),
                         down_scaler=scaler_block(OScaler(mul, base, ScalingFactor(t, base.modulus()))),
                         decrypt_scaler=scaler_block(OScaler(base, pt, ScalingFactor(t, base.modulus()))))
    return out


def decrypt_example(n=1024, nmod=2, seed=77):
    rng = random.Random(seed)
    opar = obfv.BfvParameters.default_arc(nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    va, vb = [rng.randrange(t) for _ in range(n)], [rng.randrange(t) for _ in range(n)]
    A, B = sk.encrypt(va, rng, 0), sk.encrypt(vb, rng, 0)
    ork = obfv.RelinearizationKey(sk, rng, 0, 0)
    res = obfv.Multiplicator.default(ork).multiply(A, B)
    want = obfv_negacyclic(va, vb, t, n)
    assert sk.decrypt(res) == want
    u = lambda x: np.array(x, dtype=np.uint64)
    arrays = dict(secret_key=np.array(sk.coeffs, dtype=np.int64), lhs=u(rows(A.c)), rhs=u(rows(B.c)),
                  relin_c0=u(rows(ork.ksk.c0)), relin_c1=u(rows(ork.ksk.c1)), product=u(rows(res.c)),
                  plain_lhs=u(va), plain_rhs=u(vb), plain_product=u(want))
    meta = dict(n=n, moduli=opar.moduli, plaintext=t, psi=[op.psi for op in opar.ctx[0].ops],
                product_sha256=sha(arrays["product"]), noise_bits=int(sk.measure_noise(res, want)))
    return arrays, meta


def default128_digest():
    import ref_params
    from helpers import load_engine
    fhe = load_engine("emu")
    out = {}
    for n in (4096, 8192):
        d = {}
        ref_params.check_all(fhe, False, n, batch=1, digest=d)
        out[str(n)] = dict(tables=ref_params.table_digest(n), log_q=ref_params.log_q(n), outputs=d)
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def main():
    dump("bfv_small_traces.json", {f"L{k}": small_trace(k) for k in (1, 2, 3, 6)})
    dump("scaler_constants.json", scaler_constants())
    dump("c2_digest.json", c2_digest())
    dump("default128_digest.json", default128_digest())
    arrays, meta = decrypt_example()
    np.savez_compressed(os.path.join(HERE, "decrypt_n1024.npz"), **arrays)
    dump("decrypt_n1024.json", meta)


if __name__ == "__main__":
    main()
