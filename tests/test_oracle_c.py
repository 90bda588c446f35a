"""Checks the plain-C oracle (oracle/c/fhe_oracle.c, also bench.py's CPU
baseline) bit-for-bit against the Python big-int oracle, which is itself pinned
by the reference's KATs/closed forms (tests/test_oracle_*.py)."""
import random

import numpy as np
import pytest

from fhe_oracle import bfv, coracle
from fhe_oracle.zq import Modulus, supports_opt
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import Context, Poly, Scaler, SubstitutionExponent, POWER_BASIS, NTT

Q = [4611686018282684417, 4611686018326724609, 4611686018309947393]
P = [4611686018282684417, 4611686018309947393, 4611686018257518593]


def test_c_modulus_primitives():
    rng = random.Random(1)
    L = coracle.lib()
    for p in (2, 3, 4, 1153, 4611686018326724609, 1152921504606830593, (1 << 62) - 1, 1 << 61,
              1125899906826241, 4611686018427322369):
        assert bool(L.orc_supports_opt(p)) == supports_opt(p)
        for _ in range(200):
            a, b = rng.randrange(p), rng.randrange(p)
            assert L.orc_mod_mul(p, a, b) == a * b % p
            x = rng.getrandbits(128)
            assert L.orc_reduce_u128(p, x & ((1 << 64) - 1), x >> 64) == x % p


@pytest.mark.parametrize("n", [16, 256])
def test_c_ntt_rows(n):
    ctx = Context(Q, n)
    cc = coracle.CCtx(ctx)
    rng = random.Random(n)
    for mi, m in enumerate(Q):
        a = [rng.randrange(m) for _ in range(n)]
        assert list(cc.ntt_forward_row(mi, a)) == ctx.ops[mi].forward(a)
        assert list(cc.ntt_forward_row(mi, a, lazy=True)) == ctx.ops[mi].forward_lazy(a)
        assert list(cc.ntt_backward_row(mi, ctx.ops[mi].forward(a))) == a


def test_c_ntt_row_8192():
    """One full-size row (C2 shape) against the Python butterflies."""
    n, m = 8192, 1152921504606830593
    ctx = Context([m], n)
    cc = coracle.CCtx(ctx)
    rng = random.Random(8192)
    a = [rng.randrange(m) for _ in range(n)]
    f = ctx.ops[0].forward(a)
    assert list(cc.ntt_forward_row(0, a)) == f
    assert list(cc.ntt_backward_row(0, f)) == a


def test_c_poly_ops_and_switch_down_substitute():
    n = 32
    mods = [1153, 4611686018326724609, 4611686018309947393, 4611686018232352769]
    ctx = Context(mods, n)
    cc = coracle.CCtx(ctx)
    rng = random.Random(5)
    a, b = bfv.random_poly(ctx, NTT, rng), bfv.random_poly(ctx, NTT, rng)
    assert cc.poly_add(a.coefficients, b.coefficients).tolist() == a.add(b).coefficients
    assert cc.poly_sub(a.coefficients, b.coefficients).tolist() == a.sub(b).coefficients
    assert cc.poly_mul(a.coefficients, b.coefficients).tolist() == a.mul(b).coefficients
    assert cc.poly_neg(a.coefficients).tolist() == a.neg().coefficients
    bs = b.into_ntt_shoup()
    assert cc.shoup(b.coefficients).tolist() == bs.coefficients_shoup
    assert cc.poly_mul_shoup(a.coefficients, bs.coefficients, bs.coefficients_shoup).tolist() == a.mul(bs).coefficients
    pb = bfv.random_poly(ctx, POWER_BASIS, rng)
    assert cc.poly_switch_down(pb.coefficients).tolist() == pb.switch_down().coefficients
    assert cc.poly_ntt_forward(pb.coefficients).tolist() == pb.into_ntt().coefficients
    assert cc.poly_ntt_backward(a.coefficients).tolist() == a.into_power_basis().coefficients
    for e in (3, 5, 2 * n - 1, 2 * n + 3):
        se = SubstitutionExponent(ctx, e)
        assert cc.poly_substitute(e, a.coefficients, True).tolist() == a.substitute(se).coefficients
        assert cc.poly_substitute(e, pb.coefficients, False).tolist() == pb.substitute(se).coefficients


@pytest.mark.parametrize("num,den", [(1, 1), (2, 1), (3, 2), (100, 101), (1000, 4),
                                     (4611686018326724610, 1001), (1, 4611686018326724610)])
def test_c_scaler(num, den):
    frm, to = Context(Q, 16), Context(P, 16)
    rng = random.Random(num + den)
    sc = Scaler(frm, to, ScalingFactor(num, den))
    csc = coracle.CScaler(sc)
    for _ in range(3):
        p = bfv.random_poly(frm, POWER_BASIS, rng)
        assert csc.scale(p.coefficients, False).tolist() == sc.scale(p).coefficients
        pn = p.into_ntt()
        assert csc.scale(pn.coefficients, True).tolist() == sc.scale(pn).coefficients
        col = [r[0] for r in p.coefficients]
        assert csc.rns_scale(col, 3).tolist() == sc.scaler.scale(col, 3, 0)
        assert csc.rns_scale(col, 2, 1).tolist() == sc.scaler.scale(col, 2, 1)


def test_c_scaler_tiny_moduli():
    """The reference's RNS scaler tests use non-prime/tiny moduli (scaler.rs:385)."""
    from fhe_oracle.rns import RnsContext, RnsScaler
    from fhe_oracle import rq

    class Fake:  # minimal rq.Scaler look-alike around an RnsScaler
        pass

    q = RnsContext([4, 4611686018326724609, 1153])
    r = RnsContext([4, 4611686018326724609, 1153, 4611686018309947393, 4611686018282684417])
    rng = random.Random(2)
    for n, d in ((1, 1), (3, 4), (1000, 101), (2, 4611686018326724610)):
        s = RnsScaler(q, r, ScalingFactor(n, d))
        f = Fake()
        f.scaler, f.number_common_moduli = s, 0
        f.frm, f.to = Fake(), Fake()
        csc = coracle.CScaler.__new__(coracle.CScaler)

        class Cx:
            pass
        cto = Cx()
        cto.c = coracle.OrcCtx()
        mod_arr = coracle.arr(r.moduli_u64)
        cto.c.moduli = coracle.ptr(mod_arr)
        cto.L = 5
        coracle.CScaler.__init__(csc, f, cfrom=Cx(), cto=cto)
        for _ in range(50):
            x = [rng.randrange(m) for m in q.moduli_u64]
            assert csc.rns_scale(x, 5).tolist() == s.scale(x, 5, 0)


@pytest.mark.parametrize("nmod,n,mod_switch", [(2, 16, False), (3, 16, True), (4, 64, False)])
def test_c_multiply_and_key_switch(nmod, n, mod_switch):
    rng = random.Random(nmod * 100 + n)
    par = bfv.BfvParameters.default_arc(nmod, n)
    sk = bfv.SecretKey.random(par, rng)
    rk = bfv.RelinearizationKey(sk, rng)
    m = bfv.Multiplicator.default(rk)
    if mod_switch:
        m.enable_mod_switching()
    cm = coracle.CMul.from_oracle(m)
    t = par.plaintext
    ca = sk.encrypt([rng.randrange(t) for _ in range(n)], rng)
    cb = sk.encrypt([rng.randrange(t) for _ in range(n)], rng)
    want = m.multiply(ca, cb)
    got = cm.multiply([p.coefficients for p in ca.c], [p.coefficients for p in cb.c])
    assert got.tolist() == [p.coefficients for p in want.c]
    # key switch alone + no-relin multiply
    c2 = bfv.random_poly(par.context_at_level(0), POWER_BASIS, rng)
    w0, w1 = rk.ksk.key_switch(c2)
    g0, g1 = cm.rk.key_switch(c2.coefficients)
    assert g0.tolist() == w0.coefficients and g1.tolist() == w1.coefficients
    m2 = bfv.Multiplicator(ScalingFactor.one(), ScalingFactor.one(), m.mul_ctx.moduli,
                           ScalingFactor(t, par.context_at_level(0).modulus()), par)
    cm2 = coracle.CMul.from_oracle(m2)
    got3 = cm2.multiply([p.coefficients for p in ca.c], [p.coefficients for p in cb.c])
    assert got3.tolist() == [p.coefficients for p in ca.mul(cb).c]


def test_c_galois():
    n = 32
    rng = random.Random(77)
    par = bfv.BfvParameters.default_arc(3, n)
    sk = bfv.SecretKey.random(par, rng)
    for e in (3, 2 * n - 1):
        gk = bfv.GaloisKey(sk, e, 0, 0, rng)
        ca = sk.encrypt([rng.randrange(par.plaintext) for _ in range(n)], rng)
        ck = coracle.CKsk.from_oracle(gk.ksk)
        got = ck.galois_relinearize(e, [p.coefficients for p in ca.c])
        assert got.tolist() == [p.coefficients for p in gk.relinearize(ca).c]


def test_c_synth_matches_python():
    from fhe_oracle import synth
    ctx = Context(Q, 16)
    cc = coracle.CCtx(ctx)
    assert cc.synth_poly(0xF4E50002, 5, 1).tolist() == synth.synth_rows(0xF4E50002, 5, 1, Q, 16)
