"""Pins oracle rq and BFV layers against the reference's tests:
crates/fhe-math/src/rq/scaler.rs:154-204 (Scaler closed form, PowerBasis and Ntt),
crates/fhe-math/src/rq/mod.rs:947-1122 (substitute x->x^3 formula, switch_down,
switch_down_to, switch), crates/fhe-math/src/rq/ops.rs:587-939 (ops vs Modulus),
crates/fhe/src/bfv/ops/mul.rs:263-418, ops/mod.rs:656-731 (mul / square decrypt),
keys/key_switching_key.rs:532-633, keys/relinearization_key.rs:171-273,
keys/galois_key.rs:186-256."""
import random

import pytest

from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import (Context, Poly, Scaler, Switcher, SubstitutionExponent,
                           POWER_BASIS, NTT, NTT_SHOUP)
from fhe_oracle import bfv

MODULI = [1153, 4611686018326724609, 4611686018309947393, 4611686018232352769,
          4611686018171535361]
Q = [4611686018282684417, 4611686018326724609, 4611686018309947393]
P = [4611686018282684417, 4611686018309947393, 4611686018257518593]


def rand_poly(ctx, rng, rep=POWER_BASIS):
    return bfv.random_poly(ctx, rep, rng)


def test_rq_moduli_constant_matches_reference():
    """rq/mod.rs:687-693."""
    import re, os
    src = "/root/reference/crates/fhe-math/src/rq/mod.rs"
    if os.path.exists(src):
        text = open(src).read()
        m = re.search(r"const MODULI: &\[u64; 5\] = &\[(.*?)\];", text, re.S)
        assert [int(x) for x in re.findall(r"\d+", m.group(1))] == MODULI


@pytest.mark.parametrize("n,d", [(n, d) for n in (1, 2, 3, 100, 1000, 4611686018326724610)
                                 for d in (1, 2, 3, 4, 100, 101, 1000, 1001, 4611686018326724610)])
def test_rq_scaler_closed_form(n, d):
    """rq/scaler.rs:154-204."""
    frm, to = Context(Q, 16), Context(P, 16)
    rng = random.Random(n * 31 + d)
    scaler = Scaler(frm, to, ScalingFactor(n, d))
    for _ in range(4):
        poly = rand_poly(frm, rng)
        expected = []
        for i in poly.to_biguints():
            if i >= (frm.modulus() >> 1):
                if d & 1 == 0:
                    e = to.modulus() - (((frm.modulus() - i) * n + ((d >> 1) - 1)) // d) % to.modulus()
                else:
                    e = to.modulus() - (((frm.modulus() - i) * n + (d >> 1)) // d) % to.modulus()
            else:
                e = ((i * n + (d >> 1)) // d) % to.modulus()
            expected.append(e)
        assert scaler.scale(poly).to_biguints() == expected
        assert scaler.scale(poly.into_ntt()).into_power_basis().to_biguints() == expected


def test_scaler_common_prefix():
    """rq/scaler.rs:35-43: shared prefix only when the factor is one."""
    frm, to = Context(Q[:2], 16), Context(Q[:2] + P[1:], 16)
    assert Scaler(frm, to, ScalingFactor.one()).number_common_moduli == 2
    assert Scaler(frm, to, ScalingFactor(2, 1)).number_common_moduli == 0
    assert Scaler(frm, to, ScalingFactor(7, 7)).number_common_moduli == 2
    assert Scaler(Context(Q, 16), Context(P, 16), ScalingFactor.one()).number_common_moduli == 1
    with pytest.raises(ValueError):
        Scaler(Context(Q, 16), Context(P, 32), ScalingFactor.one())


def test_substitute():
    """rq/mod.rs:947-1036."""
    rng = random.Random(5)
    for modulus in MODULI:
        ctx = Context([modulus], 16)
        p = rand_poly(ctx, rng)
        p_ntt, p_shoup = p.into_ntt(), p.into_ntt_shoup()
        for bad in (0, 2, 16):
            with pytest.raises(ValueError):
                SubstitutionExponent(ctx, bad)
        one = SubstitutionExponent(ctx, 1)
        assert p.substitute(one) == p and p_ntt.substitute(one) == p_ntt
        assert p_shoup.substitute(one) == p_shoup
        e3, e11 = SubstitutionExponent(ctx, 3), SubstitutionExponent(ctx, 11)
        q = p.substitute(e3)
        v = [0] * 16
        c = p.coefficients[0]
        for i in range(16):
            v[(3 * i) % 16] = (modulus - c[i]) if ((3 * i) // 16) & 1 == 1 and c[i] > 0 else c[i]
        assert q.coefficients[0] == v
        assert p_ntt.substitute(e3) == q.into_ntt()
        qs = p_shoup.substitute(e3)
        assert qs == q.into_ntt_shoup() and qs.coefficients_shoup == q.into_ntt_shoup().coefficients_shoup
        assert p.substitute(e3).substitute(e11) == p
        assert p_ntt.substitute(e3).substitute(e11) == p_ntt
    ctx = Context(MODULI, 16)
    p = rand_poly(ctx, rng)
    e3, e11 = SubstitutionExponent(ctx, 3), SubstitutionExponent(ctx, 11)
    assert p.substitute(e3).substitute(e11) == p
    assert p.into_ntt().substitute(e3).substitute(e11) == p.into_ntt()


def test_switch_down_chain():
    """rq/mod.rs:1039-1073."""
    rng = random.Random(6)
    ctx = Context(MODULI, 16)
    for _ in range(25):
        p = rand_poly(ctx, rng)
        reference = p.to_biguints()
        cur = ctx
        while cur.next_context is not None:
            den = cur.modulus()
            cur = cur.next_context
            num = cur.modulus()
            p = p.switch_down()
            assert p.ctx == cur
            got = p.to_biguints()
            assert got == [((b * num + (den >> 1)) // den) % cur.modulus() for b in reference]
            reference = got
        with pytest.raises(ValueError):
            p.switch_down()


def test_switch_down_to_and_switch():
    """rq/mod.rs:1075-1122."""
    rng = random.Random(7)
    ctx1, ctx2 = Context(MODULI, 16), Context(MODULI[:2], 16)
    for _ in range(25):
        p = rand_poly(ctx1, rng)
        ref = p.to_biguints()
        q = p.switch_down_to(ctx2)
        assert q.to_biguints() == [(b * ctx2.modulus() + (ctx1.modulus() >> 1)) // ctx1.modulus()
                                   for b in ref]
    with pytest.raises(ValueError):
        rand_poly(ctx2, rng).switch_down_to(ctx1)
    a, b = Context(MODULI[:2], 16), Context(MODULI[3:], 16)
    sw = Switcher(a, b)
    for _ in range(25):
        p = rand_poly(a, rng)
        ref = p.to_biguints()
        assert p.switch(sw).to_biguints() == [(x * b.modulus() + (a.modulus() >> 1)) // a.modulus()
                                              for x in ref]


def test_rq_ops_match_plain_arithmetic():
    """rq/ops.rs:587-939."""
    rng = random.Random(8)
    ctx = Context(MODULI, 16)
    for rep in (POWER_BASIS, NTT):
        p, q = rand_poly(ctx, rng, rep), rand_poly(ctx, rng, rep)
        for row_r, row_p, row_q, m in zip(p.add(q).coefficients, p.coefficients, q.coefficients, MODULI):
            assert row_r == [(x + y) % m for x, y in zip(row_p, row_q)]
        for row_r, row_p, row_q, m in zip(p.sub(q).coefficients, p.coefficients, q.coefficients, MODULI):
            assert row_r == [(x - y) % m for x, y in zip(row_p, row_q)]
        for row_r, row_p, m in zip(p.neg().coefficients, p.coefficients, MODULI):
            assert row_r == [(-x) % m for x in row_p]
    p, q = rand_poly(ctx, rng, NTT), rand_poly(ctx, rng, NTT)
    expect = [[x * y % m for x, y in zip(rp, rq)] for rp, rq, m in zip(p.coefficients, q.coefficients, MODULI)]
    assert p.mul(q).coefficients == expect
    assert p.mul(q.into_ntt_shoup()).coefficients == expect
    # negacyclic product through the NTT equals schoolbook (ops.rs mul tests)
    a, b = rand_poly(ctx, rng), rand_poly(ctx, rng)
    prod = a.into_ntt().mul(b.into_ntt()).into_power_basis()
    for row, ra, rb, m in zip(prod.coefficients, a.coefficients, b.coefficients, MODULI):
        exp = [0] * 16
        for i in range(16):
            for j in range(16):
                k = i + j
                if k < 16:
                    exp[k] = (exp[k] + ra[i] * rb[j]) % m
                else:
                    exp[k - 16] = (exp[k - 16] - ra[i] * rb[j]) % m
        assert row == exp


def negacyclic(a, b, t):
    n = len(a)
    exp = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                exp[k] = (exp[k] + a[i] * b[j]) % t
            else:
                exp[k - n] = (exp[k - n] - a[i] * b[j]) % t
    return exp


@pytest.mark.parametrize("nmoduli", [2, 3, 6])
def test_bfv_mul_relin_decrypts(nmoduli):
    """ops/mul.rs:263-300 (`mul`), ops/mod.rs:656-731 (`mul`, `square`),
    relinearization_key.rs:171-216."""
    rng = random.Random(nmoduli)
    par = bfv.BfvParameters.default_arc(nmoduli, 16)
    t = par.plaintext
    sk = bfv.SecretKey.random(par, rng)
    rk = bfv.RelinearizationKey(sk, rng)
    m = bfv.Multiplicator.default(rk)
    for _ in range(3):
        a = [rng.randrange(t) for _ in range(16)]
        b = [rng.randrange(t) for _ in range(16)]
        ca, cb = sk.encrypt(a, rng), sk.encrypt(b, rng)
        assert sk.decrypt(ca) == a
        exp = negacyclic(a, b, t)
        cc = m.multiply(ca, cb)
        assert len(cc) == 2 and sk.decrypt(cc) == exp
        c3 = ca.mul(cb)
        assert len(c3) == 3 and sk.decrypt(c3) == exp
        rk.relinearizes(c3)
        assert len(c3) == 2 and sk.decrypt(c3) == exp
        sq = ca.mul(ca)
        assert sk.decrypt(sq) == negacyclic(a, a, t)


def test_bfv_mul_at_level_and_mod_switch():
    """ops/mul.rs:302-367 (`mul_at_level`, mod switching)."""
    rng = random.Random(11)
    par = bfv.BfvParameters.default_arc(3, 16)
    t = par.plaintext
    sk = bfv.SecretKey.random(par, rng)
    for level in (0, 1):
        rk = bfv.RelinearizationKey(sk, rng, level, level)
        m = bfv.Multiplicator.default(rk)
        if level < par.max_level():
            m.enable_mod_switching()
        a = [rng.randrange(t) for _ in range(16)]
        b = [rng.randrange(t) for _ in range(16)]
        ca, cb = sk.encrypt(a, rng, level), sk.encrypt(b, rng, level)
        cc = m.multiply(ca, cb)
        assert cc.level == level + 1
        assert sk.decrypt(cc) == negacyclic(a, b, t)
    m_last = bfv.Multiplicator.default(bfv.RelinearizationKey(sk, rng, 1, 1))
    with pytest.raises(ValueError):
        bfv.RelinearizationKey(sk, rng, 2, 2)  # one modulus: KeySwitchingNotSupported


def test_key_switch_identity_and_levels():
    """key_switching_key.rs:532-561: ||c0 + c1*s - p*from||_inf small;
    relinearization_key.rs:219-273: all (ct level, key level) pairs."""
    rng = random.Random(12)
    par = bfv.BfvParameters.default_arc(4, 16)
    sk = bfv.SecretKey.random(par, rng)
    ctx = par.context_at_level(0)
    frm = bfv.random_poly(ctx, POWER_BASIS, rng)
    ksk = bfv.KeySwitchingKey(sk, frm, 0, 0, rng)
    s = Poly.from_i64(ctx, sk.coeffs).into_ntt()
    inp = bfv.random_poly(ctx, POWER_BASIS, rng)
    c0, c1 = ksk.key_switch(inp)
    c2 = c0.add(c1.mul(s)).sub(inp.into_ntt().mul(frm.into_ntt())).into_power_basis()
    q = ctx.modulus()
    assert max(min(x.bit_length(), (q - x).bit_length()) for x in c2.to_biguints()) <= 70
    t = par.plaintext
    for ct_level in range(3):
        for key_level in range(ct_level + 1):
            rk = bfv.RelinearizationKey(sk, rng, ct_level, key_level)
            a = [rng.randrange(t) for _ in range(16)]
            b = [rng.randrange(t) for _ in range(16)]
            ca, cb = sk.encrypt(a, rng, ct_level), sk.encrypt(b, rng, ct_level)
            c3 = ca.mul(cb)
            rk.relinearizes(c3)
            assert sk.decrypt(c3) == negacyclic(a, b, t)


def test_key_switch_decomposition():
    """key_switching_key.rs:600-633 (single-modulus key level)."""
    rng = random.Random(13)
    par = bfv.BfvParameters.default_arc(3, 16)
    sk = bfv.SecretKey.random(par, rng)
    ctx = par.context_at_level(2)
    frm = bfv.random_poly(ctx, POWER_BASIS, rng)
    ksk = bfv.KeySwitchingKey(sk, frm, 2, 2, rng)
    assert ksk.log_base == 31 and len(ksk.c0) == 2
    s = Poly.from_i64(ctx, sk.coeffs).into_ntt()
    inp = bfv.random_poly(ctx, POWER_BASIS, rng)
    c0, c1 = ksk.key_switch(inp)
    c2 = c0.add(c1.mul(s)).sub(inp.into_ntt().mul(frm.into_ntt())).into_power_basis()
    q = ctx.modulus()
    assert max(min(x.bit_length(), (q - x).bit_length()) for x in c2.to_biguints()) <= 45


def test_galois_rotation():
    """galois_key.rs:186-233: with SIMD slot order, x->x^3 rotates columns
    left by one and 2N-1 swaps rows; checked here on the polynomial level:
    decrypt(relinearize(ct)) == substitute(plaintext)."""
    rng = random.Random(14)
    n = 16
    par = bfv.BfvParameters.default_arc(3, n)
    t = par.plaintext
    sk = bfv.SecretKey.random(par, rng)
    for exponent in (3, 9, 2 * n - 1, bfv.rot_to_gk_exponent(n, 5)):
        for ct_level, key_level in ((0, 0), (1, 0), (1, 1)):
            gk = bfv.GaloisKey(sk, exponent, ct_level, key_level, rng)
            a = [rng.randrange(t) for _ in range(n)]
            ca = sk.encrypt(a, rng, ct_level)
            out = gk.relinearize(ca)
            exp = [0] * n
            for i, c in enumerate(a):
                k = (i * exponent) % (2 * n)
                if k < n:
                    exp[k] = (exp[k] + c) % t
                else:
                    exp[k - n] = (exp[k - n] - c) % t
            assert sk.decrypt(out) == exp
