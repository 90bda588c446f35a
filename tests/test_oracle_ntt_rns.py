"""Pins oracle NTT and RNS layers against the reference's tests:
crates/fhe-math/src/ntt/mod.rs:32-112 (bijection, lazy == reduced),
crates/fhe-math/src/rns/mod.rs:195-250 (modulus, project/lift KATs),
crates/fhe-math/src/rns/scaler.rs:380-473 (closed-form rounding, same/different bases)."""
import random

import pytest

from fhe_oracle.zq import Modulus
from fhe_oracle.ntt import NttOperator, supports_ntt, bitrev
from fhe_oracle.rns import RnsContext, RnsScaler, ScalingFactor


@pytest.mark.parametrize("size", [8, 32, 1024])
@pytest.mark.parametrize("p", [1153, 4611686018326724609, 1152921504606830593 % (1 << 62)])
def test_ntt_bijection_and_lazy(size, p):
    """ntt/mod.rs:50-112."""
    if not supports_ntt(p, size):
        pytest.skip("modulus does not support this size")
    q = Modulus(p)
    op = NttOperator(q, size)
    rng = random.Random(size * 7 + p % 1000)
    a = [rng.randrange(p) for _ in range(size)]
    f = op.forward(a)
    assert op.backward(f) == a
    lazy = op.forward_lazy(a)
    assert all(x < 4 * p for x in lazy)
    assert [x % p for x in lazy] == f


def test_ntt_is_evaluation_at_odd_powers():
    """NTT(a)[i] = a(psi^(2*bitrev(i)+1)) -- SURVEY.md Appendix B.2; negacyclic
    convolution theorem."""
    p, n = 1153, 16
    q = Modulus(p)
    op = NttOperator(q, n)
    rng = random.Random(3)
    a = [rng.randrange(p) for _ in range(n)]
    b = [rng.randrange(p) for _ in range(n)]
    fa = op.forward(a)
    for i in range(n):
        x = pow(op.psi, 2 * bitrev(i, 4) + 1, p)
        assert fa[i] == sum(c * pow(x, k, p) for k, c in enumerate(a)) % p
    fb = op.forward(b)
    prod = op.backward([x * y % p for x, y in zip(fa, fb)])
    exp = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                exp[k] = (exp[k] + a[i] * b[j]) % p
            else:
                exp[k - n] = (exp[k - n] - a[i] * b[j]) % p
    assert prod == exp


def test_ntt_custom_psi_tables():
    """Tables are a pure function of psi (native.rs:41-56): any primitive
    root gives a consistent operator."""
    p, n = 4611686018326724609, 8
    q = Modulus(p)
    base = NttOperator(q, n)
    other = NttOperator(q, n, psi=pow(base.psi, 3, p))
    a = list(range(1, n + 1))
    assert other.backward(other.forward(a)) == a
    assert other.forward(a) != base.forward(a)


def test_rns_modulus_and_project_lift():
    """rns/mod.rs:195-250."""
    assert RnsContext([2]).modulus() == 2
    assert RnsContext([2, 5]).modulus() == 10
    rns = RnsContext([4, 15, 1153])
    product = 4 * 15 * 1153
    assert rns.modulus() == product
    for value, rests in ((0, [0, 0, 0]), (4, [0, 4, 4]), (15, [3, 0, 15]),
                         (1153, [1, 13, 0]), (product - 1, [3, 14, 1152])):
        assert rns.project(value) == rests
        assert rns.lift(rests) == value
    rng = random.Random(0)
    for _ in range(100):
        b = rng.randrange(product)
        assert rns.lift(rns.project(b)) == b
    with pytest.raises(ValueError):
        RnsContext([])
    with pytest.raises(ValueError):
        RnsContext([4, 6])


NUMS = [1, 2, 3, 100, 1000, 4611686018326724610]
DENS = [1, 2, 3, 4, 100, 101, 1000, 1001, 4611686018326724610]
Q3 = [4, 4611686018326724609, 1153]
R10 = Q3 + [4611686018309947393, 4611686018282684417, 4611686018257518593, 4611686018232352769,
            4611686018171535361, 4611686018106523649, 4611686018058289153]


def _expected(x_lift, x_sign, n, d, to_modulus):
    """scaler.rs:398-414 / 458-468."""
    if x_sign:
        if d % 2 == 0:
            return to_modulus - ((x_lift * n + ((d >> 1) - 1)) // d) % to_modulus
        return to_modulus - ((x_lift * n + (d >> 1)) // d) % to_modulus
    return (x_lift * n + (d >> 1)) // d


@pytest.mark.parametrize("to_moduli,ntests", [(Q3, 120), (R10, 40)])
def test_rns_scaler_closed_form(to_moduli, ntests):
    """scaler.rs:380-473 (scale_same_context / scale_different_contexts)."""
    q = RnsContext(Q3)
    r = RnsContext(to_moduli)
    rng = random.Random(len(to_moduli))
    for n in NUMS:
        for d in DENS:
            scaler = RnsScaler(q, r, ScalingFactor(n, d))
            for _ in range(ntests):
                x = [rng.getrandbits(64) % m for m in Q3]
                x_lift = q.lift(x)
                x_sign = x_lift >= (q.modulus() >> 1)
                if x_sign:
                    x_lift = q.modulus() - x_lift
                z = scaler.scale_new(x, len(to_moduli))
                assert z == r.project(_expected(x_lift, x_sign, n, d, r.modulus()))


def test_rns_scaler_starting_index_slices():
    """scaler.rs:249-258: scale(rests, out, starting_index) fills a slice of
    the target residues (used by rq::Scaler for the non-common rows)."""
    q = RnsContext(Q3)
    r = RnsContext(R10)
    scaler = RnsScaler(q, r, ScalingFactor.one())
    x = [3, 123456789, 77]
    full = scaler.scale_new(x, 10)
    assert scaler.scale(x, 7, 3) == full[3:]
    assert scaler.scale(x, 2, 4) == full[4:6]
    assert full[:3] == x  # factor one, shared prefix: extension keeps residues
