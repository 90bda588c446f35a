"""Pins oracle/fhe_oracle/zq.py against the reference's own tests:
crates/fhe-math/src/zq/mod.rs:823-1160 (closed forms vs plain % arithmetic),
crates/fhe-math/src/zq/primes.rs:67-123 (NFLlib KATs, edge cases) and
crates/fhe/src/bfv/parameters.rs:840-870 (generate_moduli KATs)."""
import random

import pytest

from fhe_oracle.zq import Modulus, generate_prime, supports_opt, is_prime
from fhe_oracle.bfv import generate_moduli

NFL_62 = [
    4611686018326724609, 4611686018309947393, 4611686018282684417, 4611686018257518593,
    4611686018232352769, 4611686018171535361, 4611686018106523649, 4611686018058289153,
    4611686018051997697, 4611686017974403073, 4611686017812922369, 4611686017781465089,
    4611686017773076481, 4611686017678704641, 4611686017666121729, 4611686017647247361,
    4611686017590624257, 4611686017554972673, 4611686017529806849, 4611686017517223937,
]


def test_nfl_62bit_primes():
    """primes.rs:67-101."""
    generated, upper = [], ((1 << 64) - 1) >> 2
    while len(generated) != 20:
        p = generate_prime(62, 2 * 1048576, upper)
        assert p is not None
        upper = p
        generated.append(p)
    assert generated == NFL_62


def test_prime_edge_cases():
    """primes.rs:103-122."""
    with pytest.raises(AssertionError):
        generate_prime(62, 2 * 1048576, (1 << 62) + 1)
    assert generate_prime(10, 2048, 1 << 10) is None
    assert generate_prime(11, 16, 1033) is None


def test_generate_moduli_kats():
    """parameters.rs:840-870."""
    assert generate_moduli([62, 62, 62, 61, 60, 11], 16) == [
        4611686018427387617, 4611686018427387329, 4611686018427387073,
        2305843009213693921, 1152921504606845473, 2017]


def test_is_prime_small():
    """fhe-util/src/lib.rs:252-267 style."""
    primes = [2, 3, 5, 7, 11, 13, 1153, 4611686018326724609]
    composites = [0, 1, 4, 9, 15, 1155, 4611686018326724607, 3215031751, 341550071728321]
    assert all(is_prime(p) for p in primes)
    assert not any(is_prime(c) for c in composites)


MODS = [2, 3, 17, 1987, 4611686018326724609, 1152921504606830593, (1 << 62) - 1, 1 << 61,
        4611686018427322369, 1125899906826241]


@pytest.mark.parametrize("p", MODS)
def test_scalar_ops_closed_form(p):
    """zq/mod.rs:823-960: add/sub/mul/neg/shoup/reduce vs % arithmetic (the
    asserts inside the oracle check the lazy-range invariants on every call)."""
    rng = random.Random(p)
    q = Modulus(p)
    for _ in range(300):
        a, b = rng.randrange(p), rng.randrange(p)
        assert q.add(a, b) == (a + b) % p
        assert q.sub(a, b) == (a - b) % p
        assert q.mul(a, b) == (a * b) % p
        assert q.neg(a) == (-a) % p
        bs = q.shoup(b)
        assert q.mul_shoup(a, b, bs) == (a * b) % p
        x64 = rng.getrandbits(64)
        assert q.lazy_mul_shoup(x64, b, bs) % p == (x64 * b) % p
        assert q.reduce(x64) == x64 % p
        x128 = rng.getrandbits(128)
        assert q.reduce_u128(x128) == x128 % p
        i64 = rng.randrange(-(1 << 63), 1 << 63)
        assert q.reduce_i64(i64) == i64 % p
        if q.supports_opt:
            assert q.mul_opt(a, b) == (a * b) % p
            assert q.reduce_opt(x64) == x64 % p
        assert q.lazy_reduce_vec([x64])[0] % p == x64 % p
        assert q.lazy_reduce_vec([x64])[0] < 2 * p


def test_invalid_modulus():
    """zq/mod.rs:83-85."""
    for p in (0, 1, 1 << 62, (1 << 64) - 1):
        with pytest.raises(ValueError):
            Modulus(p)


def test_inv_pow():
    """zq/mod.rs:1164-1190."""
    for p in (3, 1153, 4611686018326724609):
        q = Modulus(p)
        assert q.inv(0) is None
        rng = random.Random(p)
        for _ in range(50):
            a = rng.randrange(1, p)
            assert q.mul(a, q.inv(a)) == 1
            n = rng.randrange(p)
            assert q.pow(a, n) == pow(a, n, p)
    assert Modulus(4).inv(3) is None  # not prime


def test_supports_opt_configs():
    """All moduli of configs C1..C5 take the mul_opt branch."""
    for p in (1125899906826241, 1152921504606830593, 4611686018427322369, 1152921504578666497):
        assert supports_opt(p)
    assert not supports_opt((1 << 63) + 1)


def test_center():
    q = Modulus(7)
    assert [q.center(a) for a in range(7)] == [0, 1, 2, -4, -3, -2, -1]
