"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

CPU restatement of `fhe_math::zq::Modulus` and `fhe_math::zq::primes`
(reference: crates/fhe-math/src/zq/mod.rs, crates/fhe-math/src/zq/primes.rs),
written with Python integers.  Every lazy/Barrett/Shoup routine follows the
reference's word-level formula (not just `% p`), and asserts the reference's
debug invariants so that a transcription slip shows up as an AssertionError.

Parity: pinned by the reference's own closed-form tests (zq/mod.rs:823-1160,
restated in tests/test_oracle_zq.py) and the prime KATs (primes.rs:67-101).
"""

M64 = (1 << 64) - 1
M128 = (1 << 128) - 1


def is_prime(n: int) -> bool:
    """Exact primality for u64 (reference: fhe-util/src/lib.rs:16-18 uses
    num_bigint_dig::probably_prime, deterministic on u64).  Deterministic
    Miller-Rabin with the first 12 prime bases is exact below 3.3e24."""
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for q in small:
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x == 1 or x == n - 1:
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def leading_zeros64(x: int) -> int:
    return 64 - x.bit_length()


def supports_opt(p: int) -> bool:
    """primes.rs:10-24 (NFLlib eq. 1)."""
    lz = leading_zeros64(p)
    if lz < 1:
        return False
    middle = 1 << (3 * lz)
    left_side = (middle + 1) << 64
    middle *= (1 << lz) + 1
    middle *= p
    return left_side < middle


def generate_prime(num_bits: int, modulo: int, upper_bound: int):
    """primes.rs:30-59: largest `num_bits`-bit prime == 1 (mod modulo) that is
    < upper_bound, or None."""
    if not (10 <= num_bits <= 62):
        return None
    assert (1 << num_bits) >= upper_bound, "upper_bound larger than number of bits"
    lz = 64 - num_bits
    t = upper_bound - 1
    while t % modulo != 1 and leading_zeros64(t) == lz:
        t -= 1
    while leading_zeros64(t) == lz and not is_prime(t) and t >= modulo:
        t -= modulo
    if leading_zeros64(t) == lz and is_prime(t):
        return t
    return None


class Modulus:
    """zq/mod.rs:32-98.  p in [2, 2^62)."""

    __slots__ = ("p", "barrett_hi", "barrett_lo", "leading_zeros", "supports_opt")

    def __init__(self, p: int):
        if p < 2 or (p >> 62) != 0:
            raise ValueError(f"InvalidModulus({p})")
        barrett = (1 << 128) // p
        self.p = p
        self.barrett_hi = barrett >> 64
        self.barrett_lo = barrett & M64
        self.leading_zeros = leading_zeros64(p)
        self.supports_opt = supports_opt(p)

    def __eq__(self, other):
        return isinstance(other, Modulus) and other.p == self.p

    def __hash__(self):
        return hash(self.p)

    def __repr__(self):
        return f"Modulus({self.p})"

    # --- scalar ops (mod.rs:103-234) ---------------------------------------
    @staticmethod
    def reduce1(x: int, p: int) -> int:
        """mod.rs:659-668: x in [0, 2p) -> x mod p."""
        assert p >> 63 == 0 and x < 2 * p
        return x if x < p else x - p

    def add(self, a, b):
        assert a < self.p and b < self.p
        return self.reduce1(a + b, self.p)

    def sub(self, a, b):
        assert a < self.p and b < self.p
        return self.reduce1(a + self.p - b, self.p)

    def neg(self, a):
        assert a < self.p
        return self.reduce1(self.p - a, self.p)

    def mul(self, a, b):
        assert a < self.p and b < self.p
        return self.reduce_u128(a * b)

    def mul_opt(self, a, b):
        assert self.supports_opt and a < self.p and b < self.p
        return self.reduce_opt_u128(a * b)

    def shoup(self, a):
        """mod.rs:195-199."""
        assert a < self.p
        return ((a << 64) // self.p) & M64

    def lazy_mul_shoup(self, a, b, b_shoup):
        """mod.rs:224-234: any a < 2^64; result in [0, 2p)."""
        assert b < self.p and b_shoup == self.shoup(b) and 0 <= a <= M64
        q = (a * b_shoup) >> 64
        r = (a * b - q * self.p) & M64
        assert r < 2 * self.p
        return r

    def mul_shoup(self, a, b, b_shoup):
        return self.reduce1(self.lazy_mul_shoup(a, b, b_shoup), self.p)

    # --- reductions (mod.rs:594-752) ---------------------------------------
    def lazy_reduce_u128(self, a):
        """mod.rs:693-707."""
        assert 0 <= a <= M128
        a_lo = a & M64
        a_hi = a >> 64
        p_lo_lo = (a_lo * self.barrett_lo) >> 64
        p_hi_lo = a_hi * self.barrett_lo
        p_lo_hi = a_lo * self.barrett_hi
        q = (((p_lo_hi + p_hi_lo + p_lo_lo) & M128) >> 64) + a_hi * self.barrett_hi
        r = (a - q * self.p) & M64
        assert r < 2 * self.p and r % self.p == a % self.p
        return r

    def reduce_u128(self, a):
        return self.reduce1(self.lazy_reduce_u128(a), self.p)

    def lazy_reduce(self, a):
        """mod.rs:712-723."""
        assert 0 <= a <= M64
        p_lo_lo = (a * self.barrett_lo) >> 64
        p_lo_hi = a * self.barrett_hi
        q = (p_lo_hi + p_lo_lo) >> 64
        r = (a - q * self.p) & M64
        assert r < 2 * self.p and r % self.p == a % self.p
        return r

    def reduce(self, a):
        return self.reduce1(self.lazy_reduce(a), self.p)

    def lazy_reduce_opt_u128(self, a):
        """mod.rs:730-740."""
        assert a < self.p * self.p
        q = ((self.barrett_lo * (a >> 64) + ((a << self.leading_zeros) & M128)) & M128) >> 64
        r = (a - q * self.p) & M64
        assert r < 2 * self.p and r % self.p == a % self.p
        return r

    def reduce_opt_u128(self, a):
        assert self.supports_opt
        return self.reduce1(self.lazy_reduce_opt_u128(a), self.p)

    def lazy_reduce_opt(self, a):
        """mod.rs:744-752."""
        q = a >> (64 - self.leading_zeros)
        r = (a - q * self.p) & M64
        assert r < 2 * self.p and r % self.p == a % self.p
        return r

    def reduce_opt(self, a):
        return self.reduce1(self.lazy_reduce_opt(a), self.p)

    def lazy_reduce_vec(self, a):
        """mod.rs:756-762."""
        if self.supports_opt:
            return [self.lazy_reduce_opt(x) for x in a]
        return [self.lazy_reduce(x) for x in a]

    def reduce_i64(self, a):
        """mod.rs:479-481: reduce_u128((p << 64) + a) with a as i64 -> i128."""
        return self.reduce_u128(((self.p << 64) + a) & M128)

    def pow(self, a, n):
        """mod.rs:556-577 (square and multiply)."""
        assert a < self.p and n < self.p
        return pow(a, n, self.p)

    def inv(self, a):
        """mod.rs:582-591."""
        if not is_prime(self.p) or a == 0:
            return None
        r = self.pow(a, self.p - 2)
        assert self.mul(a, r) == 1
        return r

    def center(self, a):
        """mod.rs:445-456: a >= p>>1 maps to a - p."""
        assert a < self.p
        return a - self.p if a >= (self.p >> 1) else a

    # --- slice ops (mod.rs:240-545): identical values to the scalar ops ----
    def add_vec(self, a, b):
        return [self.add(x, y) for x, y in zip(a, b)]

    def sub_vec(self, a, b):
        return [self.sub(x, y) for x, y in zip(a, b)]

    def mul_vec(self, a, b):
        if self.supports_opt:
            return [self.mul_opt(x, y) for x, y in zip(a, b)]
        return [self.mul(x, y) for x, y in zip(a, b)]

    def neg_vec(self, a):
        return [self.neg(x) for x in a]

    def shoup_vec(self, a):
        return [self.shoup(x) for x in a]

    def mul_shoup_vec(self, a, b, b_shoup):
        return [self.mul_shoup(x, y, ys) for x, y, ys in zip(a, b, b_shoup)]

    def reduce_vec(self, a):
        return [self.reduce(x) for x in a]

    def scalar_mul_vec(self, a, b):
        bs = self.shoup(b)
        return [self.mul_shoup(x, b, bs) for x in a]
