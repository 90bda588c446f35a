"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

CPU restatement of `fhe_math::ntt::NttOperator` (native backend),
reference: crates/fhe-math/src/ntt/native.rs.

Harvey lazy Cooley-Tukey forward / Gentleman-Sande inverse negacyclic NTT with
bit-reversed twiddle tables and Shoup companions, butterfly by butterfly as in
native.rs:77-300.  Result: NTT(a)[i] = a(psi^(2*bitrev(i)+1)).

PARITY UNPINNED (single point): the reference picks the primitive 2N-th root
psi with ChaCha8Rng::seed_from_u64(0).random_range(0..p) (native.rs:320-336;
crates rand 0.10.2 / rand_chacha 0.10.0, not vendored under /root/reference),
and no reference test fixes psi or any NTT-domain value.  This oracle picks
psi = g^((p-1)/2N) for the smallest g >= 2 that yields a primitive root.
Everything PowerBasis-level is psi independent; tables cross the C ABI as
inputs so a Rust host supplies its own.
"""

from .zq import Modulus, is_prime


def supports_ntt(p: int, n: int) -> bool:
    """ntt/mod.rs:18-22."""
    assert n >= 8 and (n & (n - 1)) == 0
    return p % (2 * n) == 1 and is_prime(p)


def bitrev(i: int, logn: int) -> int:
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def oracle_primitive_root(n: int, p: Modulus) -> int:
    """Deterministic stand-in for native.rs:320-336 (see module docstring)."""
    lam = (p.p - 1) // (2 * n)
    g = 2
    while True:
        root = pow(g, lam, p.p)
        if pow(root, 2 * n, p.p) == 1 and pow(root, n, p.p) != 1:
            return root
        g += 1


class NttOperator:
    """native.rs:16-73."""

    def __init__(self, p: Modulus, size: int, psi: int = None):
        if not supports_ntt(p.p, size):
            raise ValueError("NttOperatorUnavailable")
        self.p = p
        self.p_twice = 2 * p.p
        self.size = size
        self.logn = size.bit_length() - 1
        self.size_inv = p.inv(size)
        omega = oracle_primitive_root(size, p) if psi is None else psi
        assert pow(omega, 2 * size, p.p) == 1 and pow(omega, size, p.p) != 1
        self.psi = omega
        omega_inv = p.inv(omega)
        powers = [1] * size
        powers_inv = [omega_inv] * size
        for i in range(1, size):
            powers[i] = p.mul(powers[i - 1], omega)
            powers_inv[i] = p.mul(powers_inv[i - 1], omega_inv)
        # native.rs:51-56: omegas[i] = psi^bitrev(i), zetas_inv[i] = psi^-(bitrev(i)+1)
        self.omegas = [powers[bitrev(i, self.logn)] for i in range(size)]
        self.zetas_inv = [powers_inv[bitrev(i, self.logn)] for i in range(size)]
        self.omegas_shoup = p.shoup_vec(self.omegas)
        self.zetas_inv_shoup = p.shoup_vec(self.zetas_inv)
        self.size_inv_shoup = p.shoup(self.size_inv)

    # native.rs:238-246
    def reduce3(self, a):
        assert a < 4 * self.p.p
        return Modulus.reduce1(Modulus.reduce1(a, self.p_twice), self.p.p)

    # native.rs:256-269
    def _butterfly(self, x, y, w, ws):
        assert x < 4 * self.p.p and y < 4 * self.p.p
        x = Modulus.reduce1(x, self.p_twice)
        t = self.p.lazy_mul_shoup(y, w, ws)
        y = x + self.p_twice - t
        x = x + t
        assert x < 4 * self.p.p and y < 4 * self.p.p
        return x, y

    # native.rs:288-300
    def _inv_butterfly(self, x, y, z, zs):
        assert x < self.p_twice and y < self.p_twice
        t = x
        x = Modulus.reduce1(y + t, self.p_twice)
        y = self.p.lazy_mul_shoup(self.p_twice + t - y, z, zs)
        return x, y

    def forward_lazy(self, a):
        """native.rs:142-175 (forward_vt_lazy): outputs in [0, 4p)."""
        a = list(a)
        assert len(a) == self.size
        l = self.size >> 1
        m = 1
        k = 1
        while l > 0:
            for i in range(m):
                w = self.omegas[k]
                ws = self.omegas_shoup[k]
                k += 1
                s = 2 * i * l
                for j in range(s, s + l):
                    a[j], a[j + l] = self._butterfly(a[j], a[j + l], w, ws)
            l >>= 1
            m <<= 1
        return a

    def forward(self, a):
        """native.rs:77-102 / 183-189: canonical output (identical values for
        the constant-time and _vt variants)."""
        return [self.reduce3(x) for x in self.forward_lazy(a)]

    def backward(self, a):
        """native.rs:106-132 / 197-233: canonical output."""
        a = list(a)
        assert len(a) == self.size
        k = 0
        m = self.size >> 1
        l = 1
        while m > 0:
            for i in range(m):
                s = 2 * i * l
                z = self.zetas_inv[k]
                zs = self.zetas_inv_shoup[k]
                k += 1
                for j in range(s, s + l):
                    a[j], a[j + l] = self._inv_butterfly(a[j], a[j + l], z, zs)
            l <<= 1
            m >>= 1
        return [self.p.mul_shoup(x, self.size_inv, self.size_inv_shoup) for x in a]
