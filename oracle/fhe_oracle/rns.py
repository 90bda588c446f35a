"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

CPU restatement of `fhe_math::rns::{RnsContext, ScalingFactor, RnsScaler}`
(reference: crates/fhe-math/src/rns/mod.rs, crates/fhe-math/src/rns/scaler.rs).

`RnsScaler.scale` reproduces the reference's fixed-point ALGORITHM (256-bit
wrap-around sums of 64x128-bit products, shifts, sign test on bit 191;
scaler.rs:249-352; third-party `ethnum 1.5.3` U256 = plain two's-complement
256-bit integers), not the mathematical rounding, because parity means "same
algorithm".  Pinned by the reference's closed-form tests scaler.rs:380-473
(restated in tests/test_oracle_rns.py) and the KATs rns/mod.rs:211-238.
"""

from math import gcd

from .zq import Modulus, M64

M128 = (1 << 128) - 1
M256 = (1 << 256) - 1


class RnsContext:
    """rns/mod.rs:23-149."""

    def __init__(self, moduli_u64):
        moduli_u64 = list(moduli_u64)
        if not moduli_u64:
            raise ValueError("EmptyModuli")
        for i, a in enumerate(moduli_u64):
            for j, b in enumerate(moduli_u64):
                if i != j and gcd(a, b) != 1:
                    raise ValueError(f"NonCoprimeModuli({a},{b})")
        product = 1
        for m in moduli_u64:
            product *= m
        self.moduli_u64 = moduli_u64
        self.moduli = [Modulus(m) for m in moduli_u64]
        self.product = product
        self.q_star = [product // m for m in moduli_u64]
        self.q_tilde = [pow(qs % m, -1, m) for qs, m in zip(self.q_star, moduli_u64)]
        self.q_tilde_shoup = [mod.shoup(qt) for mod, qt in zip(self.moduli, self.q_tilde)]
        # mod.rs:97: garner_i = q_star_i * q_tilde_i  (NOT reduced mod product)
        self.garner = [qs * qt for qs, qt in zip(self.q_star, self.q_tilde)]

    def __eq__(self, other):
        return isinstance(other, RnsContext) and other.moduli_u64 == self.moduli_u64

    def modulus(self):
        return self.product

    def project(self, a):
        return [a % m for m in self.moduli_u64]

    def lift(self, rests):
        assert len(rests) == len(self.moduli_u64)
        return sum(g * r for g, r in zip(self.garner, rests)) % self.product

    def get_garner(self, i):
        return self.garner[i] if 0 <= i < len(self.garner) else None


class ScalingFactor:
    """scaler.rs:18-47."""

    def __init__(self, numerator, denominator):
        assert denominator != 0
        self.numerator = numerator
        self.denominator = denominator
        self.is_one = numerator == denominator

    @staticmethod
    def one():
        return ScalingFactor(1, 1)


def _next_power_of_two_ilog2(x: int) -> int:
    return (x - 1).bit_length() if x > 1 else 0


class RnsScaler:
    """scaler.rs:52-175."""

    def __init__(self, frm: RnsContext, to: RnsContext, factor: ScalingFactor):
        self.frm = frm
        self.to = to
        self.scaling_factor = factor
        n, d = factor.numerator, factor.denominator

        gamma, self.theta_gamma_lo, self.theta_gamma_hi, self.theta_gamma_sign = \
            self._extract_projection_and_theta(to, frm.product, n, d, False)
        self.gamma = gamma
        self.gamma_shoup = [q.shoup(g) for g, q in zip(gamma, to.moduli)]

        nfrom, nto = len(frm.moduli), len(to.moduli)
        self.omega = [[0] * nfrom for _ in range(nto)]
        self.omega_shoup = [[0] * nfrom for _ in range(nto)]
        self.theta_omega_lo, self.theta_omega_hi, self.theta_omega_sign = [], [], []
        for i, garner_i in enumerate(frm.garner):
            omegas_i, lo, hi, sign = self._extract_projection_and_theta(to, garner_i, n, d, True)
            self.theta_omega_lo.append(lo)
            self.theta_omega_hi.append(hi)
            self.theta_omega_sign.append(sign)
            for j in range(nto):
                qj = to.moduli[j]
                self.omega[j][i] = qj.reduce(omegas_i[j])
                self.omega_shoup[j][i] = qj.shoup(self.omega[j][i])

        # scaler.rs:130-142: (shift + 1) + log(q * n) <= 192
        self.theta_garner_shift = min(
            min(192 - 1 - _next_power_of_two_ilog2(qi * nfrom) for qi in frm.moduli_u64), 127)
        self.theta_garner_lo, self.theta_garner_hi = [], []
        for garner_i in frm.garner:
            theta = ((garner_i << self.theta_garner_shift) + (frm.product >> 1)) // frm.product
            self.theta_garner_hi.append((theta >> 64))
            self.theta_garner_lo.append(theta & M64)
            assert (theta >> 64) <= M64

    @staticmethod
    def _extract_projection_and_theta(ctx, inp, numerator, denominator, round_up):
        """scaler.rs:183-229."""
        gamma = (numerator * inp + (denominator >> 1)) // denominator
        projected = ctx.project(gamma)
        theta = (numerator * inp) % denominator
        theta_sign = False
        if denominator > 1:
            if denominator & 1 == 1:
                if theta > (denominator >> 1):
                    theta_sign = True
                    theta = denominator - theta
            else:
                if theta >= (denominator >> 1):
                    theta_sign = True
                    theta = denominator - theta
        if round_up:
            if theta_sign:
                theta = (theta << 127) // denominator
            else:
                theta = ((theta << 127) + denominator - 1) // denominator
        elif theta_sign:
            theta = ((theta << 127) + denominator - 1) // denominator
        else:
            theta = (theta << 127) // denominator
        theta_hi = theta >> 64
        theta_lo = theta & M64
        assert theta_hi <= M64
        return projected, theta_lo, theta_hi, theta_sign

    def scale_vw(self, rests):
        """The (v, w, w_sign) fixed-point part of scaler.rs:260-313."""
        s = 0
        for lo, hi, ri in zip(self.theta_garner_lo, self.theta_garner_hi, rests):
            s = (s + ri * (lo | (hi << 64))) & M256
        s >>= self.theta_garner_shift - 1
        v = ((s & M128) + 1) // 2  # u128::div_ceil(2)

        w_sign = False
        w = 0
        if not self.scaling_factor.is_one:
            t = 0
            for lo, hi, sg, ri in zip(self.theta_omega_lo, self.theta_omega_hi,
                                      self.theta_omega_sign, rests):
                product = (ri * (lo | (hi << 64))) & M256
                t = (t - product) & M256 if sg else (t + product) & M256
            vtg = (v * (self.theta_gamma_lo | (self.theta_gamma_hi << 64))) & M256
            t = (t + vtg) & M256 if self.theta_gamma_sign else (t - vtg) & M256
            w_sign = (t >> 191) > 0
            if w_sign:
                w = ((((~t) & M256) >> 126) & M128) + 1
                w //= 2
            else:
                w = (t >> 126) & M128
                w = (w + 1) // 2
        return v, w, w_sign

    def scale(self, rests, size, starting_index=0):
        """scaler.rs:249-352.  Returns `size` residues for the target moduli
        starting_index .. starting_index+size."""
        rests = list(rests)
        assert len(rests) == len(self.frm.moduli_u64)
        assert size >= 1 and starting_index + size <= len(self.to.moduli_u64)
        v, w, w_sign = self.scale_vw(rests)
        out = []
        for i in range(size):
            qi = self.to.moduli[starting_index + i]
            omega_i = self.omega[starting_index + i]
            omega_shoup_i = self.omega_shoup[starting_index + i]
            gamma_i = self.gamma[starting_index + i]
            gamma_shoup_i = self.gamma_shoup[starting_index + i]
            yi = qi.p * 2 - qi.lazy_mul_shoup(qi.reduce_u128(v), gamma_i, gamma_shoup_i)
            if not self.scaling_factor.is_one:
                wi = qi.lazy_reduce_u128(w)
                yi += (qi.p * 2 - wi) if w_sign else wi
            for j in range(len(rests)):
                yi += qi.lazy_mul_shoup(rests[j], omega_i[j], omega_shoup_i[j])
            out.append(qi.reduce_u128(yi))
        return out

    def scale_new(self, rests, size):
        return self.scale(rests, size, 0)
