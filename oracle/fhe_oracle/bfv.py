"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

CPU restatement of the BFV layer of fhe.rs that sits on the hot path
(reference: crates/fhe/src/bfv/{parameters,ciphertext,ops/mod,ops/mul}.rs,
crates/fhe/src/bfv/keys/{secret_key,key_switching_key,relinearization_key,
galois_key,evaluation_key}.rs).  Key generation / encryption / decryption are
restated only so that tests can build *decryptable* ciphertexts and check the
functional identities the reference's tests check (ops/mul.rs:263-418,
key_switching_key.rs:532-633, galois_key.rs:186-256); the randomness source is
Python's `random.Random(seed)` (the reference draws from an unseeded OS rng, so
no reference test pins concrete ciphertext bytes).
"""

import random

from .zq import Modulus, generate_prime
from .rns import RnsContext, ScalingFactor
from .rq import (Context, Poly, Scaler, Switcher, SubstitutionExponent,
                 POWER_BASIS, NTT, NTT_SHOUP)


def generate_moduli(moduli_sizes, degree):
    """parameters.rs:391-431."""
    moduli = []
    for size in moduli_sizes:
        if size > 62 or size < 10:
            raise ValueError("InvalidModulusSize")
        upper_bound = 1 << size
        while True:
            prime = generate_prime(size, 2 * degree, upper_bound)
            if prime is None:
                raise ValueError("NotEnoughPrimes")
            if prime not in moduli:
                moduli.append(prime)
                break
            upper_bound = prime
    return moduli


def extended_basis_primes(degree, existing, count):
    """parameters.rs:660-676 == ops/mul.rs:109-125: 62-bit primes descending
    from 2^62, skipping members of `existing`."""
    ext = []
    upper_bound = 1 << 62
    while len(ext) != count:
        upper_bound = generate_prime(62, 2 * degree, upper_bound)
        if upper_bound is None:
            raise ValueError("NotEnoughPrimes")
        if upper_bound not in ext and upper_bound not in existing:
            ext.append(upper_bound)
    return ext


class MultiplicationParameters:
    """parameters.rs:793-814."""

    def __init__(self, frm, to, up_self_factor, down_factor):
        self.extender = Scaler(frm, to, up_self_factor)
        self.down_scaler = Scaler(to, frm, down_factor)
        self.frm = frm
        self.to = to


class BfvParameters:
    """parameters.rs:83-117, build() :560-738 (the parts the hot path needs)."""

    def __init__(self, degree, plaintext_modulus, moduli=None, moduli_sizes=None, variance=10):
        self.polynomial_degree = degree
        self.plaintext = plaintext_modulus
        self.variance = variance
        if moduli_sizes:
            moduli = generate_moduli(moduli_sizes, degree)
        self.moduli = list(moduli)
        self.moduli_sizes = [m.bit_length() for m in self.moduli]
        t = plaintext_modulus
        self.plaintext_mod = Modulus(t) if t < (1 << 62) else None

        # plaintext context: enough moduli for t (parameters.rs:583-598)
        t_bits = t.bit_length()
        acc, count = 0, 0
        for size in self.moduli_sizes:
            acc += size
            count += 1
            if acc >= t_bits + 60:
                break
        count = min(max(count, 1), len(self.moduli))
        top = Context(self.moduli, degree)
        self.ctx = [top.context_at_level(i) for i in range(len(self.moduli))]
        self.plaintext_context = Context(self.moduli[:count], degree)

        self.delta = []       # per level, Poly<NttShoup>
        self.q_mod_t = []
        self.plain_scaler = []
        for level, cipher_ctx in enumerate(self.ctx):
            level_moduli = cipher_ctx.moduli
            delta_rests = []
            for m in level_moduli:
                q = Modulus(m)
                inv = q.inv(q.neg(t % m))
                if inv is None:
                    raise ValueError("NonInvertible")
                delta_rests.append(inv)
            rns = RnsContext(level_moduli)
            delta = Poly.from_biguints(cipher_ctx, [rns.lift(delta_rests)]).into_ntt_shoup()
            self.delta.append(delta)
            self.q_mod_t.append(rns.modulus() % t)
            self.plain_scaler.append(
                Scaler(cipher_ctx, self.plaintext_context, ScalingFactor(t, rns.modulus())))
        self.plain_threshold = (t + 1) >> 1

        ext = extended_basis_primes(degree, self.moduli, len(self.moduli) + 1)
        self.extended_basis = ext
        self.mul_params = []
        for level, cipher_ctx in enumerate(self.ctx):
            nl = len(self.moduli) - level
            modulus_size = sum(self.moduli_sizes[:nl])
            n_moduli = -(-(modulus_size + 60) // 62)
            mul_moduli = self.moduli[:nl] + ext[:n_moduli]
            mul_ctx = Context(mul_moduli, degree)
            self.mul_params.append(MultiplicationParameters(
                cipher_ctx, mul_ctx, ScalingFactor.one(),
                ScalingFactor(t, cipher_ctx.modulus())))

    def degree(self):
        return self.polynomial_degree

    def max_level(self):
        return len(self.moduli) - 1

    def context_at_level(self, level):
        if level >= len(self.ctx):
            raise ValueError("InvalidLevel")
        return self.ctx[level]

    def level_of_context(self, ctx):
        for i, c in enumerate(self.ctx):
            if c == ctx:
                return i
        raise ValueError("ContextNotFound")

    @staticmethod
    def default_arc(num_moduli, degree):
        """parameters.rs:299-309 (test default: t=1153, 62-bit moduli)."""
        return BfvParameters(degree, 1153, moduli_sizes=[62] * num_moduli)


def sample_vec_cbd(size, variance, rng):
    """fhe-util/src/lib.rs:22-66 (centered binomial), rng.getrandbits(64) words."""
    assert 1 <= variance <= 32
    number_bits = 4 * variance
    mask_add = (((1 << 128) - 1) >> (128 - number_bits)) >> (2 * variance)
    mask_sub = mask_add << (2 * variance)

    def sample(pool):
        return bin(pool & mask_add).count("1") - bin(pool & mask_sub).count("1")

    out = []
    if number_bits <= 64:
        pool, nbits = 0, 0
        for _ in range(size):
            if nbits < number_bits:
                pool |= rng.getrandbits(64) << nbits
                nbits += 64
            out.append(sample(pool))
            pool >>= number_bits
            nbits -= number_bits
    else:
        for _ in range(size):
            pool = rng.getrandbits(64) | (rng.getrandbits(64) << 64)
            out.append(sample(pool))
    return out


def random_poly(ctx, rep, rng):
    rows = [[rng.randrange(m) for _ in range(ctx.degree)] for m in ctx.moduli]
    return Poly(ctx, rep, rows)


def small_poly(ctx, variance, rng, rep=POWER_BASIS):
    return Poly.from_i64(ctx, sample_vec_cbd(ctx.degree, variance, rng), rep)


class Ciphertext:
    """ciphertext.rs:18-30."""

    def __init__(self, par, c, level):
        self.par = par
        self.c = list(c)
        self.level = level

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        return self.c[i]

    def clone(self):
        return Ciphertext(self.par, [p.clone() for p in self.c], self.level)

    def switch_down(self):
        """ciphertext.rs:148-161."""
        if self.level >= self.par.max_level():
            raise ValueError("NoMoreContext")
        self.c = [ci.into_power_basis().switch_down().into_ntt() for ci in self.c]
        self.level += 1

    def switch_to_level(self, target):
        if target < self.level or target > self.par.max_level():
            raise ValueError("InvalidLevel")
        while self.level < target:
            self.switch_down()

    def add(self, other):
        """ops/mod.rs:15-70."""
        assert self.level == other.level and len(self) == len(other)
        return Ciphertext(self.par, [a.add(b) for a, b in zip(self.c, other.c)], self.level)

    def sub(self, other):
        assert self.level == other.level and len(self) == len(other)
        return Ciphertext(self.par, [a.sub(b) for a, b in zip(self.c, other.c)], self.level)

    def neg(self):
        return Ciphertext(self.par, [a.neg() for a in self.c], self.level)

    def mul(self, rhs):
        """ops/mod.rs:259-358: tensor without relinearisation, any #parts (the
        squaring branch computes the same values)."""
        assert self.level == rhs.level
        mp = self.par.mul_params[self.level]
        self_c = [ci.scale(mp.extender) for ci in self.c]
        other_c = [ci.scale(mp.extender) for ci in rhs.c]
        c = [Poly.zero(mp.to, NTT) for _ in range(len(self_c) + len(other_c) - 1)]
        for i in range(len(self_c)):
            for j in range(len(other_c)):
                c[i + j] = c[i + j].add(self_c[i].mul(other_c[j]))
        c = [ci.scale(mp.down_scaler) for ci in c]
        return Ciphertext(self.par, c, rhs.level)


class SecretKey:
    """keys/secret_key.rs."""

    def __init__(self, par, coeffs):
        self.par = par
        self.coeffs = list(coeffs)

    @staticmethod
    def random(par, rng):
        return SecretKey(par, sample_vec_cbd(par.degree(), par.variance, rng))

    def _s(self, ctx):
        return Poly.from_i64(ctx, self.coeffs).into_ntt()

    def encode_poly(self, values, level=0):
        """plaintext.rs:172-197 (`to_poly`, Encoding::poly): m*(q mod t) mod t,
        NTT, times delta."""
        par = self.par
        ctx = par.context_at_level(level)
        t = par.plaintext
        v = [(x % t) * par.q_mod_t[level] % t for x in values]
        m = Poly.from_u64(ctx, v).into_ntt()
        return m.mul(par.delta[level])

    def encrypt_poly(self, p: Poly, rng):
        """secret_key.rs:100-134."""
        level = self.par.level_of_context(p.ctx)
        s = self._s(p.ctx)
        a = random_poly(p.ctx, NTT, rng)
        a_s = a.mul(s)
        b = small_poly(p.ctx, self.par.variance, rng, NTT)
        b = b.sub(a_s).add(p)
        return Ciphertext(self.par, [b, a], level)

    def encrypt(self, values, rng, level=0):
        return self.encrypt_poly(self.encode_poly(values, level), rng)

    def phase(self, ct):
        s = self._s(ct[0].ctx)
        si = s
        c = ct[0]
        for i in range(1, len(ct)):
            c = c.add(ct[i].mul(si))
            if i + 1 < len(ct):
                si = si.mul(s)
        return c.into_power_basis()

    def decrypt(self, ct):
        """secret_key.rs:198-260 (small-plaintext path); returns poly-encoded
        plaintext coefficients mod t."""
        par = self.par
        c_pb = self.phase(ct)
        d = c_pb.scale(par.plain_scaler[ct.level])
        t = par.plaintext_mod
        v = [x + t.p for x in d.coefficients[0]]
        q = Modulus(par.moduli[0])
        w = q.reduce_vec(v)
        return t.reduce_vec(w)

    def measure_noise(self, ct, values):
        """secret_key.rs:57-98."""
        m = self.encode_poly(values, ct.level)
        s = self._s(ct[0].ctx)
        si = s
        c = ct[0]
        for i in range(1, len(ct)):
            c = c.add(ct[i].mul(si))
            si = si.mul(s)
        c = c.sub(m).into_power_basis()
        q = ct[0].ctx.modulus()
        return max(min(x.bit_length(), (q - x).bit_length()) for x in c.to_biguints())


class KeySwitchingKey:
    """keys/key_switching_key.rs:22-362."""

    def __init__(self, sk, frm: Poly, ciphertext_level, ksk_level, rng):
        par = sk.par
        self.par = par
        self.ctx_ksk = par.context_at_level(ksk_level)
        self.ctx_ciphertext = par.context_at_level(ciphertext_level)
        self.ciphertext_level = ciphertext_level
        self.ksk_level = ksk_level
        assert frm.ctx == self.ctx_ksk and frm.rep == POWER_BASIS
        if len(self.ctx_ksk.moduli) == 1:
            modulus = self.ctx_ksk.moduli[0]
            log_modulus = (modulus - 1).bit_length()
            self.log_base = log_modulus // 2
            size = -(-log_modulus // self.log_base)
            self.c1 = [random_poly(self.ctx_ksk, NTT_SHOUP, rng) for _ in range(size)]
            self.c0 = self._generate_c0(sk, frm, rng, decomposition=True)
        else:
            self.log_base = 0
            size = len(self.ctx_ciphertext.moduli)
            self.c1 = [random_poly(self.ctx_ksk, NTT_SHOUP, rng) for _ in range(size)]
            self.c0 = self._generate_c0(sk, frm, rng, decomposition=False)

    @staticmethod
    def from_parts(par, c0, c1, ciphertext_level, ksk_level, log_base=0):
        """Build from given NttShoup polys (synthetic keys for throughput
        tests: key-switch arithmetic does not care whether the key is 'real')."""
        k = KeySwitchingKey.__new__(KeySwitchingKey)
        k.par = par
        k.ctx_ksk = par.context_at_level(ksk_level)
        k.ctx_ciphertext = par.context_at_level(ciphertext_level)
        k.ciphertext_level = ciphertext_level
        k.ksk_level = ksk_level
        k.log_base = log_base
        k.c0, k.c1 = list(c0), list(c1)
        return k

    def _generate_c0(self, sk, frm, rng, decomposition):
        """key_switching_key.rs:149-238."""
        ctx = self.ctx_ksk
        s = Poly.from_i64(ctx, sk.coeffs).into_ntt()
        size = len(self.c1)
        rns = RnsContext(sk.par.moduli[:size]) if not decomposition else None
        c0 = []
        for i, c1i in enumerate(self.c1):
            a_s = c1i.as_ntt().mul(s).into_power_basis()
            b = small_poly(ctx, sk.par.variance, rng)
            b = b.sub(a_s)
            if decomposition:
                b = b.add(frm.mul_scalar(1 << (i * self.log_base)))
            else:
                b = b.add(frm.mul_scalar(rns.get_garner(i)))
            c0.append(b.into_ntt_shoup())
        return c0

    def key_switch(self, p: Poly):
        """key_switching_key.rs:241-270."""
        if self.log_base != 0:
            return self._key_switch_decomposition(p)
        if p.ctx != self.ctx_ciphertext:
            raise ValueError("ParameterMismatch")
        assert p.rep == POWER_BASIS
        c0 = Poly.zero(self.ctx_ksk, NTT)
        c1 = Poly.zero(self.ctx_ksk, NTT)
        for row, c0_i, c1_i in zip(p.coefficients, self.c0, self.c1):
            c2_i = Poly.create_constant_ntt_polynomial_with_lazy_coefficients(row, self.ctx_ksk)
            c0 = c0.add(c2_i.mul(c0_i))
            c1 = c1.add(c2_i.mul(c1_i))
        return c0, c1

    def _key_switch_decomposition(self, p: Poly):
        """key_switching_key.rs:323-362."""
        if p.ctx != self.ctx_ciphertext:
            raise ValueError("ParameterMismatch")
        log_modulus = (p.ctx.moduli[0] - 1).bit_length()
        coefficients = [x for row in p.coefficients for x in row]
        mask = (1 << self.log_base) - 1
        c2i = []
        for _ in range(-(-log_modulus // self.log_base)):
            c2i.append([c & mask for c in coefficients])
            coefficients = [c >> self.log_base for c in coefficients]
        c0 = Poly.zero(self.ctx_ksk, NTT)
        c1 = Poly.zero(self.ctx_ksk, NTT)
        for row, c0_i, c1_i in zip(c2i, self.c0, self.c1):
            c2_i = Poly.create_constant_ntt_polynomial_with_lazy_coefficients(row, self.ctx_ksk)
            c0 = c0.add(c2_i.mul(c0_i))
            c1 = c1.add(c2_i.mul(c1_i))
        return c0, c1


class RelinearizationKey:
    """keys/relinearization_key.rs:24-110."""

    def __init__(self, sk=None, rng=None, ciphertext_level=0, key_level=0, ksk=None):
        if ksk is not None:
            self.ksk = ksk
            return
        par = sk.par
        ctx_relin_key = par.context_at_level(key_level)
        ctx_ciphertext = par.context_at_level(ciphertext_level)
        if len(ctx_relin_key.moduli) == 1:
            raise ValueError("KeySwitchingNotSupported")
        s = Poly.from_i64(ctx_ciphertext, sk.coeffs).into_ntt()
        s2 = s.mul(s).into_power_basis()
        s2_up = s2.switch(Switcher(ctx_ciphertext, ctx_relin_key))
        self.ksk = KeySwitchingKey(sk, s2_up, ciphertext_level, key_level, rng)

    def relinearizes_poly(self, c2: Poly):
        return self.ksk.key_switch(c2)

    def relinearizes(self, ct: Ciphertext):
        """relinearization_key.rs:69-102 (in place)."""
        if len(ct) != 3:
            raise ValueError("InvalidPolynomialCount")
        if ct.level != self.ksk.ciphertext_level:
            raise ValueError("InvalidLevel")
        c2 = ct[2].into_power_basis()
        c0, c1 = self.relinearizes_poly(c2)
        if c0.ctx != ct[0].ctx:
            c0 = c0.into_power_basis().switch_down_to(ct[0].ctx).into_ntt()
            c1 = c1.into_power_basis().switch_down_to(ct[1].ctx).into_ntt()
        ct.c = [ct[0].add(c0), ct[1].add(c1)]


class GaloisKey:
    """keys/galois_key.rs:19-123."""

    def __init__(self, sk=None, exponent=None, ciphertext_level=0, galois_key_level=0,
                 rng=None, ksk=None, par=None):
        if ksk is not None:
            self.ksk = ksk
            self.element = SubstitutionExponent(par.context_at_level(ciphertext_level), exponent)
            return
        par = sk.par
        ctx_gk = par.context_at_level(galois_key_level)
        ctx_ct = par.context_at_level(ciphertext_level)
        self.element = SubstitutionExponent(ctx_ct, exponent)
        s = Poly.from_i64(ctx_ct, sk.coeffs)
        s_sub = s.substitute(self.element)
        s_sub_up = s_sub.switch(Switcher(ctx_ct, ctx_gk))
        self.ksk = KeySwitchingKey(sk, s_sub_up, ciphertext_level, galois_key_level, rng)

    def relinearize(self, ct: Ciphertext):
        """galois_key.rs:63-86 (== relinearize_into :89-123)."""
        assert len(ct) == 2 and ct.level == self.ksk.ciphertext_level
        c2 = ct[1].substitute(self.element).into_power_basis()
        c0, c1 = self.ksk.key_switch(c2)
        if c0.ctx != ct[0].ctx:
            c0 = c0.into_power_basis().switch_down_to(ct[0].ctx).into_ntt()
            c1 = c1.into_power_basis().switch_down_to(ct[1].ctx).into_ntt()
        c0 = c0.add(ct[0].substitute(self.element))
        return Ciphertext(ct.par, [c0, c1], self.ksk.ciphertext_level)


def rot_to_gk_exponent(degree, i):
    """evaluation_key.rs:278-286: column rotation by i <-> 3^i mod 2N; row
    rotation <-> 2N-1 (:118)."""
    return pow(3, i, 2 * degree)


class Multiplicator:
    """ops/mul.rs:28-243."""

    def __init__(self, lhs_factor, rhs_factor, extended_basis, post_mul_factor, par, level=0):
        self.par = par
        self.base_ctx = par.context_at_level(level)
        self.mul_ctx = Context(extended_basis, par.degree())
        self.extender_lhs = Scaler(self.base_ctx, self.mul_ctx, lhs_factor)
        self.extender_rhs = Scaler(self.base_ctx, self.mul_ctx, rhs_factor)
        self.down_scaler = Scaler(self.mul_ctx, self.base_ctx, post_mul_factor)
        self.rk = None
        self.mod_switch = False
        self.level = level

    @staticmethod
    def default_extended_basis(par, level):
        """ops/mul.rs:101-125."""
        ctx = par.context_at_level(level)
        nl = len(ctx.moduli)
        modulus_size = sum(par.moduli_sizes[:nl])
        n_moduli = -(-(modulus_size + 60) // 62)
        return list(ctx.moduli) + extended_basis_primes(par.degree(), ctx.moduli, n_moduli)

    @staticmethod
    def default(rk: RelinearizationKey):
        """ops/mul.rs:101-138."""
        par = rk.ksk.par
        level = rk.ksk.ciphertext_level
        ctx = par.context_at_level(level)
        basis = Multiplicator.default_extended_basis(par, level)
        m = Multiplicator(ScalingFactor.one(), ScalingFactor.one(), basis,
                          ScalingFactor(par.plaintext, ctx.modulus()), par, level)
        m.enable_relinearization(rk)
        return m

    def enable_relinearization(self, rk):
        if self.par.context_at_level(rk.ksk.ciphertext_level) != self.base_ctx:
            raise ValueError("ParameterMismatch")
        self.rk = rk

    def enable_mod_switching(self):
        if self.par.context_at_level(self.par.max_level()) == self.base_ctx:
            raise ValueError("NoMoreContext")
        self.mod_switch = True

    def multiply(self, lhs: Ciphertext, rhs: Ciphertext, trace=None):
        """ops/mul.rs:165-243."""
        if lhs.level != self.level or rhs.level != self.level:
            raise ValueError("InvalidLevel")
        if len(lhs) != 2 or len(rhs) != 2:
            raise ValueError("MultiplicationPolynomialCount")
        c00 = lhs[0].scale(self.extender_lhs)
        c01 = lhs[1].scale(self.extender_lhs)
        c10 = rhs[0].scale(self.extender_rhs)
        c11 = rhs[1].scale(self.extender_rhs)
        c0 = c00.mul(c10)
        c1 = c00.mul(c11).add(c01.mul(c10))
        c2 = c01.mul(c11)
        if trace is not None:
            trace["extended"] = [c00, c01, c10, c11]
            trace["tensor"] = [c0, c1, c2]
        c0 = c0.scale(self.down_scaler)
        c1 = c1.scale(self.down_scaler)
        c2 = c2.scale(self.down_scaler)
        c = [c0, c1, c2]
        if trace is not None:
            trace["scaled"] = list(c)
        if self.rk is not None:
            c2_pb = c[2].into_power_basis()
            c0r, c1r = self.rk.relinearizes_poly(c2_pb)
            if c0r.ctx != c[0].ctx:
                c0r = c0r.into_power_basis().switch_down_to(c[0].ctx).into_ntt()
                c1r = c1r.into_power_basis().switch_down_to(c[1].ctx).into_ntt()
            c = [c[0].add(c0r), c[1].add(c1r)]
        out = Ciphertext(self.par, c, self.level)
        if self.mod_switch:
            out.switch_down()
        return out


# --------------------------------------------------------------------------------------
# "next" rows of SURVEY.md 8(f): PIR server inner loop, RGSW external product, inner sum
# --------------------------------------------------------------------------------------
def poly_dot_product(ps, qs):
    """rq/ops.rs:449-570 (`dot_product`): sum_k p_k (.) q_k over Ntt polynomials; the u128
    lazy accumulation with periodic reduce_u128 yields the canonical sum of products."""
    ps, qs = list(ps), list(qs)
    if not ps or not qs:
        raise ValueError("EmptyDotProduct")
    if len(ps) != len(qs):
        raise ValueError("DotProductLengthMismatch")
    ctx = ps[0].ctx
    if any(p.ctx != ctx for p in ps) or any(q.ctx != ctx for q in qs):
        raise ValueError("PolynomialContextMismatch")
    rows = []
    for r, qi in enumerate(ctx.q):
        max_acc = 1 << (2 * qi.leading_zeros)
        acc = [0] * ctx.degree
        num_acc = 1
        for p, q in zip(ps, qs):
            pr, qr = p.coefficients[r], q.coefficients[r]
            acc = [a + x * y for a, x, y in zip(acc, pr, qr)]
            num_acc += 1
            if len(ps) > max_acc and num_acc == max_acc:
                acc = [qi.reduce_u128(a) for a in acc]
                num_acc = 1
        rows.append([qi.reduce_u128(a) for a in acc])
    return Poly(ctx, NTT, rows)


def dot_product_scalar(cts, pts):
    """ops/dot_product.rs:54-180: sum_k ct_k * pt_k with pt_k given as Ntt polynomials
    (`Plaintext::poly_ntt`)."""
    cts, pts = list(cts), list(pts)
    if not cts or not pts:
        raise ValueError("EmptyInput")
    if len(cts) != len(pts):
        raise ValueError("OperandCountMismatch")
    nparts = len(cts[0])
    if any(len(c) != nparts for c in cts):
        raise ValueError("CiphertextPolynomialCountMismatch")
    c = [poly_dot_product([ct[i] for ct in cts], pts) for i in range(nparts)]
    return Ciphertext(cts[0].par, c, cts[0].level)


def plaintext_poly_ntt(par, values, level=0):
    """Plaintext::poly_ntt for Encoding::poly (plaintext_vec.rs:70-102): the values as
    coefficients (zero padded), reduced mod every q_i, in Ntt form -- NOT scaled by delta."""
    ctx = par.context_at_level(level)
    return Poly.from_u64(ctx, [v % par.plaintext for v in values]).into_ntt()


def mul_plain(ct, pt_poly_ntt):
    """ops/mod.rs:229-257: every part times pt.poly_ntt."""
    return Ciphertext(ct.par, [ci.mul(pt_poly_ntt) for ci in ct.c], ct.level)


class RGSWCiphertext:
    """rgsw_ciphertext.rs:19-156."""

    def __init__(self, sk=None, values=None, rng=None, level=0, ksk0=None, ksk1=None):
        if ksk0 is not None:
            self.ksk0, self.ksk1 = ksk0, ksk1
            return
        par = sk.par
        ctx = par.context_at_level(level)
        pt_poly_ntt = plaintext_poly_ntt(par, values, level)  # Plaintext::poly_ntt (unscaled)
        m = pt_poly_ntt.into_power_basis()
        m_s = Poly.from_i64(ctx, sk.coeffs).into_ntt().mul(pt_poly_ntt).into_power_basis()
        self.ksk0 = KeySwitchingKey(sk, m, level, level, rng)
        self.ksk1 = KeySwitchingKey(sk, m_s, level, level, rng)

    def external_product(self, ct: Ciphertext):
        """`&Ciphertext * &RGSWCiphertext` (rgsw_ciphertext.rs:122-156)."""
        assert len(ct) == 2 and ct.level == self.ksk0.ciphertext_level
        c0, c1 = self.ksk0.key_switch(ct[0].into_power_basis())
        c0p, c1p = self.ksk1.key_switch(ct[1].into_power_basis())
        return Ciphertext(ct.par, [c0.add(c0p), c1.add(c1p)], ct.level)


def inner_sum(ct, galois_keys, degree):
    """evaluation_key.rs:56-100 (`computes_inner_sum`): galois_keys maps exponent -> GaloisKey."""
    out = ct
    i = 1
    while i < degree // 2:
        tmp = galois_keys[rot_to_gk_exponent(degree, i)].relinearize(out)
        out = out.add(tmp)
        i *= 2
    tmp = galois_keys[2 * degree - 1].relinearize(out)
    return out.add(tmp)


def expansion_monomial(par, level, l):
    """evaluation_key.rs:467-474: -x^(N - 2^l) over the ciphertext context, in Ntt form."""
    ctx = par.context_at_level(level)
    v = [0] * par.degree()
    v[par.degree() - (1 << l)] = -1
    return Poly.from_i64(ctx, v).into_ntt()


def expands(ct, size, galois_keys):
    """EvaluationKey::expands (evaluation_key.rs:192-256), eprint 2019/1483.
    galois_keys: {element: GaloisKey}; needs (N >> l) + 1 for l < ceil(log2(size))."""
    n = ct.par.degree()
    if len(ct) != 2:
        raise ValueError("InvalidPolynomialCount")
    if size == 0 or size > n:
        raise ValueError("InvalidExpansionSize")
    level = (size - 1).bit_length()
    if level == 0:
        return [ct.clone()]
    if any((n >> l) + 1 not in galois_keys for l in range(level)):
        raise ValueError("Unsupported(Expansion)")
    out = [None] * (1 << level)
    out[0] = ct.clone()
    for l in range(level):
        monomial = expansion_monomial(ct.par, ct.level, l)
        gk = galois_keys[(n >> l) + 1]
        step = 1 << l
        for i in range(step):
            sub = gk.relinearize(out[i])
            j = step | i
            if j < size:
                t = out[i].sub(sub)
                out[j] = Ciphertext(ct.par, [t[0].mul(monomial), t[1].mul(monomial)], ct.level)
            out[i] = out[i].add(sub)
    return out[:size]
