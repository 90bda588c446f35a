"""TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.

CPU restatement of `fhe_math::rq` (reference: crates/fhe-math/src/rq/
{context,mod,ops,scaler,switcher}.rs).  Polynomials are `Poly` objects holding
`coefficients[row][col]` Python ints (row-major [L][N], rq/mod.rs:126-133) and a
representation tag.  Values equal the reference's for both constant-time and
`_vt` variants (those differ in timing only).
"""

from .zq import Modulus
from .ntt import NttOperator, bitrev
from .rns import RnsContext, RnsScaler, ScalingFactor

POWER_BASIS = "PowerBasis"
NTT = "Ntt"
NTT_SHOUP = "NttShoup"


_OP_CACHE = {}


def cached_ntt_operator(q, degree, psi):
    key = (q.p, degree, psi)
    if key not in _OP_CACHE:
        _OP_CACHE[key] = NttOperator(q, degree, psi)
    return _OP_CACHE[key]


class Context:
    """rq/context.rs:9-92."""

    def __init__(self, moduli, degree, psis=None):
        if degree < 8 or degree & (degree - 1):
            raise ValueError("InvalidPolynomialDegree")
        self.moduli = list(moduli)
        self.degree = degree
        self.logn = degree.bit_length() - 1
        self.rns = RnsContext(self.moduli)
        self.q = [Modulus(m) for m in self.moduli]
        # NttOperator::new is a pure function of (p, degree, psi): memoised, because the
        # next_context chain below would otherwise rebuild every table O(L^2) times.
        self.ops = [cached_ntt_operator(qi, degree, None if psis is None else psis[i])
                    for i, qi in enumerate(self.q)]
        self.bitrev = [bitrev(j, self.logn) for j in range(degree)]
        q_last = self.moduli[-1]
        self.inv_last_qi_mod_qj = []
        self.inv_last_qi_mod_qj_shoup = []
        for qi in self.q[:-1]:
            inv = qi.inv(qi.reduce(q_last))
            self.inv_last_qi_mod_qj.append(inv)
            self.inv_last_qi_mod_qj_shoup.append(qi.shoup(inv))
        self.next_context = (Context(self.moduli[:-1], degree,
                                     None if psis is None else psis[:-1])
                             if len(self.moduli) >= 2 else None)

    def __eq__(self, other):
        return (isinstance(other, Context) and other.moduli == self.moduli
                and other.degree == self.degree)

    def modulus(self):
        return self.rns.product

    def niterations_to(self, context):
        """context.rs:117-141."""
        if context == self:
            return 0
        n, cur = 0, self
        while cur.next_context is not None:
            n += 1
            cur = cur.next_context
            if cur == context:
                return n
        raise ValueError("ContextNotReachable")

    def context_at_level(self, i):
        if i >= len(self.moduli):
            raise ValueError("InvalidContextLevel")
        cur = self
        for _ in range(i):
            cur = cur.next_context
        return cur


class SubstitutionExponent:
    """rq/mod.rs:86-121."""

    def __init__(self, ctx: Context, exponent: int):
        exponent %= 2 * ctx.degree
        if exponent & 1 == 0:
            raise ValueError("InvalidSubstitutionExponent")
        self.ctx = ctx
        self.exponent = exponent
        power = (exponent - 1) // 2
        mask = ctx.degree - 1
        self.power_bitrev = []
        for _ in range(ctx.degree):
            self.power_bitrev.append(bitrev(power & mask, ctx.logn))
            power += exponent


class Poly:
    """rq/mod.rs:126-133."""

    def __init__(self, ctx: Context, rep: str, coefficients=None):
        self.ctx = ctx
        self.rep = rep
        self.has_lazy_coefficients = False
        n = ctx.degree
        self.coefficients = ([[0] * n for _ in ctx.q] if coefficients is None
                             else [list(r) for r in coefficients])
        assert len(self.coefficients) == len(ctx.q)
        assert all(len(r) == n for r in self.coefficients)
        self.coefficients_shoup = None
        if rep == NTT_SHOUP:
            self._compute_coefficients_shoup()

    # ---- constructors / conversions --------------------------------------
    @staticmethod
    def zero(ctx, rep):
        return Poly(ctx, rep)

    @staticmethod
    def from_i64(ctx, values, rep=POWER_BASIS):
        """convert.rs: TryConvertFrom<&[i64]> (PowerBasis), via reduce_vec_i64."""
        assert len(values) <= ctx.degree
        rows = []
        for qi in ctx.q:
            row = [qi.reduce_i64(v) for v in values] + [0] * (ctx.degree - len(values))
            rows.append(row)
        p = Poly(ctx, POWER_BASIS, rows)
        return p if rep == POWER_BASIS else p.into_ntt()

    @staticmethod
    def from_u64(ctx, values):
        """convert.rs: TryConvertFrom<&[u64]> of length <= degree (PowerBasis)."""
        assert len(values) <= ctx.degree
        rows = [[qi.reduce(v) for v in values] + [0] * (ctx.degree - len(values)) for qi in ctx.q]
        return Poly(ctx, POWER_BASIS, rows)

    @staticmethod
    def from_biguints(ctx, values):
        """convert.rs: TryConvertFrom<&[BigUint]> (PowerBasis)."""
        assert len(values) <= ctx.degree
        rows = [[v % m for v in values] + [0] * (ctx.degree - len(values)) for m in ctx.moduli]
        return Poly(ctx, POWER_BASIS, rows)

    def to_biguints(self):
        """From<&Poly> for Vec<BigUint>: per-column CRT lift."""
        return [self.ctx.rns.lift([row[j] for row in self.coefficients])
                for j in range(self.ctx.degree)]

    def clone(self):
        p = Poly(self.ctx, POWER_BASIS if self.rep != NTT_SHOUP else NTT, self.coefficients)
        p.rep = self.rep
        p.has_lazy_coefficients = self.has_lazy_coefficients
        if self.coefficients_shoup is not None:
            p.coefficients_shoup = [list(r) for r in self.coefficients_shoup]
        return p

    def __eq__(self, other):
        return (isinstance(other, Poly) and self.ctx == other.ctx and self.rep == other.rep
                and self.coefficients == other.coefficients)

    def _compute_coefficients_shoup(self):
        """mod.rs:244-258."""
        self.coefficients_shoup = [qi.shoup_vec(row)
                                   for qi, row in zip(self.ctx.q, self.coefficients)]

    # ---- representation changes (mod.rs:335-354, 535-625) ------------------
    def into_ntt(self):
        assert self.rep == POWER_BASIS
        rows = [op.forward(row) for op, row in zip(self.ctx.ops, self.coefficients)]
        return Poly(self.ctx, NTT, rows)

    def into_ntt_shoup(self):
        if self.rep == POWER_BASIS:
            rows = [op.forward(row) for op, row in zip(self.ctx.ops, self.coefficients)]
        else:
            assert self.rep == NTT
            rows = self.coefficients
        return Poly(self.ctx, NTT_SHOUP, rows)

    def into_power_basis(self):
        assert self.rep in (NTT, NTT_SHOUP)
        rows = [op.backward(row) for op, row in zip(self.ctx.ops, self.coefficients)]
        return Poly(self.ctx, POWER_BASIS, rows)

    def as_ntt(self):
        """NttShoup -> Ntt (drops the Shoup copy)."""
        assert self.rep in (NTT, NTT_SHOUP)
        return Poly(self.ctx, NTT, self.coefficients)

    @staticmethod
    def create_constant_ntt_polynomial_with_lazy_coefficients(power_basis_coefficients, ctx):
        """mod.rs:563-586: lift one PowerBasis row to every modulus of ctx,
        lazy_reduce_vec, forward_vt_lazy -> values < 4p."""
        rows = []
        for qi, op in zip(ctx.q, ctx.ops):
            rows.append(op.forward_lazy(qi.lazy_reduce_vec(power_basis_coefficients)))
        p = Poly(ctx, NTT, rows)
        p.has_lazy_coefficients = True
        return p

    # ---- ops (rq/ops.rs) ------------------------------------------------------
    def _check(self, other):
        assert self.ctx == other.ctx, "Incompatible contexts"

    def add(self, other):
        """ops.rs:10-118 (PowerBasis+PowerBasis, Ntt+Ntt)."""
        assert not self.has_lazy_coefficients and not other.has_lazy_coefficients
        assert self.rep == other.rep and self.rep in (POWER_BASIS, NTT)
        self._check(other)
        return Poly(self.ctx, self.rep, [qi.add_vec(a, b) for qi, a, b in
                                         zip(self.ctx.q, self.coefficients, other.coefficients)])

    def sub(self, other):
        assert not self.has_lazy_coefficients and not other.has_lazy_coefficients
        assert self.rep == other.rep and self.rep in (POWER_BASIS, NTT)
        self._check(other)
        return Poly(self.ctx, self.rep, [qi.sub_vec(a, b) for qi, a, b in
                                         zip(self.ctx.q, self.coefficients, other.coefficients)])

    def neg(self):
        assert not self.has_lazy_coefficients
        return Poly(self.ctx, self.rep, [qi.neg_vec(a) for qi, a in
                                         zip(self.ctx.q, self.coefficients)])

    def mul(self, other):
        """ops.rs:174-245: Ntt*Ntt (Barrett / opt) and Ntt*NttShoup (Shoup;
        accepts a lazy lhs and clears the flag)."""
        self._check(other)
        assert self.rep == NTT
        assert not other.has_lazy_coefficients
        if other.rep == NTT:
            assert not self.has_lazy_coefficients
            rows = [qi.mul_vec(a, b) for qi, a, b in
                    zip(self.ctx.q, self.coefficients, other.coefficients)]
        else:
            assert other.rep == NTT_SHOUP
            rows = [qi.mul_shoup_vec(a, b, bs) for qi, a, b, bs in
                    zip(self.ctx.q, self.coefficients, other.coefficients,
                        other.coefficients_shoup)]
        return Poly(self.ctx, NTT, rows)

    def mul_scalar(self, scalar: int):
        """ops.rs:297-352: `*= &BigUint` (projected per modulus)."""
        crt = self.ctx.rns.project(scalar)
        rows = [qi.scalar_mul_vec(a, s) for qi, a, s in zip(self.ctx.q, self.coefficients, crt)]
        return Poly(self.ctx, self.rep, rows)

    # ---- substitute (mod.rs:360-412) ----------------------------------------
    def substitute(self, i: SubstitutionExponent):
        n = self.ctx.degree
        q = Poly(self.ctx, self.rep if self.rep != NTT_SHOUP else NTT)
        if self.rep in (NTT, NTT_SHOUP):
            for q_row, p_row in zip(q.coefficients, self.coefficients):
                for j, k in zip(self.ctx.bitrev, i.power_bitrev):
                    q_row[j] = p_row[k]
            if self.rep == NTT_SHOUP:
                q.rep = NTT_SHOUP
                q.coefficients_shoup = [[0] * n for _ in self.ctx.q]
                for q_row, p_row in zip(q.coefficients_shoup, self.coefficients_shoup):
                    for j, k in zip(self.ctx.bitrev, i.power_bitrev):
                        q_row[j] = p_row[k]
        else:
            power = 0
            mask = n - 1
            for j in range(n):
                for qi, q_row, p_row in zip(self.ctx.q, q.coefficients, self.coefficients):
                    if power & n:
                        q_row[power & mask] = qi.sub(q_row[power & mask], p_row[j])
                    else:
                        q_row[power & mask] = qi.add(q_row[power & mask], p_row[j])
                power += i.exponent
        return q

    # ---- modulus switching (mod.rs:433-507) ----------------------------------
    def switch_down(self):
        """mod.rs:433-492 (eprint 2018/931 Alg. 2).  Returns a new Poly."""
        assert self.rep == POWER_BASIS
        ctx = self.ctx
        if ctx.next_context is None:
            raise ValueError("NoMoreContext")
        q_last = ctx.q[-1]
        q_last_div_2 = q_last.p // 2
        q_last_poly = [q_last.add(c, q_last_div_2) for c in self.coefficients[-1]]
        rows = []
        for coeffs, qi, inv, inv_shoup in zip(self.coefficients[:-1], ctx.q,
                                              ctx.inv_last_qi_mod_qj,
                                              ctx.inv_last_qi_mod_qj_shoup):
            q_last_div_2_mod_qi = qi.p - qi.reduce(q_last_div_2)
            row = []
            for coeff, q_last_coeff in zip(coeffs, q_last_poly):
                tmp = qi.lazy_reduce(q_last_coeff) + q_last_div_2_mod_qi
                c = coeff + 3 * qi.p - tmp
                row.append(qi.mul_shoup(c, inv, inv_shoup))
            rows.append(row)
        return Poly(ctx.next_context, POWER_BASIS, rows)

    def switch_down_to(self, context):
        n = self.ctx.niterations_to(context)
        p = self
        for _ in range(n):
            p = p.switch_down()
        assert p.ctx == context
        return p

    # ---- scale / switch (mod.rs:660-680) --------------------------------------
    def scale(self, scaler):
        return scaler.scale(self)

    def switch(self, switcher):
        return switcher.switch(self)


class Scaler:
    """rq/scaler.rs:18-127."""

    def __init__(self, frm: Context, to: Context, factor: ScalingFactor):
        if frm.degree != to.degree:
            raise ValueError("DegreeMismatch")
        self.frm = frm
        self.to = to
        ncm = 0
        if factor.is_one:
            for a, b in zip(frm.moduli, to.moduli):
                if a != b:
                    break
                ncm += 1
        self.number_common_moduli = ncm
        self.scaler = RnsScaler(frm.rns, to.rns, factor)

    def scale(self, p: Poly) -> Poly:
        if p.ctx != self.frm:
            raise ValueError("PolynomialContextMismatch")
        assert p.rep in (POWER_BASIS, NTT)
        n = self.to.degree
        ncm = self.number_common_moduli
        nto = len(self.to.q)
        new = [[0] * n for _ in range(nto)]
        for r in range(ncm):
            new[r] = list(p.coefficients[r])
        if ncm < nto:
            needs_transform = p.rep != POWER_BASIS
            if needs_transform:
                pb = [op.backward(row) for op, row in zip(p.ctx.ops, p.coefficients)]
            else:
                pb = p.coefficients
            for col in range(n):
                out = self.scaler.scale([row[col] for row in pb], nto - ncm, ncm)
                for r, v in enumerate(out):
                    new[ncm + r][col] = v
            if needs_transform:
                for r in range(ncm, nto):
                    new[r] = self.to.ops[r].forward(new[r])
        return Poly(self.to, p.rep, new)


class Switcher:
    """rq/switcher.rs:11-26."""

    def __init__(self, frm: Context, to: Context):
        self.scaler = Scaler(frm, to, ScalingFactor(to.modulus(), frm.modulus()))

    def switch(self, p: Poly) -> Poly:
        return self.scaler.scale(p)


# --------------------------------------------------------------------------------------
# Rq wire format
# --------------------------------------------------------------------------------------
def transcode_to_bytes(a, nbits):
    """crates/fhe-util/src/lib.rs:71-107."""
    assert 0 < nbits <= 64
    mask = (1 << nbits) - 1
    nbytes = -(-(len(a) * nbits) // 8)
    out = bytearray()
    cur, have, i = 0, 0, 0
    while i < len(a):
        if have < 8:
            cur |= (a[i] & mask) << have
            have += nbits
            i += 1
        while have >= 8:
            out.append(cur & 0xFF)
            cur >>= 8
            have -= 8
    if have > 0:
        assert have < 8 and len(out) == nbytes - 1
        out.append(cur & 0xFF)
    else:
        assert len(out) == nbytes and cur == 0
    return bytes(out)


def transcode_from_bytes(b, nbits):
    """crates/fhe-util/src/lib.rs:111-146."""
    assert 0 < nbits <= 64
    mask = (1 << nbits) - 1
    nelements = -(-(len(b) * 8) // nbits)
    out = []
    cur, have, i = 0, 0, 0
    while i < len(b):
        if have < nbits:
            cur |= b[i] << have
            have += 8
            i += 1
        while have >= nbits:
            out.append(cur & mask)
            cur >>= nbits
            have -= nbits
    if have > 0:
        assert len(out) == nelements - 1
        out.append(cur)
    else:
        assert len(out) == nelements and cur == 0
    return out


def modulus_wire_bits(p):
    """zq/mod.rs:783-793: p_nbits = 64 - (p - 1).leading_zeros()."""
    return (p - 1).bit_length()


def poly_to_wire(p: Poly) -> bytes:
    """`impl From<&Poly<R>> for Rq`, rq/convert.rs:17-44: the `coefficients` field (always the
    PowerBasis form, one bit-packed run per residue row)."""
    assert not p.has_lazy_coefficients
    q = p if p.rep == POWER_BASIS else p.clone().into_power_basis()
    return b"".join(transcode_to_bytes(row, modulus_wire_bits(m)) for row, m in zip(q.coefficients, p.ctx.moduli))


def poly_from_wire(ctx: Context, data: bytes, rep=POWER_BASIS) -> Poly:
    """parse_proto + TryConvertFrom<&Rq>, rq/convert.rs:46-147 (coefficients taken verbatim)."""
    n = ctx.degree
    if n % 8 or n < 8:
        raise ValueError("InvalidDegree")
    sizes = [-(-(n * modulus_wire_bits(m)) // 8) for m in ctx.moduli]
    if len(data) != sum(sizes):
        raise ValueError("InvalidCoefficientCount")
    rows, idx = [], 0
    for m, sz in zip(ctx.moduli, sizes):
        rows.append(transcode_from_bytes(data[idx:idx + sz], modulus_wire_bits(m))[:n])
        idx += sz
    p = Poly(ctx, POWER_BASIS, rows)
    return p if rep == POWER_BASIS else p.into_ntt()
