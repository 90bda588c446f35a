"""TEST INFRASTRUCTURE (oracle) -- ctypes binding of oracle/c/fhe_oracle.c.

The C file restates the same reference algorithms as the Python oracle, at a
speed that lets tests check the HIP engine at BASELINE.json's full sizes and
lets bench.py time a CPU baseline.  Constants come from the Python oracle's
objects (Context, Scaler, KeySwitchingKey, Multiplicator).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_LIB_PATH = os.path.join(_ORACLE_DIR, "_build", "libfhe_oracle.so")
_lib = None

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "c", "fhe_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_time_multiply.restype = C.c_double
        _lib.orc_time_ntt_forward.restype = C.c_double
        _lib.orc_mod_mul.restype = C.c_uint64
        _lib.orc_reduce_u128.restype = C.c_uint64
        _lib.orc_mod_mul.argtypes = [C.c_uint64] * 3
        _lib.orc_reduce_u128.argtypes = [C.c_uint64] * 3
        _lib.orc_supports_opt.argtypes = [C.c_uint64]
    return _lib


def arr(x):
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


class OrcCtx(C.Structure):
    _fields_ = [("n", C.c_uint64), ("nmod", C.c_uint64), ("moduli", u64p),
                ("omegas", u64p), ("omegas_shoup", u64p), ("zetas_inv", u64p),
                ("zetas_inv_shoup", u64p), ("size_inv", u64p), ("size_inv_shoup", u64p),
                ("inv_last", u64p), ("inv_last_shoup", u64p)]


class OrcScaler(C.Structure):
    _fields_ = [("nfrom", C.c_uint64), ("nto", C.c_uint64), ("ncommon", C.c_uint64),
                ("is_one", C.c_uint64), ("shift", C.c_uint64),
                ("gamma", u64p), ("gamma_shoup", u64p), ("omega", u64p), ("omega_shoup", u64p),
                ("theta_gamma_lo", C.c_uint64), ("theta_gamma_hi", C.c_uint64),
                ("theta_gamma_sign", C.c_uint64),
                ("theta_omega_lo", u64p), ("theta_omega_hi", u64p), ("theta_omega_sign", u64p),
                ("theta_garner_lo", u64p), ("theta_garner_hi", u64p)]


class OrcKsk(C.Structure):
    _fields_ = [("ndigits", C.c_uint64), ("c0", u64p), ("c0_shoup", u64p),
                ("c1", u64p), ("c1_shoup", u64p)]


class OrcMul(C.Structure):
    _fields_ = [("base_ctx", C.POINTER(OrcCtx)), ("mul_ctx", C.POINTER(OrcCtx)),
                ("extender_lhs", C.POINTER(OrcScaler)), ("extender_rhs", C.POINTER(OrcScaler)),
                ("down_scaler", C.POINTER(OrcScaler)), ("rk", C.POINTER(OrcKsk)),
                ("mod_switch", C.c_uint64)]


class CCtx:
    """C view of a Python-oracle rq.Context."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.n, self.L = ctx.degree, len(ctx.moduli)
        self._keep = dict(
            moduli=arr(ctx.moduli),
            omegas=arr([op.omegas for op in ctx.ops]),
            omegas_shoup=arr([op.omegas_shoup for op in ctx.ops]),
            zetas_inv=arr([op.zetas_inv for op in ctx.ops]),
            zetas_inv_shoup=arr([op.zetas_inv_shoup for op in ctx.ops]),
            size_inv=arr([op.size_inv for op in ctx.ops]),
            size_inv_shoup=arr([op.size_inv_shoup for op in ctx.ops]),
            inv_last=arr(list(ctx.inv_last_qi_mod_qj) + [0]),
            inv_last_shoup=arr(list(ctx.inv_last_qi_mod_qj_shoup) + [0]),
        )
        k = self._keep
        self.c = OrcCtx(self.n, self.L, ptr(k["moduli"]), ptr(k["omegas"]), ptr(k["omegas_shoup"]),
                        ptr(k["zetas_inv"]), ptr(k["zetas_inv_shoup"]), ptr(k["size_inv"]),
                        ptr(k["size_inv_shoup"]), ptr(k["inv_last"]), ptr(k["inv_last_shoup"]))

    def ref(self):
        return C.byref(self.c)

    def tables(self):
        return self._keep

    # -- row / poly level ----------------------------------------------------
    def ntt_forward_row(self, mi, row, lazy=False):
        a = arr(row)
        lib().orc_ntt_forward(self.ref(), C.c_uint64(mi), ptr(a), C.c_int(1 if lazy else 0))
        return a

    def ntt_backward_row(self, mi, row):
        a = arr(row)
        lib().orc_ntt_backward(self.ref(), C.c_uint64(mi), ptr(a))
        return a

    def poly_ntt_forward(self, poly):
        a = arr(poly).reshape(self.L, self.n).copy()
        lib().orc_poly_ntt_forward(self.ref(), ptr(a))
        return a

    def poly_ntt_backward(self, poly):
        a = arr(poly).reshape(self.L, self.n).copy()
        lib().orc_poly_ntt_backward(self.ref(), ptr(a))
        return a

    def _binop(self, fn, a, b):
        a = arr(a).reshape(self.L, self.n).copy()
        b = arr(b).reshape(self.L, self.n)
        fn(self.ref(), ptr(a), ptr(b))
        return a

    def poly_add(self, a, b):
        return self._binop(lib().orc_poly_add, a, b)

    def poly_sub(self, a, b):
        return self._binop(lib().orc_poly_sub, a, b)

    def poly_mul(self, a, b):
        return self._binop(lib().orc_poly_mul, a, b)

    def poly_neg(self, a):
        a = arr(a).reshape(self.L, self.n).copy()
        lib().orc_poly_neg(self.ref(), ptr(a))
        return a

    def poly_mul_shoup(self, a, b, bs):
        a = arr(a).reshape(self.L, self.n).copy()
        b, bs = arr(b).reshape(self.L, self.n), arr(bs).reshape(self.L, self.n)
        lib().orc_poly_mul_shoup(self.ref(), ptr(a), ptr(b), ptr(bs))
        return a

    def shoup(self, poly):
        a = arr(poly).reshape(self.L, self.n)
        out = np.empty_like(a)
        for r in range(self.L):
            lib().orc_shoup_vec(C.c_uint64(self.ctx.moduli[r]), ptr(a[r]), ptr(out[r]), C.c_uint64(self.n))
        return out

    def poly_switch_down(self, poly):
        a = arr(poly).reshape(self.L, self.n).copy()
        lib().orc_poly_switch_down(self.ref(), ptr(a))
        return a[: self.L - 1].copy()

    def poly_substitute(self, exponent, poly, repr_is_ntt):
        a = arr(poly).reshape(self.L, self.n)
        out = np.zeros_like(a)
        lib().orc_poly_substitute(self.ref(), C.c_uint64(exponent), ptr(a), ptr(out),
                                  C.c_int(1 if repr_is_ntt else 0))
        return out

    def synth_poly(self, seed, ct, part):
        out = np.empty((self.L, self.n), dtype=np.uint64)
        lib().orc_synth_poly(C.c_uint64(seed), C.c_uint64(ct), C.c_uint64(part), self.c.moduli,
                             C.c_uint64(self.L), C.c_uint64(self.n), ptr(out))
        return out


class CScaler:
    """C view of a Python-oracle rq.Scaler."""

    def __init__(self, scaler, cfrom=None, cto=None):
        self.scaler = scaler
        s = scaler.scaler
        self.cfrom = cfrom or CCtx(scaler.frm)
        self.cto = cto or CCtx(scaler.to)
        self._keep = dict(
            gamma=arr(s.gamma), gamma_shoup=arr(s.gamma_shoup),
            omega=arr(s.omega), omega_shoup=arr(s.omega_shoup),
            tol=arr(s.theta_omega_lo), toh=arr(s.theta_omega_hi),
            tos=arr([1 if x else 0 for x in s.theta_omega_sign]),
            tgl=arr(s.theta_garner_lo), tgh=arr(s.theta_garner_hi))
        k = self._keep
        self.c = OrcScaler(len(s.frm.moduli_u64), len(s.to.moduli_u64), scaler.number_common_moduli,
                           1 if s.scaling_factor.is_one else 0, s.theta_garner_shift,
                           ptr(k["gamma"]), ptr(k["gamma_shoup"]), ptr(k["omega"]), ptr(k["omega_shoup"]),
                           s.theta_gamma_lo, s.theta_gamma_hi, 1 if s.theta_gamma_sign else 0,
                           ptr(k["tol"]), ptr(k["toh"]), ptr(k["tos"]), ptr(k["tgl"]), ptr(k["tgh"]))

    def ref(self):
        return C.byref(self.c)

    def constants(self):
        return self._keep

    def scale(self, poly, repr_is_ntt):
        a = arr(poly).reshape(self.cfrom.L, self.cfrom.n)
        out = np.zeros((self.cto.L, self.cto.n), dtype=np.uint64)
        lib().orc_poly_scale(self.ref(), self.cfrom.ref(), self.cto.ref(), ptr(a), ptr(out),
                             C.c_int(1 if repr_is_ntt else 0))
        return out

    def rns_scale(self, rests, size, starting_index=0):
        r = arr(rests)
        out = np.zeros(size, dtype=np.uint64)
        lib().orc_rns_scale(self.ref(), self.cto.c.moduli, ptr(r), ptr(out), C.c_uint64(size),
                            C.c_uint64(starting_index))
        return out


class CKsk:
    """C view of a key-switching key: c0/c1 arrays [ndigits][Lk][n] (+Shoup)."""

    def __init__(self, c0, c0_shoup, c1, c1_shoup, cct: CCtx, cksk: CCtx):
        self.cct, self.cksk = cct, cksk
        self.c0, self.c0s, self.c1, self.c1s = arr(c0), arr(c0_shoup), arr(c1), arr(c1_shoup)
        self.ndigits = self.c0.shape[0]
        self.c = OrcKsk(self.ndigits, ptr(self.c0), ptr(self.c0s), ptr(self.c1), ptr(self.c1s))

    @staticmethod
    def from_oracle(ksk, cct=None, cksk=None):
        cct = cct or CCtx(ksk.ctx_ciphertext)
        cksk = cksk or CCtx(ksk.ctx_ksk)
        return CKsk([p.coefficients for p in ksk.c0], [p.coefficients_shoup for p in ksk.c0],
                    [p.coefficients for p in ksk.c1], [p.coefficients_shoup for p in ksk.c1], cct, cksk)

    def ref(self):
        return C.byref(self.c)

    def key_switch(self, p):
        a = arr(p).reshape(self.cct.L, self.cct.n)
        o0 = np.zeros((self.cksk.L, self.cksk.n), dtype=np.uint64)
        o1 = np.zeros_like(o0)
        lib().orc_key_switch(self.cct.ref(), self.cksk.ref(), self.ref(), ptr(a), ptr(o0), ptr(o1))
        return o0, o1

    def galois_relinearize(self, exponent, ct):
        """Key at the ciphertext level."""
        a = arr(ct).reshape(2, self.cct.L, self.cct.n)
        out = np.zeros_like(a)
        lib().orc_galois_relinearize(self.cct.ref(), self.ref(), C.c_uint64(exponent), ptr(a), ptr(out))
        return out


class CMul:
    """C view of a Multiplicator (relin key at the ciphertext level)."""

    def __init__(self, base: CCtx, mul: CCtx, ext_lhs: CScaler, ext_rhs: CScaler, down: CScaler,
                 rk: CKsk = None, mod_switch=False):
        self.base, self.mul, self.ext_lhs, self.ext_rhs, self.down, self.rk = base, mul, ext_lhs, ext_rhs, down, rk
        self.mod_switch = mod_switch
        self.c = OrcMul(C.pointer(base.c), C.pointer(mul.c), C.pointer(ext_lhs.c), C.pointer(ext_rhs.c),
                        C.pointer(down.c), C.pointer(rk.c) if rk is not None else None,
                        1 if mod_switch else 0)

    @staticmethod
    def from_oracle(m, rk_c: CKsk = None):
        base, mul = CCtx(m.base_ctx), CCtx(m.mul_ctx)
        lhs = CScaler(m.extender_lhs, base, mul)
        rhs = CScaler(m.extender_rhs, base, mul)
        down = CScaler(m.down_scaler, mul, base)
        if rk_c is None and m.rk is not None:
            rk_c = CKsk.from_oracle(m.rk.ksk, base, base)
        return CMul(base, mul, lhs, rhs, down, rk_c, m.mod_switch)

    def out_shape(self):
        parts = 2 if self.rk is not None else 3
        rows = self.base.L - 1 if self.mod_switch else self.base.L
        return parts, rows, self.base.n

    def multiply(self, lhs, rhs):
        a = arr(lhs).reshape(2, self.base.L, self.base.n)
        b = arr(rhs).reshape(2, self.base.L, self.base.n)
        out = np.zeros(self.out_shape(), dtype=np.uint64)
        lib().orc_bfv_multiply(C.byref(self.c), ptr(a), ptr(b), ptr(out))
        return out

    def time_multiply(self, lhs, rhs, count, threads=1):
        a = arr(lhs).reshape(-1, 2, self.base.L, self.base.n)
        b = arr(rhs).reshape(-1, 2, self.base.L, self.base.n)
        assert not self.mod_switch
        last = np.zeros((2, self.base.L, self.base.n), dtype=np.uint64)
        secs = lib().orc_time_multiply(C.byref(self.c), ptr(a), ptr(b), C.c_uint64(a.shape[0]),
                                       C.c_uint64(count), C.c_int(threads), ptr(last))
        return secs, last


def max_threads():
    return lib().orc_max_threads()
