"""Parity cases shared by the CPU-emulation suite (tests/test_emu_parity.py) and the GPU suite
(tests/test_gpu_parity.py).  Each case drives the engine through the C ABI (via the ctypes
mirror in fhe.rs_amd/api.py) and compares bit-for-bit with the oracle; they read like the
reference's own tests (cited per case)."""
import random

import numpy as np

from fhe_oracle import bfv as obfv
from fhe_oracle import coracle
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import (Context as OCtx, Poly, Scaler as OScaler, Switcher as OSwitcher,
                           SubstitutionExponent, POWER_BASIS, NTT)
from helpers import arr, rows, rand_poly, oracle_tables, ksk_arrays, ct_arr, Xfer

Q3 = [4611686018282684417, 4611686018326724609, 4611686018309947393]
P3 = [4611686018282684417, 4611686018309947393, 4611686018257518593]
MODULI5 = [1153, 4611686018326724609, 4611686018309947393, 4611686018232352769, 4611686018171535361]


def case_context_tables(fhe, n=32):
    """Host-only setup (device = -1): tables, Shoup twins, inv_last (rq/context.rs:42-92,
    ntt/native.rs:35-73) equal the oracle's; supplied tables are used verbatim."""
    o = OCtx(MODULI5[1:], n)
    c = fhe.Context(MODULI5[1:], n, device=-1)
    t = oracle_tables(o)
    for which, name in enumerate(("omegas", "omegas_shoup", "zetas_inv", "zetas_inv_shoup", "size_inv", "size_inv_shoup")):
        assert np.array_equal(c.table(which), np.array(t[name], dtype=np.uint64)), name
    assert c.table(6).tolist() == o.inv_last_qi_mod_qj and c.table(7).tolist() == o.inv_last_qi_mod_qj_shoup
    lvl2 = c.at_level(2)
    assert lvl2.moduli == MODULI5[1:3] and lvl2.table(6).tolist() == o.context_at_level(2).inv_last_qi_mod_qj
    assert c.niterations_to(lvl2) == 2
    # explicit tables with another primitive root
    psis = [pow(op.psi, 3, op.p.p) for op in o.ops]
    o2 = OCtx(MODULI5[1:], n, psis=psis)
    c2 = fhe.Context(MODULI5[1:], n, device=-1, tables=oracle_tables(o2))
    assert np.array_equal(c2.table(0), np.array(oracle_tables(o2)["omegas"], dtype=np.uint64))
    assert not np.array_equal(c2.table(0), c.table(0))


def case_ntt(fhe, dev, n, moduli=Q3, batch=2, coracle_ctx=None, seed=0):
    """ntt/mod.rs:50-112 (bijection) + exact values vs the oracle butterflies."""
    x = Xfer(dev)
    rng = random.Random(n * 31 + seed)
    c = fhe.Context(moduli, n)
    a = np.array([[[rng.randrange(m) for _ in range(n)] for m in moduli] for _ in range(batch)], dtype=np.uint64)
    if coracle_ctx is None:
        o = OCtx(moduli, n)
        want = np.array([[op.forward([int(v) for v in row]) for op, row in zip(o.ops, poly)] for poly in a],
                        dtype=np.uint64)
    else:
        want = np.stack([coracle_ctx.poly_ntt_forward(poly) for poly in a])
    f = x.back(c.ntt_forward(x.to(a)))
    assert np.array_equal(f, want)
    back = x.back(c.ntt_backward(x.to(f)))
    assert np.array_equal(back, a)


def case_ntt_explicit_tables(fhe, dev, n=16):
    x = Xfer(dev)
    base = OCtx(Q3, n)
    o = OCtx(Q3, n, psis=[pow(op.psi, 5, op.p.p) for op in base.ops])
    c = fhe.Context(Q3, n, tables=oracle_tables(o))
    rng = random.Random(4)
    p = rand_poly(o, POWER_BASIS, rng)
    assert np.array_equal(x.back(c.ntt_forward(x.to(arr(p)))), arr(p.into_ntt()))


def case_poly_ops(fhe, dev, n=32):
    """rq/ops.rs:587-939."""
    x = Xfer(dev)
    rng = random.Random(9)
    o = OCtx(MODULI5, n)
    c = fhe.Context(MODULI5, n)
    a, b = rand_poly(o, NTT, rng), rand_poly(o, NTT, rng)
    A = np.stack([arr(a), arr(b)])
    B = np.stack([arr(b), arr(a)])
    assert np.array_equal(x.back(c.add(x.to(A), x.to(B)))[0], arr(a.add(b)))
    assert np.array_equal(x.back(c.sub(x.to(A), x.to(B)))[1], arr(b.sub(a)))
    assert np.array_equal(x.back(c.mul(x.to(A), x.to(B)))[0], arr(a.mul(b)))
    assert np.array_equal(x.back(c.neg(x.to(A)))[1], arr(b.neg()))
    bs = b.into_ntt_shoup()
    sh = c.shoup(arr(b))
    assert np.array_equal(sh, np.array(bs.coefficients_shoup, dtype=np.uint64))
    assert np.array_equal(x.back(c.mul_shoup(x.to(arr(a)), x.to(arr(b)), x.to(sh))), arr(a.mul(bs)))
    # lazy lhs (< 4p) is accepted by the Shoup product (ops.rs:208-245)
    lazy = arr(a) + np.array([[2 * m] for m in MODULI5], dtype=np.uint64) * np.uint64(1)
    assert np.array_equal(x.back(c.mul_shoup(x.to(lazy), x.to(arr(b)), x.to(sh))), arr(a.mul(bs)))


def case_product_extremes(fhe, dev, n=64):
    """Residue products at the word boundaries (zq/mod.rs:208-260 `mul` / `mul_shoup`, ops.rs:208-245): the device forms
    its high products through the multiply-add's carry-out and full products from four 32 x 32 multiply-adds with
    no carry handling in the middle column (zq_dev.hpp mulhi64_c / mul_wide62) -- operands made of all-ones and
    all-zeros 32-bit words, 0, 1, p-1 and the lazy range up to 4p - 1 are where a dropped carry would show.
    Expected values from Python integers."""
    x = Xfer(dev)
    small = 1153 if n <= 64 else fhe.generate_prime(18, 2 * n, 1 << 18)
    moduli = [small] + [fhe.generate_prime(b, 2 * n, 1 << b) for b in (33, 45, 60, 62)] + [4611686018326724609]
    moduli = list(dict.fromkeys(moduli))
    c = fhe.Context(moduli, n)
    M32 = (1 << 32) - 1

    def specials(p, top):   # values below `top`
        cand = [0, 1, 2, p - 1, p - 2, p >> 1, M32, 1 << 32, (1 << 32) + 1, M32 << 32, (M32 << 32) | M32,
                p - M32, p - (1 << 32), (p >> 32) << 32, ((p >> 32) << 32) | M32, top - 1, top - M32, top >> 1]
        return sorted({v % top for v in cand if v >= 0})

    rng = random.Random(77)
    A, B, L = [], [], []
    for p in moduli:
        sa, sl = specials(p, p), specials(p, min(4 * p, 1 << 64))
        a = [sa[i % len(sa)] for i in range(n)]
        b = [sa[(i // 3 + 5 * i) % len(sa)] for i in range(n)]
        lz = [sl[(7 * i + i // 5) % len(sl)] for i in range(n)]
        for i in range(n - 8, n):
            a[i], b[i], lz[i] = rng.randrange(p), rng.randrange(p), rng.randrange(min(4 * p, 1 << 64))
        A.append(a), B.append(b), L.append(lz)
    An, Bn, Ln = (np.array(v, dtype=np.uint64) for v in (A, B, L))
    want = np.array([[(a * b) % p for a, b in zip(ra, rb)] for ra, rb, p in zip(A, B, moduli)], dtype=np.uint64)
    assert np.array_equal(x.back(c.mul(x.to(An[None]), x.to(Bn[None])))[0], want)
    sh = c.shoup(Bn)
    assert sh.tolist() == [[(b << 64) // p for b in rb] for rb, p in zip(B, moduli)]
    want_l = np.array([[(a * b) % p for a, b in zip(ra, rb)] for ra, rb, p in zip(L, B, moduli)], dtype=np.uint64)
    assert np.array_equal(x.back(c.mul_shoup(x.to(Ln), x.to(Bn), x.to(sh))), want_l)
    # the same extremes through the transforms (every butterfly multiplies by a twiddle): forward then inverse is the
    # identity, and the forward values match the oracle's
    o = OCtx(moduli, n)
    pw = Poly(o, POWER_BASIS, [list(r) for r in A])
    f = x.back(c.ntt_forward(x.to(An[None].copy())))[0]
    assert np.array_equal(f, arr(pw.into_ntt()))
    assert np.array_equal(x.back(c.ntt_backward(x.to(f[None].copy())))[0], An)


def case_substitute(fhe, dev, n=16):
    """rq/mod.rs:947-1036."""
    x = Xfer(dev)
    rng = random.Random(10)
    o = OCtx(MODULI5, n)
    c = fhe.Context(MODULI5, n)
    p = rand_poly(o, POWER_BASIS, rng)
    pn = p.into_ntt()
    for e in (1, 3, 11, 2 * n - 1, 2 * n + 5):
        se = SubstitutionExponent(o, e)
        assert np.array_equal(x.back(c.substitute(e, x.to(arr(p)), ntt=False)), arr(p.substitute(se)))
        assert np.array_equal(x.back(c.substitute(e, x.to(arr(pn)), ntt=True)), arr(pn.substitute(se)))
    for bad in (0, 2, 2 * n):
        try:
            c.substitute(bad, x.to(arr(p)), ntt=True)
            raise AssertionError("even exponent accepted")
        except fhe.FheError as err:
            assert err.code == -10


def case_switch_down(fhe, dev, n=16):
    """rq/mod.rs:1039-1073 + ciphertext.rs:148-161."""
    x = Xfer(dev)
    rng = random.Random(11)
    o = OCtx(MODULI5, n)
    c = fhe.Context(MODULI5, n)
    p = rand_poly(o, POWER_BASIS, rng)
    cur_o, cur_c, cur = o, c, p
    while cur_o.next_context is not None:
        nxt = cur.switch_down()
        got = x.back(cur_c.switch_down(x.to(arr(cur))))
        assert np.array_equal(got, arr(nxt))
        cur, cur_o, cur_c = nxt, cur_o.next_context, cur_c.at_level(1)
    try:
        cur_c.switch_down(x.to(arr(cur)))
        raise AssertionError("switch_down past the last context")
    except fhe.FheError as err:
        assert err.code == -8
    # ciphertext-level switch_down on 3 parts x 2 ciphertexts
    cts = [[rand_poly(o, NTT, rng) for _ in range(3)] for _ in range(2)]
    want = np.array([[arr(q.into_power_basis().switch_down().into_ntt()) for q in ct] for ct in cts], dtype=np.uint64)
    got = x.back(c.ciphertext_switch_down(x.to(np.array([[arr(q) for q in ct] for ct in cts], dtype=np.uint64))))
    assert np.array_equal(got, want)


def case_switch_down_to(fhe, dev, n=16):
    """Poly::switch_down_to (rq/mod.rs:498-507, tests :1075-1122) and Ciphertext::switch_to_level
    (bfv/ciphertext.rs:164-183) in ONE call each, against the oracle's level-by-level loops; errors as the reference."""
    x = Xfer(dev)
    rng = random.Random(12)
    o = OCtx(MODULI5, n)
    c = fhe.Context(MODULI5, n)
    ps = [rand_poly(o, POWER_BASIS, rng) for _ in range(3)]
    batch = x.to(np.array([arr(p) for p in ps], dtype=np.uint64))
    for lvl in range(len(MODULI5)):
        want = np.array([arr(p.switch_down_to(o.context_at_level(lvl))) for p in ps], dtype=np.uint64)
        got = x.back(c.switch_down_to(batch, c.at_level(lvl)))
        assert got.shape == want.shape and np.array_equal(got, want), lvl
    # from an intermediate level
    mid_o, mid_c = o.context_at_level(1), c.at_level(1)
    q = rand_poly(mid_o, POWER_BASIS, rng)
    assert np.array_equal(x.back(mid_c.switch_down_to(x.to(arr(q)[None]), c.at_level(3)))[0],
                          arr(q.switch_down_to(o.context_at_level(3))))
    # a context that is not below `from` on the chain: ContextNotReachable (rq/mod.rs:503-505)
    for bad_from, bad_to in ((mid_c, c), (c, fhe.Context(Q3, n))):
        try:
            bad_from.switch_down_to(x.to(np.zeros((1, bad_from.nmoduli, n), dtype=np.uint64)), bad_to)
            raise AssertionError("unreachable context accepted")
        except fhe.FheError as err:
            assert err.code == -9, err
    # Ciphertext::switch_to_level on 2 ciphertexts x 3 parts: every reachable number of levels
    cts = [[rand_poly(o, NTT, rng) for _ in range(3)] for _ in range(2)]
    src = x.to(np.array([[arr(q) for q in ct] for ct in cts], dtype=np.uint64))
    for levels in range(len(MODULI5)):
        want = []
        for ct in cts:
            cur = list(ct)
            for _ in range(levels):   # the reference's loop: PowerBasis -> switch_down -> Ntt per level
                cur = [q.into_power_basis().switch_down().into_ntt() for q in cur]
            want.append([arr(q) for q in cur])
        got = x.back(c.ciphertext_switch_to_level(src, levels))
        assert np.array_equal(got, np.array(want, dtype=np.uint64)), levels
    try:
        c.ciphertext_switch_to_level(src, len(MODULI5))
        raise AssertionError("level beyond the chain accepted")
    except fhe.FheError as err:
        assert err.code == -12, err


def case_device_buffers(fhe, n=32):
    """The C ABI's own device memory and streams (fhe_buf_*, fhe_stream_*): a polynomial batch is uploaded once,
    goes Ntt -> (.)^2 -> PowerBasis -> switch_down_to -> download without leaving the device, on a stream created
    through the ABI, and equals the oracle; pinned host memory round-trips; views index without copying."""
    import ctypes as C
    from fhe_rs_amd import _lib
    rng = random.Random(13)
    o = OCtx(MODULI5, n)
    c = fhe.Context(MODULI5, n)
    ps = [rand_poly(o, POWER_BASIS, rng) for _ in range(4)]
    host = np.array([arr(p) for p in ps], dtype=np.uint64)
    with fhe.Stream(0) as st:
        d = fhe.DeviceArray.from_numpy(host)
        assert d.shape == host.shape and np.array_equal(d.download(), host)
        assert np.array_equal(d[2].download(), host[2]) and np.array_equal(d[-1][1].download(), host[3][1])
        f = c.ntt_forward(d)                       # in place, on the ABI stream
        sq = c.mul(f, fhe.DeviceArray.from_numpy(f.download()))
        pb = c.ntt_backward(sq)
        low = c.switch_down_to(pb, c.at_level(2))
        st.synchronize()
        want = []
        for p in ps:
            f_o = p.into_ntt()
            want.append(arr(f_o.mul(f_o).into_power_basis().switch_down_to(o.context_at_level(2))))
        assert np.array_equal(low.download(), np.array(want, dtype=np.uint64))
        # pinned host memory + the async copies
        L = _lib.lib()
        hp = C.c_void_p()
        _lib.check(L.fhe_host_alloc(host.nbytes, C.byref(hp)))
        C.memmove(hp, host.ctypes.data, host.nbytes)
        e = fhe.DeviceArray(host.shape)
        e2 = fhe.DeviceArray(host.shape)
        _lib.check(L.fhe_buf_upload_async(C.c_void_p(e.data_ptr()), hp, host.nbytes, st.handle))
        _lib.check(L.fhe_buf_copy_async(C.c_void_p(e2.data_ptr()), C.c_void_p(e.data_ptr()), host.nbytes, st.handle))
        _lib.check(L.fhe_buf_zero_async(C.c_void_p(e.data_ptr()), host.nbytes, st.handle))
        back = np.empty_like(host)
        _lib.check(L.fhe_buf_download_async(back.ctypes.data_as(C.c_void_p), C.c_void_p(e2.data_ptr()), host.nbytes, st.handle))
        _lib.check(L.fhe_stream_sync(st.handle))
        assert np.array_equal(back, host) and not e.download().any()
        _lib.check(L.fhe_host_free(hp))
        free_b, total_b = fhe.device_mem_info(0)
        assert 0 < free_b <= total_b
        for a in (d, e, e2):
            a.free()
    st.destroy()
    assert L.fhe_buf_alloc(10 ** 6, 8, C.byref(C.c_void_p())) == -18   # no such device


def case_table_mismatch(fhe, nmod=3, n=16):
    """Handles built from different NTT tables must not be combined where Ntt-form rows cross between them
    (key switch: ct rows read as transforms under the key moduli; multiplicator: common rows of the extenders):
    same moduli but another primitive root -> ParameterMismatch instead of silently wrong ciphertexts."""
    opar, par = _params(fhe, nmod, n)
    ctx = par.context_at_level(0)
    base = OCtx(opar.moduli, n)
    other = OCtx(opar.moduli, n, psis=[pow(op.psi, 3, op.p.p) for op in base.ops])
    ctx2 = fhe.Context(opar.moduli, n, tables=oracle_tables(other))
    same = fhe.Context(opar.moduli, n, tables=oracle_tables(base))   # the same tables through another door: fine
    rng = random.Random(5)
    sk = obfv.SecretKey.random(opar, rng)
    c0, c0s, c1, c1s = ksk_arrays(obfv.RelinearizationKey(sk, rng).ksk)
    fhe.KeySwitchingKey(ctx, same, c0, c1, c0s, c1s)
    for ct_ctx, k_ctx in ((ctx, ctx2), (ctx2, ctx)):
        try:
            fhe.KeySwitchingKey(ct_ctx, k_ctx, c0, c1, c0s, c1s)
            raise AssertionError("key over different NTT tables accepted")
        except fhe.FheError as e:
            assert e.code == -11, e
    # a multiplicator whose relinearisation key lives over other tables than its level
    rk2 = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx2, ctx2, c0, c1, c0s, c1s))
    try:
        fhe.Multiplicator.default(par, rk2, 0)
        raise AssertionError("multiplicator with a key over different NTT tables accepted")
    except fhe.FheError as e:
        assert e.code == -11, e


NUMS = [1, 2, 3, 100, 1000, 4611686018326724610]
DENS = [1, 2, 3, 4, 100, 101, 1000, 1001, 4611686018326724610]


def case_scaler_grid(fhe, dev, n=16, pairs=None):
    """rq/scaler.rs:154-204 (all 54 numerator/denominator pairs, PowerBasis and Ntt), also the
    engine-derived constants vs the oracle's (rns/scaler.rs:79-229)."""
    x = Xfer(dev)
    rng = random.Random(12)
    of, ot = OCtx(Q3, n), OCtx(P3, n)
    cf, ct = fhe.Context(Q3, n), fhe.Context(P3, n)
    for num in NUMS:
        for den in DENS:
            if pairs is not None and (num, den) not in pairs:
                continue
            osc = OScaler(of, ot, ScalingFactor(num, den))
            sc = fhe.Scaler(cf, ct, num, den)
            s = osc.scaler
            assert sc.number_common_moduli == osc.number_common_moduli
            assert sc.constants(0).tolist() == s.gamma and sc.constants(1).tolist() == s.gamma_shoup
            assert sc.constants(2).tolist() == [v for r in s.omega for v in r]
            assert sc.constants(3).tolist() == [v for r in s.omega_shoup for v in r]
            assert sc.constants(4).tolist() == s.theta_omega_lo and sc.constants(5).tolist() == s.theta_omega_hi
            assert sc.constants(6).tolist() == [1 if v else 0 for v in s.theta_omega_sign]
            assert sc.constants(7).tolist() == s.theta_garner_lo and sc.constants(8).tolist() == s.theta_garner_hi
            assert sc.constants(9).tolist() == [s.theta_gamma_lo, s.theta_gamma_hi, 1 if s.theta_gamma_sign else 0,
                                                s.theta_garner_shift, 1 if s.scaling_factor.is_one else 0]
            p = rand_poly(of, POWER_BASIS, rng)
            both = np.stack([arr(p), arr(rand_poly(of, POWER_BASIS, rng))])
            got = x.back(sc.scale(x.to(both), ntt=False))
            assert np.array_equal(got[0], arr(osc.scale(p)))
            pn = p.into_ntt()
            assert np.array_equal(x.back(sc.scale(x.to(arr(pn)), ntt=True)), arr(osc.scale(pn)))
            # carry-chain extremes of the wide sums: every residue at its maximum, zero, and a mix
            ext = Poly(of, POWER_BASIS, [[(m - 1) if (i + r) % 3 else (0 if i % 2 else m - 1) for i in range(n)]
                                         for r, m in enumerate(of.moduli)])
            ext.coefficients[0][:4] = [of.moduli[0] - 1, 0, 1, of.moduli[0] // 2]
            for r in range(1, len(of.moduli)):
                ext.coefficients[r][:4] = [of.moduli[r] - 1, 0, 1, of.moduli[r] // 2]
            assert np.array_equal(x.back(sc.scale(x.to(arr(ext)), ntt=False)), arr(osc.scale(ext)))


def case_scaler_extend_and_constants_api(fhe, dev, n=16):
    """Common-prefix extension (rq/scaler.rs:35-43, 61-65), Switcher (rq/mod.rs:1101-1122) and
    the bring-your-own-constants constructor."""
    x = Xfer(dev)
    rng = random.Random(13)
    base, ext = MODULI5[1:3], MODULI5[1:3] + [MODULI5[4], Q3[0]]
    of, ot = OCtx(base, n), OCtx(ext, n)
    cf, ct = fhe.Context(base, n), fhe.Context(ext, n)
    osc = OScaler(of, ot, ScalingFactor.one())
    sc = fhe.Scaler(cf, ct, 7, 7)
    assert sc.number_common_moduli == 2 == osc.number_common_moduli
    pn = rand_poly(of, NTT, rng)
    want = arr(osc.scale(pn))
    assert np.array_equal(x.back(sc.scale(x.to(arr(pn)), ntt=True)), want)
    s = osc.scaler
    k = dict(gamma=s.gamma, gamma_shoup=s.gamma_shoup, omega=s.omega, omega_shoup=s.omega_shoup,
             theta_gamma_lo=s.theta_gamma_lo, theta_gamma_hi=s.theta_gamma_hi, theta_gamma_sign=s.theta_gamma_sign,
             theta_omega_lo=s.theta_omega_lo, theta_omega_hi=s.theta_omega_hi, theta_omega_sign=s.theta_omega_sign,
             theta_garner_lo=s.theta_garner_lo, theta_garner_hi=s.theta_garner_hi,
             theta_garner_shift=s.theta_garner_shift)
    sc2 = fhe.Scaler.from_constants(cf, ct, 2, True, k)
    assert np.array_equal(x.back(sc2.scale(x.to(arr(pn)), ntt=True)), want)
    # the kernel takes v's bits from limb 3 of the 256-bit sum on: a shift outside [97, 127] (never RnsScaler::new's
    # choice, scaler.rs:130-142) is refused, not mis-shifted
    for bad in (64, 96, 128):
        try:
            fhe.Scaler.from_constants(cf, ct, 2, True, dict(k, theta_garner_shift=bad))
            raise AssertionError("theta_garner_shift %d accepted" % bad)
        except fhe.FheError as err:
            assert err.code == -1, err.code
    a, b = OCtx(MODULI5[:2], n), OCtx(MODULI5[3:], n)
    ca, cb = fhe.Context(MODULI5[:2], n), fhe.Context(MODULI5[3:], n)
    p = rand_poly(a, POWER_BASIS, rng)
    assert np.array_equal(x.back(fhe.Switcher(ca, cb).switch(x.to(arr(p)))), arr(p.switch(OSwitcher(a, b))))
    try:
        fhe.Scaler(cf, fhe.Context(ext, 2 * n), 1, 1)
        raise AssertionError("degree mismatch accepted")
    except fhe.FheError as err:
        assert err.code == -7


PARAM_SIZES = None     # round 6: tests set this to run the parameter-set cases below on other modulus widths (the F64 kernels)


def _params(fhe, nmod, n, dev_needed=True):
    if PARAM_SIZES is not None:
        opar = obfv.BfvParameters(n, 1153 if n <= 64 else fhe.generate_prime(20, 2 * n, (1 << 20) - 1), moduli_sizes=list(PARAM_SIZES)[:nmod])
    else:
        opar = obfv.BfvParameters.default_arc(nmod, n)
    par = fhe.BfvParameters(n, opar.plaintext, moduli=opar.moduli)
    return opar, par


def case_params(fhe, nmod=3, n=16):
    """parameters.rs:560-738: per-level contexts, extended basis, mul params == oracle's."""
    opar, par = _params(fhe, nmod, n)
    assert fhe.generate_moduli([62] * nmod, n) == opar.moduli
    for level in range(nmod):
        assert par.context_at_level(level).moduli == opar.ctx[level].moduli
        assert par.mul_context_at_level(level).moduli == opar.mul_params[level].to.moduli
        e, d = par.extender(level), par.down_scaler(level)
        oe, od = opar.mul_params[level].extender.scaler, opar.mul_params[level].down_scaler.scaler
        assert e.number_common_moduli == nmod - level
        assert e.constants(2).tolist() == [v for r in oe.omega for v in r]
        assert d.constants(0).tolist() == od.gamma and d.constants(5).tolist() == od.theta_omega_hi
        assert d.constants(9).tolist() == [od.theta_gamma_lo, od.theta_gamma_hi, 1 if od.theta_gamma_sign else 0,
                                           od.theta_garner_shift, 0]


def case_mul_default_level_basis(fhe, nmod=3, n=16):
    """ops/mul.rs:101-138 vs parameters.rs:660-676: Multiplicator::default skips only the moduli of rk's level
    when it picks its 62-bit extension primes; with 62-bit ciphertext moduli a dropped top-level modulus is
    itself one of those primes, so at level > 0 the basis differs from the level's mul_params basis."""
    opar, par = _params(fhe, nmod, n)
    rng = random.Random(5)
    sk = obfv.SecretKey.random(opar, rng)
    differs = 0
    for level in range(nmod - 1):
        ork = obfv.RelinearizationKey(sk, rng, level, level)
        c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
        ctx = par.context_at_level(level)
        rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1, c0s, c1s))
        want = obfv.Multiplicator.default_extended_basis(opar, level)
        assert fhe.Multiplicator.default(par, rk, level).basis() == want
        # `&ct * &ct` (no key) uses the level's mul_params, as the reference does (ops/mod.rs:259-358)
        assert fhe.Multiplicator.default(par, None, level).basis() == opar.mul_params[level].to.moduli
        differs += want != opar.mul_params[level].to.moduli
    assert differs > 0, "the shape was meant to exercise differing bases"


def case_params_with_tables(fhe, dev, nmod=3, n=16):
    """fhe_params_create_with_tables: the host supplies NttOperator tables for every modulus (here built from a
    psi that is NOT the engine's default -- psi^3 is another primitive 2N-th root), and every Ntt-form result
    follows that evaluation order: a forward NTT equals the oracle's under the same tables, and the multiply
    pipeline run on ciphertexts transformed under those tables decrypts/compares like the default one."""
    from fhe_oracle import ntt as ontt
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import Modulus
    x = Xfer(dev)
    opar = obfv.BfvParameters.default_arc(nmod, n)
    seen = []

    def alt_op(modulus, degree):
        op = ontt.NttOperator(Modulus(modulus), degree)
        return ontt.NttOperator(Modulus(modulus), degree, psi=pow(op.psi, 3, modulus))

    def alt_tables(modulus, degree):
        seen.append(modulus)
        alt = alt_op(modulus, degree)
        return dict(omegas=alt.omegas, omegas_shoup=alt.omegas_shoup, zetas_inv=alt.zetas_inv,
                    zetas_inv_shoup=alt.zetas_inv_shoup, size_inv=alt.size_inv, size_inv_shoup=alt.size_inv_shoup)

    par = fhe.BfvParameters(n, opar.plaintext, moduli=opar.moduli, tables_fn=alt_tables)
    assert set(opar.moduli) <= set(seen) and len(set(seen)) > nmod   # ciphertext moduli and extension primes
    ref = fhe.BfvParameters(n, opar.plaintext, moduli=opar.moduli)
    ctx, rctx = par.context_at_level(0), ref.context_at_level(0)
    assert not np.array_equal(ctx.table(0), rctx.table(0))
    rng = random.Random(77)
    octx = OCtx(opar.moduli, n)
    pb = np.stack([arr(rand_poly(octx, POWER_BASIS, rng)) for _ in range(4)]).reshape(2, 2, nmod, n)
    # the same PowerBasis ciphertexts taken to Ntt form under each table set
    a_alt = x.back(ctx.ntt_forward(x.to(pb.copy())))
    a_ref = x.back(rctx.ntt_forward(x.to(pb.copy())))
    assert not np.array_equal(a_alt, a_ref)
    for r in range(nmod):   # explicit check of one row against the oracle operator with the alternative psi
        assert a_alt[0, 0, r].tolist() == alt_op(opar.moduli[r], n).forward([int(v) for v in pb[0, 0, r]])
    # multiply without relinearisation: PowerBasis results are psi-independent
    m_alt, m_ref = fhe.Multiplicator.default(par, None, 0), fhe.Multiplicator.default(ref, None, 0)
    o_alt = x.back(ctx.ntt_backward(m_alt.multiply(x.to(a_alt), x.to(a_alt))))
    o_ref = x.back(rctx.ntt_backward(m_ref.multiply(x.to(a_ref), x.to(a_ref))))
    assert np.array_equal(o_alt, o_ref)


def case_ksk_validation(fhe, nmod=3, n=16):
    """Key coefficients must be canonical and supplied Shoup twins must be the twins, in every creation path."""
    opar, par = _params(fhe, nmod, n)
    ctx = par.context_at_level(0)
    rng = random.Random(3)
    sk = obfv.SecretKey.random(opar, rng)
    c0, c0s, c1, c1s = ksk_arrays(obfv.RelinearizationKey(sk, rng).ksk)
    fhe.KeySwitchingKey(ctx, ctx, c0, c1, c0s, c1s)
    bad = c0.copy()
    bad[1, 2, 3] += np.uint64(opar.moduli[2])
    for args in ((bad, c1, None, None), (bad, c1, c0s, c1s), (c0, c1, c1s, c1s)):
        try:
            fhe.KeySwitchingKey(ctx, ctx, *args)
        except fhe.FheError as e:
            assert e.code == -1, e
        else:
            raise AssertionError("unreduced key / wrong twin accepted")


def case_key_switch_levels(fhe, dev, nmod=4, n=16):
    """key_switching_key.rs:241-320, 532-598 and relinearization_key.rs:219-273: every
    (ciphertext level, key level) pair; relinearizes incl. the switch_down_to fix-up."""
    x = Xfer(dev)
    rng = random.Random(14)
    opar, par = _params(fhe, nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    for ct_level in range(nmod - 1):
        for key_level in range(ct_level + 1):
            ork = obfv.RelinearizationKey(sk, rng, ct_level, key_level)
            c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
            cctx, kctx = par.context_at_level(ct_level), par.context_at_level(key_level)
            ksk = fhe.KeySwitchingKey(cctx, kctx, c0, c1, c0s if key_level % 2 else None, c1s if key_level % 2 else None)
            p = [rand_poly(opar.ctx[ct_level], POWER_BASIS, rng) for _ in range(3)]
            g0, g1 = ksk.key_switch(x.to(np.stack([arr(q) for q in p])))
            g0, g1 = x.back(g0), x.back(g1)
            for i, q in enumerate(p):
                w0, w1 = ork.ksk.key_switch(q)
                assert np.array_equal(g0[i], arr(w0)) and np.array_equal(g1[i], arr(w1))
            # relinearizes on 2 three-part ciphertexts
            cts = []
            for _ in range(2):
                a = sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng, ct_level)
                b = sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng, ct_level)
                cts.append(a.mul(b))
            inp = np.stack([ct_arr(c) for c in cts])
            got = x.back(fhe.RelinearizationKey(ksk).relinearizes(x.to(inp)))
            for i, c in enumerate(cts):
                ork.relinearizes(c)
                assert np.array_equal(got[i], ct_arr(c))


def case_key_switch_many_digits(fhe, dev, n=128, shapes=((50, 10), (58, 17), (62, 5), (60, 9), (45, 2), (61, 3))):
    """key_switching_key.rs:241-270 with long digit loops: the fused kernels go through the digits two per round
    (+ one single round for an odd count) and, for moduli below 2^60, leave their accumulators unreduced for up to
    eight digits before folding them back -- 9, 10 and 17 digits cross that fold once and twice.  Random residues
    and random key polynomials (the key switch is a fixed function of them) against the C oracle."""
    x = Xfer(dev)
    for bits, L in shapes:
        q = obfv.generate_moduli([bits] * L, n)
        octx = OCtx(q, n)
        cc = coracle.CCtx(octx)
        ctx = fhe.Context(q, n)
        seed = 1000 + bits * 100 + L
        c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(L)])
        c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(L)])
        ck = coracle.CKsk(c0, np.stack([cc.shoup(v) for v in c0]), c1, np.stack([cc.shoup(v) for v in c1]), cc, cc)
        ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
        p = np.stack([cc.synth_poly(seed, 1 + b, 0) for b in range(3)])
        g0, g1 = ksk.key_switch(x.to(p))
        g0, g1 = x.back(g0), x.back(g1)
        for b in range(3):
            w0, w1 = ck.key_switch(p[b])
            assert np.array_equal(g0[b], w0) and np.array_equal(g1[b], w1), (bits, L, b)
        # relinearise (the key switch adds into c0, c1 on the way out) and a rotation
        ct3 = np.stack([np.stack([cc.synth_poly(seed, 10 + b, part) for part in range(3)]) for b in range(2)])
        got = x.back(fhe.RelinearizationKey(ksk).relinearizes(x.to(ct3)))
        rot = x.back(fhe.GaloisKey(ksk, 3).relinearize(x.to(np.ascontiguousarray(ct3[:, :2]))))
        for b in range(2):
            k0, k1 = ck.key_switch(cc.poly_ntt_backward(ct3[b, 2]))
            assert np.array_equal(got[b], np.stack([cc.poly_add(ct3[b, 0], k0), cc.poly_add(ct3[b, 1], k1)])), (bits, L, b)
            assert np.array_equal(rot[b], ck.galois_relinearize(3, ct3[b, :2])), (bits, L, b)


def case_random_from_seed(fhe, dev, n=64):
    """rq/mod.rs:276-292 + bfv/ciphertext.rs:287-302: the seeded c1 of a wire ciphertext, against the oracle's
    restatement (SHA-256 -> ChaCha8 -> Lemire rejection per residue row), with moduli whose rejection rate runs
    from ~0 (a 60-bit NTT prime) over ~2 % (a 61-bit prime far from a power of two) to a 20-bit modulus, so that
    rejected draws shift the stream inside and across rows; several seeds per call."""
    from fhe_oracle import seeded
    from fhe_oracle.zq import generate_prime
    x = Xfer(dev)
    for moduli in ([generate_prime(60, 2 * n, 1 << 60)],
                   [generate_prime(61, 2 * n, 0x1400000000000000), generate_prime(60, 2 * n, 1 << 60),
                    generate_prime(20, 2 * n, 1 << 20), generate_prime(62, 2 * n, 1 << 62)]):
        ctx = fhe.Context(moduli, n)
        seeds = np.array([[(7 * b + i) & 0xFF for i in range(32)] for b in range(5)], dtype=np.uint8)
        got = x.back(ctx.random_from_seed(x.to_bytes(seeds)))
        for b in range(5):
            want = np.array(seeded.random_from_seed(moduli, n, bytes(seeds[b])), dtype=np.uint64)
            assert np.array_equal(got[b], want), (moduli, b)
    assert ctx.random_from_seed(x.to_bytes(np.zeros((0, 32), dtype=np.uint8))).shape[0] == 0


def case_key_switch_decomposition(fhe, dev, n=16):
    """key_switching_key.rs:323-362, 600-633 (single-modulus key level, base-2^k digits)."""
    x = Xfer(dev)
    rng = random.Random(15)
    opar, par = _params(fhe, 3, n)
    sk = obfv.SecretKey.random(opar, rng)
    octx = opar.ctx[2]
    frm = rand_poly(octx, POWER_BASIS, rng)
    oksk = obfv.KeySwitchingKey(sk, frm, 2, 2, rng)
    c0, c0s, c1, c1s = ksk_arrays(oksk)
    ctx = par.context_at_level(2)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1, log_base=oksk.log_base)
    p = rand_poly(octx, POWER_BASIS, rng)
    g0, g1 = ksk.key_switch(x.to(arr(p)))
    w0, w1 = oksk.key_switch(p)
    assert np.array_equal(x.back(g0), arr(w0)) and np.array_equal(x.back(g1), arr(w1))
    try:
        fhe.KeySwitchingKey(ctx, ctx, c0, c1)  # single modulus without log_base
        raise AssertionError("accepted")
    except fhe.FheError as err:
        assert err.code in (-11, -17)


def case_key_switch_decomposition_rows(fhe, dev, n, bits):
    """key_switching_key.rs:323-362 at full row sizes: a single-modulus key level, base-2^(log q / 2) digits taken from ONE
    residue row (the key switch's shift-and-mask loader, incl. the folded first stages of rows larger than LDS).  Expected
    value: the digits by numpy, transforms / products / sums by the C oracle."""
    from fhe_oracle.zq import generate_prime
    x = Xfer(dev)
    q = generate_prime(bits, 2 * n, 1 << bits)
    cc = coracle.CCtx(OCtx([q], n))
    seed = 0xF4E50099 + bits
    log_q = (q - 1).bit_length()
    log_base = log_q // 2
    nd = -(-log_q // log_base)
    c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(nd)])     # [nd][1][n]
    c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(nd)])
    p = np.stack([cc.synth_poly(seed, i, 0) for i in range(3)])                 # [3][1][n]
    ctx = fhe.Context([q], n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1, log_base=log_base)
    g0, g1 = ksk.key_switch(x.to(p))
    g0, g1 = x.back(g0), x.back(g1)
    mask = np.uint64((1 << log_base) - 1)
    for b in range(3):
        w0 = np.zeros((1, n), dtype=np.uint64)
        w1 = np.zeros((1, n), dtype=np.uint64)
        for i in range(nd):
            d = cc.poly_ntt_forward((p[b] >> np.uint64(i * log_base)) & mask)
            w0 = cc.poly_add(w0, cc.poly_mul(d, c0[i]))
            w1 = cc.poly_add(w1, cc.poly_mul(d, c1[i]))
        assert np.array_equal(g0[b], w0) and np.array_equal(g1[b], w1), (n, bits, b)


def case_galois(fhe, dev, nmod=3, n=16):
    """galois_key.rs:63-123, 186-256; evaluation_key.rs:110-170, 278-286."""
    x = Xfer(dev)
    rng = random.Random(16)
    opar, par = _params(fhe, nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    for ct_level, key_level in ((0, 0), (1, 0), (1, 1)):
        gks, ogks = [], {}
        for e in (3, pow(3, 2, 2 * n), 2 * n - 1):
            ogk = obfv.GaloisKey(sk, e, ct_level, key_level, rng)
            c0, c0s, c1, c1s = ksk_arrays(ogk.ksk)
            ksk = fhe.KeySwitchingKey(par.context_at_level(ct_level), par.context_at_level(key_level), c0, c1)
            gks.append(fhe.GaloisKey(ksk, e))
            ogks[e] = ogk
        ek = fhe.EvaluationKey(n, gks)
        cts = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng, ct_level) for _ in range(2)]
        inp = x.to(np.stack([ct_arr(c) for c in cts]))
        for got, e in ((ek.rotates_columns_by(inp, 1), 3), (ek.rotates_columns_by(inp, 2), pow(3, 2, 2 * n)),
                       (ek.rotates_rows(inp), 2 * n - 1)):
            got = x.back(got)
            for i, c in enumerate(cts):
                assert np.array_equal(got[i], ct_arr(ogks[e].relinearize(c)))
    try:
        fhe.GaloisKey(gks[0].ksk, 4)
        raise AssertionError("even exponent accepted")
    except fhe.FheError as err:
        assert err.code == -10


def case_multiply(fhe, dev, nmod=3, n=16, batch=3, level=0, chunk=0, streams=1):
    """ops/mul.rs:165-243, 263-367 and ops/mod.rs:259-358: Multiplicator::default with
    relinearisation (+/- modulus switching) and the `&ct * &ct` tensor without it; results
    equal the oracle bit for bit AND decrypt to the plaintext product."""
    x = Xfer(dev)
    rng = random.Random(17 + nmod + level)
    opar, par = _params(fhe, nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    ork = obfv.RelinearizationKey(sk, rng, level, level)
    c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
    ctx = par.context_at_level(level)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1, c0s, c1s))
    t = opar.plaintext
    A = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng, level) for _ in range(batch)]
    B = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng, level) for _ in range(batch)]
    lhs, rhs = x.to(np.stack([ct_arr(c) for c in A])), x.to(np.stack([ct_arr(c) for c in B]))
    for mod_switch in ([False, True] if level < opar.max_level() else [False]):
        om = obfv.Multiplicator.default(ork)
        if mod_switch:
            om.enable_mod_switching()
        m = fhe.Multiplicator.default(par, rk, level, mod_switch).set_chunk(chunk).set_streams(streams)
        assert m.options() == dict(chunk=chunk, streams=streams)
        got = x.back(m.multiply(lhs, rhs))
        for i in range(batch):
            want = om.multiply(A[i], B[i])
            assert np.array_equal(got[i], ct_arr(want)), (mod_switch, i)
    m3 = fhe.Multiplicator.default(par, None, level).set_chunk(chunk).set_streams(streams)
    got = x.back(m3.multiply(lhs, rhs))
    for i in range(batch):
        assert np.array_equal(got[i], ct_arr(A[i].mul(B[i])))


def case_multiply_square(fhe, dev, nmod=3, n=32, batch=5):
    """`&c1 * &c1` (the reference's bench ID "square", bfv/ops/mod.rs:259-358 squaring branch; benches/bfv.rs:232) on
    the fused pipeline: the SAME buffer as both operands takes the engine's squaring shortcut (operand extended
    once) and must equal the general multiplication of two equal ciphertexts and the oracle -- with and without
    relinearisation, with modulus switching."""
    x = Xfer(dev)
    opar, par = _params(fhe, nmod, n)
    rng = random.Random(77)
    sk = obfv.SecretKey.random(opar, rng)
    cts = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng) for _ in range(batch)]
    a = np.array([ct_arr(c) for c in cts], dtype=np.uint64)
    ork = obfv.RelinearizationKey(sk, rng)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, *ksk_arrays(ork.ksk)[::2]))
    for use_rk, ms in ((None, False), (rk, False), (rk, True)):
        m = fhe.Multiplicator.default(par, use_rk, 0, ms)
        om = obfv.Multiplicator.default(ork) if use_rk is not None else None
        da = x.to(a)
        sq = x.back(m.multiply(da, da))                       # same buffer: the shortcut
        gen = x.back(m.multiply(x.to(a), x.to(a.copy())))     # two buffers: the general pipeline
        assert np.array_equal(sq, gen)
        if om is not None and ms:
            om.enable_mod_switching()
        for i, c in enumerate(cts):
            want = ct_arr(om.multiply(c, c)) if om is not None else ct_arr(c.mul(c))
            assert np.array_equal(sq[i], want), (use_rk is not None, ms, i)


def case_multiply_host_sliced(fhe, nmod=3, n=32, batch=7):
    """fhe_bfv_mul on host pointers with a batch that goes through in slices (upload, pipeline and download of
    successive slices overlap on three internal streams; emulation build: slices of 2 pairs): every ciphertext against
    the oracle -- two operands, and one buffer as both (the squaring shortcut inside the slices)."""
    opar, par = _params(fhe, nmod, n)
    rng = random.Random(91)
    sk = obfv.SecretKey.random(opar, rng)
    ca = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng) for _ in range(batch)]
    cb = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng) for _ in range(batch)]
    a = np.array([ct_arr(c) for c in ca], dtype=np.uint64)
    b = np.array([ct_arr(c) for c in cb], dtype=np.uint64)
    ork = obfv.RelinearizationKey(sk, rng)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, *ksk_arrays(ork.ksk)[::2]))
    m = fhe.Multiplicator.default(par, rk, 0)
    om = obfv.Multiplicator.default(ork)
    out = m.multiply(a, b)
    sq = m.multiply(a, a)
    assert np.shares_memory(a, a)
    for i in range(batch):
        assert np.array_equal(out[i], ct_arr(om.multiply(ca[i], cb[i]))), i
        assert np.array_equal(sq[i], ct_arr(om.multiply(ca[i], ca[i]))), i
    # the other two host-pointer calls that go through in slices: relinearise (three parts in, two out) and a rotation
    m3 = fhe.Multiplicator.default(par, None, 0)
    t3 = m3.multiply(a, b)
    rel = rk.relinearizes(t3)
    ogk = obfv.GaloisKey(sk, 3, 0, 0, rng)
    gk = fhe.GaloisKey(fhe.KeySwitchingKey(ctx, ctx, *ksk_arrays(ogk.ksk)[::2]), 3)
    rot = gk.relinearize(a)
    for i in range(batch):
        c3 = ca[i].mul(cb[i])
        assert np.array_equal(t3[i], ct_arr(c3)), i
        ork.relinearizes(c3)
        assert np.array_equal(rel[i], ct_arr(c3)), i
        assert np.array_equal(rot[i], ct_arr(ogk.relinearize(ca[i]))), i


def case_multiply_custom_factors(fhe, dev, n=16):
    """ops/mul.rs:369-418 (`different_mul_strategy`): rhs pre-scaled by P/Q, post-scale t/P."""
    x = Xfer(dev)
    rng = random.Random(18)
    opar, par = _params(fhe, 3, n)
    sk = obfv.SecretKey.random(opar, rng)
    ork = obfv.RelinearizationKey(sk, rng)
    octx = opar.ctx[0]
    ext = obfv.extended_basis_primes(n, opar.moduli, 4)
    pprod = 1
    for v in ext:
        pprod *= v
    basis = ext  # the second strategy multiplies in the fresh basis P only
    # HPS "second strategy": lhs factor 1, rhs factor P/Q, post factor t/P, basis = q ++ P
    full = opar.moduli + ext
    om = obfv.Multiplicator(ScalingFactor.one(), ScalingFactor(pprod, octx.modulus()), full,
                            ScalingFactor(opar.plaintext, pprod), opar)
    om.enable_relinearization(ork)
    ctx = par.context_at_level(0)
    mctx = fhe.Context(full, n)
    el = fhe.Scaler(ctx, mctx, 1, 1)
    er = fhe.Scaler(ctx, mctx, pprod, octx.modulus())
    dn = fhe.Scaler(mctx, ctx, opar.plaintext, pprod)
    c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
    m = fhe.Multiplicator(el, er, dn, rk)
    t = opar.plaintext
    a = sk.encrypt([rng.randrange(t) for _ in range(n)], rng)
    b = sk.encrypt([rng.randrange(t) for _ in range(n)], rng)
    got = x.back(m.multiply(x.to(ct_arr(a)), x.to(ct_arr(b))))
    assert np.array_equal(got, ct_arr(om.multiply(a, b)))
    del basis


def case_dot_product_and_mul_plain(fhe, dev, n=16, count=7):
    """ops/dot_product.rs tests + rq/ops.rs:880-939 (`dot_product` == sum of products) and
    ops/mod.rs `ct * pt`: bit-exact vs the oracle and decrypts to sum_k m_k * p_k."""
    x = Xfer(dev)
    rng = random.Random(19)
    opar, par = _params(fhe, 3, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    ctx = par.context_at_level(0)
    cts = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng) for _ in range(count)]
    pts = [obfv.plaintext_poly_ntt(opar, [rng.randrange(t) for _ in range(n)]) for _ in range(count)]
    want = obfv.dot_product_scalar(cts, pts)
    C = np.stack([ct_arr(c) for c in cts])            # [count, 2, L, N]
    P = np.stack([arr(p) for p in pts])               # [count, L, N]
    got = x.back(ctx.dot_product_scalar(x.to(C), x.to(P)))
    assert np.array_equal(got, ct_arr(want))
    # batched: two dot products sharing the ciphertexts (the MulPIR server loop shape)
    P2 = np.stack([P, P[::-1].copy()])
    got2 = x.back(ctx.dot_product_scalar(x.to(C), x.to(P2)))
    assert np.array_equal(got2[0], ct_arr(want))
    assert np.array_equal(got2[1], ct_arr(obfv.dot_product_scalar(cts, pts[::-1])))
    # polynomial dot product (parts == 1)
    pp = x.back(ctx.dot_product_scalar(x.to(C[:, :1].copy()), x.to(P)))
    assert np.array_equal(pp[0], arr(obfv.poly_dot_product([c[0] for c in cts], pts)))
    # ct * pt, per-ciphertext and shared plaintext
    got3 = x.back(ctx.mul_plain(x.to(C), x.to(P)))
    for k in range(count):
        assert np.array_equal(got3[k], ct_arr(obfv.mul_plain(cts[k], pts[k])))
    got4 = x.back(ctx.mul_plain(x.to(C), x.to(P[0])))
    assert np.array_equal(got4[3], ct_arr(obfv.mul_plain(cts[3], pts[0])))
    try:
        ctx.dot_product_scalar(x.to(C[:0].copy()), x.to(P[:0].copy()))
        raise AssertionError("empty dot product accepted")
    except fhe.FheError as err:
        assert err.code == -19


def case_rgsw_and_inner_sum(fhe, dev, n=16):
    """rgsw_ciphertext.rs:122-156 (+ its tests :190-245) and evaluation_key.rs:56-100."""
    x = Xfer(dev)
    rng = random.Random(20)
    opar, par = _params(fhe, 3, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    for level in (0, 1):
        ctx = par.context_at_level(level)
        org = obfv.RGSWCiphertext(sk, [rng.randrange(t) for _ in range(n)], rng, level)
        ks = []
        for oksk in (org.ksk0, org.ksk1):
            c0, c0s, c1, c1s = ksk_arrays(oksk)
            ks.append(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
        rg = fhe.RGSWCiphertext(*ks)
        cts = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng, level) for _ in range(3)]
        got = x.back(rg.external_product(x.to(np.stack([ct_arr(c) for c in cts]))))
        for i, c in enumerate(cts):
            assert np.array_equal(got[i], ct_arr(org.external_product(c)))
    # inner sum at level 0
    ogks, gks, i = {}, [], 1
    seq = []
    while i < n // 2:
        seq.append(obfv.rot_to_gk_exponent(n, i))
        i *= 2
    seq.append(2 * n - 1)
    ctx = par.context_at_level(0)
    for e in seq:
        ogks[e] = obfv.GaloisKey(sk, e, 0, 0, rng)
        c0, c0s, c1, c1s = ksk_arrays(ogks[e].ksk)
        gks.append(fhe.GaloisKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1), e))
    ek = fhe.EvaluationKey(n, gks)
    cts = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng) for _ in range(2)]
    got = x.back(ek.computes_inner_sum(x.to(np.stack([ct_arr(c) for c in cts]))))
    for i, c in enumerate(cts):
        assert np.array_equal(got[i], ct_arr(obfv.inner_sum(c, ogks, n)))
    try:
        fhe.EvaluationKey(n, gks[:-1]).computes_inner_sum(x.to(ct_arr(cts[0])))
        raise AssertionError("inner sum without the row-rotation key accepted")
    except fhe.FheError as err:
        assert err.code == -11


def case_expand(fhe, dev, n=16, nmod=3):
    """evaluation_key.rs:192-256 and its tests :798-885: oblivious expansion.  The oracle result
    satisfies the reference test's closed form (ct_i decrypts to 2^level * v_i at x^0); the engine
    equals the oracle bit for bit, for power-of-two and ragged sizes, batched and unbatched."""
    x = Xfer(dev)
    rng = random.Random(23)
    opar, par = _params(fhe, nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    logn = n.bit_length() - 1
    for level in (0, 1):
        ctx = par.context_at_level(level)
        ogks, gks = {}, []
        for l in range(logn):
            e = (n >> l) + 1
            ogks[e] = obfv.GaloisKey(sk, e, level, level, rng)
            c0, c0s, c1, c1s = ksk_arrays(ogks[e].ksk)
            gks.append(fhe.GaloisKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1), e))
        ek = fhe.EvaluationKey(n, gks)
        assert ek.supports_expansion(logn) and not fhe.EvaluationKey(n, gks[:2]).supports_expansion(3)
        for size in ((1, 2, 5, n) if level == 0 else (3, 8)):
            lv = (size - 1).bit_length()
            v = [rng.randrange(t) for _ in range(1 << lv)]
            ct = sk.encrypt(v, rng, level)
            want = obfv.expands(ct, size, ogks)
            assert len(want) == size
            for vi, c in zip(v, want):                       # the reference test's closed form
                assert sk.decrypt(c) == [(vi << lv) % t] + [0] * (n - 1)
            got = x.back(ek.expands(x.to(ct_arr(ct)), size))
            assert got.shape == (size, 2, len(ctx.moduli), n)
            for i, c in enumerate(want):
                assert np.array_equal(got[i], ct_arr(c)), (level, size, i)
        # batched: [batch, 2, L, N] -> [size, batch, 2, L, N]
        cts = [sk.encrypt([rng.randrange(t) for _ in range(4)], rng, level) for _ in range(3)]
        got = x.back(ek.expands(x.to(np.stack([ct_arr(c) for c in cts])), 3))
        for b, c in enumerate(cts):
            for i, w in enumerate(obfv.expands(c, 3, ogks)):
                assert np.array_equal(got[i, b], ct_arr(w)), (level, b, i)
    ct = x.to(ct_arr(cts[0]))
    for size, code in ((0, -20), (n + 1, -20)):
        try:
            ek.expands(ct, size)
            raise AssertionError("invalid expansion size accepted")
        except fhe.FheError as err:
            assert err.code == code
    try:
        fhe.EvaluationKey(n, gks[:1]).expands(ct, 4)
        raise AssertionError("expansion without the needed Galois keys accepted")
    except fhe.FheError as err:
        assert err.code == -21


def case_wire_format(fhe, dev, n=32):
    """rq/convert.rs:17-147 + zq/mod.rs:783-793 + fhe-util lib.rs:71-148: bit-packed Rq payload.
    Engine bytes == oracle bytes for 62/61/51/20-bit moduli, PowerBasis and Ntt sources; decoding
    (incl. into_ntt on arrival) inverts it; the reference's round-trip tests (lib.rs:323-370)."""
    from fhe_oracle.rq import poly_to_wire, poly_from_wire, transcode_to_bytes, transcode_from_bytes
    x = Xfer(dev)
    rng = random.Random(29)
    assert transcode_from_bytes(transcode_to_bytes([1, 2, 3, 4], 4), 4)[:4] == [1, 2, 3, 4]   # lib.rs:357-362
    assert transcode_to_bytes([], 8) == b"" and transcode_from_bytes(b"", 8) == []             # lib.rs:365-371
    moduli = [obfv.generate_moduli([b], n)[0] for b in (62, 61, 51, 20)]
    o = OCtx(moduli, n)
    c = fhe.Context(moduli, n)
    assert c.serialized_size == sum(((m - 1).bit_length() * n) // 8 for m in moduli)
    polys = [rand_poly(o, POWER_BASIS, rng) for _ in range(3)]
    polys[0].coefficients = [[m - 1] * n for m in moduli]                 # all-ones bit patterns
    polys[1].coefficients = [[0] * n for _ in moduli]
    want = [poly_to_wire(p) for p in polys]
    pb = np.stack([arr(p) for p in polys])
    got = x.back_bytes(c.serialize(x.to(pb)))
    for g, w in zip(got, want):
        assert g.tobytes() == w
    ntt = [p.clone().into_ntt() for p in polys]
    got2 = x.back_bytes(c.serialize(x.to(np.stack([arr(p) for p in ntt])), from_ntt=True))
    assert all(g.tobytes() == w for g, w in zip(got2, want))
    data = np.stack([np.frombuffer(w, dtype=np.uint8) for w in want])
    assert np.array_equal(x.back(c.deserialize(x.to_bytes(data))), pb)
    back_ntt = x.back(c.deserialize(x.to_bytes(data), to_ntt=True))
    for i, w in enumerate(want):
        assert np.array_equal(back_ntt[i], arr(poly_from_wire(o, w, NTT)))
        assert np.array_equal(back_ntt[i], arr(ntt[i]))
    # arbitrary bytes decode like the reference (masking only): compare with the oracle
    junk = bytes(rng.randrange(256) for _ in range(c.serialized_size))
    dec = x.back(c.deserialize(x.to_bytes(np.frombuffer(junk, dtype=np.uint8)[None])))[0]
    assert np.array_equal(dec, arr(poly_from_wire(o, junk)))
    # a borrowed level context (drops the last modulus)
    c1, o1 = c.at_level(1), o.context_at_level(1)
    p1 = rand_poly(o1, POWER_BASIS, rng)
    w1 = poly_to_wire(p1)
    assert c1.serialized_size == len(w1)
    assert x.back_bytes(c1.serialize(x.to(arr(p1)[None])))[0].tobytes() == w1
    try:
        c.deserialize(x.to_bytes(data[:, :-1]))
        raise AssertionError("short payload accepted")
    except fhe.FheError as err:
        assert err.code == -1


def case_decrypt(fhe, dev, n=16, nmod=3):
    """secret_key.rs:198-247: the decrypt-side scaler.  2- and 3-part ciphertexts at levels 0 and 1;
    engine output == oracle SecretKey::decrypt == the encrypted values / their product."""
    x = Xfer(dev)
    rng = random.Random(31)
    opar, par = _params(fhe, nmod, n)
    assert par.plaintext_context().moduli == opar.plaintext_context.moduli
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    for level in (0, 1):
        s_ntt = x.to(arr(sk._s(opar.ctx[level])))
        vals = [[rng.randrange(t) for _ in range(n)] for _ in range(3)]
        cts = [sk.encrypt(v, rng, level) for v in vals]
        got = x.back(par.decrypt(s_ntt, x.to(np.stack([ct_arr(c) for c in cts])), level))
        for g, v, c in zip(got, vals, cts):
            assert g.tolist() == sk.decrypt(c) == v
        prod = cts[0].mul(cts[1])                               # 3 parts: phase uses s^2
        got3 = x.back(par.decrypt(s_ntt, x.to(ct_arr(prod)), level))
        assert got3.tolist() == sk.decrypt(prod)


def case_tensor_any_parts(fhe, dev, n=16, nmod=3):
    """ops/mod.rs:259-358: `&ct * &ct` for 1-, 2- and 3-part operands (incl. the squaring branch,
    which computes the same values); results equal the oracle and decrypt to the product."""
    x = Xfer(dev)
    rng = random.Random(37)
    opar, par = _params(fhe, nmod, n)
    sk = obfv.SecretKey.random(opar, rng)
    t = opar.plaintext
    m = fhe.Multiplicator.default(par, None, 0)
    va, vb, vc = ([rng.randrange(t) for _ in range(n)] for _ in range(3))
    A, B, C = (sk.encrypt(v, rng, 0) for v in (va, vb, vc))
    AB = A.mul(B)                                                     # 3 parts
    one = obfv.Ciphertext(opar, [A.c[0]], 0)                          # 1 part
    for lhs, rhs in ((A, B), (AB, C), (C, AB), (AB, AB), (A, A), (one, B), (one, one)):
        want = lhs.mul(rhs)
        got = x.back(m.tensor(x.to(ct_arr(lhs)[None]), x.to(ct_arr(rhs)[None])))[0]
        assert got.shape[0] == len(lhs) + len(rhs) - 1
        assert np.array_equal(got, ct_arr(want)), (len(lhs), len(rhs))
    # batched, and the 2 x 2 case agrees with the fused Multiplicator pipeline
    l2 = x.to(np.stack([ct_arr(A), ct_arr(B)]))
    r2 = x.to(np.stack([ct_arr(B), ct_arr(C)]))
    assert np.array_equal(x.back(m.tensor(l2, r2)), x.back(m.multiply(l2, r2)))
    # (a * b) * c decrypts to the triple product (4 parts, phase uses s^3)
    ABC = AB.mul(C)
    got = x.back(m.tensor(x.to(ct_arr(AB)[None]), x.to(ct_arr(C)[None])))[0]
    assert np.array_equal(got, ct_arr(ABC))
    s_ntt = x.to(arr(sk._s(opar.ctx[0])))
    assert x.back(par.decrypt(s_ntt, x.to(got), 0)).tolist() == sk.decrypt(ABC)


def case_extender_narrow_sums(fhe, dev, n=32):
    """Factor-one extension Q -> Q*P with 60-bit Q and 62-bit P (the C2 shape): every output sum
    stays below 2^(2k+1), so the engine takes the single-word Barrett branch of the scaler
    (engine.hpp narrow_mask); residues at their maxima included.  rq/scaler.rs:55-127."""
    x = Xfer(dev)
    rng = random.Random(41)
    opar = obfv.BfvParameters(n, 1153 if n <= 64 else obfv.generate_moduli([20], n)[0], moduli_sizes=[60] * 4)
    par = fhe.BfvParameters(n, opar.plaintext, moduli=opar.moduli)
    oext, ext = opar.mul_params[0].extender, par.extender(0)
    base = opar.ctx[0]
    polys = [rand_poly(base, POWER_BASIS, rng) for _ in range(3)]
    polys.append(Poly(base, POWER_BASIS, [[m - 1] * n for m in base.moduli]))
    polys.append(Poly(base, POWER_BASIS, [[(m - 1) if i % 2 else 0 for i in range(n)] for m in base.moduli]))
    got = x.back(ext.scale(x.to(np.stack([arr(p) for p in polys])), ntt=False))
    for g, p in zip(got, polys):
        assert np.array_equal(g, arr(oext.scale(p)))


def case_errors(fhe):
    """Error conventions (include/fhe_hip.h status codes <-> fhe_math::Error variants)."""
    def code(fn):
        try:
            fn()
        except fhe.FheError as e:
            return e.code
        return 0
    assert code(lambda: fhe.Context([], 16, device=-1)) == -14          # EmptyModuli
    assert code(lambda: fhe.Context([1 << 62], 16, device=-1)) == -3    # InvalidModulus
    assert code(lambda: fhe.Context(Q3, 12, device=-1)) == -4           # InvalidPolynomialDegree
    assert code(lambda: fhe.Context(Q3, 4, device=-1)) == -4
    assert code(lambda: fhe.Context([1153, 1153], 16, device=-1)) == -15  # NonCoprimeModuli
    assert code(lambda: fhe.Context([1153], 1024, device=-1)) == -5     # NttOperatorUnavailable
    c = fhe.Context(Q3, 16, device=-1)
    assert code(lambda: c.at_level(3)) == -12
    assert code(lambda: c.ntt_forward(np.zeros((3, 16), dtype=np.uint64))) == -18  # host-only handle
    assert fhe.generate_prime(62, 2 * 1048576, (1 << 62)) == 4611686018326724609
    assert fhe.generate_prime(11, 16, 1033) is None
    assert fhe.supports_opt(4611686018326724609) and not fhe.supports_opt(1153)
    assert fhe.is_prime(1153) and not fhe.is_prime(1155)
    assert code(lambda: c.random_from_seed(np.zeros((1, 32), dtype=np.uint8))) == -18


def case_option_errors(fhe):
    """Per-handle multiply options accept only 1 or 2 streams; a failing host table callback aborts the
    parameter set (needs a device or the emulator: parameter sets always carry device tables)."""
    def code(fn):
        try:
            fn()
        except fhe.FheError as e:
            return e.code
        return 0
    opar, par = _params(fhe, 2, 16)
    m = fhe.Multiplicator.default(par, None, 0)
    assert m.options() == dict(chunk=0, streams=2)
    assert code(lambda: m.set_streams(3)) == -1 and code(lambda: m.set_streams(0)) == -1
    assert m.set_streams(1).set_chunk(7).options() == dict(chunk=7, streams=1)
    # a host table callback that fails aborts the parameter set with NttOperatorUnavailable
    assert code(lambda: fhe.BfvParameters(16, opar.plaintext, moduli=opar.moduli, tables_fn=lambda q, n: 1 / 0)) == -5


def case_workspace_bounds(fhe, make_stream, kill_stream, nstreams=40, nmod=3, n=64, batch=4):
    """Round 4: the engine's process-global scratch is bounded.
    (a) Foreign streams -- created and destroyed by the host without telling the engine (`make_stream` / `kill_stream`:
        raw hipStreamCreate / hipStreamDestroy on the GPU, the emulator's stream table on CPU) -- each run one two-stream
        multiply (which also creates an internal second stream and its scratch) and vanish.  The engine cannot see a
        stream die (hipStreamQuery on a destroyed handle crashes in this runtime), so the bound is the LRU one: with
        `total_bytes` = two streams' footprint what it holds stays there, internal streams stay <= 32 + the tag
        streams, and the results stay bit-identical.
    (b) fhe_stream_destroy gives back the blocks of the stream's internal second stream too.
    (c) limits: idle blocks are evicted least-recently-used first; a call larger than the limit still runs."""
    import fhe_oracle.bfv as obfv_
    opar, par = _params(fhe, nmod, n)
    rng = random.Random(5)
    sk = obfv_.SecretKey.random(opar, rng)
    ork = obfv_.RelinearizationKey(sk, rng)
    c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
    ctx = par.context_at_level(0)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1, c0s, c1s))
    t = opar.plaintext
    A = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng) for _ in range(batch)]
    B = [sk.encrypt([rng.randrange(t) for _ in range(n)], rng) for _ in range(batch)]
    a_h, b_h = np.stack([ct_arr(c) for c in A]), np.stack([ct_arr(c) for c in B])
    m = fhe.Multiplicator.default(par, rk, 0).set_chunk(1).set_streams(2)      # 4 chunks on two streams
    om = obfv_.Multiplicator.default(ork)
    want = np.stack([ct_arr(om.multiply(A[i], B[i])) for i in range(batch)])
    fhe.workspace_trim()
    fhe.workspace_set_limit(0, 0)
    assert fhe.workspace_stats()["held_bytes"] == 0

    def one_call(stream):
        with stream:
            a, b = fhe.DeviceArray.from_numpy(a_h), fhe.DeviceArray.from_numpy(b_h)
            out = m.multiply(a, b)
            stream.synchronize()
            got = out.download()
            for x in (a, b, out):
                x.free()
        return got

    # (a) one stream's footprint, then many short-lived foreign streams under a bound of two footprints
    h0 = make_stream()
    assert np.array_equal(one_call(fhe.Stream.foreign(h0)), want)
    one = fhe.workspace_stats()
    assert one["held_bytes"] > 0 and one["owners"] == 2 and one["internal_streams"] >= 1, one
    kill_stream(h0)
    fhe.workspace_set_limit(0, 2 * one["held_bytes"])
    for i in range(nstreams):
        h = make_stream()
        assert np.array_equal(one_call(fhe.Stream.foreign(h)), want), i
        kill_stream(h)
        st = fhe.workspace_stats()
        assert st["in_use_bytes"] == 0 and st["held_bytes"] <= 2 * one["held_bytes"], (i, st, one)
        assert st["internal_streams"] <= 32 + 3, st
    # (b) an ABI-made stream takes everything it owned with it
    fhe.workspace_set_limit(0, 0)
    fhe.workspace_trim()
    s = fhe.Stream(0)
    assert np.array_equal(one_call(s), want)
    assert fhe.workspace_stats()["held_bytes"] > 0
    s.destroy()
    st = fhe.workspace_stats()
    assert (st["held_bytes"], st["in_use_bytes"], st["blocks"], st["owners"]) == (0, 0, 0, 0), st
    # (c) limits: total cap of one stream's footprint -> a second live stream evicts the first one's idle blocks
    s1, s2 = fhe.Stream(0), fhe.Stream(0)
    fhe.workspace_set_limit(0, one["held_bytes"])
    assert np.array_equal(one_call(s1), want)
    assert np.array_equal(one_call(s2), want)
    st = fhe.workspace_stats()
    assert st["held_bytes"] <= one["held_bytes"] and st["in_use_bytes"] == 0, st
    fhe.workspace_set_limit(1, 0)        # a per-stream cap below any block: nothing is retained, calls still run
    assert np.array_equal(one_call(s1), want)
    assert fhe.workspace_stats()["held_bytes"] == 0
    fhe.workspace_set_limit(0, 0)
    assert np.array_equal(one_call(s1), want)
    assert fhe.workspace_stats()["held_bytes"] > 0
    s1.destroy()
    s2.destroy()
    fhe.workspace_trim()
    fhe.workspace_set_limit()            # back to the defaults


def case_scaler_many_wide_moduli(fhe, dev, n=16, counts=(9, 24, 40), factors=((3, 7), (1, 46116860181065), (5, 1))):
    """non-unit factors over MANY 62-bit source moduli, where the fixed-point sum t of RnsScaler::scale
    (rns/scaler.rs:278-313) leaves the range in which its sign and w are what the mathematics says: the reference then
    returns what its bit tests (`t >> 191 > 0`, the low 128 bits of `t >> 126`) give, and so must the engine -- 9
    moduli stay on the fast instance, 24 and 40 take the one that reproduces those tests (scaler_upload's bound)."""
    from fhe_oracle.zq import generate_prime
    x = Xfer(dev)
    rng = random.Random(31)
    primes, up = [], 1 << 62
    while len(primes) < max(counts) + 3:
        up = generate_prime(62, 2 * n, up)
        primes.append(up)
    to_mods = primes[-3:]
    for cnt in counts:
        src = primes[:cnt]
        of, ot = OCtx(src, n), OCtx(to_mods, n)
        cf, ct = fhe.Context(src, n), fhe.Context(to_mods, n)
        for num, den in factors:
            osc = OScaler(of, ot, ScalingFactor(num, den))
            sc = fhe.Scaler(cf, ct, num, den)
            polys = [rand_poly(of, POWER_BASIS, rng) for _ in range(3)]
            polys.append(Poly(of, POWER_BASIS, [[m - 1] * n for m in src]))
            polys.append(Poly(of, POWER_BASIS, [[(m - 1) if (i + r) % 2 else 0 for i in range(n)] for r, m in enumerate(src)]))
            got = x.back(sc.scale(x.to(np.stack([arr(p) for p in polys])), ntt=False))
            for i, p in enumerate(polys):
                assert np.array_equal(got[i], arr(osc.scale(p))), (cnt, num, den, i)


# ------------------------------------------------------------------------------------------------ round 6: F64 kernels
def _f64_labels(fhe, fn):
    """Runs fn() with the library's launch profiler on; returns (result, set of launch labels)."""
    fhe.prof_reset()
    fhe.prof_enable(True)
    try:
        out = fn()
    finally:
        fhe.prof_enable(False)
    return out, set(fhe.prof_report())


def case_f64_ntt(fhe, dev, n, sizes=(50, 50, 49, 48, 44, 36), seed=11):
    """Moduli below 2^50 take the FP64-FMA instances of ntt_kernel (csrc/zq_f64.hpp; whole rows of 4096 ... 16384 points):
    forward and inverse transforms against the C oracle's butterflies (M/ntt/native.rs:142-233), on random residues and
    on the word patterns where a lost rounding would show (0, 1, p - 1, alternating), per launch class (a basis with a
    50-bit prime runs class 3, 49-bit class 4, the rest class 5); the same calls with the F64 option off (integer narrow
    kernels) give the same words, and the profiler says which kernels really ran."""
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    x = Xfer(dev)
    rng = np.random.default_rng(seed + n)
    assert fhe.get_f64()
    for group in ([s for s in sizes], [s for s in sizes if s <= 49], [s for s in sizes if s <= 48]):
        q = obfv.generate_moduli(group, n)
        cc = coracle.CCtx(OCtx(q, n))
        c = fhe.Context(q, n)
        qa = np.array(q, dtype=np.uint64)[:, None]
        rand = np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in q])
        alt = np.where(np.arange(n)[None, :] % 2 == 0, qa - 1, 0).astype(np.uint64)
        polys = np.stack([rand, np.broadcast_to(qa - 1, (len(q), n)).copy(), alt, np.ones((len(q), n), dtype=np.uint64),
                          np.zeros((len(q), n), dtype=np.uint64)])
        want = np.stack([cc.poly_ntt_forward(p) for p in polys])
        f, labels = _f64_labels(fhe, lambda: x.back(c.ntt_forward(x.to(polys))))
        assert np.array_equal(f, want), (n, group)
        assert "ntt_fwd_f64" in labels and "ntt_fwd" not in labels, labels
        b, labels = _f64_labels(fhe, lambda: x.back(c.ntt_backward(x.to(f))))
        assert np.array_equal(b, polys), (n, group)
        assert "ntt_inv_f64" in labels and "ntt_inv" not in labels, labels
        fhe.set_f64(False)
        try:
            f2, labels = _f64_labels(fhe, lambda: x.back(c.ntt_forward(x.to(polys))))
            assert np.array_equal(f2, want) and "ntt_fwd" in labels and "ntt_fwd_f64" not in labels, labels
            assert np.array_equal(x.back(c.ntt_backward(x.to(f2))), polys)
        finally:
            fhe.set_f64(True)
    # one modulus of 51 bits in the launch: integer kernels for the whole launch
    q = obfv.generate_moduli([51, 44], n)
    c = fhe.Context(q, n)
    p = np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in q])[None]
    f, labels = _f64_labels(fhe, lambda: x.back(c.ntt_forward(x.to(p))))
    assert "ntt_fwd" in labels and "ntt_fwd_f64" not in labels
    assert np.array_equal(x.back(c.ntt_backward(x.to(f))), p)


def case_f64_key_switch(fhe, dev, n, sizes, batch=2, seed=0xF4E5F640, exps=(3,), mode=1):
    """The fused key switch's F64 instances (every key modulus below 2^50, RNS digits): key_switch of arbitrary digit rows,
    relinearisation (own Ntt row + both addends) and Galois rotations (gathering loader) against the C oracle
    (F/bfv/keys/key_switching_key.rs:241-320, relinearization_key.rs:69-102, galois_key.rs:63-123), forced FUSED so that
    small batches reach it; then the same with the F64 option off.  `sizes` with more than ~10 moduli of 50 bits walk the
    accumulator fold.  mode = 2 (KS_UNFUSED): stage A's F64 instances (ks_ntt_kernel), stage B unchanged."""
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    import full_size
    x = Xfer(dev)
    q = obfv.generate_moduli(list(sizes), n)
    L = len(q)
    cc = coracle.CCtx(OCtx(q, n))
    ck = full_size.host_key(cc, seed, L)
    c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(L)])
    c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(L)])
    ctx = fhe.Context(q, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1).set_mode(mode)      # 1: KS_FUSED, 2: KS_UNFUSED
    lab_f64, lab_int = ("key_switch_fused_f64", "key_switch_fused") if mode == 1 else ("ks_digit_ntt_f64", "ks_digit_ntt")
    p = np.stack([np.stack([cc.synth_poly(seed, i, 0)]) for i in range(batch)])[:, 0]     # [batch][L][N], residues < q_i
    # extremes in the first polynomial: q_i - 1 everywhere in row 0, zeros in the last row
    p[0, 0, :] = np.uint64(q[0] - 1)
    p[0, L - 1, :] = 0
    ct3 = np.stack([np.stack([cc.synth_poly(seed, i, part) for part in range(3)]) for i in range(batch)])
    rk = fhe.RelinearizationKey(ksk)
    for f64 in (True, False):
        fhe.set_f64(f64)
        try:
            (g0, g1), labels = _f64_labels(fhe, lambda: ksk.key_switch(x.to(p)))
            g0, g1 = x.back(g0), x.back(g1)
            assert (lab_f64 in labels) == f64 and (lab_int in labels) == (not f64), (labels, f64)
            for i in range(batch):
                w0, w1 = ck.key_switch(p[i])
                assert np.array_equal(g0[i], w0) and np.array_equal(g1[i], w1), (n, sizes, f64, i)
            got, labels = _f64_labels(fhe, lambda: x.back(rk.relinearizes(x.to(ct3))))
            assert (lab_f64 in labels) == f64, (labels, f64)
            for i in range(batch):
                k0, k1 = ck.key_switch(cc.poly_ntt_backward(ct3[i, 2]))
                want = np.stack([cc.poly_add(ct3[i, 0], k0), cc.poly_add(ct3[i, 1], k1)])
                assert np.array_equal(got[i], want), (n, sizes, f64, "relinearize", i)
            for e in exps:
                r, labels = _f64_labels(fhe, lambda: x.back(fhe.GaloisKey(ksk, e).relinearize(x.to(ct3[:, :2].copy()))))
                assert (lab_f64 in labels) == f64, (labels, f64)
                for i in range(batch):
                    assert np.array_equal(r[i], ck.galois_relinearize(e, ct3[i, :2])), (n, sizes, f64, "rotate", e, i)
        finally:
            fhe.set_f64(True)
