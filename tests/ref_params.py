"""The reference's own stock parameter sets -- `BfvParameters::default_parameters_128`
(crates/fhe/src/bfv/parameters.rs:218-251), the sets every reference bench and example iterates over
(crates/fhe/benches/bfv.rs:27: `default_parameters_128(20)`) -- through the engine, for every operation the
bench file times on the hot path (benches/bfv.rs:167-286), each ciphertext against the plain-C oracle.

These are the only full-size constants the reference holds: explicit primes (not the "first prime below 2^bits"
of `generate_moduli`), of widths no other test uses (27, 54, 36/37, 43/44, 48/49 bits), with extended bases that mix
them with 62-bit primes (n = 8192: K = 5 + 5).  Inputs are the shared counter-based synthetic residues
(oracle/fhe_oracle/synth.py), keys are synthetic uniform polynomials (a key switch is a fixed function of them).

Shared by tests/test_emu_parity.py (kernel sources under host emulation, small batches) and
tests/test_gpu_parity.py (HIP build on the MI355X)."""
import numpy as np

from fhe_oracle import bfv as obfv
from fhe_oracle import coracle
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import Context as OCtx, Scaler as OScaler
from fhe_oracle.zq import generate_prime
from helpers import Xfer

# parameters.rs:222-251 (data: the moduli of the five sets)
DEFAULT_128 = {
    1024: [0x7e00001],
    2048: [0x3fffffff000001],
    4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
    8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
    16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
            0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001],
}
SEED = 0xF4E5D128          # synthetic-input seed of this family of tests (+ n)


def log_q(n):
    """The `log(q)` of the Criterion IDs (benches/bfv.rs:58: sum of moduli_sizes)."""
    return sum(int(m).bit_length() for m in DEFAULT_128[n])


def plaintext_modulus(n, bits=20):
    """default_parameters_128(20): generate_prime(20, 2n, 2^20 - 1) (parameters.rs:256-260)."""
    return generate_prime(bits, 2 * n, (1 << 64) - 1 >> (64 - bits))


_cache = {}


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).tobytes()).hexdigest()


def _note(digest, name, want):
    """digest: optional dict collecting SHA-256 of the ORACLE's outputs per operation (tests/golden/default128_digest.json
    freezes them: make_golden.py; test_golden.py compares a later run's with the committed ones)."""
    if digest is not None:       # (callers pass ciphertext 0 only: the digest does not depend on the batch size)
        digest[name] = sha(np.stack([np.asarray(w) for w in want]))


def level(n, lvl=0):
    """Oracle objects of one level of a stock set (Multiplicator::default shape; mul.rs:101-138 = parameters.rs:660-676
    when the key sits at the ciphertext's level and no ciphertext modulus is a 62-bit prime)."""
    key = (n, lvl)
    if key not in _cache:
        import full_size
        _cache[key] = full_size.oracle_level(n, DEFAULT_128[n], plaintext_modulus(n), lvl)
    return _cache[key]


def synth_ct(cc, seed, i, part0, nparts):
    return np.stack([cc.synth_poly(seed, i, part0 + p) for p in range(nparts)])


def synth_key(cc, seed, slot, ndigits=None):
    """A synthetic key-switching key [ndigits][Lk][N] x (c0, c1): generator 'ciphertext' index 1000 + slot."""
    nd = cc.L if ndigits is None else ndigits
    c0 = np.stack([cc.synth_poly(seed, 1000 + slot, 2 * i) for i in range(nd)])
    c1 = np.stack([cc.synth_poly(seed, 1000 + slot, 2 * i + 1) for i in range(nd)])
    ck = coracle.CKsk(c0, np.stack([cc.shoup(v) for v in c0]), c1, np.stack([cc.shoup(v) for v in c1]), cc, cc)
    return c0, c1, ck


FORCE_KS_MODE = None     # tests set this to run every key switch of the checks below in one strategy (fhe_ksk_set_mode)


def _ksk(fhe, ctx, c0, c1):
    k = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
    return k.set_mode(FORCE_KS_MODE) if FORCE_KS_MODE else k


def params(fhe, n):
    par = fhe.BfvParameters(n, plaintext_modulus(n), moduli=DEFAULT_128[n])
    assert par.moduli == DEFAULT_128[n]
    return par


def check_mul(fhe, dev, n, relin, mod_switch=False, batch=2, lvl=0, seed_off=0, digest=None):
    """`&ct * &ct` (benches "mul"; relin = False) and Multiplicator::default(rk).multiply ("mul_and_relin")."""
    x = Xfer(dev)
    par = params(fhe, n)
    o = level(n, lvl)
    cb = o["cb"]
    ctx = par.context_at_level(lvl)
    assert ctx.moduli == o["base"].moduli and par.mul_context_at_level(lvl).moduli == o["mul"].moduli
    seed = SEED + n + seed_off
    rk, ck = None, None
    if relin:
        c0, c1, ck = synth_key(cb, seed, lvl)
        rk = fhe.RelinearizationKey(_ksk(fhe, ctx, c0, c1))
    m = fhe.Multiplicator.default(par, rk, lvl, mod_switch)
    if relin:
        assert m.basis() == o["mul"].moduli
    lhs = np.stack([synth_ct(cb, seed, i, 0, 2) for i in range(batch)])
    rhs = np.stack([synth_ct(cb, seed, i, 2, 2) for i in range(batch)])
    got = x.back(m.multiply(x.to(lhs), x.to(rhs)))
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], ck, mod_switch)
    want = [cm.multiply(lhs[i], rhs[i]) for i in range(batch)]
    for i in range(batch):
        assert np.array_equal(got[i], want[i]), (n, relin, mod_switch, i)
    _note(digest, "mul_and_relin" if relin else "mul", want[:1])
    # "square" (benches/bfv.rs:232): the same buffer as both operands
    d = x.to(lhs)
    sq = x.back(m.multiply(d, d))
    wsq = cm.multiply(lhs[0], lhs[0])
    assert np.array_equal(sq[0], wsq), (n, "square")
    _note(digest, "square_relin" if relin else "square", [wsq])
    return got


def check_relin_rotate(fhe, dev, n, batch=2, digest=None):
    """"relinearize", "rotate_rows", "rotate_columns" (benches/bfv.rs:167-195)."""
    x = Xfer(dev)
    q = DEFAULT_128[n]
    cc = level(n)["cb"]
    ctx = params(fhe, n).context_at_level(0)
    seed = SEED + n + 1
    c0, c1, ck = synth_key(cc, seed, 0)
    rk = fhe.RelinearizationKey(_ksk(fhe, ctx, c0, c1))
    ct3 = np.stack([synth_ct(cc, seed, i, 0, 3) for i in range(batch)])
    got = x.back(rk.relinearizes(x.to(ct3)))
    gks, cks = [], {}
    for slot, e in ((1, 2 * n - 1), (2, 3)):
        g0, g1, gk = synth_key(cc, seed, slot)
        gks.append(fhe.GaloisKey(_ksk(fhe, ctx, g0, g1), e))
        cks[e] = gk
    ek = fhe.EvaluationKey(n, gks)
    ct2 = np.ascontiguousarray(ct3[:, :2])
    rows_ = x.back(ek.rotates_rows(x.to(ct2)))
    cols_ = x.back(ek.rotates_columns_by(x.to(ct2), 1))
    wr, wrows, wcols = [], [], []
    for i in range(batch):
        k0, k1 = ck.key_switch(cc.poly_ntt_backward(ct3[i, 2]))
        wr.append(np.stack([cc.poly_add(ct3[i, 0], k0), cc.poly_add(ct3[i, 1], k1)]))
        wrows.append(cks[2 * n - 1].galois_relinearize(2 * n - 1, ct2[i]))
        wcols.append(cks[3].galois_relinearize(3, ct2[i]))
        assert np.array_equal(got[i], wr[i]), (n, "relinearize", i)
        assert np.array_equal(rows_[i], wrows[i]), (n, "rotate_rows", i)
        assert np.array_equal(cols_[i], wcols[i]), (n, "rotate_columns", i)
    assert len(q) == ctx.nmoduli
    _note(digest, "relinearize", wr[:1])
    _note(digest, "rotate_rows", wrows[:1])
    _note(digest, "rotate_columns", wcols[:1])


def check_inner_sum(fhe, dev, n, batch=1, digest=None):
    """"inner_sum" (benches/bfv.rs:197-202; evaluation_key.rs:56-100): log2(n/2) column rotations + the row rotation."""
    x = Xfer(dev)
    cc = level(n)["cb"]
    ctx = params(fhe, n).context_at_level(0)
    seed = SEED + n + 2
    seq, i = [], 1
    while i < n // 2:
        seq.append(pow(3, i, 2 * n))
        i *= 2
    seq.append(2 * n - 1)
    gks, cks = [], {}
    for slot, e in enumerate(seq):
        g0, g1, gk = synth_key(cc, seed, slot)
        gks.append(fhe.GaloisKey(_ksk(fhe, ctx, g0, g1), e))
        cks[e] = gk
    ek = fhe.EvaluationKey(n, gks)
    cts = np.stack([synth_ct(cc, seed, b, 0, 2) for b in range(batch)])
    got = x.back(ek.computes_inner_sum(x.to(cts)))
    for b in range(batch):
        out = cts[b]
        for e in seq:
            tmp = cks[e].galois_relinearize(e, out)
            out = np.stack([cc.poly_add(out[0], tmp[0]), cc.poly_add(out[1], tmp[1])])
        assert np.array_equal(got[b], out), (n, "inner_sum", b)
        if b == 0:
            _note(digest, "inner_sum", [out])


def check_expand(fhe, dev, n, size=16, digest=None):
    """"expand_{i}" (benches/bfv.rs:204-218: `ek.expands(&c1, 1 << i)`, i <= 4 above n = 2048); evaluation_key.rs:192-256."""
    x = Xfer(dev)
    cc = level(n)["cb"]
    ctx = params(fhe, n).context_at_level(0)
    seed = SEED + n + 3
    lv = (size - 1).bit_length()
    gks, cks = [], {}
    for l in range(lv):
        e = (n >> l) + 1
        g0, g1, gk = synth_key(cc, seed, l)
        gks.append(fhe.GaloisKey(_ksk(fhe, ctx, g0, g1), e))
        cks[e] = gk
    ek = fhe.EvaluationKey(n, gks)
    ct = synth_ct(cc, seed, 0, 0, 2)
    got = x.back(ek.expands(x.to(ct), size))
    out = [None] * (1 << lv)
    out[0] = ct
    for l in range(lv):
        mono = np.zeros((cc.L, n), dtype=np.uint64)               # -x^(N - 2^l), evaluation_key.rs:467-474
        mono[:, n - (1 << l)] = np.array(cc.ctx.moduli, dtype=np.uint64) - np.uint64(1)
        mono = cc.poly_ntt_forward(mono)
        e = (n >> l) + 1
        step = 1 << l
        for i in range(step):
            sub = cks[e].galois_relinearize(e, out[i])
            j = step | i
            if j < size:
                out[j] = np.stack([cc.poly_mul(cc.poly_sub(out[i][p], sub[p]), mono) for p in range(2)])
            out[i] = np.stack([cc.poly_add(out[i][p], sub[p]) for p in range(2)])
    assert got.shape[0] == size
    for i in range(size):
        assert np.array_equal(got[i], out[i]), (n, "expand", size, i)
    _note(digest, "expand_%d" % lv, out[:size])


def second_strategy(n):
    """benches/bfv.rs:257-277 ("mul_and_relin_2"): extended basis = q ++ ceil(log q / 62) fresh 62-bit primes P;
    lhs factor 1, rhs factor P/Q, post-multiplication factor t/P."""
    key = (n, "mul2")
    if key in _cache:
        return _cache[key]
    q = DEFAULT_128[n]
    t = plaintext_modulus(n)
    nmoduli = -(-log_q(n) // 62)
    ext = obfv.extended_basis_primes(n, q, nmoduli)
    base, mul = OCtx(q, n), OCtx(q + ext, n)
    P = 1
    for v in ext:
        P *= v
    Q = base.modulus()
    cb, cm = coracle.CCtx(base), coracle.CCtx(mul)
    el = OScaler(base, mul, ScalingFactor.one())
    er = OScaler(base, mul, ScalingFactor(P, Q))
    dn = OScaler(mul, base, ScalingFactor(t, P))
    out = dict(q=q, ext=ext, P=P, Q=Q, t=t, cb=cb, cm=cm, cel=coracle.CScaler(el, cb, cm),
               cer=coracle.CScaler(er, cb, cm), cdn=coracle.CScaler(dn, cm, cb))
    _cache[key] = out
    return out


def check_mul2(fhe, dev, n, batch=2, digest=None):
    x = Xfer(dev)
    s = second_strategy(n)
    cb = s["cb"]
    par = params(fhe, n)
    ctx = par.context_at_level(0)
    mctx = fhe.Context(s["q"] + s["ext"], n)
    el = fhe.Scaler(ctx, mctx, 1, 1)
    er = fhe.Scaler(ctx, mctx, s["P"], s["Q"])
    dn = fhe.Scaler(mctx, ctx, s["t"], s["P"])
    seed = SEED + n + 4
    c0, c1, ck = synth_key(cb, seed, 0)
    rk = fhe.RelinearizationKey(_ksk(fhe, ctx, c0, c1))
    m = fhe.Multiplicator(el, er, dn, rk)
    lhs = np.stack([synth_ct(cb, seed, i, 0, 2) for i in range(batch)])
    rhs = np.stack([synth_ct(cb, seed, i, 2, 2) for i in range(batch)])
    got = x.back(m.multiply(x.to(lhs), x.to(rhs)))
    cm = coracle.CMul(s["cb"], s["cm"], s["cel"], s["cer"], s["cdn"], ck, False)
    want = [cm.multiply(lhs[i], rhs[i]) for i in range(batch)]
    for i in range(batch):
        assert np.array_equal(got[i], want[i]), (n, "mul_and_relin_2", i)
    _note(digest, "mul_and_relin_2", want[:1])


def check_chain(fhe, dev, n, batch=2, digest=None):
    """The leveled chain: multiply + relinearise + modulus switch (mul.rs:165-243 with enable_mod_switching,
    ciphertext.rs:148-161) from level 0 until one modulus is left, every level's output feeding the next."""
    x = Xfer(dev)
    par = params(fhe, n)
    L = len(DEFAULT_128[n])
    seed = SEED + n + 5
    cur, want = None, None
    for lvl in range(L - 1):
        o = level(n, lvl)
        cb = o["cb"]
        ctx = par.context_at_level(lvl)
        c0, c1, ck = synth_key(cb, seed, lvl)
        rk = fhe.RelinearizationKey(_ksk(fhe, ctx, c0, c1))
        m = fhe.Multiplicator.default(par, rk, lvl, True)
        assert m.basis() == o["mul"].moduli, lvl
        cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], ck, True)
        rhs = np.stack([synth_ct(cb, seed + 16 * (lvl + 1), i, 2, 2) for i in range(batch)])
        if cur is None:
            want = [synth_ct(cb, seed, i, 0, 2) for i in range(batch)]
            cur = x.to(np.stack(want))
        cur = m.multiply(cur, x.to(rhs))
        got = x.back(cur)
        for i in range(batch):
            want[i] = cm.multiply(want[i], rhs[i])
            assert np.array_equal(got[i], want[i]), (n, "chain level", lvl, i)
    assert got.shape[-2] == 1
    _note(digest, "chain_to_one_modulus", want[:1])


def check_all(fhe, dev, n, batch=2, expand_size=16, inner_sum=True, digest=None):
    """Every hot-path Criterion ID of benches/bfv.rs for one stock set."""
    check_mul(fhe, dev, n, relin=False, batch=batch, digest=digest)
    if len(DEFAULT_128[n]) == 1:
        return
    check_mul(fhe, dev, n, relin=True, batch=batch, digest=digest)
    check_relin_rotate(fhe, dev, n, batch=batch, digest=digest)
    if inner_sum:
        check_inner_sum(fhe, dev, n, digest=digest)
    check_expand(fhe, dev, n, expand_size, digest=digest)
    check_mul2(fhe, dev, n, batch=batch, digest=digest)
    check_chain(fhe, dev, n, batch=batch, digest=digest)


def table_digest(n):
    """SHA-256 of the NttOperator tables of the multiplication basis (oracle psi rule) of one stock set."""
    ops = level(n)["mul"].ops
    return dict(moduli=DEFAULT_128[n], mul_moduli=level(n)["mul"].moduli, plaintext=plaintext_modulus(n), psi=[op.psi for op in ops],
                omegas=sha([op.omegas for op in ops]), zetas_inv=sha([op.zetas_inv for op in ops]))


def cpu_port_ms(n, min_s=0.15):
    """Single-thread time of the plain-C port (oracle/c/fhe_oracle.c: the reference's algorithms and pass structure) for
    the hot-path Criterion IDs of benches/bfv.rs on one stock set, one ciphertext per call as Criterion times them.
    bench.py's cpu_baseline leg prints these beside the GPU's numbers.  ms per call."""
    import time
    o = level(n)
    cb = o["cb"]
    seed = SEED + n
    out = {}

    def timed(fn):
        fn()
        reps, t0 = 0, time.perf_counter()
        while reps < 2 or time.perf_counter() - t0 < min_s:
            fn()
            reps += 1
        return round((time.perf_counter() - t0) / reps * 1e3, 4)

    a, b = synth_ct(cb, seed, 0, 0, 2), synth_ct(cb, seed, 0, 2, 2)
    out["add_ct"] = timed(lambda: [cb.poly_add(a[p], b[p]) for p in range(2)])
    out["sub_ct"] = timed(lambda: [cb.poly_sub(a[p], b[p]) for p in range(2)])
    out["neg"] = timed(lambda: [cb.poly_neg(a[p]) for p in range(2)])
    plain = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], None, False)
    out["mul"] = timed(lambda: plain.multiply(a, b))
    out["square"] = timed(lambda: plain.multiply(a, a))
    if cb.L == 1:
        return out
    c0, c1, ck = synth_key(cb, seed, 0)
    c3 = plain.multiply(a, b)

    def relin(c):
        k0, k1 = ck.key_switch(cb.poly_ntt_backward(c[2]))
        return np.stack([cb.poly_add(c[0], k0), cb.poly_add(c[1], k1)])
    out["relinearize"] = timed(lambda: relin(c3))
    out["rotate_rows"] = timed(lambda: ck.galois_relinearize(2 * n - 1, a))
    out["rotate_columns"] = timed(lambda: ck.galois_relinearize(3, a))
    seq, i = [], 1
    while i < n // 2:
        seq.append(pow(3, i, 2 * n))
        i *= 2
    seq.append(2 * n - 1)

    def inner():
        cur = a
        for e in seq:
            tmp = ck.galois_relinearize(e, cur)
            cur = np.stack([cb.poly_add(cur[0], tmp[0]), cb.poly_add(cur[1], tmp[1])])
        return cur
    out["inner_sum"] = timed(inner)
    monos = []
    for l in range(4):
        mono = np.zeros((cb.L, n), dtype=np.uint64)
        mono[:, n - (1 << l)] = np.array(cb.ctx.moduli, dtype=np.uint64) - np.uint64(1)
        monos.append(cb.poly_ntt_forward(mono))

    def expand(lv):
        size = 1 << lv
        res = [None] * size
        res[0] = a
        for l in range(lv):
            e = (n >> l) + 1
            step = 1 << l
            for i in range(step):
                sub = ck.galois_relinearize(e, res[i])
                res[step | i] = np.stack([cb.poly_mul(cb.poly_sub(res[i][p], sub[p]), monos[l]) for p in range(2)])
                res[i] = np.stack([cb.poly_add(res[i][p], sub[p]) for p in range(2)])
        return res
    for lv in range(1, 5):
        out["expand_%d" % lv] = timed(lambda: expand(lv))
    out["mul_then_relinearize"] = timed(lambda: relin(plain.multiply(a, b)))
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], ck, False)
    out["mul_and_relin"] = timed(lambda: cm.multiply(a, b))
    s = second_strategy(n)
    cm2 = coracle.CMul(s["cb"], s["cm"], s["cel"], s["cer"], s["cdn"], ck, False)
    out["mul_and_relin_2"] = timed(lambda: cm2.multiply(a, b))
    return out
