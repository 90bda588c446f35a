"""The committed golden vectors of tests/golden/ (made by tests/golden/make_golden.py) against
every implementation in the tree: the Python oracle, the plain-C oracle, the kernel sources
under host emulation, and -- `gpu`-marked -- the HIP build on the MI355X.  None of these reads
/root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from fhe_oracle import bfv as obfv
from fhe_oracle import coracle
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import Context as OCtx, Poly, Scaler as OScaler, NTT, NTT_SHOUP

from helpers import load_engine, Xfer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).tobytes()).hexdigest()


def u(x):
    return np.array(x, dtype=np.uint64)


TRACES = load("bfv_small_traces.json")
RELIN_CASES = [k for k in TRACES if "relin_key" in TRACES[k]]


# ------------------------------------------------------------------ Python oracle ----
def _opar(g):
    par = obfv.BfvParameters(g["n"], g["plaintext"], moduli=g["moduli"])
    assert [op.psi for op in par.ctx[0].ops] == g["psi"]            # the oracle's psi rule is part of the fixture
    assert par.mul_params[0].to.moduli == g["mul_moduli"]
    assert [op.psi for op in par.mul_params[0].to.ops] == g["mul_psi"]
    return par


def _oct(par, rows_, level=0):
    ctx = par.context_at_level(level)
    return obfv.Ciphertext(par, [Poly(ctx, NTT, r) for r in rows_], level)


def _oksk(par, key):
    ctx = par.context_at_level(0)
    return obfv.KeySwitchingKey.from_parts(par, [Poly(ctx, NTT_SHOUP, r) for r in key["c0"]],
                                           [Poly(ctx, NTT_SHOUP, r) for r in key["c1"]], 0, 0)


def _rows(polys):
    return [[[int(v) for v in r] for r in p.coefficients] for p in polys]


@pytest.mark.parametrize("case", list(TRACES))
def test_python_oracle_reproduces_small_traces(case):
    g = TRACES[case]
    par = _opar(g)
    A, B = _oct(par, g["lhs"]), _oct(par, g["rhs"])
    assert _rows(A.mul(B).c) == g["tensor_product_3part"]
    if case not in RELIN_CASES:
        return
    ork = obfv.RelinearizationKey(ksk=_oksk(par, g["relin_key"]))
    trace = {}
    res = obfv.Multiplicator.default(ork).multiply(A, B, trace=trace)
    for stage in ("extended", "tensor", "scaled"):
        assert _rows(trace[stage]) == g[stage], stage
    assert _rows(res.c) == g["multiply_relin"]
    m2 = obfv.Multiplicator.default(ork)
    m2.enable_mod_switching()
    assert _rows(m2.multiply(A, B).c) == g["multiply_relin_modswitch"]
    ogk = obfv.GaloisKey(exponent=3, ksk=_oksk(par, g["galois_key_e3"]), par=par)
    assert _rows(ogk.relinearize(A).c) == g["rotate_e3"]


def test_scaler_constants_python_oracle():
    for name, g in load("scaler_constants.json").items():
        n, q, ext, t = g["n"], g["moduli"], g["ext"], g["plaintext"]
        assert obfv.generate_moduli([m.bit_length() for m in q], n) == q
        base, mul, pt = OCtx(q, n), OCtx(q + ext, n), OCtx([t], n) if (t - 1) % (2 * n) == 0 else None
        for key, s in (("extender", OScaler(base, mul, ScalingFactor.one())),
                       ("down_scaler", OScaler(mul, base, ScalingFactor(t, base.modulus())))):
            r = s.scaler
            assert [int(v) for v in r.gamma] == g[key]["gamma"], (name, key)
            assert r.theta_garner_shift == g[key]["theta_garner_shift"]
            assert [int(v) for v in r.omega[0]] == g[key]["omega_row0"]
            assert s.number_common_moduli == g[key]["number_common_moduli"]


# ----------------------------------------------------------------------- C oracle ----
def _c_level(g):
    base, mul = OCtx(g["moduli"], g["n"]), OCtx(g["mul_moduli"], g["n"])
    cb, cm = coracle.CCtx(base), coracle.CCtx(mul)
    el = coracle.CScaler(OScaler(base, mul, ScalingFactor.one()), cb, cm)
    dn = coracle.CScaler(OScaler(mul, base, ScalingFactor(g["plaintext"], base.modulus())), cm, cb)
    return cb, cm, el, dn


@pytest.mark.parametrize("case", RELIN_CASES)
def test_c_oracle_reproduces_small_traces(case):
    g = TRACES[case]
    cb, cm, el, dn = _c_level(g)
    c0, c1 = u(g["relin_key"]["c0"]), u(g["relin_key"]["c1"])
    ck = coracle.CKsk(c0, np.stack([cb.shoup(x) for x in c0]), c1, np.stack([cb.shoup(x) for x in c1]), cb, cb)
    got = coracle.CMul(cb, cm, el, el, dn, ck, False).multiply(u(g["lhs"]), u(g["rhs"]))
    assert np.array_equal(got, u(g["multiply_relin"]))
    got = coracle.CMul(cb, cm, el, el, dn, ck, True).multiply(u(g["lhs"]), u(g["rhs"]))
    assert np.array_equal(got, u(g["multiply_relin_modswitch"]))
    g0, g1 = u(g["galois_key_e3"]["c0"]), u(g["galois_key_e3"]["c1"])
    gk = coracle.CKsk(g0, np.stack([cb.shoup(x) for x in g0]), g1, np.stack([cb.shoup(x) for x in g1]), cb, cb)
    assert np.array_equal(gk.galois_relinearize(3, u(g["lhs"])), u(g["rotate_e3"]))


def test_c_oracle_reproduces_c2_digest():
    import full_size
    g = load("c2_digest.json")
    n, q, t, seed = g["n"], g["moduli"], g["plaintext"], g["seed"]
    assert obfv.generate_moduli([60] * 4, n) == q and full_size.plaintext_modulus(n) == t
    o = full_size.oracle_level(n, q, t, 0)
    assert o["mul"].moduli == g["mul_moduli"] and [op.psi for op in o["mul"].ops] == g["psi"]
    ops = o["mul"].ops
    assert sha([op.omegas for op in ops]) == g["tables_sha256"]["omegas"]
    assert sha([op.zetas_inv_shoup for op in ops]) == g["tables_sha256"]["zetas_inv_shoup"]
    cb = o["cb"]
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], full_size.host_key(cb, seed, len(q)), False)
    e = g["outputs"][0]
    lhs = np.stack([cb.synth_poly(seed, e["ct"], p) for p in (0, 1)])
    rhs = np.stack([cb.synth_poly(seed, e["ct"], p) for p in (2, 3)])
    assert sha(np.stack([lhs, rhs])) == e["input_sha256"]
    r = cm.multiply(lhs, rhs)
    assert sha(r) == e["output_sha256"] and np.array_equal(r[:, :, :64], u(e["head"]))


# ------------------------------------------------- engine (emulated here, HIP on the GPU) ----
def _engine_small_traces(fhe, dev, case):
    g = TRACES[case]
    x = Xfer(dev)
    par = fhe.BfvParameters(g["n"], g["plaintext"], moduli=g["moduli"])
    ctx, mctx = par.context_at_level(0), par.mul_context_at_level(0)
    assert mctx.moduli == g["mul_moduli"]
    lhs, rhs = x.to(u([g["lhs"]])), x.to(u([g["rhs"]]))
    # &ct * &ct
    assert np.array_equal(x.back(fhe.Multiplicator.default(par, None, 0).multiply(lhs, rhs))[0], u(g["tensor_product_3part"]))
    if case not in RELIN_CASES:
        return
    # stage by stage through the primitive entry points (Scaler::scale, Poly ops)
    ext = par.extender(0)
    e = [x.back(ext.scale(x.to(u(p)), True)) for p in g["lhs"] + g["rhs"]]
    assert np.array_equal(np.stack(e), u(g["extended"]))
    # Poly ops are the *Assign forms (rq/ops.rs:10-245): device operands are updated in place
    c = lambda i: x.to(e[i])
    t0 = mctx.mul(c(0), c(2))
    t1 = mctx.add(mctx.mul(c(0), c(3)), mctx.mul(c(1), c(2)))
    t2 = mctx.mul(c(1), c(3))
    assert np.array_equal(np.stack([x.back(t0), x.back(t1), x.back(t2)]), u(g["tensor"]))
    dn = par.down_scaler(0)
    sc = [x.back(dn.scale(x.to(u(p)), True)) for p in g["tensor"]]
    assert np.array_equal(np.stack(sc), u(g["scaled"]))
    # whole operation
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, u(g["relin_key"]["c0"]), u(g["relin_key"]["c1"])))
    assert np.array_equal(x.back(fhe.Multiplicator.default(par, rk, 0).multiply(lhs, rhs))[0], u(g["multiply_relin"]))
    assert np.array_equal(x.back(fhe.Multiplicator.default(par, rk, 0, True).multiply(lhs, rhs))[0],
                          u(g["multiply_relin_modswitch"]))
    gk = fhe.GaloisKey(fhe.KeySwitchingKey(ctx, ctx, u(g["galois_key_e3"]["c0"]), u(g["galois_key_e3"]["c1"])), 3)
    assert np.array_equal(x.back(gk.relinearize(lhs))[0], u(g["rotate_e3"]))


@pytest.mark.parametrize("case", list(TRACES))
def test_emulated_kernels_reproduce_small_traces(case):
    _engine_small_traces(load_engine("emu"), False, case)


def test_engine_host_setup_reproduces_scaler_constants():
    """RnsScaler::new (scaler.rs:79-229) runs on the host side of the engine: no device needed."""
    fhe = load_engine("emu")
    for name, g in load("scaler_constants.json").items():
        n, q, ext, t = g["n"], g["moduli"], g["ext"], g["plaintext"]
        base, mul = fhe.Context(q, n, device=-1), fhe.Context(q + ext, n, device=-1)
        for key, s in (("extender", fhe.Scaler(base, mul, 1, 1)),
                       ("down_scaler", fhe.Scaler(mul, base, t, int(np.prod([int(m) for m in q], dtype=object))))):
            flat = np.concatenate([s.constants(w) for w in range(10)])
            assert sha(flat) == g[key]["sha256"], (name, key)
            assert s.number_common_moduli == g[key]["number_common_moduli"]


def _default128(fhe, dev, n, batch):
    """The reference's stock sets (tests/ref_params.py): the oracle's outputs -- to which the engine's are compared on
    the way -- hash to the committed digests, whatever the batch size of the run."""
    import ref_params
    g = load("default128_digest.json")[str(n)]
    assert g["tables"] == json.loads(json.dumps(ref_params.table_digest(n))) and g["log_q"] == ref_params.log_q(n)
    assert g["tables"]["moduli"] == ref_params.DEFAULT_128[n]
    d = {}
    ref_params.check_all(fhe, dev, n, batch=batch, digest=d)
    assert d == g["outputs"]


@pytest.mark.parametrize("n", [4096, 8192])
def test_emulated_kernels_reproduce_default128_digest(n):
    _default128(load_engine("emu"), False, n, 1)


# ----------------------------------------------------------------------------- GPU ----
@pytest.fixture(scope="module")
def hip():
    eng = load_engine("hip")
    assert eng.device_count() >= 1, "no HIP device visible"
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("dev", [False, True])
@pytest.mark.parametrize("case", list(TRACES))
def test_hip_reproduces_small_traces(hip, case, dev):
    _engine_small_traces(hip, dev, case)


@pytest.mark.gpu
def test_hip_reproduces_c2_digest(hip):
    """BASELINE config C2 at full size: device-generated synthetic inputs, ct x ct + relinearize,
    SHA-256 of whole output ciphertexts against the committed digests."""
    import torch
    import full_size
    g = load("c2_digest.json")
    n, q, t, seed = g["n"], g["moduli"], g["plaintext"], g["seed"]
    par = hip.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    for which, name in ((0, "omegas"), (1, "omegas_shoup"), (2, "zetas_inv"), (3, "zetas_inv_shoup")):
        assert sha(par.mul_context_at_level(0).table(which)) == g["tables_sha256"][name], name
    c0, c1 = full_size.device_key(ctx, seed, len(q))
    assert sha(np.stack([full_size.u64(c0), full_size.u64(c1)])) == g["key_sha256"]
    mul = hip.Multiplicator.default(par, hip.RelinearizationKey(hip.KeySwitchingKey(ctx, ctx, c0, c1)), 0)
    batch = 1024
    lhs, rhs = ctx.synth_uniform(seed, 0, 0, 2, batch), ctx.synth_uniform(seed, 0, 2, 2, batch)
    out = mul.multiply(lhs, rhs)
    torch.cuda.synchronize()
    for e in g["outputs"]:
        i = e["ct"]
        assert sha(np.stack([full_size.u64(lhs[i]), full_size.u64(rhs[i])])) == e["input_sha256"]
        r = full_size.u64(out[i])
        assert sha(r) == e["output_sha256"], f"ciphertext {i}"
        assert np.array_equal(r[:, :, :64], u(e["head"])) and np.array_equal(r[:, :, -64:], u(e["tail"]))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 8192])
def test_hip_reproduces_default128_digest(hip, n):
    """n = 8192 / log q = 218 (and n = 4096 / log q = 109) of default_parameters_128: every hot-path Criterion ID, `_dev`
    entry points, batch 3; also the engine's own tables of the multiplication basis against the committed hashes."""
    g = load("default128_digest.json")[str(n)]
    par = hip.BfvParameters(n, g["tables"]["plaintext"], moduli=g["tables"]["moduli"])
    mctx = par.mul_context_at_level(0)
    assert mctx.moduli == g["tables"]["mul_moduli"]
    assert sha(mctx.table(0)) == g["tables"]["omegas"] and sha(mctx.table(2)) == g["tables"]["zetas_inv"]
    _default128(hip, True, n, 3)


@pytest.mark.gpu
def test_hip_product_decrypts(hip):
    """N=1024 secret-key example: the engine's ct x ct + relinearize equals the stored product and
    decrypts (oracle SecretKey, test side only) to the plaintext product."""
    meta = load("decrypt_n1024.json")
    z = np.load(os.path.join(GOLD, "decrypt_n1024.npz"))
    n, q, t = meta["n"], meta["moduli"], meta["plaintext"]
    par = hip.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    rk = hip.RelinearizationKey(hip.KeySwitchingKey(ctx, ctx, z["relin_c0"], z["relin_c1"]))
    got = hip.Multiplicator.default(par, rk, 0).multiply(z["lhs"][None], z["rhs"][None])[0]
    assert np.array_equal(np.asarray(got), z["product"]) and sha(got) == meta["product_sha256"]
    opar = obfv.BfvParameters(n, t, moduli=q)
    sk = obfv.SecretKey(opar, [int(v) for v in z["secret_key"]])
    oct_ = obfv.Ciphertext(opar, [Poly(opar.ctx[0], NTT, r.tolist()) for r in np.asarray(got)], 0)
    assert sk.decrypt(oct_) == [int(v) for v in z["plain_product"]]
