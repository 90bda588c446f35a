"""Full-size parity helpers for the GPU suite: BASELINE.json's configs on synthetic
ciphertexts (uniform residues from the shared splitmix64 counter generator), the HIP engine
against the plain-C oracle (oracle/c/fhe_oracle.c, itself checked against the Python oracle
in tests/test_oracle_c.py)."""
import numpy as np

from fhe_oracle import bfv as obfv
from fhe_oracle import coracle, synth
from fhe_oracle.rns import ScalingFactor
from fhe_oracle.rq import Context as OCtx, Scaler as OScaler
from fhe_oracle.zq import generate_prime

_cache = {}


def plaintext_modulus(n):
    """crates/fhe/benches/bfv.rs:28 / parameters.rs:256-260: a 20-bit prime = 1 mod 2N."""
    return generate_prime(20, 2 * n, 1 << 20)


def oracle_level(n, q, t, level):
    """Oracle + C-oracle objects of one multiplication level (Multiplicator::default shape)."""
    key = (n, tuple(q), t, level)
    if key in _cache:
        return _cache[key]
    ql = q[: len(q) - level]
    sizes = [m.bit_length() for m in ql]
    n_ext = -(-(sum(sizes) + 60) // 62)
    ext = obfv.extended_basis_primes(n, ql, n_ext)
    base, mul = OCtx(ql, n), OCtx(ql + ext, n)
    el = OScaler(base, mul, ScalingFactor.one())
    dn = OScaler(mul, base, ScalingFactor(t, base.modulus()))
    cb, cm = coracle.CCtx(base), coracle.CCtx(mul)
    out = dict(base=base, mul=mul, cb=cb, cm=cm, cel=coracle.CScaler(el, cb, cm), cdn=coracle.CScaler(dn, cm, cb))
    _cache[key] = out
    return out


def u64(t):
    return t.cpu().numpy().view(np.uint64)


def device_key(ctx, seed, ndigits):
    """Synthetic key on the device: digit i uses generator parts 8+2i (c0) and 9+2i (c1)."""
    k = ctx.synth_uniform(seed, 0, 8, 2 * ndigits, 1)[0]          # [2*ndigits, Lk, N]
    k = k.reshape(ndigits, 2, ctx.nmoduli, ctx.degree)
    return k[:, 0].contiguous(), k[:, 1].contiguous()


def host_key(cctx, seed, ndigits):
    c0 = np.stack([cctx.synth_poly(seed, 0, 8 + 2 * i) for i in range(ndigits)])
    c1 = np.stack([cctx.synth_poly(seed, 0, 9 + 2 * i) for i in range(ndigits)])
    c0s = np.stack([cctx.shoup(x) for x in c0])
    c1s = np.stack([cctx.shoup(x) for x in c1])
    return coracle.CKsk(c0, c0s, c1, c1s, cctx, cctx)


def check_mul(fhe, n, sizes, batch, relin, cfg, sample=None, mod_switch=False, ct0=0, moduli=None):
    """ct0: index of the first ciphertext of this batch in the global synthetic stream (a rank's shard of a
    sharded batch starts at rank * batch_per_gpu, fhe.rs_amd/shard.py).  moduli: explicit primes instead of
    generate_moduli(sizes) (the reference's stock sets, tests/ref_params.py)."""
    import torch
    q = list(moduli) if moduli is not None else obfv.generate_moduli(sizes, n)
    t = plaintext_modulus(n)
    seed = synth.seed_for_config(cfg)
    par = fhe.BfvParameters(n, t, moduli=q)
    assert par.moduli == q
    ctx = par.context_at_level(0)
    o = oracle_level(n, q, t, 0)
    assert par.mul_context_at_level(0).moduli == o["mul"].moduli
    rk, crk = None, None
    if relin:
        c0, c1 = device_key(ctx, seed, len(q))
        rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
        crk = host_key(o["cb"], seed, len(q))
    m = fhe.Multiplicator.default(par, rk, 0, mod_switch)
    lhs = ctx.synth_uniform(seed, ct0, 0, 2, batch)
    rhs = ctx.synth_uniform(seed, ct0, 2, 2, batch)
    out = m.multiply(lhs, rhs)
    torch.cuda.synchronize()
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], crk, mod_switch)
    for i in (sample or range(batch)):
        l = np.stack([o["cb"].synth_poly(seed, ct0 + i, 0), o["cb"].synth_poly(seed, ct0 + i, 1)])
        r = np.stack([o["cb"].synth_poly(seed, ct0 + i, 2), o["cb"].synth_poly(seed, ct0 + i, 3)])
        assert np.array_equal(u64(lhs[i]), l), "device generator != oracle generator"
        want = cm.multiply(l, r)
        assert np.array_equal(u64(out[i]), want), f"ciphertext {i} differs from the oracle"


def check_mul_host(fhe, n, sizes, batch, relin, cfg, mod_switch=False):
    """check_mul through the host-pointer entry points (numpy in / out): what the CPU-side emulation of
    the kernel sources can run."""
    q = obfv.generate_moduli(sizes, n)
    t = plaintext_modulus(n)
    seed = synth.seed_for_config(cfg)
    par = fhe.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    o = oracle_level(n, q, t, 0)
    rk, crk = None, None
    if relin:
        crk = host_key(o["cb"], seed, len(q))
        rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, crk.c0, crk.c1))
    m = fhe.Multiplicator.default(par, rk, 0, mod_switch)
    lhs = np.stack([np.stack([o["cb"].synth_poly(seed, i, 0), o["cb"].synth_poly(seed, i, 1)]) for i in range(batch)])
    rhs = np.stack([np.stack([o["cb"].synth_poly(seed, i, 2), o["cb"].synth_poly(seed, i, 3)]) for i in range(batch)])
    out = m.multiply(lhs, rhs)
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], crk, mod_switch)
    for i in range(batch):
        assert np.array_equal(np.asarray(out[i]).view(np.uint64), cm.multiply(lhs[i], rhs[i])), f"ciphertext {i} differs"


def check_batch_properties(fhe, n, sizes, batch, cfg):
    import torch
    q = obfv.generate_moduli(sizes, n)
    t = plaintext_modulus(n)
    seed = synth.seed_for_config(cfg)
    par = fhe.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = device_key(ctx, seed, len(q))
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
    m = fhe.Multiplicator.default(par, rk, 0)
    lhs = ctx.synth_uniform(seed, 0, 0, 2, batch)
    rhs = ctx.synth_uniform(seed, 0, 2, 2, batch)
    out = m.multiply(lhs, rhs)
    # (1) batch consistency: any pair alone == the same pair inside the batch (chunking/strides)
    for i in (0, batch // 3, batch - 1):
        one = m.multiply(lhs[i:i + 1].contiguous(), rhs[i:i + 1].contiguous())
        assert torch.equal(one[0], out[i])
    # (2) commutativity of the tensor+relin pipeline in its operands
    sw = m.multiply(rhs[:8].contiguous(), lhs[:8].contiguous())
    assert torch.equal(sw, out[:8])
    # (3) NTT round trip and linearity over the whole batch
    x = lhs.clone()
    ctx.ntt_backward(x)
    assert not torch.equal(x, lhs)
    ctx.ntt_forward(x)
    assert torch.equal(x, lhs)
    a, b = lhs.clone(), rhs.clone()
    ctx.ntt_backward(a)
    ctx.ntt_backward(b)
    s = ctx.add(a, b)                      # a += b (PowerBasis)
    ctx.ntt_forward(s)
    assert torch.equal(s, ctx.add(lhs.clone(), rhs))
    # (4) all outputs canonical
    mods = torch.tensor(q, dtype=torch.int64, device=out.device).view(1, 1, len(q), 1)
    assert bool((out < mods).all()) and bool((out >= 0).all())


def check_relin_rotate(fhe, n, sizes, batch, cfg, sample=None):
    import torch
    q = obfv.generate_moduli(sizes, n)
    seed = synth.seed_for_config(cfg)
    L = len(q)
    ctx = fhe.Context(q, n)
    octx = OCtx(q, n)
    cc = coracle.CCtx(octx)
    c0, c1 = device_key(ctx, seed, L)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
    ck = host_key(cc, seed, L)
    ct3 = ctx.synth_uniform(seed, 0, 0, 3, batch)
    got = fhe.RelinearizationKey(ksk).relinearizes(ct3)
    rots = {e: fhe.GaloisKey(ksk, e).relinearize(ct3[:, :2].contiguous()) for e in (3, 2 * n - 1)}
    torch.cuda.synchronize()
    for i in (sample or sorted({0, batch - 1})):
        parts = [cc.synth_poly(seed, i, p) for p in range(3)]
        k0, k1 = ck.key_switch(cc.poly_ntt_backward(parts[2]))
        want = np.stack([cc.poly_add(parts[0], k0), cc.poly_add(parts[1], k1)])
        assert np.array_equal(u64(got[i]), want), f"relinearize {i}"
        for e, r in rots.items():
            assert np.array_equal(u64(r[i]), ck.galois_relinearize(e, np.stack(parts[:2]))), f"rotate e={e} ct {i}"


def check_relin_rotate_host(fhe, n, sizes, batch, cfg):
    """check_relin_rotate through the host-pointer entry points (numpy in / out), every ciphertext."""
    q = obfv.generate_moduli(sizes, n)
    seed = synth.seed_for_config(cfg)
    L = len(q)
    ctx = fhe.Context(q, n)
    cc = coracle.CCtx(OCtx(q, n))
    ck = host_key(cc, seed, L)
    ksk = fhe.KeySwitchingKey(ctx, ctx, ck.c0, ck.c1)
    ct3 = np.stack([np.stack([cc.synth_poly(seed, i, p) for p in range(3)]) for i in range(batch)])
    got = fhe.RelinearizationKey(ksk).relinearizes(ct3)
    ct2 = np.ascontiguousarray(ct3[:, :2])
    rots = {e: fhe.GaloisKey(ksk, e).relinearize(ct2) for e in (3, 2 * n - 1)}
    for i in range(batch):
        k0, k1 = ck.key_switch(cc.poly_ntt_backward(ct3[i, 2]))
        want = np.stack([cc.poly_add(ct3[i, 0], k0), cc.poly_add(ct3[i, 1], k1)])
        assert np.array_equal(np.asarray(got[i]).view(np.uint64), want), f"relinearize {i}"
        for e, r in rots.items():
            assert np.array_equal(np.asarray(r[i]).view(np.uint64), ck.galois_relinearize(e, ct2[i])), f"rotate e={e} ct {i}"


def check_random_shape_host(fhe, idx, big=False):
    """check_random_shape on host buffers (the emulation build of the kernel sources runs it without a GPU)."""
    n, sizes, batch = random_shape(idx, big)
    cfg = 100 + idx
    L = len(sizes)
    check_mul_host(fhe, n, sizes, batch, relin=L >= 2, cfg=cfg, mod_switch=L >= 2 and idx % 2 == 0)
    if L >= 2:
        check_relin_rotate_host(fhe, n, sizes, batch, cfg)


def check_chain(fhe, n, sizes, batch, levels, cfg):
    import torch
    q = obfv.generate_moduli(sizes, n)
    t = plaintext_modulus(n)
    seed = synth.seed_for_config(cfg)
    par = fhe.BfvParameters(n, t, moduli=q)
    cur = None
    want = None
    for level in range(levels):
        o = oracle_level(n, q, t, level)
        ctx = par.context_at_level(level)
        Ll = ctx.nmoduli
        c0, c1 = device_key(ctx, seed + level, Ll)
        rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
        m = fhe.Multiplicator.default(par, rk, level, mod_switch=True)
        crk = host_key(o["cb"], seed + level, Ll)
        cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], crk, True)
        rhs = ctx.synth_uniform(seed + level, 0, 2, 2, batch)
        if cur is None:
            cur = ctx.synth_uniform(seed, 0, 0, 2, batch)
            want = [np.stack([o["cb"].synth_poly(seed, i, 0), o["cb"].synth_poly(seed, i, 1)]) for i in range(batch)]
        cur = m.multiply(cur, rhs)
        torch.cuda.synchronize()
        for i in range(batch):
            r = np.stack([o["cb"].synth_poly(seed + level, i, 2), o["cb"].synth_poly(seed + level, i, 3)])
            want[i] = cm.multiply(want[i], r)
            assert np.array_equal(u64(cur[i]), want[i]), f"level {level} ciphertext {i}"


def random_shape(idx, big=False):
    """A deterministic 'random' parameter shape: degree, modulus sizes (mixed widths), batch.  big: rows larger than LDS
    (N = 32768 / 65536, up to five moduli, up to three ciphertexts) -- the sub-block / part forms of every kernel."""
    import random
    rng = random.Random(0xC0FFEE + idx)
    if big == "f64":
        # round 6: shapes that take the FP64-FMA kernels (whole rows of 4096 ... 16384 points, every modulus below 2^50),
        # widths from VERDICT r05 #3's list and the launch-class boundaries (48 / 49 / 50 bits), up to twelve digits
        n = 1 << rng.randrange(12, 15)
        L = rng.randrange(1, 13 if n == 4096 else 8)
        sizes = [rng.choice([27, 30, 36, 40, 43, 44, 47, 48, 49, 50]) for _ in range(L)]
        return n, sizes, rng.randrange(1, 5)
    if big == "f64wide":
        # round 6: N = 8192 launches of MORE than one workgroup per CU (256 CUs) -- the 512-thread x 16-coefficient F64
        # key-switch instance (engine.hpp launch_ks_fused), which batches of 1 ... 4 never reach
        L = rng.randrange(2, 8)
        sizes = [rng.choice([27, 30, 36, 40, 43, 44, 47, 48, 49, 50]) for _ in range(L)]
        return 8192, sizes, 256 // L + rng.randrange(1, 9)
    if big:
        n = 1 << rng.randrange(15, 17)
        L = rng.randrange(1, 6)
        sizes = [rng.choice([45, 54, 58, 60, 61, 62]) for _ in range(L)]
        return n, sizes, rng.randrange(1, 4)
    n = 1 << rng.randrange(5, 15)
    L = rng.randrange(1, 7)
    sizes = [rng.choice([36, 45, 50, 54, 58, 60, 61, 62]) for _ in range(L)]
    batch = rng.randrange(1, 6)
    return n, sizes, batch


def check_random_shape(fhe, idx, big=False):
    """Multiply (+relinearise, +modulus switch when the chain allows), relinearise and rotations on
    a pseudo-random parameter shape, every ciphertext against the C oracle."""
    n, sizes, batch = random_shape(idx, big)
    cfg = 100 + idx
    L = len(sizes)
    if big == "f64wide":   # (sampled ciphertexts: first, two inside, last)
        smp = sorted({0, batch // 3, 2 * batch // 3, batch - 1})
        check_relin_rotate(fhe, n, sizes, batch, cfg, sample=smp)
        if idx % 4 == 0:
            check_mul(fhe, n, sizes, batch, relin=True, cfg=cfg, sample=smp)
        return
    check_mul(fhe, n, sizes, batch, relin=L >= 2, cfg=cfg, mod_switch=L >= 2 and idx % 2 == 0)
    if L >= 2:
        check_relin_rotate(fhe, n, sizes, batch, cfg)
