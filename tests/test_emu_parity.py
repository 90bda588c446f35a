"""CPU suite: the engine's host logic and its *kernel sources compiled for the host*
(tests/emu, fiber-emulated workgroups) against the oracle.  This validates indexing, barrier
placement, pipelines and the C ABI without a GPU; `tests/test_gpu_parity.py` runs the same
cases on the MI355X through the real HIP build."""
import pytest

import cases
from helpers import load_engine


@pytest.fixture(scope="module")
def fhe():
    return load_engine("emu")


def test_context_tables(fhe):
    cases.case_context_tables(fhe)


@pytest.mark.parametrize("n", [8, 16, 32, 64, 256, 1024])
def test_ntt(fhe, n):
    cases.case_ntt(fhe, False, n, batch=2 if n < 1024 else 1)


@pytest.mark.parametrize("n", [256, 1024])
def test_ntt_narrow_moduli(fhe, n):
    """Moduli below 2^60: the bound-tracked forward / inverse passes (radix-16 groups at these sizes), whose
    range checks trap in this build."""
    from fhe_oracle.zq import generate_prime
    mods = [generate_prime(60, 2 * n, 1 << 60), generate_prime(50, 2 * n, 1 << 50), generate_prime(36, 2 * n, 1 << 36)]
    cases.case_ntt(fhe, False, n, moduli=mods, batch=2, seed=5)


def test_ntt_explicit_tables(fhe):
    cases.case_ntt_explicit_tables(fhe, False)


def test_poly_ops(fhe):
    cases.case_poly_ops(fhe, False)


def test_product_extremes(fhe):
    cases.case_product_extremes(fhe, False)


def test_substitute(fhe):
    cases.case_substitute(fhe, False)


def test_switch_down(fhe):
    cases.case_switch_down(fhe, False)


def test_switch_down_to(fhe):
    cases.case_switch_down_to(fhe, False)


def test_switch_down_to_abi_buffers(fhe):
    with fhe.Stream(0):
        cases.case_switch_down_to(fhe, "abi")


def test_device_buffers_and_streams(fhe):
    cases.case_device_buffers(fhe)


def test_table_mismatch_is_rejected(fhe):
    cases.case_table_mismatch(fhe)


def test_scaler_grid(fhe):
    cases.case_scaler_grid(fhe, False)


def test_scaler_extend_switcher_constants(fhe):
    cases.case_scaler_extend_and_constants_api(fhe, False)


def test_params(fhe):
    cases.case_params(fhe)


def test_key_switch_levels(fhe):
    cases.case_key_switch_levels(fhe, False)


def test_key_switch_decomposition(fhe):
    cases.case_key_switch_decomposition(fhe, False)


def test_galois(fhe):
    cases.case_galois(fhe, False)


@pytest.mark.parametrize("nmod,level,chunk", [(2, 0, 0), (3, 0, 2), (3, 1, 0), (4, 1, 1)])
def test_multiply(fhe, nmod, level, chunk):
    cases.case_multiply(fhe, False, nmod=nmod, level=level, chunk=chunk)


def test_multiply_two_streams(fhe):
    """fhe_mul_set_streams(2): chunks alternate between the caller's stream and the internal one (5 pairs in chunks
    of 2 and 1: three and five chunks, odd counts included); same results."""
    cases.case_multiply(fhe, False, nmod=3, level=0, chunk=2, streams=2, batch=5)
    cases.case_multiply(fhe, False, nmod=2, level=0, chunk=1, streams=2, batch=5)


def test_internal_streams_are_parked_and_reused(fhe):
    """An internal second stream that the engine gives up (fhe_workspace_trim here) is parked and taken again by the next
    call that needs one -- a re-created stream can land on the caller's own hardware queue and serialise the two lanes of
    a multiply (profiles/r06_y_parked_streams_*_ab.jsonl).  White box on the emulator's stream table, which
    hands out consecutive one-byte slots: between two probes exactly ONE slot (the first probe's) has been taken if
    the second multiply created no stream."""
    def probe():
        s = fhe.Stream(0)
        v = s.handle.value
        s.destroy()
        return v
    a = fhe.Stream(0)
    with a:
        cases.case_multiply(fhe, "abi", nmod=2, n=16, batch=5, chunk=2, streams=2)
    assert fhe.workspace_stats()["internal_streams"] >= 1
    first = probe()
    fhe.workspace_trim()
    assert fhe.workspace_stats()["internal_streams"] == 0
    with a:
        cases.case_multiply(fhe, "abi", nmod=2, n=16, batch=5, chunk=2, streams=2)
    assert probe() - first == 1
    a.destroy()


@pytest.mark.parametrize("n", [128, 512])
def test_multiply_vector_tile_paths(fhe, n):
    """Sizes whose tiles take the 16-byte-chunk (CH > 0) load/MAC/store paths of the kernels."""
    cases.case_multiply(fhe, False, nmod=2, n=n, batch=2)


def test_galois_vector_tile_path(fhe):
    cases.case_galois(fhe, False, nmod=3, n=128)


@pytest.mark.parametrize("bits", [62, 60])
@pytest.mark.parametrize("n", [32768, 65536])
def test_ntt_split_kernels(fhe, n, bits):
    """N > 16384: global radix stages + LDS kernel on 8192-point sub-blocks (vs the C oracle); 60-bit moduli take the
    bound-tracked narrow passes in the LDS halves (round 4), whose range checks trap in this build."""
    from fhe_oracle import coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    mods = [generate_prime(bits, 2 * n, 1 << bits)]
    cases.case_ntt(fhe, False, n, moduli=mods, batch=1, coracle_ctx=coracle.CCtx(OCtx(mods, n)))


@pytest.mark.parametrize("n", [32768, 65536])
def test_multiply_split_rows(fhe, n):
    """N >= 32768: rows larger than LDS -- fused tensor + inverse NTT on 8192-point sub-blocks followed by the
    global inverse stages, split fused key switch, split NTTs -- against the C oracle."""
    import full_size
    full_size.check_mul_host(fhe, n=n, sizes=[60, 60], batch=1, relin=True, cfg=7)


def test_random_parameter_shapes(fhe):
    """48 pseudo-random shapes (degree 8..4096, 1-6 moduli of mixed widths 36..62 bits, ragged batches 1..11):
    multiply (+relinearise, +modulus switch on every other shape) against the C oracle.  Ragged batches hit
    the tails of the XCD-grouped grids; mixed widths hit both the narrow and the general transforms."""
    import random
    import full_size
    for idx in range(48):
        rng = random.Random(0xBEEF + idx)
        n = 1 << rng.randrange(3, 13)
        L = rng.randrange(1, 7)
        sizes = [rng.choice([36, 45, 50, 54, 58, 59, 60, 61, 62]) for _ in range(L)]
        batch = rng.randrange(1, 12)
        full_size.check_mul_host(fhe, n=n, sizes=sizes, batch=batch, relin=L >= 2, cfg=300 + idx,
                                 mod_switch=(L >= 2 and idx % 2 == 0))


def test_multiply_custom_factors(fhe):
    cases.case_multiply_custom_factors(fhe, False)


def test_dot_product_and_mul_plain(fhe):
    cases.case_dot_product_and_mul_plain(fhe, False)


def test_rgsw_and_inner_sum(fhe):
    cases.case_rgsw_and_inner_sum(fhe, False)


def test_expand(fhe):
    cases.case_expand(fhe, False)


@pytest.mark.parametrize("n", [32, 128, 1024])
def test_wire_format(fhe, n):
    """n = 32: the byte-granular kernels; n >= 128: the word-granular ones (rows are whole 16-byte words)."""
    cases.case_wire_format(fhe, False, n=n)


def test_tensor_any_parts(fhe):
    cases.case_tensor_any_parts(fhe, False)


def test_extender_narrow_sums(fhe):
    cases.case_extender_narrow_sums(fhe, False)


def test_decrypt(fhe):
    cases.case_decrypt(fhe, False)


def test_workspace_trim(fhe):
    cases.case_multiply(fhe, False, nmod=2)
    assert fhe.workspace_trim() > 0
    assert fhe.workspace_trim() == 0
    cases.case_multiply(fhe, False, nmod=2)


def test_mul_default_level_basis(fhe):
    cases.case_mul_default_level_basis(fhe)


def test_params_with_tables(fhe):
    cases.case_params_with_tables(fhe, False)


def test_ksk_validation(fhe):
    cases.case_ksk_validation(fhe)


def test_key_switch_many_digits(fhe):
    cases.case_key_switch_many_digits(fhe, False)


def test_random_from_seed(fhe):
    cases.case_random_from_seed(fhe, False)
    cases.case_random_from_seed(fhe, False, n=2048)


def test_errors(fhe):
    cases.case_errors(fhe)
    cases.case_option_errors(fhe)


@pytest.mark.parametrize("n", [8192, 16384, 32768, 65536])
def test_key_switch_large_rows(fhe, n):
    """N = 8192: the fused key switch keeps its c1 accumulators in LDS behind the row tile and
    prefetches the next digit; N = 16384: both accumulator sets in registers; N >= 32768: one
    workgroup per 16384-point part of a row with the first one / two stages folded into the loader
    (ks_fused_kernel<14, ..., G0>).  Synthetic key and input vs the C oracle."""
    import numpy as np
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    import full_size
    seed = 0xF4E50077
    q = obfv.generate_moduli([60, 60], n)
    cc = coracle.CCtx(OCtx(q, n))
    ck = full_size.host_key(cc, seed, len(q))
    c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(len(q))])
    c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(len(q))])
    ctx = fhe.Context(q, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
    p = np.stack([cc.synth_poly(seed, i, 0) for i in range(2)])     # [2][L][N], any residues < q_i
    g0, g1 = ksk.key_switch(p)
    for i in range(2):
        w0, w1 = ck.key_switch(p[i])
        assert np.array_equal(np.asarray(g0[i]), w0) and np.array_equal(np.asarray(g1[i]), w1)


def test_params_with_short_tables_fail_cleanly(fhe):
    """A host callback that returns a table shorter than `degree` must fail creation (NttOperatorUnavailable), not
    hand uninitialised memory to the device as twiddles."""
    from fhe_oracle.rq import Context as OCtx
    from helpers import oracle_tables
    import numpy as np

    def short(modulus, degree):
        t = oracle_tables(OCtx([modulus], degree))
        d = {k: np.array(v[0], dtype=np.uint64) for k, v in t.items() if k not in ("size_inv", "size_inv_shoup")}
        d["zetas_inv"] = d["zetas_inv"][: degree // 2]
        d["size_inv"], d["size_inv_shoup"] = t["size_inv"][0], t["size_inv_shoup"][0]
        return d
    try:
        fhe.BfvParameters(16, 1153, moduli_sizes=[50, 50], tables_fn=short)
        raise AssertionError("short host table accepted")
    except fhe.FheError as e:
        assert e.code == -5, e


def test_multiply_square_shortcut(fhe):
    cases.case_multiply_square(fhe, False)
    cases.case_multiply_host_sliced(fhe)
    with fhe.Stream(0):
        cases.case_multiply_square(fhe, "abi", batch=3)


# ---- round 4: the unfused key switch (ks_ntt_kernel + ks_mac_kernel), forced through fhe_ksk_set_mode ----
def _unfused(fhe, mode=None, w_budget=0):
    return fhe.KeySwitchingKey.forced_mode(fhe.KeySwitchingKey.UNFUSED if mode is None else mode, w_budget)


def test_unfused_key_switch_levels_and_digits(fhe):
    """Every (ciphertext level, key level) pair, long digit loops (more than sixteen terms: the accumulator is folded
    in between), Galois keys (the caller's Ntt rows stand in for one transform per key modulus), RGSW, inner sum."""
    with _unfused(fhe):
        cases.case_key_switch_levels(fhe, False)
        cases.case_key_switch_many_digits(fhe, False)
        cases.case_galois(fhe, False)
        cases.case_galois(fhe, False, nmod=3, n=128)
        cases.case_rgsw_and_inner_sum(fhe, False)
        cases.case_expand(fhe, False)


def test_unfused_small_w_budget(fhe):
    """A W budget of a few rows: groups of one key modulus over chunks of one polynomial (every loop of the host
    side, ragged tails included)."""
    with _unfused(fhe, w_budget=1):
        cases.case_key_switch_levels(fhe, False)
        cases.case_multiply(fhe, False, nmod=3, level=0, chunk=2, batch=5)
    with _unfused(fhe, w_budget=3 * 5 * 16 * 8):
        cases.case_key_switch_levels(fhe, False)
        cases.case_galois(fhe, False)


@pytest.mark.parametrize("nmod,level,chunk", [(2, 0, 0), (3, 1, 0), (4, 1, 1)])
def test_unfused_multiply(fhe, nmod, level, chunk):
    with _unfused(fhe):
        cases.case_multiply(fhe, False, nmod=nmod, level=level, chunk=chunk)
        cases.case_multiply(fhe, False, nmod=2, n=512, batch=2)


def test_multiply_forward_transform_rides_on_unfused_stage_a(fhe):
    """Round 5: when the multiply's key switch runs unfused at the ciphertext's level, the forward transform of (c0, c1)
    is done by extra workgroups of the key switch's stage-A launch (`KsExtraFwd`); when stage A is more than one launch
    (W budget of one row: one key modulus x one polynomial per launch) the rows get their own launch first.  Both,
    and the fused form, against the oracle -- general multiply, the squaring shortcut, with modulus switching, and the
    stock n = 4096 set (an LDS-row instance) through the C oracle."""
    import ref_params
    for w in (0, 1):
        with _unfused(fhe, w_budget=w):
            cases.case_multiply_square(fhe, False)
            cases.case_multiply(fhe, False, nmod=3, level=0, batch=2)
            ref_params.check_mul(fhe, False, 4096, relin=True, batch=2)
    with _unfused(fhe, mode=fhe.KeySwitchingKey.FUSED):
        cases.case_multiply_square(fhe, False)
        ref_params.check_mul(fhe, False, 4096, relin=True, batch=2)


def test_unfused_decomposition_keys_stay_fused(fhe):
    with _unfused(fhe):
        cases.case_key_switch_decomposition(fhe, False)


@pytest.mark.parametrize("n,mode", [(8192, 2), (16384, 2), (16384, 3), (32768, 2), (65536, 2), (16384, 1), (32768, 1), (65536, 1),
                                    (32768, 4), (65536, 4)])
def test_key_switch_strategies_large_rows(fhe, n, mode):
    """Unfused: whole-row tiles up to N = 16384, 8192-point sub-block tiles with the first stages folded into the loader
    above (and at 16384 in mode UNFUSED_SUB).  Fused (mode 1): whole rows up to 16384, 16384-point parts with one / two
    folded stages above.  Synthetic key and input vs the C oracle; 60-bit moduli (narrow passes, residue-row loader),
    62 + 61 bits (general passes, generic loader), 62 + 62 bits (general passes, residue-row loader)."""
    import numpy as np
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    import full_size
    seed = 0xF4E50078
    p62 = generate_prime(62, 2 * n, 1 << 62)
    for q in (obfv.generate_moduli([60, 60], n), [p62, generate_prime(61, 2 * n, 1 << 61)], [p62, generate_prime(62, 2 * n, p62)]):
        cc = coracle.CCtx(OCtx(q, n))
        ck = full_size.host_key(cc, seed, len(q))
        c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(len(q))])
        c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(len(q))])
        ctx = fhe.Context(q, n)
        with _unfused(fhe, mode):
            ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
        assert ksk.mode()["mode"] == mode
        p = np.stack([cc.synth_poly(seed, i, 0) for i in range(2)])
        g0, g1 = ksk.key_switch(p)
        for i in range(2):
            w0, w1 = ck.key_switch(p[i])
            assert np.array_equal(np.asarray(g0[i]), w0) and np.array_equal(np.asarray(g1[i]), w1)
        if n > 16384 and mode not in (1, 4):
            break   # (one modulus set is enough at the emulator's speed; the fused form has a loader per kind of set)


def test_unfused_random_parameter_shapes(fhe):
    """24 of the pseudo-random shapes of test_random_parameter_shapes with the unfused strategy forced (mixed modulus
    widths: all three lift forms; ragged batches: the tails of the XCD-grouped stage-B grid)."""
    import random
    import full_size
    with _unfused(fhe):
        for idx in range(0, 48, 2):
            rng = random.Random(0xBEEF + idx)
            n = 1 << rng.randrange(3, 13)
            L = rng.randrange(1, 7)
            sizes = [rng.choice([36, 45, 50, 54, 58, 59, 60, 61, 62]) for _ in range(L)]
            batch = rng.randrange(1, 12)
            if L < 2:
                continue
            full_size.check_mul_host(fhe, n=n, sizes=sizes, batch=batch, relin=True, cfg=300 + idx, mod_switch=(idx % 4 == 0))


def test_workspace_bounds(fhe):
    import ctypes as C
    from fhe_rs_amd import _lib
    L = _lib.lib()
    L.fhe_emu_stream_create.restype = C.c_void_p
    L.fhe_emu_stream_destroy.argtypes = [C.c_void_p]
    cases.case_workspace_bounds(fhe, lambda: L.fhe_emu_stream_create(), lambda h: L.fhe_emu_stream_destroy(h))


def test_scaler_many_wide_moduli(fhe):
    cases.case_scaler_many_wide_moduli(fhe, False)


EVERY_SCALER_INSTANCE = (2, 3, 5, 7, 8, 10, 11, 13, 14, 15, 16, 18, 19, 22, 23, 26, 27, 30, 31, 32)


def test_scaler_every_instance(fhe):
    """Round 6: scale_kernel<NF> has an instance per source-basis size of BASELINE's configs, the reference's stock sets
    and the levels of C5's chain (engine.hpp scale_kernel_nf); every instance on its exact fit and on a padded one, factor
    one (the PLAIN instances) and a non-unit factor, against the oracle's RnsScaler::scale (rns/scaler.rs:249-352)."""
    cases.case_scaler_many_wide_moduli(fhe, False, counts=EVERY_SCALER_INSTANCE, factors=((1, 1), (3, 7)))


@pytest.mark.parametrize("n,bits", [(16384, 60), (32768, 60), (32768, 62), (65536, 60), (65536, 62)])
def test_key_switch_decomposition_rows(fhe, n, bits):
    """Single-modulus key levels (base-2^k digits of one residue row) on whole rows and on the 16384-point parts of rows
    larger than LDS (and on their 8192-point sub-blocks: FUSED_SUB), narrow (60-bit) and general (62-bit) passes."""
    for mode in (1, 4) if n > 16384 else (1,):
        with fhe.KeySwitchingKey.forced_mode(mode):
            cases.case_key_switch_decomposition_rows(fhe, False, n, bits)



# ---- round 5: the reference's own stock parameter sets (BfvParameters::default_parameters_128, parameters.rs:218-251) ----
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_reference_default_parameter_sets(fhe, n):
    """Every hot-path Criterion ID of crates/fhe/benches/bfv.rs:167-286 -- mul, square, mul_and_relin, relinearize,
    rotate_rows, rotate_columns, inner_sum, expand_4, mul_and_relin_2 -- plus the leveled multiply + modulus-switch chain
    down to one modulus, on the explicit primes of default_parameters_128 (27 / 54 / 36-37 / 43-44-bit rows with 62-bit
    extension rows; K = 10 at n = 8192), one ciphertext per call, against the plain-C oracle."""
    import ref_params
    ref_params.check_all(fhe, False, n, batch=1)


@pytest.mark.parametrize("mode", [1, 2])
def test_reference_default_parameter_sets_key_switch_modes(fhe, mode):
    """n = 4096 / log q = 109 and n = 8192 / log q = 218 with each key-switch strategy forced."""
    import ref_params
    with fhe.KeySwitchingKey.forced_mode(mode):
        ref_params.check_mul(fhe, False, 8192, relin=True, batch=1)
        ref_params.check_relin_rotate(fhe, False, 4096, batch=2)
        ref_params.check_chain(fhe, False, 4096, batch=1)


def test_reference_default_parameter_set_n16384(fhe):
    """n = 16384 / log q = 438 (nine 48/49-bit rows, K = 18): multiply + relinearise (+ the HPS second strategy) and the
    key switches; the long rotation chains of this set run on the GPU suite."""
    import ref_params
    ref_params.check_mul(fhe, False, 16384, relin=True, batch=1)
    ref_params.check_relin_rotate(fhe, False, 16384, batch=1)
    ref_params.check_mul2(fhe, False, 16384, batch=1)


@pytest.mark.parametrize("n,mode", [(4096, 1), (4096, 2), (16384, 1), (32768, 1), (32768, 4), (32768, 2)])
def test_galois_folded_substitution_rows(fhe, n, mode):
    """Round 5: from N = 4096 on, GaloisKey::relinearize reads c1 / c0 through the Ntt-domain substitution inside the
    inverse transform's loader (whole rows, and the 8192-point sub-blocks of rows larger than LDS) and inside the key
    switch (own-row read and addend: fused whole rows, 16384-point parts, 8192-point sub-blocks, and the unfused MAC) --
    no separate permutation pass.  Exponents 3, 2N - 1 and N + 1 (the expansion's first), 60- and 62-bit rows, against the
    C oracle."""
    import numpy as np
    from fhe_oracle import coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    import full_size
    q = [generate_prime(60, 2 * n, 1 << 60), generate_prime(62, 2 * n, 1 << 62)]
    cc = coracle.CCtx(OCtx(q, n))
    seed = 0xF4E50B00 + n
    ck = full_size.host_key(cc, seed, len(q))
    ctx = fhe.Context(q, n)
    ksk = fhe.KeySwitchingKey(ctx, ctx, ck.c0, ck.c1).set_mode(mode)
    ct = np.stack([np.stack([cc.synth_poly(seed, b, p) for p in range(2)]) for b in range(2)])
    for e in (3, 2 * n - 1, n + 1):
        got = fhe.GaloisKey(ksk, e).relinearize(ct)
        for b in range(2):
            assert np.array_equal(np.asarray(got[b]), ck.galois_relinearize(e, ct[b])), (n, mode, e, b)


# ---- round 6: the FP64-FMA instances for moduli below 2^50 (csrc/zq_f64.hpp); range traps active in this build ----
def test_f64_ntt(fhe):
    cases.case_f64_ntt(fhe, False, 4096)


def test_f64_ntt_n16384(fhe):
    cases.case_f64_ntt(fhe, False, 16384, sizes=(50, 49, 48))


@pytest.mark.parametrize("n,sizes", [(4096, (36, 36, 37)), (8192, (43, 43, 44, 44, 44)), (4096, (50, 50, 49, 48)),
                                     (16384, (48, 49, 49))])
def test_f64_key_switch(fhe, n, sizes):
    """The reference's stock n = 4096 / 8192 widths (parameters.rs:222-242), a 50-bit basis and a 16384-point tile."""
    cases.case_f64_key_switch(fhe, False, n, sizes, exps=(3, 2 * n - 1) if n == 4096 else (3,))


def test_f64_key_switch_accumulator_fold(fhe):
    """Twelve 50-bit moduli: more digits than an F64 accumulator holds before it is reduced (class 3: nine terms)."""
    cases.case_f64_key_switch(fhe, False, 4096, (50,) * 12, batch=1, exps=())


@pytest.mark.parametrize("n,sizes", [(4096, (36, 36, 37)), (8192, (43, 43, 44, 44, 44)), (4096, (50, 49, 48))])
def test_f64_key_switch_unfused(fhe, n, sizes):
    """KS_UNFUSED: stage A (ks_ntt_kernel) on its F64 instances, W in canonical words, stage B unchanged."""
    cases.case_f64_key_switch(fhe, False, n, sizes, mode=2)


def test_f64_multiply_stock_sets(fhe):
    """The reference's stock n = 4096 set through Multiplicator::multiply with every F64 kernel on the way (the
    ciphertext rows' transforms, the key switch, the forward transform riding on the unfused stage A), then with the option
    off: same words (C oracle)."""
    import ref_params
    for f64 in (True, False):
        fhe.set_f64(f64)
        try:
            for mode in (1, 2):
                ref_params.FORCE_KS_MODE = mode
                ref_params.check_mul(fhe, False, 4096, relin=True, batch=2)
        finally:
            ref_params.FORCE_KS_MODE = None
            fhe.set_f64(True)
