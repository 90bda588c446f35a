"""GPU suite (`pytest -m gpu`): the parity tests proper.  Every case calls the HIP build
through the C ABI on the MI355X and compares bit-for-bit with the oracle -- host-pointer and
device-pointer (`_dev`, torch tensors on the current stream) entry points, small sizes
against the Python oracle, BASELINE.json's full sizes against the plain-C oracle, plus
size-independent properties (round trips, linearity) at full batch."""
import numpy as np
import pytest

import cases
from helpers import HIP_LIB, load_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fhe():
    eng = load_engine("hip")
    from fhe_rs_amd import _lib
    assert _lib.loaded_path() == HIP_LIB, "GPU tests must run on the HIP build"
    assert eng.device_count() >= 1, "no HIP device visible"
    return eng


@pytest.mark.parametrize("dev", [False, True])
@pytest.mark.parametrize("n", [8, 16, 64, 1024, 4096])
def test_ntt_small(fhe, n, dev):
    cases.case_ntt(fhe, dev, n, batch=3)


@pytest.mark.parametrize("n,mods", [(8192, [1152921504606830593, 1152921504606748673, 4611686018427322369]),
                                    (16384, [1152921504606748673, 4611686018427322369]),
                                    (32768, [1152921504606584833, 4611686018427322369]),
                                    (65536, [4611686018427322369 - 0])])
def test_ntt_full_sizes(fhe, n, mods):
    from fhe_oracle import coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    mods = [m if m % (2 * n) == 1 else generate_prime(62, 2 * n, 1 << 62) for m in mods]
    mods = list(dict.fromkeys(mods))
    cases.case_ntt(fhe, True, n, moduli=mods, batch=5, coracle_ctx=coracle.CCtx(OCtx(mods, n)))


def test_ntt_explicit_tables(fhe):
    cases.case_ntt_explicit_tables(fhe, True)


@pytest.mark.parametrize("dev", [False, True])
def test_poly_ops(fhe, dev):
    cases.case_poly_ops(fhe, dev)


@pytest.mark.parametrize("n", [64, 8192])
def test_product_extremes(fhe, n):
    cases.case_product_extremes(fhe, True, n)


@pytest.mark.parametrize("dev", [False, True])
def test_substitute(fhe, dev):
    cases.case_substitute(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_switch_down(fhe, dev):
    cases.case_switch_down(fhe, dev)


@pytest.mark.parametrize("dev", [False, True, "abi"])
def test_switch_down_to(fhe, dev):
    if dev == "abi":
        with fhe.Stream(0):
            cases.case_switch_down_to(fhe, dev)
    else:
        cases.case_switch_down_to(fhe, dev)


def test_device_buffers_and_streams(fhe):
    cases.case_device_buffers(fhe)


def test_table_mismatch_is_rejected(fhe):
    cases.case_table_mismatch(fhe)


@pytest.mark.parametrize("what", ["multiply", "galois", "key_switch", "wire"])
def test_abi_owned_buffers(fhe, what):
    """The `_dev` entry points on buffers and a stream that come from the C ABI itself (no torch allocator)."""
    with fhe.Stream(0) as st:
        if what == "multiply":
            cases.case_multiply(fhe, "abi", nmod=3, n=64, batch=5)
        elif what == "galois":
            cases.case_galois(fhe, "abi")
        elif what == "key_switch":
            cases.case_key_switch_levels(fhe, "abi")
        else:
            cases.case_wire_format(fhe, "abi")
        st.synchronize()


def test_scaler_grid(fhe):
    cases.case_scaler_grid(fhe, True)


def test_scaler_grid_host_api(fhe):
    cases.case_scaler_grid(fhe, False, pairs={(1, 1), (3, 4), (1000, 101), (4611686018326724610, 1001)})


def test_scaler_extend_switcher_constants(fhe):
    cases.case_scaler_extend_and_constants_api(fhe, True)


def test_params(fhe):
    cases.case_params(fhe)


@pytest.mark.parametrize("dev", [False, True])
def test_key_switch_levels(fhe, dev):
    cases.case_key_switch_levels(fhe, dev)


def test_key_switch_decomposition(fhe):
    cases.case_key_switch_decomposition(fhe, True)


@pytest.mark.parametrize("dev", [False, True])
def test_galois(fhe, dev):
    cases.case_galois(fhe, dev)


@pytest.mark.parametrize("nmod,n,level,chunk,dev", [(2, 16, 0, 0, True), (3, 16, 0, 2, False), (3, 64, 1, 0, True),
                                                    (4, 32, 1, 1, True), (6, 16, 0, 0, True)])
def test_multiply(fhe, nmod, n, level, chunk, dev):
    cases.case_multiply(fhe, dev, nmod=nmod, n=n, level=level, chunk=chunk)


@pytest.mark.parametrize("dev", [False, True])
def test_multiply_square_shortcut(fhe, dev):
    cases.case_multiply_square(fhe, dev)
    cases.case_multiply_square(fhe, dev, nmod=2, n=1024, batch=3)


def test_multiply_host_pointer_sliced(fhe):
    """fhe_bfv_mul on host arrays with a batch large enough for the sliced path (N = 4096, 2 moduli: slices of 256
    pairs, 773 pairs = three slices and a ragged fourth): bit-identical to the `_dev` entry point on the same inputs
    for every ciphertext, sampled ciphertexts against the C oracle (full_size.check_mul does the latter on the device
    path; the generator's output is the common input)."""
    import torch
    import full_size
    n, sizes, batch, cfg = 4096, [60, 60], 773, 4242
    full_size.check_mul(fhe, n, sizes, batch, relin=True, cfg=cfg, sample=(0, 255, 256, 511, 512, 767, 768, 772))
    q = full_size.obfv.generate_moduli(sizes, n)
    par = fhe.BfvParameters(n, full_size.plaintext_modulus(n), moduli=q)
    ctx = par.context_at_level(0)
    seed = full_size.synth.seed_for_config(cfg)
    c0, c1 = full_size.device_key(ctx, seed, len(q))
    m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1)), 0)
    lhs, rhs = ctx.synth_uniform(seed, 0, 0, 2, batch), ctx.synth_uniform(seed, 0, 2, 2, batch)
    dev_out = m.multiply(lhs, rhs)
    torch.cuda.synchronize()
    hl, hr = lhs.cpu().numpy().view(np.uint64), rhs.cpu().numpy().view(np.uint64)
    host_out = m.multiply(hl, hr)
    assert np.array_equal(host_out, dev_out.cpu().numpy().view(np.uint64))
    assert np.array_equal(m.multiply(hl, hl), m.multiply(lhs, lhs).cpu().numpy().view(np.uint64))   # squaring, sliced
    # relinearise (773 three-part ciphertexts: slices of 168) and a rotation through the same sliced host path
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
    t3 = fhe.Multiplicator.default(par, None, 0).multiply(lhs, rhs)
    assert np.array_equal(rk.relinearizes(t3.cpu().numpy().view(np.uint64)), rk.relinearizes(t3).cpu().numpy().view(np.uint64))
    gk = fhe.GaloisKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1), 3)
    assert np.array_equal(gk.relinearize(hl), gk.relinearize(lhs).cpu().numpy().view(np.uint64))


def test_multiply_custom_factors(fhe):
    cases.case_multiply_custom_factors(fhe, True)


@pytest.mark.parametrize("dev", [False, True])
def test_dot_product_and_mul_plain(fhe, dev):
    cases.case_dot_product_and_mul_plain(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_rgsw_and_inner_sum(fhe, dev):
    cases.case_rgsw_and_inner_sum(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_expand(fhe, dev):
    cases.case_expand(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_tensor_any_parts(fhe, dev):
    cases.case_tensor_any_parts(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_extender_narrow_sums(fhe, dev):
    cases.case_extender_narrow_sums(fhe, dev)
    cases.case_extender_narrow_sums(fhe, dev, n=2048)


@pytest.mark.parametrize("dev", [False, True])
def test_decrypt(fhe, dev):
    cases.case_decrypt(fhe, dev)


@pytest.mark.parametrize("dev", [False, True])
def test_wire_format(fhe, dev):
    cases.case_wire_format(fhe, dev)
    cases.case_wire_format(fhe, dev, n=8192)


def test_mul_default_level_basis(fhe):
    cases.case_mul_default_level_basis(fhe)


def test_params_with_tables(fhe):
    cases.case_params_with_tables(fhe, True)


def test_ksk_validation(fhe):
    cases.case_ksk_validation(fhe)


def test_key_switch_many_digits(fhe):
    cases.case_key_switch_many_digits(fhe, True)
    cases.case_key_switch_many_digits(fhe, True, n=8192, shapes=((60, 9), (50, 11), (62, 4)))


@pytest.mark.parametrize("dev", [False, True])
def test_random_from_seed(fhe, dev):
    cases.case_random_from_seed(fhe, dev)
    cases.case_random_from_seed(fhe, dev, n=2048)


def test_errors(fhe):
    cases.case_errors(fhe)
    cases.case_option_errors(fhe)


def test_empty_batch(fhe):
    import torch
    c = fhe.Context(cases.Q3, 16)
    e = torch.empty((0, 3, 16), dtype=torch.int64, device="cuda")
    assert c.ntt_forward(e).shape[0] == 0
    assert c.ntt_forward(np.zeros((0, 3, 16), dtype=np.uint64)).shape[0] == 0


# ---------------------------------------------------------------- BASELINE.json configs ----
def test_config_c1_ct_times_ct_n4096(fhe):
    """configs[0]: N=4096, one 50-bit modulus, `&ct * &ct` (no relin: one modulus)."""
    import full_size
    full_size.check_mul(fhe, n=4096, sizes=[50], batch=4, relin=False, cfg=1)


def test_config_c2_mul_relin_n8192(fhe):
    """configs[1]: N=8192, 4x60-bit moduli, ct x ct + relinearise; sampled against the C oracle."""
    import full_size
    full_size.check_mul(fhe, n=8192, sizes=[60] * 4, batch=70, relin=True, cfg=2, sample=(0, 1, 33, 69))


def test_config_c2_properties_full_batch(fhe):
    """Full batch (1024): size-independent properties -- batch consistency (every ciphertext
    pair gives the same result alone as inside the batch) and NTT round trip."""
    import full_size
    full_size.check_batch_properties(fhe, n=8192, sizes=[60] * 4, batch=1024, cfg=2)


def test_config_c3_relin_rotate_n16384(fhe):
    """configs[2]: N=16384, 8x60-bit moduli, relinearise + rotation (Galois key switch)."""
    import full_size
    full_size.check_relin_rotate(fhe, n=16384, sizes=[60] * 8, batch=6, cfg=3)


def test_config_c3_bench_batch_n16384(fhe):
    """configs[2] at the batch the bench line's `other_configs` times it at (512: other grid / wave counts than a
    handful of ciphertexts): 33 ciphertexts spread over the batch against the C oracle, all three operations."""
    import full_size
    full_size.check_relin_rotate(fhe, n=16384, sizes=[60] * 8, batch=512, cfg=3,
                                 sample=tuple(range(0, 512, 16)) + (511,))


def test_config_c4_per_gpu_shard_n8192(fhe):
    """configs[3]: batch 65,536 sharded over 8 GPUs = 8,192 pairs per GPU.  One rank's shard (rank 3: global
    ciphertexts 24,576...32,767 of the synthetic stream) on this GPU: 8 GiB of ciphertexts in, 4 GiB out, 64
    chunks on two streams; 70 ciphertexts incl. the first / last of the shard and both sides of chunk boundaries,
    each whole output against the C oracle."""
    import full_size
    per_gpu, rank = 8192, 3
    sample = (0, 1, 127, 128, 129, 255, 256, 4095, 4096, 4097, 8063, 8064, 8190, 8191) + tuple(range(37, 8192, 146))
    full_size.check_mul(fhe, n=8192, sizes=[60] * 4, batch=per_gpu, relin=True, cfg=4, sample=sample,
                        ct0=rank * per_gpu)
    fhe.workspace_trim()


def test_config_c5_chain_n32768(fhe):
    """configs[4]: N=32768, 16x60-bit moduli, the deep chain: multiply + relinearise + modulus switch at EVERY
    level 0...14 (L_l = 16...2 moduli, K_l = 33...5: every scaler instantiation, rows that do not fit LDS),
    each level's output feeding the next, 4 ciphertexts, every one against the C oracle at every level."""
    import full_size
    full_size.check_chain(fhe, n=32768, sizes=[60] * 16, batch=4, levels=15, cfg=5)
    fhe.workspace_trim()


@pytest.mark.parametrize("batch", [16, 64])
def test_config_c5_bench_batches_n32768(fhe, batch):
    """C5 at the batches bench.py times (16: one chunk; 64: the two-stream cut of plan_chunks, 8 chunks of 8):
    level-0 multiply + relinearise + modulus switch, sampled ciphertexts (first, last, both sides of chunk
    boundaries) against the C oracle."""
    import full_size
    sample = sorted({0, 1, 7, 8, batch // 2 - 1, batch // 2, batch - 2, batch - 1})
    full_size.check_mul(fhe, n=32768, sizes=[60] * 16, batch=batch, relin=True, cfg=5, sample=sample, mod_switch=True)
    fhe.workspace_trim()


def _bench_lines(stdout):
    """(record, detail) of a bench.py run: the record is the LAST stdout line and must be what the driver can keep whole
    (< 2,000 characters); everything else bench.py measured is the `DETAIL ` line before it."""
    import json
    lines = stdout.splitlines()
    assert lines and lines[-1].startswith("{") and len(lines[-1]) < 2000, (len(lines[-1]) if lines else 0)
    detail = [l for l in lines if l.startswith("DETAIL {")]
    assert len(detail) == 1 and not [l for l in lines[:-1] if l.startswith("{")]     # nothing else can be mistaken for it
    rec, det = json.loads(lines[-1]), json.loads(detail[0][len("DETAIL "):])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"):
        assert rec[k] == det[k], k
    assert rec["vs_baseline"] is None and rec["roofline"]["bound"] == "hbm" and rec["roofline"]["frac"] == det["roofline"]["frac"]
    return rec, det


def test_bench_record_is_compact():
    """the default-flags shape of the command the driver runs (extras ON, CPU leg ON), at a small step
    count: the last stdout line parses, is below 2,000 characters and carries `roofline`, `cpu_baseline`, `ntt` and the
    other configs' binding ceilings; the full result is in the DETAIL line and in bench_detail.json."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    rec, det = _bench_lines(r.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["workload"].startswith("C2:") and rec["config"]["batch_per_gpu"] == 1024
    rf = rec["roofline"]
    for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_observed_this_run", "kernel_sum_ms_per_step",
              "kernel_sum_le_step", "launches", "avg_launch_ms", "algorithmic_bytes_per_launch", "binding",
              "frac_of_binding_ceiling", "frac_hbm_whole_op", "dominant_symbol"):
        assert k in rf, k
    assert rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["kernel_sum_le_step"] is True
    # the dominant kernel by SYMBOL is a real kernel instantiation, named as rocprofv3 names it
    assert "_kernel<" in rf["dominant_symbol"]["kernel"] and 0 < rf["dominant_symbol"]["share_of_kernel_time"] < 1
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"] and cb["unit"] == "ops/s"
    assert rec["ntt"]["row_ntt_per_s"] == 4 * rec["ntt"]["poly_ntt_per_s"] or abs(rec["ntt"]["row_ntt_per_s"] / rec["ntt"]["poly_ntt_per_s"] - 4) < 1e-3
    assert rec["parity"].startswith("bit-identical")
    # binding ceilings of the other BASELINE configs and the stock sets: [frac_hbm, frac_int_issue, ops/s]
    for k in ("C3_relinearize", "C3_rotate_columns", "C5_level0", "C5_chain", "stock8192_mul_and_relin", "stock16384_relinearize"):
        assert k in rec["configs"] and 0 < rec["configs"][k][0] < 1 and 0 < rec["configs"][k][1] < 1.05, (k, rec["configs"].get(k))
    assert not det.get("errors"), det.get("errors")
    assert det["binding_ceilings"]["C3_relinearize"]["binding"] in ("hbm", "int_issue")
    syms = [e["kernel"] for e in det["roofline"]["by_symbol"]]
    assert len(syms) == len(set(syms)) and sum("ntt_kernel<false, 13" in x for x in syms) == 2     # wide and narrow apart
    on_disk = json.load(open(os.path.join(root, "bench_detail.json")))
    assert on_disk["value"] == rec["value"] and "other_configs" in on_disk


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` starts its own two ranks; on a one-GPU box they share the device and
    rendezvous over gloo (RCCL needs one device per rank).  The line must say n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "128", "--no-cpu", "--no-extras"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec, line = _bench_lines(r.stdout)
    assert rec["n_gpus"] == 2 and rec["multi_gpu"]["data_path_collectives"] == 0 and "efficiency_vs_1gpu_same_batch" in rec["multi_gpu"]
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["value"] > 0
    # round 4: the line says who ran where and what one GPU does alone on the same per-GPU batch
    mg = line["multi_gpu"]
    assert len(mg["ranks"]) == 2 and {r["rank"] for r in mg["ranks"]} == {0, 1} and mg["data_path_collectives"] == 0
    assert mg["solo_rank0_ops_per_s"] > 0 and 0 < mg["efficiency_vs_1gpu_same_batch"] < 1.5
    assert mg["one_device_per_rank"] == (mg["distinct_devices"] == 2)
    assert line["repeats"] == 3 and len(line["value_all"]) == 3 and line["value_min"] <= line["value"] <= line["value_max"]
    assert line["roofline"]["kernel_sum_le_step"] in (True, False) and line["roofline"]["traffic_observed_this_run"] is False


def test_bench_eight_ranks_on_one_gpu():
    """The world size the driver's SCALE run uses: `python bench.py --gpus 8` spawns eight ranks (they share this box's
    one GPU and rendezvous over gloo): the spawn, the NUMA pinning, the identity gather, the solo / n1_reference legs and
    the efficiency block at world size 8, on a batch small enough for eight processes on one device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--batch", "32", "--no-cpu", "--no-extras", "--sustain", "0.2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rec, line = _bench_lines(r.stdout)
    # the compact record of an N > 1 run (what SCALE_rNN.json keeps): world size, who ran where in summary, the one-GPU reference
    mgc = rec["multi_gpu"]
    assert rec["n_gpus"] == 8 and rec["config"]["dist_world_size"] == 8 and rec["config"]["global_batch"] == 256
    for k in ("dist_backend", "distinct_devices", "one_device_per_rank", "efficiency_vs_1gpu_same_batch", "data_path_collectives",
              "rank_max_over_min"):
        assert k in mgc, k
    assert "cpu_baseline" not in rec and "ranks" not in mgc and rec["scaling"] == "weak"
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 256 and line["value"] > 0
    mg = line["multi_gpu"]
    assert len(mg["ranks"]) == 8 and {x["rank"] for x in mg["ranks"]} == set(range(8)) and mg["data_path_collectives"] == 0
    assert mg["n1_reference"]["batch_32"]["value"] > 0 and mg["solo_rank0_ops_per_s"] > 0
    assert len(line["per_rank"]["ops_per_s"]) == 8 and line["sustained"]["steps"] >= 20
    assert "cpu_baseline" not in line and line["scaling"] == "weak"


def test_max_degree_n65536(fhe):
    """Largest degree the reference accepts (parameters.rs MAX_DEGREE = 65536): ct x ct + relin,
    rows go through the two-kernel NTT and the unfused key switch."""
    import full_size
    full_size.check_mul(fhe, n=65536, sizes=[60, 60], batch=2, relin=True, cfg=6)


@pytest.mark.parametrize("streams", [1, 2])
def test_concurrent_streams_share_handles(fhe, streams):
    """Handles are immutable and may be shared by concurrent callers working on different
    buffers (one HIP stream per calling thread) -- SURVEY.md 8(b) threading contract.  streams = 2: every
    caller's multiply also forks onto its own internal stream (chunks of 5 pairs)."""
    import threading
    import torch
    import full_size
    from fhe_oracle import bfv as obfv, synth
    n, sizes = 4096, [60, 60, 60]
    q = obfv.generate_moduli(sizes, n)
    par = fhe.BfvParameters(n, full_size.plaintext_modulus(n), moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = full_size.device_key(ctx, 77, len(q))
    m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1)), 0)
    ins = [(ctx.synth_uniform(100 + i, 0, 0, 2, 24), ctx.synth_uniform(100 + i, 0, 2, 2, 24)) for i in range(4)]
    want = [m.multiply(a, b) for a, b in ins]
    torch.cuda.synchronize()
    got, errs = [None] * 4, []
    # options live on the handle: set once here, read atomically by every concurrent call
    m.set_streams(streams)
    if streams == 2:
        m.set_chunk(5)

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(3):
                    got[i] = m.multiply(*ins[i])
            st.synchronize()
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    # flipping the handle's options while calls are in flight is allowed (each call reads them once on entry)
    for k in range(6):
        m.set_chunk((0, 5, 7)[k % 3])
    [t.join() for t in th]
    assert not errs, errs
    for i in range(4):
        assert torch.equal(got[i], want[i])


@pytest.mark.parametrize("idx", range(48))
def test_random_parameter_shapes(fhe, idx):
    """Sweep of degrees 32..16384, 1..6 moduli of mixed widths (36..62 bits), small ragged batches:
    ct x ct (+relinearise, +mod switch), relinearise, rotations -- bit-exact vs the C oracle."""
    import full_size
    full_size.check_random_shape(fhe, idx)


def test_workspace_trim(fhe):
    """Scratch buffers persist between calls; trimming frees the idle ones and the next call regrows them."""
    cases.case_multiply(fhe, True, nmod=2)
    assert fhe.workspace_trim() > 0
    assert fhe.workspace_trim() == 0
    cases.case_multiply(fhe, True, nmod=2)


def test_c2_full_batch_against_oracle(fhe):
    """BASELINE config C2 at the benchmark's batch size (1024 pairs, one launch per pipeline step):
    96 ciphertexts spread over the batch, each whole output compared with the C oracle."""
    import full_size
    full_size.check_mul(fhe, n=8192, sizes=[60] * 4, batch=1024, relin=True, cfg=2,
                        sample=tuple(range(0, 1024, 11)) + (1023, 1022, 1021))


def test_dev_entry_points_are_graph_capturable(fhe):
    """The `_dev` entry points enqueue kernels only (no allocation, copy or synchronisation once the
    stream's workspace exists), so a call sequence can be captured into a hipGraph and replayed --
    the launch-bound small-batch case (one ciphertext pair: ten launches of a few microseconds each)."""
    import torch
    import full_size
    from fhe_oracle import bfv as obfv
    n, sizes = 8192, [60] * 4
    q = obfv.generate_moduli(sizes, n)
    par = fhe.BfvParameters(n, full_size.plaintext_modulus(n), moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = full_size.device_key(ctx, 5, len(q))
    ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1)
    m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(ksk), 0)
    gk = fhe.GaloisKey(ksk, 3)
    a, b = ctx.synth_uniform(9, 0, 0, 2, 2), ctx.synth_uniform(9, 0, 2, 2, 2)
    want_m = m.multiply(a, b)
    want_r = gk.relinearize(want_m)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up on the capture stream: its workspace blocks get allocated
        gk.relinearize(m.multiply(a, b))
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        got_m = m.multiply(a, b)
        got_r = gk.relinearize(got_m)
    got_m.zero_()
    got_r.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_m, want_m) and torch.equal(got_r, want_r)
    # new inputs through the captured graph (static input buffers, as with any graph)
    a2, b2 = ctx.synth_uniform(10, 0, 0, 2, 2), ctx.synth_uniform(10, 0, 2, 2, 2)
    want2 = m.multiply(a2, b2)
    a.copy_(a2)
    b.copy_(b2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_m, want2)


def test_multiply_two_streams(fhe):
    """Two-stream mode (the handle's default; fhe_mul_set_streams) at C2's shape: chunks of 128 pairs alternating between two streams give bit-identical
    ciphertexts to the single-stream pipeline (which the other tests tie to the oracle), repeatedly, also when
    the call is captured into a hipGraph (the internal stream forks from / joins the capturing stream)."""
    import torch
    import full_size
    from fhe_oracle import bfv as obfv
    n, sizes, batch = 8192, [60] * 4, 600          # 600 pairs: five chunks of 120
    q = obfv.generate_moduli(sizes, n)
    par = fhe.BfvParameters(n, full_size.plaintext_modulus(n), moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = full_size.device_key(ctx, 11, len(q))
    m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1)), 0)
    a, b = ctx.synth_uniform(21, 0, 0, 2, batch), ctx.synth_uniform(21, 0, 2, 2, batch)
    m.set_streams(1)
    want = m.multiply(a, b)
    torch.cuda.synchronize()
    m.set_streams(2)
    assert m.options()["streams"] == 2
    for _ in range(3):
        got = m.multiply(a, b)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    # stream order for the caller: work enqueued right after the call sees the complete result
    got = m.multiply(a, b)
    chk = (got != want).sum()
    assert int(chk.item()) == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.multiply(a, b)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = m.multiply(a, b)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_release_library_ignores_lab_environment(tmp_path):
    """Round 2's library read kernel-selection and wrong-result switches from the environment on every launch.  The
    release build reads none: with every one of those variables set, a fresh process still matches the oracle."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    names = ["FHE_DEBUG_NTT_NOMEM", "FHE_DEBUG_KS_NOMEM", "FHE_DEBUG_SYNC", "FHE_NO_NARROW", "FHE_NTT_SWAP", "FHE_NTT_CPT8",
             "FHE_KS_VARIANT", "FHE_KS14_PLAN", "FHE_KS_PERSIST", "FHE_KS_SPLIT14", "FHE_NO_KS_XHAT", "FHE_NO_SKIP_COPY",
             "FHE_NO_TENSOR_FUSION"]
    names += [n.replace("FHE_", "FHE_LAB_") for n in names]
    env = dict(os.environ, **{n: "1" for n in names})
    code = ("import sys; sys.path[:0] = [r'%s', r'%s/oracle', r'%s/tests'];"
            "import fhe_rs_amd as fhe, cases;"
            "cases.case_multiply(fhe, True, nmod=3, n=64, batch=4);"
            "cases.case_key_switch_levels(fhe, True);"
            "cases.case_ntt(fhe, True, 4096, batch=2);"
            "print('lab-env ok')") % (ROOT, ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "lab-env ok" in r.stdout, r.stdout + r.stderr


# ---- round 4: the unfused key switch (ks_ntt_kernel + ks_mac_kernel).  Every case that goes through a key switch is
# run with each evaluation strategy forced on the handle (fhe_ksk_set_mode), so that whichever one the engine's
# per-shape choice (FHE_KS_AUTO) lands on has been compared with the oracle at that shape. ----
KS_MODES = {"auto": 0, "fused": 1, "unfused": 2, "unfused_sub": 3, "fused_sub": 4}


@pytest.mark.parametrize("dev", [False, True])
def test_unfused_key_switch_small(fhe, dev):
    with fhe.KeySwitchingKey.forced_mode(2):
        cases.case_key_switch_levels(fhe, dev)
        cases.case_key_switch_many_digits(fhe, dev)
        cases.case_galois(fhe, dev)
        cases.case_galois(fhe, dev, nmod=3, n=128)
        cases.case_rgsw_and_inner_sum(fhe, dev)
        cases.case_expand(fhe, dev)
        cases.case_key_switch_decomposition(fhe, dev)     # (stays on the fused kernel: base-2^k digits)
    with fhe.KeySwitchingKey.forced_mode(2, w_budget=1):   # one polynomial x one key modulus per launch pair
        cases.case_key_switch_levels(fhe, dev)
        cases.case_multiply(fhe, dev, nmod=3, level=0, chunk=2, batch=5)


@pytest.mark.parametrize("mode", ["fused", "unfused"])
@pytest.mark.parametrize("sizes,n,batch", [([60, 58, 62, 50], 4096, 7), ([61, 45, 36], 2048, 33), ([62] * 6, 1024, 5),
                                           ([60, 60], 256, 3)])
def test_multiply_key_switch_modes(fhe, mode, sizes, n, batch):
    """Mixed modulus widths (the lift takes its one-, two-subtraction and Barrett forms), with modulus switch."""
    import full_size
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        full_size.check_mul(fhe, n=n, sizes=sizes, batch=batch, relin=True, cfg=40 + len(sizes), mod_switch=True)


@pytest.mark.parametrize("mode", ["fused", "unfused"])
def test_config_c2_key_switch_modes(fhe, mode):
    import full_size
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        full_size.check_mul(fhe, n=8192, sizes=[60] * 4, batch=70, relin=True, cfg=2)


@pytest.mark.parametrize("mode", ["fused", "unfused", "unfused_sub"])
def test_config_c3_key_switch_modes(fhe, mode):
    """configs[2] (relinearise + both rotations, N = 16384, 8 moduli) with every key-switch strategy: batch 6, and
    batch 96 with a W budget that cuts it into three chunks of polynomials and four groups of key moduli."""
    import full_size
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        full_size.check_relin_rotate(fhe, n=16384, sizes=[60] * 8, batch=6, cfg=3)
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode], w_budget=32 * 8 * 2 * 16384 * 8):
        full_size.check_relin_rotate(fhe, n=16384, sizes=[60] * 8, batch=96, cfg=3, sample=(0, 1, 31, 32, 33, 63, 64, 95))


@pytest.mark.parametrize("mode", ["fused", "fused_sub", "unfused"])
def test_config_c5_key_switch_modes(fhe, mode):
    """configs[4]: the 15-level chain (4 ciphertexts) and the bench batch 16 at level 0, with each strategy (FHE_KS_AUTO
    -- test_config_c5_chain_n32768, test_config_c5_bench_batches_n32768 -- mixes the two fused forms by launch size)."""
    import full_size
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        full_size.check_chain(fhe, n=32768, sizes=[60] * 16, batch=4, levels=15, cfg=5)
        full_size.check_mul(fhe, n=32768, sizes=[60] * 16, batch=16, relin=True, cfg=5, sample=(0, 7, 8, 15), mod_switch=True)
    fhe.workspace_trim()


@pytest.mark.parametrize("n,mode", [(8192, 2), (16384, 2), (16384, 3), (32768, 2), (65536, 2), (16384, 1), (32768, 1), (65536, 1),
                                    (32768, 4), (65536, 4), (32768, 0), (65536, 0)])
def test_key_switch_strategies_large_rows(fhe, n, mode):
    """Every key-switch strategy (2 / 3 unfused, 1 fused: 16384-point parts with one / two folded stages above N = 16384)
    on synthetic keys and inputs vs the C oracle: 60-bit moduli (narrow passes, residue-row loader), 62 + 61 bits (general
    passes, generic loader), 62 + 62 bits (general passes, residue-row loader), 60 + 58 + 60 bits (narrow, generic)."""
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    import full_size
    import torch
    seed = 0xF4E50078
    p62 = generate_prime(62, 2 * n, 1 << 62)
    for q in (obfv.generate_moduli([60, 60, 60], n), [p62, generate_prime(61, 2 * n, 1 << 61)], [p62, generate_prime(62, 2 * n, p62)],
              obfv.generate_moduli([60, 58, 60], n)):
        cc = coracle.CCtx(OCtx(q, n))
        ck = full_size.host_key(cc, seed, len(q))
        c0 = np.stack([cc.synth_poly(seed, 0, 8 + 2 * i) for i in range(len(q))])
        c1 = np.stack([cc.synth_poly(seed, 0, 9 + 2 * i) for i in range(len(q))])
        ctx = fhe.Context(q, n)
        ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1).set_mode(mode)
        p = np.stack([cc.synth_poly(seed, i, 0) for i in range(3)])
        g0, g1 = ksk.key_switch(torch.from_numpy(p.view(np.int64)).cuda())
        for i in range(3):
            w0, w1 = ck.key_switch(p[i])
            assert np.array_equal(full_size.u64(g0[i]), w0) and np.array_equal(full_size.u64(g1[i]), w1)


def _raw_hip_streams():
    """hipStreamCreate / hipStreamDestroy straight from the HIP runtime: streams the engine is never told about."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def make():
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        return s.value

    def kill(h):
        assert hip.hipStreamDestroy(C.c_void_p(h)) == 0
    return make, kill


def test_workspace_bounds(fhe):
    make, kill = _raw_hip_streams()
    cases.case_workspace_bounds(fhe, make, kill, nstreams=60, n=256)


def test_workspace_1000_foreign_streams_c2(fhe):
    """1,000 short-lived streams of the host's own (never announced to the engine, destroyed behind its
    back), each doing one C2-shaped two-stream multiply, under fhe_workspace_set_limit(total = 2 footprints): the
    device's free memory stays within 2x of one stream's footprint, the internal second streams stay bounded, and every
    result is bit-identical to the first (which is checked against the C oracle)."""
    import full_size
    import torch
    from fhe_oracle import bfv as obfv, coracle, synth
    make, kill = _raw_hip_streams()
    n, sizes, batch = 8192, [60] * 4, 64
    q = obfv.generate_moduli(sizes, n)
    t = full_size.plaintext_modulus(n)
    seed = synth.seed_for_config(2)
    par = fhe.BfvParameters(n, t, moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = full_size.device_key(ctx, seed, len(q))
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1))
    m = fhe.Multiplicator.default(par, rk, 0)
    lhs, rhs = ctx.synth_uniform(seed, 0, 0, 2, batch), ctx.synth_uniform(seed, 0, 2, 2, batch)
    torch.cuda.synchronize()
    fhe.workspace_trim()
    torch.cuda.empty_cache()
    la, ra = fhe.DeviceArray.from_numpy(full_size.u64(lhs)), fhe.DeviceArray.from_numpy(full_size.u64(rhs))
    del lhs, rhs
    torch.cuda.empty_cache()
    free0, _ = fhe.device_mem_info(0)

    def one(handle):
        with fhe.Stream.foreign(handle) as st:
            out = m.multiply(la, ra)
            st.synchronize()
            got = out.download()
            out.free()
        return got
    h = make()
    first = one(h)
    o = full_size.oracle_level(n, q, t, 0)
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], full_size.host_key(o["cb"], seed, len(q)), False)
    for i in (0, 31, 63):
        l = np.stack([o["cb"].synth_poly(seed, i, 0), o["cb"].synth_poly(seed, i, 1)])
        r = np.stack([o["cb"].synth_poly(seed, i, 2), o["cb"].synth_poly(seed, i, 3)])
        assert np.array_equal(first[i], cm.multiply(l, r))
    free1, _ = fhe.device_mem_info(0)
    footprint = free0 - free1
    held1 = fhe.workspace_stats()["held_bytes"]
    assert footprint > 0 and held1 > 0
    kill(h)
    # what a host that churns streams of its own does: a retention bound of two streams' footprint
    fhe.workspace_set_limit(0, 2 * held1)
    worst = 0
    for it in range(1000):
        h = make()
        got = one(h)
        kill(h)
        if it % 50 == 0:
            assert np.array_equal(got, first), it
            fr, _ = fhe.device_mem_info(0)
            worst = max(worst, free0 - fr)
        assert fhe.workspace_stats()["held_bytes"] <= 2 * held1
    assert worst <= 2 * footprint + (64 << 20), (worst, footprint)
    assert fhe.workspace_stats()["internal_streams"] <= 32 + 3
    fhe.workspace_trim()
    fhe.workspace_set_limit()


@pytest.mark.parametrize("dev", [False, True])
def test_scaler_many_wide_moduli(fhe, dev):
    cases.case_scaler_many_wide_moduli(fhe, dev)


def test_scaler_every_instance(fhe):
    """Round 6: every scale_kernel<NF> instance (engine.hpp scale_kernel_nf: BASELINE's bases, the stock sets, the levels of
    C5's chain) on its exact fit and on a padded one, factor one (PLAIN) and a non-unit factor, vs the oracle."""
    cases.case_scaler_many_wide_moduli(fhe, True, counts=(2, 3, 5, 7, 8, 10, 11, 13, 14, 15, 16, 18, 19, 22, 23, 26, 27, 30, 31, 32),
                                       factors=((1, 1), (3, 7)))


@pytest.mark.parametrize("n", [32768, 65536])
def test_ntt_split_rows_narrow_moduli(fhe, n):
    """Rows larger than LDS over moduli below 2^60 only: the LDS halves of both transforms take the narrow passes
    (forward: input below 4p behind the global stages) -- two 60-bit primes and a 50-bit one, vs the C oracle."""
    from fhe_oracle import coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_oracle.zq import generate_prime
    p0 = generate_prime(60, 2 * n, 1 << 60)
    mods = [p0, generate_prime(60, 2 * n, p0), generate_prime(50, 2 * n, 1 << 50)]
    cases.case_ntt(fhe, True, n, moduli=mods, batch=5, coracle_ctx=coracle.CCtx(OCtx(mods, n)))


@pytest.mark.parametrize("mode", ["auto", "fused", "fused_sub", "unfused"])
def test_relin_rotate_rows_larger_than_lds(fhe, mode):
    """Relinearise and both rotations at N = 32768 (rows larger than LDS): the key switch reads the caller's Ntt rows for
    one transform per key modulus and adds the substituted c0 on the way out -- the sub-block offsets of those paths."""
    import full_size
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        full_size.check_relin_rotate(fhe, n=32768, sizes=[60, 60, 60, 58], batch=3, cfg=8)


@pytest.mark.parametrize("n,bits", [(16384, 60), (32768, 60), (32768, 62), (65536, 60), (65536, 62)])
def test_key_switch_decomposition_rows(fhe, n, bits):
    """Single-modulus key levels (base-2^k digits of one residue row) on whole rows and on the 16384-point parts / the
    8192-point sub-blocks of rows larger than LDS, narrow (60-bit) and general (62-bit) passes."""
    for mode in (1, 4) if n > 16384 else (1,):
        with fhe.KeySwitchingKey.forced_mode(mode):
            cases.case_key_switch_decomposition_rows(fhe, True, n, bits)


def test_key_switch_auto_picks_strategy_by_launch_size(fhe):
    """What FHE_KS_AUTO does (engine.hpp ks_use_unfused / key_switch_polys; profiles/r04_ks_small_batches_all_modes.txt,
    r04_ks_small_launch_ab.txt; round 5: r05_ks_rounds12_ab.jsonl), read back from the library's per-launch profiler: launches
    with at most 0.6 of the compute units' worth of fused workgroups run the unfused kernels, larger ones the fused kernel
    unless a nearly empty last round of workgroups follows one or two full ones; a key with fewer than three
    digits stays fused, on 8192-point sub-blocks while those fit the device at once (N >= 32768).  Values are compared
    with the oracle by every other test of this file; this one pins the choice."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def kernels_of(ctx, L, batch):
        kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
        rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
        ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
        rk.relinearizes(ct3)
        torch.cuda.synchronize()
        fhe.prof_reset()
        fhe.prof_enable(True)
        try:
            rk.relinearizes(ct3)
            torch.cuda.synchronize()
            return set(fhe.prof_report())
        finally:
            fhe.prof_enable(False)
            fhe.prof_reset()

    n, L = 8192, 4
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    small, large = cus // (2 * L), cus // L          # 2 x batch x L <= CUs  /  batch x L = CUs
    assert {"ks_digit_ntt", "ks_mac"} <= kernels_of(ctx, L, small) and "key_switch_fused" not in kernels_of(ctx, L, small)
    ks = kernels_of(ctx, L, large)
    assert "key_switch_fused" in ks and "ks_mac" not in ks
    # round 5 (profiles/r05_ks_rounds12_ab.jsonl): one workgroup per CU makes the fused launch's time a step function of
    # its rounds -- after one full round a last round at most half full goes to the unfused kernels (1.125 and 1.5
    # rounds), a fuller one stays fused (1.75 rounds), and so do exactly two rounds
    for rounds8, want_unfused in ((9, True), (12, True), (14, False), (16, False)):
        ks = kernels_of(ctx, L, rounds8 * cus // (8 * L))
        assert ("ks_mac" in ks) == want_unfused and ("key_switch_fused" in ks) == (not want_unfused), (rounds8, ks)
    # round 6 (profiles/r06_m_f64_ks_modes_grid.jsonl, r06_n_f64_ks_modes_grid_after.jsonl): the F64 instances (moduli below 2^50 -- the reference's stock sets) have
    # narrower windows after a full round: up to 0.6 of a round as the integer kernels (N = 8192 fused launches of at most one
    # workgroup per CU run the 1024-thread instance), then up to 4/9 (N = 8192, the 512-thread instance) / a quarter (N = 16384)
    # of the round after the first; never after two full rounds (r06_p_f64_ks_modes_grid_two_geometries.jsonl)
    import ref_params
    for n, cases_ in ((8192, ((9, 16, True), (3, 4, False), (5, 4, True), (3, 2, False), (17, 8, False))),
                      (16384, ((9, 16, True), (3, 4, False), (9, 8, True), (3, 2, False), (17, 8, False)))):
        q = ref_params.DEFAULT_128[n]
        ctx = fhe.Context(q, n)
        for num, den, want_unfused in cases_:
            ks = kernels_of(ctx, len(q), num * cus // (den * len(q)))
            assert ("ks_digit_ntt_f64" in ks) == want_unfused and ("key_switch_fused_f64" in ks) == (not want_unfused), (n, num, den, ks)
    n, L = 32768, 2                                  # two digits: never unfused; sub-blocks while they fit at once
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    ks = kernels_of(ctx, L, cus // (4 * L))          # batch x L x 4 sub-blocks = CUs
    assert "key_switch_fused_sub" in ks and "ks_mac" not in ks and "key_switch_fused" not in ks
    ks = kernels_of(ctx, L, cus // (4 * L) + 1)
    assert "key_switch_fused" in ks and "key_switch_fused_sub" not in ks



# ---- round 5: the reference's own stock parameter sets (BfvParameters::default_parameters_128, parameters.rs:218-251;
# what crates/fhe/benches/bfv.rs:27 iterates over) ----
@pytest.mark.parametrize("dev", [False, True])
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384])
def test_reference_default_parameter_sets(fhe, n, dev):
    """Every hot-path Criterion ID of benches/bfv.rs:167-286 (mul, square, mul_and_relin, relinearize, rotate_rows,
    rotate_columns, inner_sum, expand_4, mul_and_relin_2) and the leveled multiply + modulus-switch chain down to one
    modulus, on the explicit primes of each stock set, host-pointer and `_dev` entry points, vs the plain-C oracle."""
    import ref_params
    ref_params.check_all(fhe, dev, n, batch=3)


@pytest.mark.parametrize("mode", ["fused", "unfused"])
@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_reference_default_parameter_sets_key_switch_modes(fhe, n, mode):
    import ref_params
    with fhe.KeySwitchingKey.forced_mode(KS_MODES[mode]):
        ref_params.check_mul(fhe, True, n, relin=True, batch=3)
        ref_params.check_relin_rotate(fhe, True, n, batch=3)
        ref_params.check_inner_sum(fhe, True, n)
        ref_params.check_expand(fhe, True, n, 16)
        ref_params.check_chain(fhe, True, n, batch=2)


@pytest.mark.parametrize("n,batch", [(4096, 1024), (8192, 1024), (16384, 256)])
def test_reference_default_parameter_sets_bench_batches(fhe, n, batch):
    """The stock sets at the batches bench.py's `reference_default_128` entries time (device-generated inputs): sampled
    ciphertexts of multiply + relinearise against the C oracle."""
    import full_size
    import ref_params
    full_size.check_mul(fhe, n=n, sizes=None, moduli=ref_params.DEFAULT_128[n], batch=batch, relin=True, cfg=0x128,
                        sample=(0, 1, batch // 2 - 1, batch // 2, batch - 1))
    fhe.workspace_trim()


def test_workspace_limit_bounds_device_memory(fhe):
    """fhe_workspace_set_limit bounds the DEVICE MEMORY the process holds for scratch, not just the engine's
    table: after one large call under a small `total_bytes`, the scratch pool's reserved bytes (what the driver says,
    fhe_workspace_pool_stats) come back under the bound once the stream has been synchronised -- torch or any other
    allocator in the process can have the memory -- and results stay bit-identical to the unbounded run."""
    import torch
    import full_size
    from fhe_oracle import bfv as obfv
    n, sizes, batch = 8192, [60] * 4, 256
    q = obfv.generate_moduli(sizes, n)
    par = fhe.BfvParameters(n, full_size.plaintext_modulus(n), moduli=q)
    ctx = par.context_at_level(0)
    c0, c1 = full_size.device_key(ctx, 31, len(q))
    m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1)), 0).set_streams(1)
    a, b = ctx.synth_uniform(31, 0, 0, 2, batch), ctx.synth_uniform(31, 0, 2, 2, batch)
    fhe.workspace_trim()
    fhe.workspace_set_limit(0, 0)                      # unbounded: the pool keeps the call's whole footprint
    want = m.multiply(a, b)
    torch.cuda.synchronize()
    big = fhe.workspace_pool_stats(0)
    held = fhe.workspace_stats()["held_bytes"]
    assert held > (256 << 20) and big["scratch_reserved_bytes"] >= held, (big, held)
    limit = 64 << 20
    fhe.workspace_set_limit(0, limit)                  # evicts the idle blocks and trims the pool
    torch.cuda.synchronize()
    st = fhe.workspace_pool_stats(0)
    assert fhe.workspace_stats()["held_bytes"] <= limit
    assert st["scratch_reserved_bytes"] <= limit + (32 << 20), (st, big)
    got = m.multiply(a, b)                             # a call larger than the bound still runs ...
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    st = fhe.workspace_pool_stats(0)                   # ... and its blocks are not kept afterwards
    assert fhe.workspace_stats()["held_bytes"] <= limit
    assert st["scratch_reserved_bytes"] <= limit + (32 << 20), st
    # the ABI's own buffers live in the other pool and are untouched by the scratch bound
    with fhe.Stream(0) as s:
        d = fhe.DeviceArray((1 << 24,))
        assert fhe.workspace_pool_stats(0)["buffers_used_bytes"] >= (1 << 27)
        d.free()
        s.synchronize()
    s.destroy()
    fhe.workspace_set_limit()
    fhe.workspace_trim()


@pytest.mark.parametrize("n", [4096, 16384])
def test_galois_in_place_and_folded_substitution(fhe, n):
    """Round 5: from N = 4096 on a rotation reads c0 / c1 through the Ntt-domain substitution inside the inverse transform
    and the key switch (no separate permutation pass) -- which gathers from the input while the output is written.  A
    caller that rotates IN PLACE (out == ct through the `_dev` entry point) must therefore take the copying path: both
    forms against the C oracle, every key-switch strategy."""
    import ctypes as C
    import torch
    import full_size
    from fhe_oracle import bfv as obfv, coracle
    from fhe_oracle.rq import Context as OCtx
    from fhe_rs_amd import _lib
    q = obfv.generate_moduli([60, 58, 60], n)
    cc = coracle.CCtx(OCtx(q, n))
    seed = 0xF4E50A00 + n
    ck = full_size.host_key(cc, seed, len(q))
    ctx = fhe.Context(q, n)
    c0, c1 = full_size.device_key(ctx, seed, len(q))
    for mode in (0, 1, 2):
        ksk = fhe.KeySwitchingKey(ctx, ctx, c0, c1).set_mode(mode)
        for e in (3, 2 * n - 1, n + 1):
            ct = ctx.synth_uniform(seed, 0, 0, 2, 3)
            want = [ck.galois_relinearize(e, full_size.u64(ct[i])) for i in range(3)]
            out = fhe.GaloisKey(ksk, e).relinearize(ct)
            torch.cuda.synchronize()
            for i in range(3):
                assert np.array_equal(full_size.u64(out[i]), want[i]), (mode, e, i, "out of place")
            same = ct.clone()
            _lib.check(_lib.lib().fhe_bfv_galois_dev(ksk._h, e, C.c_void_p(same.data_ptr()), C.c_void_p(same.data_ptr()), 3,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            for i in range(3):
                assert np.array_equal(full_size.u64(same[i]), want[i]), (mode, e, i, "in place")


def test_measurement_aids(fhe):
    """The measurement aids behind bench.py's ceilings (not on the path): every fhe_ubench_int kind returns a positive
    rate, fhe_ubench_scaler runs the scaler's own kernel instance, fhe_ubench_copy streams; bad arguments are refused with
    the ABI's status codes instead of crashing."""
    for kind in fhe.UBENCH_KINDS:
        assert fhe.ubench_int(kind, 0.005) > 1e9, kind
    par = fhe.BfvParameters(4096, fhe.generate_prime(20, 8192, 1 << 20), moduli_sizes=[60, 60])
    assert fhe.ubench_scaler(par.extender(0), 0.005) > 1e8 and fhe.ubench_scaler(par.down_scaler(0), 0.005) > 1e8
    assert fhe.ubench_copy(1 << 26, 0.005) > 1e11

    def code(fn):
        try:
            fn()
        except fhe.FheError as e:
            return e.code
        return 0
    assert code(lambda: fhe.ubench_int(99, 0.01)) == -1 and code(lambda: fhe.ubench_int(0, 0.0)) == -1
    assert code(lambda: fhe.ubench_int(0, 0.01, device=77)) == -18
    assert code(lambda: fhe.ubench_copy(16, 0.01)) == -1 and code(lambda: fhe.ubench_copy(1 << 20, 100.0)) == -1
    assert code(lambda: fhe.ubench_scaler(par.extender(0), 0.0)) == -1
    host_only = fhe.Context(par.moduli, 4096, device=-1)
    assert code(lambda: fhe.ubench_scaler(fhe.Scaler(host_only, host_only, 3, 7), 0.01)) == -18
    st = fhe.workspace_pool_stats(0)
    assert set(st) == {"scratch_reserved_bytes", "scratch_used_bytes", "buffers_reserved_bytes", "buffers_used_bytes"}
    assert fhe.workspace_pool_stats(99)["scratch_reserved_bytes"] == 0


# ---- round 6: the FP64-FMA instances for moduli below 2^50 (csrc/zq_f64.hpp) ----------------------------------------
@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_f64_ntt(fhe, n):
    """Forward / inverse transforms on the F64 instances (launch classes 3 / 4 / 5) against the C oracle, the integer
    kernels on the same inputs, and the profiler's word on which kernels ran."""
    cases.case_f64_ntt(fhe, True, n)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("n,sizes", [(4096, (36, 36, 37)), (8192, (43, 43, 44, 44, 44)), (4096, (50, 50, 49, 48)),
                                     (16384, (48, 48, 48, 49, 49, 49, 49, 49, 49)), (8192, (27, 36, 43, 44, 48, 49, 50))])
def test_f64_key_switch(fhe, n, sizes, mode):
    """The key switch on its F64 instances -- fused (mode 1) and stage A of the unfused form (mode 2) -- with the
    reference's stock widths (parameters.rs:222-251), a 50-bit basis, and every width of VERDICT r05 #3's list in one
    basis; key_switch, relinearise, two rotations, against the C oracle, then with the option off."""
    cases.case_f64_key_switch(fhe, True, n, sizes, batch=3, exps=(3, 2 * n - 1), mode=mode)


@pytest.mark.parametrize("sizes", [(43, 43, 44, 44, 44), (50, 49, 48, 27)])
def test_f64_key_switch_two_workgroups_per_cu(fhe, sizes):
    """N = 8192: launches of more than one workgroup per CU run the 512-thread x 16-coefficient instance (two workgroups per
    CU; engine.hpp launch_ks_fused), smaller ones the tile's own 1024 threads -- the batch-3 cases above.  key_switch,
    relinearise, two rotations (the gathering loaders of the GAL instance) of enough ciphertexts for the former, against the C
    oracle, every ciphertext; then with the option off (the integer kernel on the same launch)."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cases.case_f64_key_switch(fhe, True, 8192, sizes, batch=cus // len(sizes) + 2, exps=(3, 2 * 8192 - 1), mode=1)


def test_f64_key_switch_accumulator_fold(fhe):
    """Sixteen 50-bit moduli: more digits than an F64 accumulator holds before it is reduced (class 3: nine terms)."""
    cases.case_f64_key_switch(fhe, True, 4096, (50,) * 16, batch=2, exps=(3,))


@pytest.mark.parametrize("n", [4096, 8192])
def test_stock_sets_on_the_integer_kernels(fhe, n):
    """The stock sets run on the F64 kernels by default (every other test of this file); with fhe_engine_set_f64(0) they
    take the integer narrow kernels as in rounds 1-5: every bench ID again, both key-switch strategies."""
    import ref_params
    fhe.set_f64(False)
    try:
        for mode in (1, 2):
            ref_params.FORCE_KS_MODE = mode
            ref_params.check_all(fhe, True, n, batch=2, expand_size=4, inner_sum=(mode == 1))
    finally:
        ref_params.FORCE_KS_MODE = None
        fhe.set_f64(True)


def test_f64_multiply_uses_every_f64_kernel(fhe):
    """Multiplicator::multiply on the stock n = 8192 set at a batch that fills the device: the launch labels of one call
    are the F64 instances for every narrow-row kernel (inverse / forward transforms of the ciphertext rows, the tensor's
    narrow run, the fused key switch) and the integer kernels only for the 62-bit extension rows and the scalers."""
    import ref_params
    par = ref_params.params(fhe, 8192)
    ctx = par.context_at_level(0)
    kk = ctx.synth_uniform(5, 0, 8, 2 * ctx.nmoduli, 1)[0].reshape(ctx.nmoduli, 2, ctx.nmoduli, ctx.degree)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
    m = fhe.Multiplicator.default(par, rk, 0)
    a, b = ctx.synth_uniform(5, 0, 0, 2, 256), ctx.synth_uniform(5, 0, 2, 2, 256)
    want = m.multiply(a, b).clone()
    fhe.prof_reset()
    fhe.prof_enable(True)
    got = m.multiply(a, b)
    fhe.prof_enable(False)
    labels = set(fhe.prof_report())
    # (the key switch: fused, or -- KS_AUTO's choice for launches that do not fill the device -- unfused with the forward
    # transform of (c0, c1) riding on its stage A; either way on the F64 instances)
    assert {"ntt_inv_f64", "tensor_intt_f64"} <= labels, labels
    assert ("key_switch_fused_f64" in labels and "ntt_fwd_f64" in labels) or "ks_digit_ntt_f64" in labels, labels
    for integer_label in ("key_switch_fused", "ks_digit_ntt", "tensor_intt_narrow", "ntt_inv"):
        assert integer_label not in labels, labels
    import torch
    assert torch.equal(got, want)
    fhe.set_f64(False)
    try:
        assert torch.equal(m.multiply(a, b), want)
    finally:
        fhe.set_f64(True)


@pytest.mark.parametrize("n", [4096, 8192])
def test_f64_callers_match_the_integer_kernels(fhe, n):
    """The callers either side of the path on an F64-eligible basis (the stock sets' moduli): RGSW external product, oblivious
    expansion, inner sum, rotations by several steps and decryption run their transforms and key switches on the F64
    instances; every output must equal, word for word, what the integer kernels give on the same inputs (those are tied
    to the oracle by the small-size cases of this file and by the stock-set checks of tests/ref_params.py)."""
    import torch
    import ref_params
    par = ref_params.params(fhe, n)
    ctx = par.context_at_level(0)
    L = ctx.nmoduli

    def key(seed):
        kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
        return fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())
    ksk, ksk2 = key(31), key(32)
    ct = ctx.synth_uniform(31, 0, 0, 2, 24)
    s_ntt = ctx.synth_uniform(31, 77, 0, 1, 1)[0, 0].contiguous()
    seq, i = [], 1
    while i < n // 2:
        seq.append(pow(3, i, 2 * n))
        i *= 2
    seq.append(2 * n - 1)
    levels = 4
    ek = fhe.EvaluationKey(n, [fhe.GaloisKey(ksk, e) for e in set(seq + [(n >> lv) + 1 for lv in range(levels)])])
    rgsw = fhe.RGSWCiphertext(ksk, ksk2)

    def run():
        return [rgsw.external_product(ct).clone(), ek.computes_inner_sum(ct).clone(), ek.expands(ct[:3].contiguous(), 1 << levels).clone(),
                ek.rotates_columns_by(ct, 1).clone(), ek.rotates_rows(ct).clone(), par.decrypt(s_ntt, ct, 0).clone()]
    fhe.prof_reset()
    fhe.prof_enable(True)
    got = run()
    fhe.prof_enable(False)
    labels = set(fhe.prof_report())
    assert "ntt_inv_f64" in labels and ("key_switch_fused_f64" in labels or "ks_digit_ntt_f64" in labels), labels
    fhe.set_f64(False)
    try:
        want = run()
    finally:
        fhe.set_f64(True)
    for g, w, name in zip(got, want, ("rgsw", "inner_sum", "expand", "rotate_columns", "rotate_rows", "decrypt")):
        assert torch.equal(g, w), (n, name)
