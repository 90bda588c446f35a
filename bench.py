#!/usr/bin/env python3
"""bench.py -- BFV ct x ct + relinearise throughput (and NTT/s) on MI355X: BASELINE.json's metric.

One "step" = one pass of the hot path (`Multiplicator::default(rk).multiply`, fhe.rs
crates/fhe/src/bfv/ops/mul.rs:165-243) over one batch of synthetic ciphertext pairs per GPU.
Workload (config C2, BASELINE.json configs[1]): N = 8192, 4 x 60-bit RNS moduli
(K = 9 rows in the extended basis), batch = 1024 ciphertext pairs per GPU, relin key at level 0.
Inputs (and the synthetic relin key) are generated ON the device by the shared splitmix64
counter generator and are resident in HBM before the timed region starts.  Setup also makes one
call of the path so that the stream's workspace exists (a one-time hipMalloc); the W warmup steps
and the K timed steps follow.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), independent
ciphertexts sharded by rank, no data-path collective ("weak" scaling: per-GPU batch is fixed);
barrier + synchronize on both sides of the timed region, MAX over ranks.  `python bench.py --gpus N`
starts its own N ranks when it is not already running under torchrun (WORLD_SIZE unset); under
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.  When fewer than N
devices are visible (a 1-GPU box) the ranks share devices and rendezvous over gloo.

`--gpus N` with N > 1 measures BASELINE.json configs[3] (C4: batch 65536 over 8 GPUs = 8192 pairs per GPU) unless
`--batch` says otherwise; N = 1 measures C2 (1024 pairs).  Each rank pins its host thread to its GPU's NUMA node.

Prints ONE JSON line on rank 0:
  value / ms_per_step   the K timed steps, single-stream mode with the library's per-launch HIP events on
                        (per-kernel durations are exact there; `roofline` comes from this region)
  event_free            the same K steps without the events (their overhead, stated rather than assumed)
  default_mode          the same K steps in the handle's default two-stream mode (kernels of the two streams
                        overlap, so per-kernel durations are not attributable there: informational)
  ntt                   forward NTT of [batch*2][4][8192] (fhe-math/benches/ntt.rs:12-38): Poly-NTT/s, row-NTT/s
  other_configs         C3 relinearise / rotations (batch 512) and C5 level-0 multiply+relin+mod-switch (batch 16)
  roofline              dominant kernel, HIP-event timed on the launching stream; `roofline.int_issue`: the second
                        ceiling SURVEY 8(d) asks for -- register-resident butterfly / multiply rates measured in this
                        process (fhe_ubench_int) and every NTT-type kernel as a fraction of them
  host_api              the drop-in host-pointer entry point (fhe_bfv_mul: H2D + compute + D2H per call) at batch 1
                        and at the bench batch, and the `_dev` path on buffers / a stream owned through the C ABI
  per_rank              (N > 1) every rank's own ops/s and the max/min skew
  cpu_baseline          (N = 1) the plain-C port of the reference algorithm on this box's host cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_DEGREE = 8192
MODULI_SIZES = [60, 60, 60, 60]
BATCH_PER_GPU = 1024        # C2 (configs[1]): the single-GPU workload the metric is quoted on
BATCH_PER_GPU_SHARDED = 8192  # C4 (configs[3]): 65536 ciphertext pairs over 8 GPUs
SEED = 0xF4E50002           # 0xF4E50000 + cfg
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def stage_model_rows(L, K, Lk):
    """SURVEY.md §8(d) stage model, in units of R = 8N bytes per ct x ct + relin."""
    return 22 * K + 7 * L + L * Lk + 4 * Lk


def cpu_baseline(n, sizes, t, seed, budget_s):
    """Times the oracle's plain-C restatement of Multiplicator::multiply (same algorithm and
    pass structure as the reference's single-threaded Rust) on this host: single thread, then
    all cores batch-parallel (one ciphertext pair per task).  Checker code is timed here, never
    shipped: this is the only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from fhe_oracle import bfv as obfv, coracle
    import full_size
    q = obfv.generate_moduli(sizes, n)
    o = full_size.oracle_level(n, q, t, 0)
    crk = full_size.host_key(o["cb"], seed, len(q))
    cm = coracle.CMul(o["cb"], o["cm"], o["cel"], o["cel"], o["cdn"], crk, False)
    npairs = 16
    lhs = np.stack([np.stack([o["cb"].synth_poly(seed, i, 0), o["cb"].synth_poly(seed, i, 1)]) for i in range(npairs)])
    rhs = np.stack([np.stack([o["cb"].synth_poly(seed, i, 2), o["cb"].synth_poly(seed, i, 3)]) for i in range(npairs)])
    cm.time_multiply(lhs, rhs, 2, 1)  # warm up (page in tables)
    s1, _ = cm.time_multiply(lhs, rhs, 24, 1)
    single = 24 / s1
    threads = coracle.max_threads()
    # calibrate the all-core rate on a short run, then size the timed sample to the budget
    cal_n = threads * 2
    cal_s, _ = cm.time_multiply(lhs, rhs, cal_n, threads)
    count = int(max(threads * 2, min(budget_s, 30.0) * cal_n / cal_s))
    count -= count % npairs          # the last op then is pair npairs-1 (used for the parity spot check)
    sN, last = cm.time_multiply(lhs, rhs, count, threads)
    return dict(value=round(count / sN, 2), unit="ops/s", cores=threads, kind="port",
                sample_short=f"{count} C2 ct x ct + relin ops, {threads} OpenMP threads, {sN:.1f} s (plain-C port)",
                sample=f"{count} ct x ct + relinearise ops of the C2 workload (16 distinct synthetic pairs cycled), "
                       f"{threads} OpenMP threads batch-parallel, {sN:.1f} s",
                single_thread_ops_per_s=round(single, 2)), cm, (lhs, rhs, last, count, npairs)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (one process per GPU, the same
    environment contract torchrun gives: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), let rank 0
    print the line, and return non-zero if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            rc = p.wait() or rc
            if rc:
                break
    finally:
        for p in procs:        # a failed rank must not leave the others waiting in a collective
            if p.poll() is None:
                p.kill()
    return rc


def pin_to_gpu_numa_node(torch, dev):
    """Pins this rank's host threads to the CPUs of its GPU's NUMA node (launch latency and the host-pointer copies
    otherwise cross the socket interconnect on a two-socket box).  Returns what was done, for the JSON line."""
    try:
        props = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return dict(pci=bdf, numa_node=None, pinned=False)
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return dict(pci=bdf, numa_node=node, pinned=False)
        os.sched_setaffinity(0, cpus)
        return dict(pci=bdf, numa_node=node, pinned=True, cpus=len(cpus))
    except Exception as e:   # no sysfs entry / no permission: run unpinned, say so
        return dict(numa_node=None, pinned=False, error=str(e)[:80])


def read_sclk_mhz():
    """Current engine clock from sysfs (the level pp_dpm_sclk marks with '*'); None when not exposed."""
    import glob
    best = None
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    best = max(best or 0, int(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
        except Exception:
            pass
    return best


def sclk_under_load(work, max_seconds=6.0):
    """Engine clock while `work()` keeps the GPU busy: `rocm-smi --showclocks` (which reads the SMU's current gfxclk,
    not the DPM level table) runs once as a subprocess while `work` is called in a loop.  None if rocm-smi is missing."""
    import re
    import shutil
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.Popen([exe, "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        t0 = time.perf_counter()
        while p.poll() is None and time.perf_counter() - t0 < max_seconds:
            work()
        out = p.communicate(timeout=10)[0]
        vals = [int(v) for v in re.findall(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)]
        return max(vals) if vals else None
    except Exception:
        return None


def bf_row(n):
    """Butterflies of one row transform of length n."""
    return (n // 2) * (n.bit_length() - 1)


def key_switch_work(n, L, rates, narrow=True, inverse_rows=None, f64=False):
    """Integer work of ONE key switch of a level-0 polynomial (key_switching_key.rs:241-320; relinearization_key.rs:69-102,
    galois_key.rs:63-123 wrap it): the inverse transform of the switched polynomial (L rows; relinearise: c2, rotation: the
    substituted c1), L (L - 1) digit transforms forward -- digit j under key modulus j is the caller's own Ntt-form row
    (`xhat`, kernels_ks.hpp) -- and two accumulator sets of L x L Shoup multiply-accumulates per coefficient.
    (count, rate) pairs in lane-operations.  f64 (round 6): every modulus below 2^50 -- the FP64-FMA kernels' own
    register-resident rates (csrc/zq_f64.hpp), so the ceiling is the FP64 issue ceiling."""
    fwd = rates["f64_fwd_butterfly"] if f64 else rates["fwd_butterfly_narrow"] if narrow else rates["fwd_butterfly"]
    inv = rates["f64_inv_butterfly"] if f64 else rates["inv_butterfly"]
    mac = rates["f64_mac"] if f64 else rates["shoup_mac"]
    inv_rows = L if inverse_rows is None else inverse_rows
    return [(inv_rows * bf_row(n), inv), (L * (L - 1) * bf_row(n), fwd), (2 * L * L * n, mac)]


def mul_relin_work(n, L, K, rates, col_ext, col_down, narrow=True, mod_switch=False, f64=False):
    """Integer work of one Multiplicator::multiply with relinearisation (mul.rs:165-243) by kernel family, as (count, rate)
    pairs: the L ciphertext primes take the narrow forward passes when they are below 2^60, the K - L extension primes
    (62-bit) the wide ones; inverse passes are priced with the one inverse rate.  mod_switch: + Ciphertext::switch_down of
    the two result polynomials (2L inverse rows, 2(L-1) forward rows, 2(L-1) lazy Shoup products per coefficient)."""
    b = bf_row(n)
    fw, fn_, iv = rates["fwd_butterfly"], rates["fwd_butterfly_narrow"], rates["inv_butterfly"]
    fq = fn_ if narrow else fw
    work = {
        "ntt_fwd": [(4 * (K - L) * b, fw), (2 * L * b, fq)],
        "ntt_inv": [(4 * L * b, iv)],
        # fused tensor + inverse NTT: per row of the extended basis two single products (slots 0, 2) and one double
        # product (slot 1) per coefficient, then three inverse row transforms
        "tensor_intt": [(2 * K * n, rates["tensor_mul"]), (K * n, rates["tensor_mac2"]), (3 * K * b, iv)],
        # fused key switch of c2 (all L x L digit transforms) + two accumulator sets of L x L Shoup MACs
        "key_switch_fused": [(L * L * b, fq), (2 * L * L * n, rates["shoup_mac"])],
    }
    if f64:   # round 6: the L ciphertext rows (all below 2^50) on the FP64 kernels; the K - L extension rows as before
        ff, fi, fm, fmac = rates["f64_fwd_butterfly"], rates["f64_inv_butterfly"], rates["f64_mulmod"], rates["f64_mac"]
        work["ntt_fwd"] = [(4 * (K - L) * b, fw), (2 * L * b, ff)]
        work["ntt_inv"] = [(4 * L * b, fi)]
        work["tensor_intt"] = [(2 * (K - L) * n, rates["tensor_mul"]), ((K - L) * n, rates["tensor_mac2"]), (3 * (K - L) * b, iv),
                               (4 * L * n, fm), (3 * L * b, fi)]       # 2 single + 1 double product = 4 exact products per coefficient
        work["key_switch_fused"] = [(L * L * b, ff), (2 * L * L * n, fmac)]
    if col_ext:
        work["scale_extend"] = [(4 * n, col_ext)]                # 4 operand polynomials, one column per coefficient
        work["scale_down"] = [(3 * n, col_down)]                 # 3 tensor slots
    if mod_switch:
        work["mod_switch"] = [(2 * L * b, iv), (2 * (L - 1) * b, fq), (2 * (L - 1) * n, rates["shoup_lazy"])]
    return work


def ideal_seconds(work):
    parts = work.values() if isinstance(work, dict) else [work]
    return sum(c / r for ps in parts for c, r in ps)


def ceiling_entry(ops_per_s, stage_bytes, ideal_s):
    """One config against both ceilings of SURVEY 8(d): the HBM stage model at 8 TB/s and the integer-issue ceiling
    1 / (its arithmetic at this box's register-resident rates)."""
    hbm_ops, int_ops = HBM_PEAK_GBS * 1e9 / stage_bytes, 1.0 / ideal_s
    return dict(ops_per_s=round(ops_per_s, 1), frac_hbm=round(ops_per_s / hbm_ops, 4), frac_int_issue=round(ops_per_s / int_ops, 4),
                binding="int_issue" if int_ops < hbm_ops else "hbm", ceiling_ops_per_s=round(min(hbm_ops, int_ops), 1),
                frac_of_binding_ceiling=round(ops_per_s / min(hbm_ops, int_ops), 4),
                ideal_int_us_per_op=round(ideal_s * 1e6, 3), stage_model_bytes_per_op=int(stage_bytes))


def binding_ceilings(other, rates, c2_value, n, L, K):
    """`{frac_hbm, frac_int_issue, binding}` for every BASELINE config and the reference's stock sets, from
    this run's own timings (`other_configs`) and this process's register-resident rates (`roofline.int_issue.rates`).
    Shapes: relinearization_key.rs:69-102, galois_key.rs:63-123, mul.rs:165-243."""
    out = {}
    sc = rates.get("scaler_cols_per_s", {})
    if sc.get("C2"):
        out["C2"] = ceiling_entry(c2_value, stage_model_rows(L, K, L) * 8 * n, ideal_seconds(mul_relin_work(n, L, K, rates, *sc["C2"])))
    n3, L3 = 16384, 8
    for name, rows in (("C3_relinearize", 2 * L3 + L3 * L3 + 4 * L3), ("C3_rotate_columns", 2 * L3 + L3 * L3 + 3 * L3),
                       ("C3_rotate_rows", 2 * L3 + L3 * L3 + 3 * L3)):
        if other.get(name):
            out[name] = ceiling_entry(other[name]["ops_per_s"], rows * 8 * n3, ideal_seconds(key_switch_work(n3, L3, rates)))
    n5 = 32768
    for name in ("C5_level0_mul_relin_modswitch", "C5_level0_mul_relin_modswitch_batch64"):
        e = other.get(name)
        if e and e.get("shape"):
            L5, K5, ce, cd = e["shape"]
            out[name.replace("_mul_relin_modswitch", "")] = ceiling_entry(
                e["ops_per_s"], e["stage_model_bytes_per_op"], ideal_seconds(mul_relin_work(n5, L5, K5, rates, ce, cd, mod_switch=True)))
    ch = other.get("C5_chain_15_levels")
    if ch and ch.get("levels_shape"):
        ideal = 0.0
        for Ll, Kl, ce, cd in ch["levels_shape"]:
            ideal += ideal_seconds(mul_relin_work(n5, Ll, Kl, rates, ce, cd, mod_switch=True))
            ideal += ideal_seconds([(2 * Ll * bf_row(n5), rates["inv_butterfly"]), (2 * (Ll - 1) * bf_row(n5), rates["fwd_butterfly_narrow"]),
                                    (2 * (Ll - 1) * n5, rates["shoup_lazy"])])        # the second operand's own switch_down
        out["C5_chain"] = ceiling_entry(ch["chains_per_s"], ch["stage_model_bytes_per_chain"], ideal)
    for key, st in (other.get("reference_default_128") or {}).items():
        if not isinstance(st, dict) or "ids" not in st or not st.get("scaler_cols_per_s"):
            continue
        ns, Ls, Ks = st["degree"], st["moduli"], st["mul_basis_rows"]
        ce, cd = st["scaler_cols_per_s"]
        ids, tag = st["ids"], "stock%d_" % ns
        f64 = bool(st.get("f64"))
        ks = ideal_seconds(key_switch_work(ns, Ls, rates, f64=f64))
        for idn, rows in (("relinearize", 2 * Ls + Ls * Ls + 4 * Ls), ("rotate_columns", 2 * Ls + Ls * Ls + 3 * Ls)):
            if idn in ids and "batch_ops_per_s" in ids[idn]:
                out[tag + idn] = ceiling_entry(ids[idn]["batch_ops_per_s"], rows * 8 * ns, ks)
        if "mul_and_relin" in ids and "batch_ops_per_s" in ids["mul_and_relin"]:
            out[tag + "mul_and_relin"] = ceiling_entry(ids["mul_and_relin"]["batch_ops_per_s"], stage_model_rows(Ls, Ks, Ls) * 8 * ns,
                                                       ideal_seconds(mul_relin_work(ns, Ls, Ks, rates, ce, cd, f64=f64)))
    out["note"] = ("ops_per_s: this run; frac_hbm: SURVEY 8(d) stage-model bytes x ops/s over 8 TB/s; frac_int_issue: ops/s x (the "
                   "op's butterflies, MACs, tensor products and scaler columns at this process's register-resident rates; stock "
                   "sets, round 6: their rows below 2^50 at the FP64-FMA kernels' rates, i.e. the issue ceiling of the "
                   "instructions they really run); binding: the lower ceiling; lifts, final reductions, loads / stores are not priced")
    return out


def int_issue_roofline(fhe, dev, prof, n, L, K, batch, steps, pipeline_step=None, sync=None, par=None, value_per_gpu=None,
                       stage_bytes=None):
    """"measure it (microbench v_mad_u64_u32 throughput) and report both ceilings".  Runs the library's
    no-HBM loops -- the butterflies, the tensor products, the key switch's Shoup multiply-accumulate (register resident)
    and the two scale_kernel instances of the multiply on one L2-resident polynomial -- for ~0.3 s in this process,
    samples the engine clock while they run, and prices EVERY kernel family of the timed region against them:
    frac_of_ceiling = (time its arithmetic would take at those rates) / (its HIP-event time).  Round 5: the numerators
    cover the whole operation (rounds 1-4 counted transform butterflies only), so `whole_op` is the operation's integer
    ceiling and `binding` says which of the two ceilings -- integer issue or the HBM stage model -- is the lower one."""
    import threading
    clocks, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            v = read_sclk_mhz()
            if v:
                clocks.append(v)
            stop.wait(0.01)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    rates = {k: fhe.ubench_int(k, 0.03, dev) for k in ("mad_u64_u32", "mul_lo_u32", "mul_hi_u32", "shoup_lazy",
                                                        "fwd_butterfly", "fwd_butterfly_narrow", "inv_butterfly",
                                                        "shoup_mac", "tensor_mul", "tensor_mac2", "f64_fma", "f64_mulmod",
                                                        "f64_fwd_butterfly", "f64_inv_butterfly", "f64_mac")}
    col_ext = col_down = None
    if par is not None:
        col_ext = fhe.ubench_scaler(par.extender(0), 0.03)
        col_down = fhe.ubench_scaler(par.down_scaler(0), 0.03)
    stop.set()
    th.join()
    # the clock the SMU reports while the integer pipe is saturated, and while the real pipeline runs
    sclk_ubench = sclk_under_load(lambda: fhe.ubench_int("fwd_butterfly_narrow", 0.05, dev))
    sclk_pipeline = None
    if pipeline_step is not None:
        def work():
            for _ in range(4):
                pipeline_step()
            sync()
        sclk_pipeline = sclk_under_load(work)
    fw, fn_, iv = rates["fwd_butterfly"], rates["fwd_butterfly_narrow"], rates["inv_butterfly"]
    mac, tm, tm2 = rates["shoup_mac"], rates["tensor_mul"], rates["tensor_mac2"]
    # work per ct x ct + relin by kernel family: (count, rate) pairs; counts in lane-operations (mul_relin_work)
    work = mul_relin_work(n, L, K, rates, col_ext, col_down)
    rates["scaler_cols_per_s"] = {"C2": [col_ext, col_down]} if col_ext else {}
    per_kernel, ideal_op_s = {}, 0.0
    for name, parts in work.items():
        ideal_one = sum(c / rate for c, rate in parts)           # seconds per operation at the no-HBM rates
        ideal_op_s += ideal_one
        if name in prof and prof[name][1] > 0:
            d = dict(ideal_us_per_op=round(ideal_one * 1e6, 4),
                     frac_of_ceiling=round(ideal_one * batch * steps / (prof[name][1] * 1e-3), 4))
            if name in ("ntt_fwd", "ntt_inv", "tensor_intt", "key_switch_fused"):
                rows = sum(c for c, _ in parts if c % bf_row(n) == 0 and c >= bf_row(n))
                d["butterflies_per_s"] = round(rows * batch * steps / (prof[name][1] * 1e-3), 0)
            per_kernel[name] = d
    out = dict(butterflies_per_s_ceiling=dict(forward_wide=round(fw, 0), forward_narrow_lt_2p60=round(fn_, 0),
                                              inverse=round(iv, 0)),
               row_ntt_per_s_ceiling=dict(forward_wide=round(fw / bf_row(n), 0), forward_narrow_lt_2p60=round(fn_ / bf_row(n), 0),
                                          inverse=round(iv / bf_row(n), 0)),
               rates=rates,
               mad_u64_u32_per_s=round(rates["mad_u64_u32"], 0), mul_lo_u32_per_s=round(rates["mul_lo_u32"], 0),
               mul_hi_u32_per_s=round(rates["mul_hi_u32"], 0), shoup_lazy_per_s=round(rates["shoup_lazy"], 0),
               shoup_mac_per_s=round(mac, 0), tensor_mul_per_s=round(tm, 0), tensor_mac2_per_s=round(tm2, 0),
               scale_extend_columns_per_s=round(col_ext, 0) if col_ext else None,
               scale_down_columns_per_s=round(col_down, 0) if col_down else None,
               sclk_mhz_observed=sclk_ubench or (round(sum(clocks) / len(clocks)) if clocks else None),
               sclk_mhz_during_pipeline=sclk_pipeline,
               sclk_source=("rocm-smi --showclocks (SMU current gfxclk) sampled once under each load" if sclk_ubench
                            else "sysfs pp_dpm_sclk level table (the active LEVEL, not the instantaneous clock)"),
               sclk_dpm_level_mhz=(round(sum(clocks) / len(clocks)) if clocks else None), per_kernel=per_kernel,
               note="fhe_ubench_int / fhe_ubench_scaler, this process, this box; numerators: transform butterflies, tensor "
                    "products, key-switch MACs (register-resident loops) and scaler columns (scale_kernel on one "
                    "L2-resident polynomial); lifts, final reductions and loads / stores are not priced")
    if col_ext and value_per_gpu:
        ceil_ops = 1.0 / ideal_op_s
        out["whole_op"] = dict(ceiling_ops_per_s=round(ceil_ops, 1), ideal_us_per_op=round(ideal_op_s * 1e6, 3),
                               frac=round(value_per_gpu / ceil_ops, 4),
                               what="1 / sum over the six kernel families of (their arithmetic at the no-HBM rates): the "
                                    "operation's integer-issue ceiling if the chip did nothing but that arithmetic")
        if stage_bytes:
            hbm_ops = HBM_PEAK_GBS * 1e9 / stage_bytes
            out["hbm_whole_op"] = dict(ceiling_ops_per_s=round(hbm_ops, 1), frac=round(value_per_gpu / hbm_ops, 4))
            out["binding"] = "int_issue" if ceil_ops < hbm_ops else "hbm"
            out["value_over_min_ceiling"] = round(value_per_gpu / min(ceil_ops, hbm_ops), 4)
    return out


COMPACT_LIMIT = 1900        # the driver keeps a 2,000-character tail of stdout: the record must fit in it whole, with margin
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_record(result):
    """The ONE line the driver parses (round 5's 24.6 KB line was cut by the driver's 2,000-character
    tail and left no record).  Takes the full result (everything bench.py measured; that goes to bench_detail.json and
    to an earlier `DETAIL ` stdout line) and returns a JSON string below COMPACT_LIMIT characters holding exactly what
    the record needs: the contract's keys, `roofline`, `cpu_baseline`, `ntt`, the N > 1 summary and the binding-ceiling
    fractions of the other BASELINE configs.  Optional blocks are dropped in a fixed order should it ever not fit; the
    contract's keys, `roofline` and `cpu_baseline` never are."""
    rec = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                         "scaling", "vs_baseline", "dtype", "data", "value_min", "value_max"))
    rec["config"] = _pick(result.get("config", {}), ("workload", "batch_per_gpu", "global_batch", "parallelism",
                                                     "dist_backend", "dist_world_size"))
    rf = result.get("roofline", {})
    rec["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                 "traffic_observed_this_run", "kernel_sum_ms_per_step", "kernel_sum_le_step", "launches",
                                 "avg_launch_ms", "algorithmic_bytes_per_launch", "binding", "frac_of_binding_ceiling",
                                 "frac_hbm_whole_op", "frac_int_issue_whole_op"))
    ds = rf.get("dominant_by_symbol")
    if ds:
        rec["roofline"]["dominant_symbol"] = _pick(ds, ("kernel", "avg_launch_ms", "frac", "share_of_kernel_time"))
        rec["roofline"]["dominant_symbol"]["kernel"] = str(ds.get("kernel", ""))[:64]
    if "cpu_baseline" in result:
        cb = result["cpu_baseline"]
        rec["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_ops_per_s"))
        rec["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:96]
    if "ntt" in result:
        rec["ntt"] = _pick(result["ntt"], ("poly_ntt_per_s", "row_ntt_per_s", "frac"))
    if "parity_spot_check" in result:
        rec["parity"] = "bit-identical to the oracle (spot check)"
    mg = result.get("multi_gpu")
    if mg:
        rec["multi_gpu"] = _pick(mg, ("rccl_world_size", "dist_backend", "distinct_devices", "one_device_per_rank",
                                      "efficiency_vs_1gpu_same_batch", "data_path_collectives"))
        n1 = (mg.get("n1_reference") or {}).get("batch_%d" % BATCH_PER_GPU)
        if n1:
            rec["multi_gpu"]["n1_batch_1024_value"] = n1.get("value")
        if result.get("per_rank"):
            rec["multi_gpu"]["rank_max_over_min"] = result["per_rank"].get("max_over_min")
    if result.get("binding_ceilings"):
        # {config: [frac of 8 TB/s (stage model), frac of the integer-issue ceiling, ops/s]} for the BASELINE configs and the
        # reference's stock n = 8192 / 16384 sets; the binding ceiling and every other ID: the detail file
        keep = ("C3_relinearize", "C3_rotate_columns", "C5_level0", "C5_chain", "stock8192_mul_and_relin",
                "stock8192_relinearize", "stock16384_mul_and_relin", "stock16384_relinearize")
        rate = lambda v: None if v.get("ops_per_s") is None else int(round(v["ops_per_s"]))
        rec["configs"] = {k: [v.get("frac_hbm"), v.get("frac_int_issue"), rate(v)] for k, v in result["binding_ceilings"].items()
                          if k in keep and isinstance(v, dict)}
    if result.get("errors"):
        rec["errors"] = len(result["errors"])
    rec["detail"] = os.path.basename(DETAIL_FILE)
    dumps = lambda r: json.dumps(r, separators=(",", ":"))
    line = dumps(rec)
    for drop in ("configs", "parity", "ntt", "multi_gpu"):      # never reached with today's keys; a guard, not a plan
        if len(line) < COMPACT_LIMIT:
            break
        rec.pop(drop, None)
        line = dumps(rec)
    if len(line) >= COMPACT_LIMIT:
        rec["roofline"].pop("dominant_symbol", None)
        rec["config"]["workload"] = rec["config"].get("workload", "")[:100]
        line = dumps(rec)
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(result):
    """Writes the full result to bench_detail.json, prints it as a `DETAIL ` line (so that it cannot be mistaken for the
    record) and prints the compact record as the LAST stdout line."""
    try:
        with open(DETAIL_FILE, "w") as f:
            json.dump(result, f, indent=1)
    except OSError as e:              # a read-only checkout must not cost the record
        result.setdefault("errors", []).append("bench_detail.json not written: %s" % e)
    sys.stdout.write("DETAIL " + json.dumps(result) + "\n")
    sys.stdout.write(compact_record(result) + "\n")
    sys.stdout.flush()


def short_symbol(sym):
    """`void fhe::k::ntt_kernel<false, 13, true, 1, false>(unsigned long const*, ...)` -> `ntt_kernel<false, 13, true, 1, false>`."""
    s = sym[5:] if sym.startswith("void ") else sym
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            s = s[:i]
            break
    return s.replace("fhe::k::", "")


def workload_name(world, batch):
    if world > 1 and batch == BATCH_PER_GPU_SHARDED:
        return (f"C4: BFV n=8192, 4x60-bit RNS moduli (K=9), batch={batch * world} ct x ct + relinearize sharded over "
                f"{world} GPUs ({batch} per GPU)")
    tag = "C2" if batch == BATCH_PER_GPU else "C2 shape"
    return f"{tag}: BFV n=8192, 4x60-bit RNS moduli (K=9), batch={batch} ct x ct + relinearize per GPU"


def host_api_numbers(fhe, _lib, torch, mul, ctx, par, rk, batch, n, L, value_hint=None):
    """What a host that keeps `Poly.coefficients` in its own memory gets from the drop-in entry points.
    host_pointer: fhe_bfv_mul (pageable host arrays in, host array out: H2D + pipeline + D2H inside every call).
    abi_device_buffers: fhe_bfv_mul_dev on buffers from fhe_buf_alloc and a stream from fhe_stream_create (no torch
    allocator or stream involved): the device-resident path a Rust / C host uses."""
    import numpy as np
    out = {}
    m2 = fhe.Multiplicator.default(par, rk, 0)          # product defaults (two streams)
    for b in (1, batch):
        lh = ctx.synth_uniform(SEED, 0, 0, 2, b).cpu().numpy().view(np.uint64)
        rh = ctx.synth_uniform(SEED, 0, 2, 2, b).cpu().numpy().view(np.uint64)
        # (the entry point itself, on arrays that exist and have been touched: the Python wrapper would allocate a
        # fresh zeroed result per call, whose page faults cost more than the copy)
        oh = np.ones((b,) + tuple(m2.multiply(lh[:1], rh[:1]).shape[1:]), dtype=np.uint64)
        hcall = lambda: _lib.check(_lib.lib().fhe_bfv_mul(m2._h, lh.ctypes.data_as(_lib.u64p), rh.ctypes.data_as(_lib.u64p),
                                                          oh.ctypes.data_as(_lib.u64p), b))
        hcall()
        reps = 20 if b == 1 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            hcall()
        dt = (time.perf_counter() - t0) / reps
        moved = b * (2 * 2 + 2) * L * n * 8
        out[f"host_pointer_batch{b}"] = dict(ops_per_s=round(b / dt, 1), ms_per_call=round(dt * 1e3, 3),
                                            pcie_GBps=round(moved / dt / 1e9, 2))
        del lh, rh, oh
    # the same entry point on PINNED host memory (fhe_host_alloc): what a host gets when it keeps its coefficient
    # storage in page-locked memory -- the copies then run at the link's DMA rate instead of being staged by HIP
    import ctypes as C
    L_ = _lib.lib()
    ct_bytes = batch * 2 * L * n * 8
    ptrs = []
    for _ in range(3):
        pp = C.c_void_p()
        _lib.check(L_.fhe_host_alloc(ct_bytes, C.byref(pp)))
        ptrs.append(pp)
    try:
        arrs = [np.frombuffer((C.c_uint8 * ct_bytes).from_address(pp.value), dtype=np.uint64).reshape(batch, 2, L, n)
                for pp in ptrs]
        arrs[0][:] = ctx.synth_uniform(SEED, 0, 0, 2, batch).cpu().numpy().view(np.uint64)
        arrs[1][:] = ctx.synth_uniform(SEED, 0, 2, 2, batch).cpu().numpy().view(np.uint64)
        call = lambda: _lib.check(L_.fhe_bfv_mul(m2._h, arrs[0].ctypes.data_as(_lib.u64p), arrs[1].ctypes.data_as(_lib.u64p),
                                                 arrs[2].ctypes.data_as(_lib.u64p), batch))
        call()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        dt = (time.perf_counter() - t0) / 3
        out[f"host_pointer_batch{batch}_pinned"] = dict(ops_per_s=round(batch / dt, 1), ms_per_call=round(dt * 1e3, 3),
                                                       pcie_GBps=round(batch * 6 * L * n * 8 / dt / 1e9, 2),
                                                       note="operands and result in fhe_host_alloc memory")
        del arrs
    finally:
        for pp in ptrs:
            L_.fhe_host_free(pp)
    with fhe.Stream(ctx.device) as st:
        la, ra = ctx.synth_uniform(SEED, 0, 0, 2, batch), ctx.synth_uniform(SEED, 0, 2, 2, batch)
        assert isinstance(la, fhe.DeviceArray)
        m2.multiply(la, ra)
        st.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            o = m2.multiply(la, ra)
        st.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["abi_device_buffers"] = dict(ops_per_s=round(batch / dt, 1), ms_per_call=round(dt * 1e3, 3), batch=batch,
                                         note="fhe_buf_alloc(_async) / fhe_stream_create / fhe_bfv_mul_dev; every call allocates its result stream-ordered and drops the previous one")
        for x in (la, ra, o):
            x.free()
    st.destroy()
    return out


def key_for(fhe, ctx, seed):
    L = ctx.nmoduli
    kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
    return fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())


def make_timeit(torch, reps=3):
    """ms per call of fn(): one untimed call, then `reps` calls between two events on torch's current stream (the
    stream every `_dev` call of this file is issued on)."""
    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    return timeit


def graph_replay_ms(torch, fn, reps=50):
    """ms per replay of `fn`'s launches captured into a hipGraph (torch.cuda.CUDAGraph on a side stream, after two warm-up
    calls there so that the stream's workspace exists); None when capture is not possible.  VERDICT r04 #4 asked for the
    eager / graph pair: a call's time is the sum of its kernels' single-workgroup latencies (gaps 0.2-0.9 us per launch,
    profiles/r05_latency_breakdown_*.json), so a replay buys nothing -- the line says so with numbers."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 4)
    except Exception:
        return None


def reference_bench_ids(fhe, torch, par, rk, batch, timeit):
    """The rest of the reference's own Criterion IDs on the C2 parameter set (informational, same process):
    crates/fhe/benches/bfv.rs:219-245 `mul` (&c1 * &c2, three parts, no relinearisation), `square` (&c1 * &c1),
    `mul_then_relinearize` (the two calls in sequence; `mul_and_relin` -- Multiplicator::multiply -- is `value`);
    crates/fhe-math/benches/rns.rs:32-53 `scaler` / `scaler_as_converter` (3 -> 4 moduli of the reference's lists, per
    coefficient column = one RnsScaler::scale call); benches/rq.rs:214-260 `mul_shoup_assign` and the two
    `change_representation` IDs (= Poly-NTT/s backward; forward is the line's `ntt`)."""
    out = {}
    n, ctx = par.degree, par.context_at_level(0)
    L, K = ctx.nmoduli, par.mul_context_at_level(0).nmoduli
    R = 8 * n
    a, b = ctx.synth_uniform(SEED, 0, 0, 2, batch), ctx.synth_uniform(SEED, 0, 2, 2, batch)
    plain = fhe.Multiplicator.default(par, None, 0)

    def entry(ms, rows, units=batch, unit_name="ops_per_s"):
        gbs = units * rows * R / ms / 1e6 if rows else None
        d = {unit_name: round(units / ms * 1e3, 1), "ms": round(ms, 3)}
        if rows:
            d.update(stage_model_bytes_per_op=rows * R, stage_model_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
        return d
    out["bfv/mul"] = entry(timeit(lambda: plain.multiply(a, b)), 22 * K + 9 * L)          # SURVEY 8(d) "&ct*&ct no relin"
    # &c1 * &c1: the same buffer as both operands -> the operand is extended once (engine's squaring shortcut);
    # stage model: extension of 2 polynomials instead of 4
    out["bfv/square"] = entry(timeit(lambda: plain.multiply(a, a)), 22 * K + 9 * L - (4 * L + 2 * K + 4 * (K - L)))
    out["bfv/square_general_tensor"] = entry(timeit(lambda: plain.tensor(a, a)), 22 * K + 9 * L)
    c3 = plain.multiply(a, b)
    out["bfv/relinearize"] = entry(timeit(lambda: rk.relinearizes(c3)), 2 * L + L * L + 4 * L)
    out["bfv/mul_then_relinearize"] = entry(timeit(lambda: rk.relinearizes(plain.multiply(a, b))),
                                            22 * K + 9 * L + 2 * L + L * L + 4 * L)
    # the reference's Criterion IDs time ONE ciphertext per call: the same calls at batch 1 (informational; caller's stream)
    a1, b1, c31 = a[:1].contiguous(), b[:1].contiguous(), c3[:1].contiguous()
    m1 = fhe.Multiplicator.default(par, rk, 0)
    out["single_ciphertext_latency_c2"] = dict(mul_and_relin_ms=round(timeit(lambda: m1.multiply(a1, b1)), 4),
                                               relinearize_ms=round(timeit(lambda: rk.relinearizes(c31)), 4),
                                               mul_and_relin_graph_replay_ms=graph_replay_ms(torch, lambda: m1.multiply(a1, b1)),
                                               relinearize_graph_replay_ms=graph_replay_ms(torch, lambda: rk.relinearizes(c31)),
                                               note="one ciphertext (pair) per call, inputs resident; *_graph_replay_ms: the same "
                                                    "launches replayed from a captured hipGraph")
    del c3, plain, a1, b1, c31, m1
    # fhe-math/benches/rns.rs: the reference's 3 -> 4 modulus lists, PowerBasis columns of [batch*2] polynomials
    q3 = [4611686018326724609, 4611686018309947393, 4611686018282684417]
    p4 = [4611686018257518593, 4611686018232352769, 4611686018171535361, 4611686018106523649]
    cq, cp = fhe.Context(q3, n), fhe.Context(p4, n)
    x = cq.synth_uniform(SEED, 0, 0, 1, batch * 2)
    cols = batch * 2 * n
    for name, num, den in (("rns/scaler/3->4", 1, 46116860181065), ("rns/scaler_as_converter/3->4", 1, 1)):
        sc = fhe.Scaler(cq, cp, num, den)
        ms = timeit(lambda: sc.scale(x, ntt=False))
        out[name] = dict(columns_per_s=round(cols / ms * 1e3, 0), ms=round(ms, 3),
                         GBps=round(cols * 7 * 8 / ms / 1e6, 1), frac=round(cols * 7 * 8 / ms / 1e6 / HBM_PEAK_GBS, 4))
    del x, cq, cp
    # fhe-math/benches/rq.rs
    polys = a.view(batch * 2, L, n)
    q8 = ctx.synth_uniform(SEED, 0, 2, 2, 4).view(8, L, n)              # eight distinct NttShoup operands, cycled
    reps8 = (batch * 2 + 7) // 8
    qs = q8.repeat(reps8, 1, 1)[: batch * 2].contiguous()
    qshoup = torch.from_numpy(ctx.shoup(q8.cpu().numpy().view("uint64")).view("int64")).cuda().repeat(reps8, 1, 1)[: batch * 2].contiguous()
    ms = timeit(lambda: ctx.mul_shoup(polys, qs, qshoup))
    out["rq/mul_shoup_assign"] = dict(polys_per_s=round(batch * 2 / ms * 1e3, 1), ms=round(ms, 3),
                                      GBps=round(batch * 2 * 4 * L * R / ms / 1e6, 1),
                                      frac=round(batch * 2 * 4 * L * R / ms / 1e6 / HBM_PEAK_GBS, 4))
    # fhe-math/benches/rq.rs:120-148 (`rq_add_assign`, `rq_sub_assign`, `rq_mul_assign`, `rq_neg`: element-wise on Ntt polys;
    # the non-assign forms compute the same values into a new Poly) and benches/zq.rs:10-57 (the same operations on one
    # residue row: a Poly with one modulus).  Bytes: two operands read, one written (neg: one read, one written).
    other = b.view(batch * 2, L, n)
    for name, fn, rows_rw in (("rq_add_assign", lambda: ctx.add(polys, other), 3), ("rq_sub_assign", lambda: ctx.sub(polys, other), 3),
                              ("rq_mul_assign", lambda: ctx.mul(polys, other), 3), ("rq_neg", lambda: ctx.neg(polys), 2)):
        ms = timeit(fn)
        out[f"rq/{name}"] = dict(polys_per_s=round(batch * 2 / ms * 1e3, 1), ms=round(ms, 3),
                                 GBps=round(batch * 2 * rows_rw * L * R / ms / 1e6, 1),
                                 frac=round(batch * 2 * rows_rw * L * R / ms / 1e6 / HBM_PEAK_GBS, 4),
                                 workload=f"{batch * 2} polys of {L} x {n}; zq/*_vec is the same kernel on one row")
    # rq_dot_product/opt (rq.rs:150-195): 256 x 256 polynomial dot product, L = 4
    pv, qv = ctx.synth_uniform(SEED, 0, 0, 1, 256), ctx.synth_uniform(SEED, 300, 0, 1, 256).reshape(256, L, n)
    ms = timeit(lambda: ctx.dot_product_scalar(pv, qv))
    out["rq_dot_product/opt/256"] = dict(ms=round(ms, 4), dot_products_per_s=round(1e3 / ms, 1),
                                         GBps=round((2 * 256 + 1) * L * R / ms / 1e6, 1),
                                         note="ONE dot product of 256 polynomial pairs per call (Criterion's shape): a launch that does not fill the device")
    del pv, qv
    ms = timeit(lambda: ctx.ntt_backward(polys))
    out["rq/change_representation/Ntt_to_PowerBasis"] = dict(
        poly_ntt_per_s=round(batch * 2 / ms * 1e3, 1), row_ntt_per_s=round(batch * 2 * L / ms * 1e3, 1), ms=round(ms, 3),
        frac=round(batch * 2 * 2 * L * R / ms / 1e6 / HBM_PEAK_GBS, 4))
    return out


def other_configs(fhe, torch, reps=3):
    """Informational (never `value`): the other single-GPU configs of BASELINE.json on this box, same process.
    C3 (fhe/benches/bfv.rs:167-194): N=16384, 8x60-bit, relinearise 3->2 and the two rotations, batch 512.
    C5 (bfv.rs:247-255 shape at the top of a deep chain): N=32768, 16x60-bit, multiply + relinearise +
    modulus switch at level 0, batch 16 and 64.  Stage-model bytes per op: SURVEY.md §8(d)."""
    timeit = make_timeit(torch, reps)
    out = {}
    n, L, batch = 16384, 8, 512
    ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
    ksk = key_for(fhe, ctx, 0xF4E50003)
    rk, gk3, gkr = fhe.RelinearizationKey(ksk), fhe.GaloisKey(ksk, 3), fhe.GaloisKey(ksk, 2 * n - 1)
    ct3 = ctx.synth_uniform(0xF4E50003, 0, 0, 3, batch)
    ct2 = ct3[:, :2].contiguous()
    R = 8 * n
    for name, fn, rows in (("C3_relinearize", lambda: rk.relinearizes(ct3), 2 * L + L * L + 4 * L),
                           ("C3_rotate_columns", lambda: gk3.relinearize(ct2), 2 * L + L * L + 3 * L),
                           ("C3_rotate_rows", lambda: gkr.relinearize(ct2), 2 * L + L * L + 3 * L)):
        ms = timeit(fn)
        gbs = batch * rows * R / ms / 1e6
        out[name] = dict(workload=f"n=16384, 8x60-bit, batch {batch}", ops_per_s=round(batch / ms * 1e3, 1),
                         ms=round(ms, 3), stage_model_bytes_per_op=rows * R, stage_model_GBps=round(gbs, 1),
                         frac=round(gbs / HBM_PEAK_GBS, 4))
    # one ciphertext (the reference's own benches are single-ciphertext calls): the launch does not fill the device with
    # fused workgroups, FHE_KS_AUTO takes the unfused key switch (profiles/r04_ks_small_batches_all_modes.txt)
    one3, one2 = ct3[:1].contiguous(), ct2[:1].contiguous()
    lat = {"C3_relinearize_ms": round(timeit(lambda: rk.relinearizes(one3)), 4),
           "C3_rotate_columns_ms": round(timeit(lambda: gk3.relinearize(one2)), 4),
           "C3_relinearize_graph_replay_ms": graph_replay_ms(torch, lambda: rk.relinearizes(one3))}
    del ct3, ct2, one3, one2, rk, gk3, gkr, ksk, ctx
    n, L = 32768, 16
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    q = fhe.generate_moduli([60] * L, n)
    K = L + (60 * L + 60 + 61) // 62
    ext, upper = [], 1 << 62
    while len(ext) < K - L:            # Multiplicator::default's extension primes (mul.rs:110-126)
        upper = fhe.generate_prime(62, 2 * n, upper)
        if upper not in q:
            ext.append(upper)
    ctx, mctx = fhe.Context(q, n), fhe.Context(q + ext, n)
    Q = 1
    for m in q:
        Q *= m
    extender, down = fhe.Scaler(ctx, mctx, 1, 1), fhe.Scaler(mctx, ctx, t, Q)
    mul = fhe.Multiplicator(extender, extender, down, fhe.RelinearizationKey(key_for(fhe, ctx, 0xF4E50005)), True)
    rows = 22 * K + 7 * L + L * L + 4 * L + 12 * L - 6
    c5_shape = [L, K, fhe.ubench_scaler(extender, 0.03), fhe.ubench_scaler(down, 0.03)]
    for batch in (16, 64):
        a, b = ctx.synth_uniform(0xF4E50005, 0, 0, 2, batch), ctx.synth_uniform(0xF4E50005, 0, 2, 2, batch)
        ms = timeit(lambda: mul.multiply(a, b))
        gbs = batch * rows * 8 * n / ms / 1e6
        out["C5_level0_mul_relin_modswitch" + ("" if batch == 16 else f"_batch{batch}")] = dict(
            workload=f"n=32768, 16x60-bit (K={K}), batch {batch}", ops_per_s=round(batch / ms * 1e3, 1), ms=round(ms, 3),
            stage_model_bytes_per_op=rows * 8 * n, stage_model_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4),
            shape=c5_shape)
        del a, b
    a, b = ctx.synth_uniform(0xF4E50005, 0, 0, 2, 1), ctx.synth_uniform(0xF4E50005, 0, 2, 2, 1)
    lat["C5_level0_mul_relin_modswitch_ms"] = round(timeit(lambda: mul.multiply(a, b)), 4)
    lat["C5_level0_mul_relin_modswitch_graph_replay_ms"] = graph_replay_ms(torch, lambda: mul.multiply(a, b), 20)
    out["single_ciphertext_latency"] = dict(lat, note="one ciphertext (pair) per call, caller's stream, inputs resident")
    return out


def next_rows(fhe, torch, par, timeit):
    """SURVEY.md 8(f) -- the callers and data formats either side of the path -- on the C2 parameter set, same process
    (informational, never `value`).  IDs follow the reference's benches: crates/fhe/benches/bfv_optimized_ops.rs
    (`dot_product/opt`), benches/bfv.rs (`inner_sum`, `expand_*`, `mul_and_relin_2` :257-286, `decrypt`),
    benches/bfv_rgsw.rs (external product), fhe-math/benches/rq.rs (serialisation).  Every entry carries the
    stage-model bytes it is priced with (R = 8N bytes per residue row) and its fraction of 8 TB/s."""
    import numpy as np
    out = {}
    n, ctx = par.degree, par.context_at_level(0)
    L = ctx.nmoduli
    R = 8 * n

    def entry(ms, units, rows_per_unit, unit="ops_per_s", note=None, workload=None):
        gbs = units * rows_per_unit * R / ms / 1e6
        d = {unit: round(units / ms * 1e3, 1), "ms": round(ms, 3), "stage_model_bytes_per_unit": int(rows_per_unit * R),
             "stage_model_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        if workload:
            d["workload"] = workload
        if note:
            d["note"] = note
        return d

    # The box's own streaming rates, read + write bytes counted: (a) torch's 2 GiB device-to-device copy (hipMemcpyDtoD:
    # ~5.1 TB/s on these boxes -- NOT a ceiling, the path's own streaming kernels beat it), (b) the library's
    # 16-byte-per-lane copy kernel with streaming loads / stores (fhe_ubench_copy; the guide's "achievable" HBM rate is
    # ~6.3 TB/s).  The streaming rows below carry their fraction of the FASTER of the two next to the fraction of the
    # nominal 8 TB/s.
    src = torch.empty(1 << 28, dtype=torch.int64, device=f"cuda:{ctx.device}")
    dst = torch.empty_like(src)
    copy_ms = timeit(lambda: dst.copy_(src))
    copy_gbs = 2 * src.numel() * 8 / copy_ms / 1e6
    del src, dst
    kern_gbs = fhe.ubench_copy(1 << 31, 0.05, ctx.device) / 1e9
    out["hbm_streaming_rates"] = dict(
        torch_d2d_copy_GBps=round(copy_gbs, 1), torch_d2d_copy_frac=round(copy_gbs / HBM_PEAK_GBS, 4),
        copy16_kernel_GBps=round(kern_gbs, 1), copy16_kernel_frac=round(kern_gbs / HBM_PEAK_GBS, 4),
        note="2 GiB each, read + write bytes; torch D2D is hipMemcpyDtoD and is not a ceiling; `frac_of_measured_stream_rate` "
             "below is against the faster of the two")
    stream_gbs = max(copy_gbs, kern_gbs)
    _entry = entry

    def entry(ms, units, rows_per_unit, unit="ops_per_s", note=None, workload=None):   # noqa: F811
        d = _entry(ms, units, rows_per_unit, unit, note, workload)
        d["frac_of_measured_stream_rate"] = round(d["stage_model_GBps"] / stream_gbs, 4)
        return d

    ksk = key_for(fhe, ctx, SEED + 0x100)
    # -- PIR server loop: dot_product_scalar, 256 query ciphertexts shared by 32 database rows (tools/bench_kernels.py)
    count, rowsdb = 256, 32
    q = ctx.synth_uniform(SEED, 0, 0, 2, count)
    db = ctx.synth_uniform(SEED, 1000, 0, count, rowsdb).reshape(rowsdb, count, L, n)
    ms = timeit(lambda: ctx.dot_product_scalar(q, db))
    out["pir/dot_product_scalar"] = entry(ms, rowsdb * count, (rowsdb * count * L + count * 2 * L + rowsdb * 2 * L) / (rowsdb * count),
                                          unit="ct_pt_mac_per_s", workload=f"{count} query cts (shared) x {rowsdb} db rows",
                                          note="unique bytes: the db rows once, the queries once, the outputs")
    del q, db
    # -- ct x pt, one plaintext per ciphertext
    b = 1024
    ct = ctx.synth_uniform(SEED, 0, 0, 2, b)
    pt = ctx.synth_uniform(SEED, 0, 2, 1, b).reshape(b, L, n)
    ms = timeit(lambda: ctx.mul_plain(ct, pt))
    out["pir/mul_plain"] = entry(ms, b, 2 * L + L + 2 * L, workload=f"batch {b}, one plaintext per ciphertext")
    del pt
    # -- wire format (fhe-math rq/convert.rs): PowerBasis polys <-> bit-packed payload (60 of 64 bits per coefficient)
    polys = ct.view(b * 2, L, n)
    packed_rows = L * 60 / 64
    ms = timeit(lambda: ctx.serialize(polys))
    out["wire/serialize"] = entry(ms, b * 2, L + packed_rows, unit="polys_per_s", workload=f"{b * 2} polys")
    blob = ctx.serialize(polys)
    ms = timeit(lambda: ctx.deserialize(blob))
    out["wire/deserialize"] = entry(ms, b * 2, L + packed_rows, unit="polys_per_s")
    ms = timeit(lambda: ctx.deserialize(blob, to_ntt=True))
    out["wire/deserialize_into_ntt"] = entry(ms, b * 2, L + packed_rows + 2 * L, unit="polys_per_s")
    del blob
    # -- seeded c1 (Poly::random_from_seed): SHA-256 + ChaCha8 + rejection sampling per polynomial; compute-bound
    seeds = torch.arange(b * 2 * 32, dtype=torch.int64, device=ct.device).to(torch.uint8).reshape(b * 2, 32)
    ms = timeit(lambda: ctx.random_from_seed(seeds))
    out["wire/seed_expand"] = entry(ms, b * 2, L, unit="polys_per_s", note="bytes = the polynomial written; the kernel is ChaCha / SHA bound, not HBM bound")
    del seeds, polys
    # -- decrypt (phase, inverse NTT, ciphertext->plaintext scaler, final reduction)
    s_ntt = ctx.synth_uniform(SEED, 77, 0, 1, 1)[0, 0].contiguous()
    pc = par.plaintext_context().nmoduli
    ms = timeit(lambda: par.decrypt(s_ntt, ct, 0))
    out["decrypt"] = entry(ms, b, (2 * L + L) + 2 * L + (L + pc) + (pc + 1), workload=f"batch {b}, two parts")
    # -- RGSW external product: two fused key switches per ciphertext
    rgsw = fhe.RGSWCiphertext(ksk, key_for(fhe, ctx, SEED + 0x101))
    ms = timeit(lambda: rgsw.external_product(ct))
    out["rgsw/external_product"] = entry(ms, b, 2 * (2 * L) + 2 * (L * L + 2 * L) + 2 * L, workload=f"batch {b}")
    del rgsw
    # -- inner sum: log2(N/2) column rotations + one row rotation, each a Galois key switch + add
    seq, i = [], 1
    while i < n // 2:
        seq.append(pow(3, i, 2 * n))
        i *= 2
    seq.append(2 * n - 1)
    gal_rows = 2 * (2 * L) + 2 * L + (L * L + 4 * L)         # substitute (2 polys r+w), inverse NTT, fused key switch + addend
    ek = fhe.EvaluationKey(n, [fhe.GaloisKey(ksk, e) for e in seq])
    bs = 256
    cs = ct[:bs].contiguous()
    ms = timeit(lambda: ek.computes_inner_sum(cs))
    out["inner_sum"] = entry(ms, bs, len(seq) * (gal_rows + 3 * 2 * L), workload=f"batch {bs}, {len(seq)} Galois key switches each")
    del ek, cs
    # -- oblivious expansion of ONE ciphertext to N outputs (13 levels; 8,191 Galois key switches; 4 GiB out)
    levels = n.bit_length() - 1
    ek = fhe.EvaluationKey(n, [fhe.GaloisKey(ksk, (n >> l) + 1) for l in range(levels)])
    one = ct[:1].contiguous()
    ms = timeit(lambda: ek.expands(one, n))
    out["pir/expand"] = entry(ms, n - 1, gal_rows + 8 * L, unit="galois_applications_per_s", workload=f"1 ciphertext -> {n} (levels: {levels})",
                              note=f"{round(ms, 2)} ms per expansion = {round(1e3 / ms, 1)} expansions/s")
    del ek, one, ct
    # -- mul_and_relin_2: HPS second strategy (rhs pre-scaled by P/Q, product scaled by t/P; benches/bfv.rs:257-286)
    qs = par.moduli
    nm = (sum(int(m).bit_length() for m in qs) + 61) // 62
    ext, upper = [], (1 << 64) - 1 >> 2
    while len(ext) < nm:
        upper = fhe.generate_prime(62, 2 * n, upper)
        if upper not in qs:
            ext.append(upper)
    Q, P = 1, 1
    for m_ in qs:
        Q *= int(m_)
    for m_ in ext:
        P *= int(m_)
    mctx = fhe.Context(list(qs) + ext, n)
    mul2 = fhe.Multiplicator(fhe.Scaler(ctx, mctx, 1, 1), fhe.Scaler(ctx, mctx, P, Q), fhe.Scaler(mctx, ctx, int(par.plaintext), P),
                             fhe.RelinearizationKey(ksk))
    a, bb = ctx.synth_uniform(SEED, 0, 0, 2, b), ctx.synth_uniform(SEED, 0, 2, 2, b)
    K2 = L + nm
    ms = timeit(lambda: mul2.multiply(a, bb))
    out["bfv/mul_and_relin_2"] = entry(ms, b, stage_model_rows(L, K2, L), workload=f"batch {b}, basis q ++ P ({K2} rows), rhs factor P/Q, post factor t/P")
    return out


def reference_default_128(fhe, torch, cpu_ms=None, sets=(4096, 8192, 16384)):
    """The reference's OWN workload: every hot-path Criterion ID of crates/fhe/benches/bfv.rs:100-286 on the stock sets of
    BfvParameters::default_parameters_128(20) (parameters.rs:218-251; explicit primes, log q = 109 / 218 / 438), keyed the
    way Criterion names them ("<id>" under "n=<n>/log(q)=<logq>").  Per ID: (a) `single_ms` -- one ciphertext per call,
    inputs resident, what `b.iter(|| ...)` times; (b) `batch_ops_per_s` at `batch` ciphertexts per call; and, when the CPU
    leg ran, `cpu_port_single_thread_ms` -- the plain-C port of the reference algorithm for the same ID on this box's
    host (one thread, as the reference runs) -- with the two ratios.  Parity of every ID on these sets:
    tests/ref_params.py (GPU suite) and tests/golden/default128_digest.json."""
    out = {}
    for n in sets:
        q = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
             8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
             16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                     0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}[n]
        logq = sum(int(m).bit_length() for m in q)
        t = fhe.generate_prime(20, 2 * n, (1 << 20) - 1)
        par = fhe.BfvParameters(n, t, moduli=q)
        ctx = par.context_at_level(0)
        L, K = ctx.nmoduli, par.mul_context_at_level(0).nmoduli
        batch = 1024 if n <= 8192 else 256
        seed = 0xF4E5D128 + n
        ksk = key_for(fhe, ctx, seed)
        rk = fhe.RelinearizationKey(ksk)
        seq, i = [], 1
        while i < n // 2:
            seq.append(pow(3, i, 2 * n))
            i *= 2
        seq.append(2 * n - 1)
        ek = fhe.EvaluationKey(n, [fhe.GaloisKey(ksk, e) for e in set(seq + [3] + [(n >> l) + 1 for l in range(4)])])
        plain = fhe.Multiplicator.default(par, None, 0)
        mul = fhe.Multiplicator.default(par, rk, 0)
        nm = (logq + 61) // 62                                   # benches/bfv.rs:258-266
        ext, upper = [], (1 << 64) - 1 >> 2
        while len(ext) < nm:
            upper = fhe.generate_prime(62, 2 * n, upper)
            if upper not in q:
                ext.append(upper)
        Q, P = 1, 1
        for m_ in q:
            Q *= m_
        for m_ in ext:
            P *= m_
        mctx = fhe.Context(q + ext, n)
        mul2 = fhe.Multiplicator(fhe.Scaler(ctx, mctx, 1, 1), fhe.Scaler(ctx, mctx, P, Q), fhe.Scaler(mctx, ctx, t, P), rk)
        ids = {}
        for b, key, reps in ((1, "single_ms", 20), (batch, "batch", 3)):
            timeit = make_timeit(torch, reps)
            a, bb = ctx.synth_uniform(seed, 0, 0, 2, b), ctx.synth_uniform(seed, 0, 2, 2, b)
            c3 = plain.multiply(a, bb)
            be = min(b, 64)                                     # expansions write 2^i ciphertexts per input
            ae = a[:be].contiguous()
            todo = [("add_ct", lambda: ctx.add(a, bb)), ("sub_ct", lambda: ctx.sub(a, bb)), ("neg", lambda: ctx.neg(a)),
                    ("relinearize", lambda: rk.relinearizes(c3)), ("rotate_rows", lambda: ek.rotates_rows(a)),
                    ("rotate_columns", lambda: ek.rotates_columns_by(a, 1)), ("inner_sum", lambda: ek.computes_inner_sum(a))]
            todo += [("expand_%d" % lv, (lambda lv=lv: ek.expands(ae, 1 << lv))) for lv in range(1, 5)]
            todo += [("mul", lambda: plain.multiply(a, bb)), ("square", lambda: plain.multiply(a, a)),
                     ("mul_then_relinearize", lambda: rk.relinearizes(plain.multiply(a, bb))),
                     ("mul_and_relin", lambda: mul.multiply(a, bb)), ("mul_and_relin_2", lambda: mul2.multiply(a, bb))]
            for name, fn in todo:
                # (batch entries: the median of three timings -- one timing of three calls moved by up to 8 % between the legs
                # of one process, r06_final5: 167.9 k here against 182.8 k in tools/f64_ab.py's three repetitions)
                ms = timeit(fn) if b == 1 else sorted(timeit(fn) for _ in range(3))[1]
                d = ids.setdefault(name, {})
                if b == 1:
                    d["single_ms"] = round(ms, 4)
                    if name == "mul_and_relin":
                        d["single_graph_replay_ms"] = graph_replay_ms(torch, fn)
                else:
                    nb = be if name.startswith("expand_") else b
                    d["batch"] = nb
                    d["batch_ops_per_s"] = round(nb / ms * 1e3, 1)
            # (add / sub / neg are the *Assign forms here: `a` changes, which the timings do not care about)
            del a, bb, c3, ae
            fhe.workspace_trim()
            torch.cuda.empty_cache()
        if cpu_ms and n in cpu_ms:
            for name, d in ids.items():
                c = cpu_ms[n].get(name)
                if c:
                    d["cpu_port_single_thread_ms"] = c
                    d["single_call_speedup_vs_cpu_port_thread"] = round(c / d["single_ms"], 1)
                    d["batch_speedup_vs_cpu_port_thread"] = round(d["batch_ops_per_s"] * c / 1e3, 1)
        stage = stage_model_rows(L, K, L) * 8 * n
        ids["mul_and_relin"]["stage_model_bytes_per_op"] = stage
        ids["mul_and_relin"]["frac"] = round(stage * ids["mul_and_relin"]["batch_ops_per_s"] / 1e9 / HBM_PEAK_GBS, 4)
        out[f"n={n}/log(q)={logq}"] = dict(degree=n, moduli=len(q), mul_basis_rows=K, plaintext=t, ids=ids, f64=fhe.get_f64(),
                                           scaler_cols_per_s=[fhe.ubench_scaler(par.extender(0), 0.03),
                                                              fhe.ubench_scaler(par.down_scaler(0), 0.03)])
        del par, ctx, ksk, rk, ek, plain, mul, mul2, mctx
        fhe.workspace_trim()
        torch.cuda.empty_cache()
    out["note"] = ("Criterion group `bfv` of crates/fhe/benches/bfv.rs on default_parameters_128(20); single_ms = one ciphertext per "
                   "call on the caller's stream with inputs resident (what b.iter times), batch_ops_per_s = the same call on a "
                   "batch; cpu_port_single_thread_ms = oracle/c/fhe_oracle.c (kind: port), one host thread")
    return out


def c5_chain(fhe, torch, batch=16):
    """BASELINE.json configs[4] as written: the DEEP chain -- multiply + relinearise + modulus switch at every level
    0 ... 14 of N = 32768, 16 x 60-bit (L_l = 16 ... 2), each level's output feeding the next (a squaring chain on
    synthetic ciphertexts).  Total time of the 15 levels and every level's own rate."""
    n, L = 32768, 16
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
    levels = L - 1
    muls, rows, shapes = [], [], []
    for lv in range(levels):
        ctx = par.context_at_level(lv)
        rk = fhe.RelinearizationKey(key_for(fhe, ctx, 0xF4E50005 + lv))
        muls.append(fhe.Multiplicator.default(par, rk, lv, mod_switch=True))
        Ll, Kl = ctx.nmoduli, par.mul_context_at_level(lv).nmoduli
        rows.append(22 * Kl + 7 * Ll + Ll * Ll + 4 * Ll + 12 * Ll - 6)
        rows[-1] += 12 * Ll - 6                        # the second operand's own modulus switch (below)
        shapes.append([Ll, Kl, fhe.ubench_scaler(par.extender(lv), 0.02), fhe.ubench_scaler(par.down_scaler(lv), 0.02)])
    c0 = par.context_at_level(0)
    x0, y0 = c0.synth_uniform(0xF4E50005, 0, 0, 2, batch), c0.synth_uniform(0xF4E50005, 0, 2, 2, batch)
    ctxs = [par.context_at_level(lv) for lv in range(levels)]

    def chain(evs=None):
        x, y = x0, y0
        for i, m in enumerate(muls):
            xn = m.multiply(x, y)                      # level i -> i + 1 (relinearised, modulus-switched)
            y = ctxs[i].ciphertext_switch_down(y)      # the other operand follows one level down
            x = xn
            if evs is not None:
                evs[i + 1].record()
        return x
    chain()                      # workspace of every level exists
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(levels + 1)]
    reps, per_level = 3, [0.0] * levels
    for _ in range(reps):
        evs[0].record()
        chain(evs)
        torch.cuda.synchronize()
        for i in range(levels):
            per_level[i] += evs[i].elapsed_time(evs[i + 1]) / reps
    total = sum(per_level)
    gbs = batch * sum(rows) * 8 * n / total / 1e6
    return dict(workload=f"n=32768, 16x60-bit, {levels} levels (L = 16 ... 2), batch {batch}; each level: x <- Multiplicator(x, y) with relinearisation + modulus switch, y <- switch_down(y)",
                total_ms=round(total, 3), chains_per_s=round(batch / total * 1e3, 1), level_ops_per_s=round(batch * levels / total * 1e3, 1),
                per_level_ms=[round(v, 3) for v in per_level], per_level_ops_per_s=[round(batch / v * 1e3, 1) for v in per_level],
                stage_model_bytes_per_chain=int(sum(rows) * 8 * n), stage_model_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4),
                levels_shape=shapes,
                note="two distinct operand batches; per-level time includes the second operand's own modulus switch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0,
                    help="ciphertext pairs per GPU per step (default: 1024 = C2 with one GPU, 8192 = C4's shard with more)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip event_free / default_mode / ntt / other_configs")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="streams of the TIMED region (1: exact per-kernel durations; 2: the handle's default mode)")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds of a sustained event-free leg after the timed region (0: none)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="rendezvous only: start the ranks, all-reduce their ranks, print a JSON line (no GPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    nranks = max(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
    if not args.batch:
        args.batch = BATCH_PER_GPU if nranks == 1 else BATCH_PER_GPU_SHARDED

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL ("nccl") with one device per rank; ranks that share a device (a 1-GPU box, the CPU spawn check)
        # rendezvous over gloo -- only the barrier and the MAX-of-times reduction go through it either way
        backend = os.environ.get("BENCH_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % ndev)
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank % ndev}"))
        else:
            dist.init_process_group(backend=backend)
        world, rank = dist.get_world_size(), dist.get_rank()
    if args.spawn_check:
        total = rank
        if dist is not None:
            tsum = torch.tensor([rank], dtype=torch.int64)
            if backend == "nccl":
                tsum = tsum.cuda()
            dist.all_reduce(tsum)
            total = int(tsum.item())
            dist.barrier()
            dist.destroy_process_group()
        assert total == world * (world - 1) // 2
        if rank == 0:
            print(json.dumps(dict(spawn_check="ok", n_gpus=world, dist_backend=backend, devices_visible=ndev,
                                  workload=workload_name(world, args.batch), batch_per_gpu=args.batch)))
        return

    import fhe_rs_amd as fhe
    from fhe_rs_amd import _lib
    assert _lib.lib() is not None and _lib.loaded_path().endswith("libfhe_hip.so")
    assert ndev > 0, "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    if world > 1:
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus)
        if ndev >= world:
            assert dist.get_backend() == "nccl", "one device per rank: the ranks must rendezvous over RCCL"
    all_cpus = os.sched_getaffinity(0)
    pin = pin_to_gpu_numa_node(torch, dev)

    # ---- setup (untimed): parameters, device tables, synthetic key + inputs in HBM ----------
    n, batch = N_DEGREE, args.batch
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=MODULI_SIZES, device=dev)
    ctx = par.context_at_level(0)
    L, K = ctx.nmoduli, par.mul_context_at_level(0).nmoduli
    rk = fhe.RelinearizationKey(key_for(fhe, ctx, SEED))
    mul = fhe.Multiplicator.default(par, rk, 0)
    default_opts = mul.options()                      # the product default (two streams)
    mul.set_chunk(args.chunk).set_streams(args.streams)
    from fhe_rs_amd.shard import shard_bounds, timed_steps
    ct0, ct1 = shard_bounds(world * batch, rank, world)  # this rank's block of independent ciphertexts
    assert ct1 - ct0 == batch
    lhs = ctx.synth_uniform(SEED, ct0, 0, 2, batch)
    rhs = ctx.synth_uniform(SEED, ct0, 2, 2, batch)
    out = torch.empty((batch, 2, L, n), dtype=torch.int64, device=f"cuda:{dev}")
    stream = torch.cuda.current_stream().cuda_stream
    import ctypes as C

    def step():
        _lib.check(_lib.lib().fhe_bfv_mul_dev(mul._h, C.c_void_p(lhs.data_ptr()), C.c_void_p(rhs.data_ptr()),
                                              C.c_void_p(out.data_ptr()), batch, C.c_void_p(stream)))

    def timed(fn, steps):
        return timed_steps(fn, steps, torch.cuda.synchronize, dist, f"cuda:{dev}")

    # setup, not a step: the first call on a stream allocates that stream's workspace (~3 GiB from the engine's pool)
    # and loads the kernels' code objects -- one-time state, like the tables and the key above
    step()
    torch.cuda.synchronize()
    # The CPU-baseline leg runs FIRST (it used to be 9 of the run's 14.6 s at the END, so a sampler that
    # looks at the GPU every few seconds saw an idle device); its parity spot check happens after the timed region.
    cpu_leg, cpu_default128 = None, None
    if world == 1 and not args.no_cpu:
        os.sched_setaffinity(0, all_cpus)   # the CPU baseline gets every host core, not just the GPU's NUMA node
        cpu_leg = cpu_baseline(n, MODULI_SIZES, t, SEED, args.cpu_seconds)
        if not args.no_extras:
            # the reference's own Criterion IDs on its stock parameter sets, one host thread (part of the CPU leg: the
            # only other place this file touches oracle/)
            import ref_params
            cpu_default128 = {nn: ref_params.cpu_port_ms(nn, 0.1) for nn in (4096, 8192, 16384)}
        if pin.get("pinned"):
            pin_to_gpu_numa_node(torch, dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # N > 1: rank 0's rate on the SAME per-GPU batch while every other rank idles at a barrier -- the one-GPU
    # reference the scaling efficiency of this very run is computed against, and who is who
    solo_rate, identities, n1_reference = None, None, None
    if dist is not None:
        dist.barrier()
        if rank == 0:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            solo_rate = batch * args.steps / dt
            # what ONE GPU does alone, at this run's per-GPU batch and at C2's 1,024 (the N = 1 line of BENCH / SCALE is
            # measured at 1,024: this is the number to cross-check it against from inside an N > 1 run)
            n1_reference = {f"batch_{batch}": dict(value=round(solo_rate, 1), ms_per_step=round(dt / args.steps * 1e3, 3))}
            if batch > BATCH_PER_GPU:
                b1 = BATCH_PER_GPU

                def step_1024():
                    _lib.check(_lib.lib().fhe_bfv_mul_dev(mul._h, C.c_void_p(lhs.data_ptr()), C.c_void_p(rhs.data_ptr()),
                                                          C.c_void_p(out.data_ptr()), b1, C.c_void_p(stream)))
                for _ in range(2):
                    step_1024()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step_1024()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                n1_reference[f"batch_{b1}"] = dict(value=round(b1 * args.steps / dt, 1), ms_per_step=round(dt / args.steps * 1e3, 3))
            n1_reference["note"] = "rank 0 alone (every other rank idle at a barrier), wall clock around K steps + synchronize"
        dist.barrier()
        props = torch.cuda.get_device_properties(dev)
        me = dict(rank=rank, local_rank=local_rank, device=dev, name=props.name,
                  pci="%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id),
                  uuid=str(getattr(props, "uuid", "")), numa_node=pin.get("numa_node"), pid=os.getpid())
        identities = [None] * world
        dist.all_gather_object(identities, me)
    fhe.prof_reset()
    fhe.prof_enable(True)   # HIP events carried by every kernel launch, on the launching stream
    # the timed region: K steps between barrier + synchronize, REPEATS times back to back; `value` is the median
    # repeat, the spread is in the line (value_min / value_max)
    REPEATS = 3
    elapsed_all = [timed(step, args.steps) for _ in range(REPEATS)]
    fhe.prof_enable(False)
    prof_entries = fhe.prof_entries()                 # (label, kernel symbol, launches, ms) per instantiation
    # the library labels the two tensor_intt instances separately (rows below 2^60: narrow passes); the roofline's
    # families fold them, `dominant_by_symbol` below keeps every instantiation apart
    prof = {}
    for k_, (n_, ms_) in fhe.prof_report().items():
        fam = k_[:-len("_narrow")] if k_.endswith("_narrow") else k_
        a_ = prof.get(fam, (0, 0.0))
        prof[fam] = (a_[0] + n_, a_[1] + ms_)
    elapsed = sorted(elapsed_all)[REPEATS // 2]
    prof_steps = args.steps * REPEATS    # the per-kernel event sums cover every repeat
    # every rank's own elapsed time for the same K steps (its barrier-to-barrier time is the slowest rank's)
    per_rank = None
    if dist is not None:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        on_gpu = dist.get_backend() == "nccl"
        tt = torch.tensor([mine], dtype=torch.float64, device=f"cuda:{dev}" if on_gpu else "cpu")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rates = [batch * args.steps / float(x.item()) for x in allt]
        per_rank = dict(ops_per_s=[round(r, 1) for r in rates], max_over_min=round(max(rates) / min(rates), 4),
                        note="each rank's own un-barriered K steps right after the timed region")

    # A sustained, event-free leg (every rank, no barrier): the timed region above is a fraction of a second, too short
    # for a utilisation sampler that looks every few seconds -- this keeps the device busy with the same step for
    # `--sustain` seconds and reports the rate it held (`gpu_busy` saw 0 of 3 samples).
    sustained = None
    if args.sustain > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_sus = 0
        while True:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            n_sus += 20
            if time.perf_counter() - t0 >= args.sustain:
                break
        dt = time.perf_counter() - t0
        sustained = dict(seconds=round(dt, 2), steps=n_sus, ops_per_s_this_rank=round(batch * n_sus / dt, 1),
                         note="same step, library profiler off, back to back; this rank only")

    extras = {}
    if not args.no_extras:
        # (1) the same K steps without the per-launch events
        e1 = timed(step, args.steps)
        extras["event_free"] = dict(value=round(world * batch * args.steps / e1, 1), unit="ops/s",
                                    ms_per_step=round(e1 / args.steps * 1e3, 3), streams=args.streams,
                                    note="same K steps, library profiler off: the cost of the per-launch HIP events "
                                         "inside the timed region is the difference to `value`")
        # (2) the handle's default mode (two streams); first calls allocate the second stream's workspace
        if args.streams == 1 and default_opts["streams"] == 2:
            mul.set_streams(2)
            for _ in range(max(args.warmup, 1) + 1):
                step()
            e2 = timed(step, args.steps)
            mul.set_streams(1)
            extras["default_mode"] = dict(
                streams=2, value=round(world * batch * args.steps / e2, 1), unit="ops/s",
                ms_per_step=round(e2 / args.steps * 1e3, 3),
                note="fhe_mul's default: chunks of a batch alternate between the caller's stream and an internal "
                     "one; kernels of the two streams overlap, so their HIP-event / rocprofv3 durations are not "
                     "attributable per kernel -- `value` and `roofline` therefore come from the single-stream region")
        # (3) NTT/s (BASELINE.json metric "(and NTT/s)"): forward NTT of this rank's [batch*2][L][N] polynomials,
        # in place, K launches; one NTT = forward transform of one Poly = L row-NTTs
        polys = lhs.view(batch * 2, L, n)

        def ntt_step():
            _lib.check(_lib.lib().fhe_ntt_forward_dev(ctx._h, C.c_void_p(polys.data_ptr()), batch * 2, C.c_void_p(stream)))
        ntt_step()
        e3 = timed(ntt_step, args.steps)
        poly_rate = world * batch * 2 * args.steps / e3
        extras["ntt"] = dict(poly_ntt_per_s=round(poly_rate, 1), row_ntt_per_s=round(poly_rate * L, 1),
                             algorithmic_bytes_per_poly=2 * L * 8 * n,
                             achieved_GBps=round(poly_rate / world * 2 * L * 8 * n / 1e9, 1),
                             frac=round(poly_rate / world * 2 * L * 8 * n / 1e9 / HBM_PEAK_GBS, 4),
                             workload=f"forward NTT of [{batch * 2}][{L}][{n}] per GPU, in place, {args.steps} launches",
                             ms_per_launch=round(e3 / args.steps * 1e3, 4))
        # (the transform is a bijection on canonical residues: restore the inputs for the parity spot check)
        lhs = ctx.synth_uniform(SEED, ct0, 0, 2, batch)

        # (4) the drop-in host-pointer entry point and the `_dev` path on ABI-owned buffers (rank 0, one GPU)
        if world == 1:
            extras["host_api"] = host_api_numbers(fhe, _lib, torch, mul, ctx, par, rk, batch, n, L, value_hint=None)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ops = world * batch * args.steps
    value = ops / elapsed
    R = 8 * n
    # ---- roofline of the dominant kernel (by total HIP-event time in the timed region) ----
    # algorithmic bytes per ct x ct + relin handled by each kernel family (stage model):
    # each row-NTT reads and writes its row once (2R); the fused key switch reads L rows, writes
    # 2*Lk rows and reads 2 addend rows per key modulus (the key itself is cache resident).
    alg_rows = {
        "ntt_inv": 2 * (4 * L),                       # extend: inverse NTT of the 4 input polynomials
        "tensor_intt": 4 * K + 3 * K,                 # fused tensor + inverse NTT (SURVEY §8d "7K"): the 4 operand rows
                                                      # once (the slots' re-reads are L2 hits, see the kernel), 3 rows out
        "ntt_fwd": 2 * (4 * (K - L) + 2 * L),         # new rows of the 4 extended polys + (c0, c1)
        "key_switch_fused": L * L + 4 * L,            # L digit rows per key modulus, 2 addend rows + 2 output rows
        "scale_extend": 4 * (L + (K - L)),            # extend 4 polys: L rows in, K-L new rows out
        "scale_down": 3 * (K + L),                    # down-scale 3 polys: K rows in, L rows out
        "tensor": 7 * K,
        "copy_rows": 2 * 4 * L,
    }
    dominant = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, 1e-9))
    dname, (dlaunches, dms) = dominant
    dbytes_total = alg_rows.get(dname, 0) * R * batch * prof_steps
    achieved = dbytes_total / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(dname)
            traffic_source = ("profiles/roofline_traffic.json (builder's rocprofv3 --pmc passes, "
                              + str(tj.get("_source", "see profiles/")) + "): not observed in this run")
        except Exception:
            traffic = None
    per_kernel = {}
    for k, v in sorted(prof.items()):
        kb = alg_rows.get(k, 0) * R * batch * prof_steps
        per_kernel[k] = dict(launches=v[0], ms=round(v[1], 3),
                             frac=round(kb / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[1] > 0 and kb else None)
    kernel_sum_ms = sum(v[1] for v in prof.values())
    # the dominant kernel by SYMBOL (the name rocprofv3 lists it under): one entry per instantiation, so the wide and the
    # narrow ntt_kernel<false, 13, ...> behind the label "ntt_fwd" are two candidates, not one (VERDICT r05 weak #5 ii)
    def symbol_rows(label, sym):
        args = [a.strip() for a in sym[sym.find("<") + 1:sym.find(">")].split(",")] if "<" in sym else []
        if label == "ntt_fwd" and len(args) >= 3:      # ntt_kernel<INV, LOGM, NARROW, ...>: extension rows wide, (c0, c1) narrow
            return 2 * (2 * L) if args[2] == "true" else 2 * (4 * (K - L))
        return dict(alg_rows, tensor_intt=7 * (K - L), tensor_intt_narrow=7 * L).get(label, 0)
    by_symbol = [dict(kernel=short_symbol(sym), label=label, launches=cnt, ms=round(ms, 3), avg_launch_ms=round(ms / max(cnt, 1), 4),
                      achieved=round(symbol_rows(label, sym) * R * batch * prof_steps / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0,
                      frac=round(symbol_rows(label, sym) * R * batch * prof_steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else 0.0,
                      share_of_kernel_time=round(ms / kernel_sum_ms, 4) if kernel_sum_ms else None)
                 for label, sym, cnt, ms in prof_entries]
    by_symbol.sort(key=lambda d: -d["ms"])
    dominant_by_symbol = by_symbol[0] if by_symbol else None
    roofline = dict(bound="hbm", kernel=dname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                    traffic_observed_this_run=False,
                    # the per-kernel event sums cover EVERY repeat, so they are compared with the mean step time over every
                    # repeat (the line's ms_per_step is the MEDIAN repeat's: a slow first repeat must not fail this check)
                    kernel_sum_ms_per_step=round(kernel_sum_ms / prof_steps, 4),
                    mean_ms_per_step_all_repeats=round(sum(elapsed_all) / prof_steps * 1e3, 4),
                    kernel_sum_le_step=bool(kernel_sum_ms / prof_steps <= sum(elapsed_all) / prof_steps * 1e3 * 1.001),
                    timed_steps_behind_kernel_times=prof_steps,
                    launches=dlaunches, avg_launch_ms=round(dms / max(dlaunches, 1), 4), streams=args.streams,
                    algorithmic_bytes_per_launch=int(dbytes_total / max(dlaunches, 1)),
                    whole_op=dict(stage_model_bytes_per_op=stage_model_rows(L, K, L) * R,
                                  achieved=round(stage_model_rows(L, K, L) * R * value / world / 1e9, 1),
                                  frac=round(stage_model_rows(L, K, L) * R * value / world / 1e9 / HBM_PEAK_GBS, 4)),
                    kernels=per_kernel, kernel_is="dominant FAMILY (instances of one kernel template summed)",
                    dominant_by_symbol=dominant_by_symbol, by_symbol=by_symbol)
    if not args.no_extras:
        roofline["int_issue"] = int_issue_roofline(fhe, dev, prof, n, L, K, batch, prof_steps, pipeline_step=step,
                                                   sync=torch.cuda.synchronize, par=par, value_per_gpu=value / world,
                                                   stage_bytes=stage_model_rows(L, K, L) * R)
        if "binding" in roofline["int_issue"]:
            roofline["binding"] = roofline["int_issue"]["binding"]
            roofline["frac_of_binding_ceiling"] = roofline["int_issue"]["value_over_min_ceiling"]
            roofline["frac_hbm_whole_op"] = roofline["int_issue"]["hbm_whole_op"]["frac"]
            roofline["frac_int_issue_whole_op"] = roofline["int_issue"]["whole_op"]["frac"]

    result = {
        "metric": "BFV ct x ct + relinearize ops/s (n=8192, 4x60-bit moduli)",
        "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "repeats": REPEATS, "value_min": round(ops / max(elapsed_all), 1), "value_max": round(ops / min(elapsed_all), 1),
        "value_all": [round(ops / e, 1) for e in elapsed_all], "value_is": "median of the repeats",
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload_name(world, batch),
                   "batch_per_gpu": batch, "global_batch": batch * world, "parallelism": f"batch-sharded x{world}",
                   "streams_in_timed_region": args.streams, "dist_backend": backend,
                   "dist_world_size": world if dist is not None else 1, "devices_visible": ndev, "host_pinning": pin},
        "roofline": roofline,
    }
    result.update(extras)
    if sustained is not None:
        result["sustained"] = sustained
    if per_rank is not None:
        result["per_rank"] = per_rank
    if dist is not None:
        distinct = len({(i["pci"], i["uuid"]) for i in identities})
        result["multi_gpu"] = dict(
            rccl_world_size=world if backend == "nccl" else None, dist_backend=backend, ranks=identities,
            distinct_devices=distinct, one_device_per_rank=bool(distinct == world),
            solo_rank0_ops_per_s=round(solo_rate, 1), solo_note="rank 0, same per-GPU batch, all other ranks idle at a barrier",
            efficiency_vs_1gpu_same_batch=round(value / (world * solo_rate), 4),
            n1_reference=n1_reference, data_path_collectives=0)

    if cpu_leg is not None:
        cb, cm, (clhs, crhs, last, count, npairs) = cpu_leg
        cb["when"] = "before the GPU legs of this run"
        result["cpu_baseline"] = cb
        result["speedup_vs_cpu_all_cores"] = round(value / cb["value"], 1)
        result["speedup_vs_cpu_single_thread"] = round(value / cb["single_thread_ops_per_s"], 1)
        # parity spot check of the GPU output (last pass over the same inputs) against the same oracle
        import numpy as np
        step()
        torch.cuda.synchronize()
        i = (count - 1) % npairs
        assert np.array_equal(out[i].cpu().numpy().view(np.uint64), last), "GPU result differs from the oracle"
        result["parity_spot_check"] = f"ciphertext {i} bit-identical to the oracle"

    if world == 1 and not args.no_extras:
        del lhs, rhs, out, mul
        other = result["other_configs"] = {}

        def leg(name, fn):
            """One informational leg: its failure is recorded in the detail file and never costs the record."""
            fhe.workspace_trim()
            torch.cuda.empty_cache()
            try:
                return fn()
            except Exception as e:     # noqa: BLE001
                result.setdefault("errors", []).append(f"{name}: {type(e).__name__}: {str(e)[:200]}")
                return None
        other.update(leg("reference_bench_ids", lambda: reference_bench_ids(fhe, torch, par, rk, batch, make_timeit(torch))) or {})
        del rk
        other.update(leg("other_configs", lambda: other_configs(fhe, torch)) or {})
        other.update(leg("next_rows", lambda: next_rows(fhe, torch, par, make_timeit(torch))) or {})
        other["C5_chain_15_levels"] = leg("c5_chain", lambda: c5_chain(fhe, torch))
        other["reference_default_128"] = leg("reference_default_128", lambda: reference_default_128(fhe, torch, cpu_default128))
        rates = (roofline.get("int_issue") or {}).get("rates")
        if rates:
            result["binding_ceilings"] = leg("binding_ceilings", lambda: binding_ceilings(other, rates, value / world, n, L, K))

    if dist is not None:
        dist.destroy_process_group()
    emit(result)


if __name__ == "__main__":
    main()
