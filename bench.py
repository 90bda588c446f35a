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

Prints ONE JSON line on rank 0:
  value / ms_per_step   the K timed steps, single-stream mode with the library's per-launch HIP events on
                        (per-kernel durations are exact there; `roofline` comes from this region)
  event_free            the same K steps without the events (their overhead, stated rather than assumed)
  default_mode          the same K steps in the handle's default two-stream mode (kernels of the two streams
                        overlap, so per-kernel durations are not attributable there: informational)
  ntt                   forward NTT of [batch*2][4][8192] (fhe-math/benches/ntt.rs:12-38): Poly-NTT/s, row-NTT/s
  other_configs         C3 relinearise / rotations (batch 512) and C5 level-0 multiply+relin+mod-switch (batch 16)
  roofline              dominant kernel, HIP-event timed on the launching stream
  cpu_baseline          (N = 1) the plain-C port of the reference algorithm on this box's host cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_DEGREE = 8192
MODULI_SIZES = [60, 60, 60, 60]
BATCH_PER_GPU = 1024
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


def key_for(fhe, ctx, seed):
    L = ctx.nmoduli
    kk = ctx.synth_uniform(seed, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, ctx.degree)
    return fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous())


def other_configs(fhe, torch, reps=3):
    """Informational (never `value`): the other single-GPU configs of BASELINE.json on this box, same process.
    C3 (fhe/benches/bfv.rs:167-194): N=16384, 8x60-bit, relinearise 3->2 and the two rotations, batch 512.
    C5 (bfv.rs:247-255 shape at the top of a deep chain): N=32768, 16x60-bit, multiply + relinearise +
    modulus switch at level 0, batch 16.  Stage-model bytes per op: SURVEY.md §8(d)."""
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
    del ct3, ct2, rk, gk3, gkr, ksk, ctx
    n, L, batch = 32768, 16, 16
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
    a, b = ctx.synth_uniform(0xF4E50005, 0, 0, 2, batch), ctx.synth_uniform(0xF4E50005, 0, 2, 2, batch)
    ms = timeit(lambda: mul.multiply(a, b))
    rows = 22 * K + 7 * L + L * L + 4 * L + 12 * L - 6
    gbs = batch * rows * 8 * n / ms / 1e6
    out["C5_level0_mul_relin_modswitch"] = dict(
        workload=f"n=32768, 16x60-bit (K={K}), batch {batch}", ops_per_s=round(batch / ms * 1e3, 1), ms=round(ms, 3),
        stage_model_bytes_per_op=rows * 8 * n, stage_model_GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="ciphertext pairs per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip event_free / default_mode / ntt / other_configs")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="streams of the TIMED region (1: exact per-kernel durations; 2: the handle's default mode)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="rendezvous only: start the ranks, all-reduce their ranks, print a JSON line (no GPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

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
            print(json.dumps(dict(spawn_check="ok", n_gpus=world, dist_backend=backend, devices_visible=ndev)))
        return

    import fhe_rs_amd as fhe
    from fhe_rs_amd import _lib
    assert _lib.lib() is not None and _lib.loaded_path().endswith("libfhe_hip.so")
    assert ndev > 0, "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = local_rank % ndev
    torch.cuda.set_device(dev)

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

    # setup, not a step: the first call on a stream allocates that stream's workspace (hipMalloc of ~3 GiB)
    # and loads the kernels' code objects -- one-time state, like the tables and the key above
    step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fhe.prof_reset()
    fhe.prof_enable(True)   # HIP events around every kernel launch, on the launching stream
    elapsed = timed(step, args.steps)
    fhe.prof_enable(False)
    prof = fhe.prof_report()

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
    dbytes_total = alg_rows.get(dname, 0) * R * batch * args.steps
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
        kb = alg_rows.get(k, 0) * R * batch * args.steps
        per_kernel[k] = dict(launches=v[0], ms=round(v[1], 3),
                             frac=round(kb / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[1] > 0 and kb else None)
    roofline = dict(bound="hbm", kernel=dname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                    launches=dlaunches, avg_launch_ms=round(dms / max(dlaunches, 1), 4), streams=args.streams,
                    algorithmic_bytes_per_launch=int(dbytes_total / max(dlaunches, 1)),
                    whole_op=dict(stage_model_bytes_per_op=stage_model_rows(L, K, L) * R,
                                  achieved=round(stage_model_rows(L, K, L) * R * value / world / 1e9, 1),
                                  frac=round(stage_model_rows(L, K, L) * R * value / world / 1e9 / HBM_PEAK_GBS, 4)),
                    kernels=per_kernel)

    result = {
        "metric": "BFV ct x ct + relinearize ops/s (n=8192, 4x60-bit moduli)",
        "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C2: BFV n=8192, 4x60-bit RNS moduli (K=9), batch={batch} ct x ct + relinearize per GPU",
                   "batch_per_gpu": batch, "global_batch": batch * world, "parallelism": f"batch-sharded x{world}",
                   "streams_in_timed_region": args.streams, "dist_backend": backend, "devices_visible": ndev},
        "roofline": roofline,
    }
    result.update(extras)

    if world == 1 and not args.no_cpu:
        cb, cm, (clhs, crhs, last, count, npairs) = cpu_baseline(n, MODULI_SIZES, t, SEED, args.cpu_seconds)
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
        del lhs, rhs, out, mul, rk
        fhe.workspace_trim()
        torch.cuda.empty_cache()
        result["other_configs"] = other_configs(fhe, torch)

    if dist is not None:
        dist.destroy_process_group()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
