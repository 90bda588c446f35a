#!/usr/bin/env python3
"""bench.py -- BFV ct x ct + relinearise throughput on MI355X (BASELINE.json's metric).

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
barrier + synchronize on both sides of the timed region, MAX over ranks.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the
launching stream) and, at N=1, `cpu_baseline` (the plain-C port of the reference algorithm
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="ciphertext pairs per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-two-stream", action="store_true", help="skip the informational two-stream measurement")
    ap.add_argument("--streams", type=int, default=1,
                    help="2: chunks alternate between two streams (fhe_set_streams); per-kernel durations then overlap")
    args = ap.parse_args()

    import torch
    import fhe_rs_amd as fhe
    from fhe_rs_amd import _lib
    assert _lib.lib() is not None and _lib.loaded_path().endswith("libfhe_hip.so")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL ("nccl") on the GPU box; BENCH_DIST_BACKEND=gloo lets several ranks share one GPU in tests
        dist.init_process_group(backend=os.environ.get("BENCH_DIST_BACKEND", "nccl"))
        world = dist.get_world_size()
        rank = dist.get_rank()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)

    # ---- setup (untimed): parameters, device tables, synthetic key + inputs in HBM ----------
    n, batch = N_DEGREE, args.batch
    t = fhe.generate_prime(20, 2 * n, 1 << 20)
    par = fhe.BfvParameters(n, t, moduli_sizes=MODULI_SIZES, device=dev)
    ctx = par.context_at_level(0)
    L, K = ctx.nmoduli, par.mul_context_at_level(0).nmoduli
    kk = ctx.synth_uniform(SEED, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
    rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
    mul = fhe.Multiplicator.default(par, rk, 0)
    if args.chunk:
        fhe.set_chunk(args.chunk)
    fhe.set_streams(args.streams)
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

    # setup, not a step: the first call on a stream allocates that stream's workspace (hipMalloc of ~3 GiB)
    # and loads the kernels' code objects -- one-time state, like the tables and the key above
    step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fhe.prof_reset()
    fhe.prof_enable(True)   # HIP events around every kernel launch, on the launching stream
    elapsed = timed_steps(step, args.steps, torch.cuda.synchronize, dist, f"cuda:{dev}")
    fhe.prof_enable(False)
    prof = fhe.prof_report()

    # Informational second measurement (never `value`): the same K steps in the library's two-stream mode
    # (fhe_set_streams(2)).  Kernels of the two streams overlap there, so per-kernel
    # durations stop being attributable -- which is why the line's `value` and `roofline` come from the
    # single-stream region above.
    throughput_mode = None
    if args.streams == 1 and not args.no_two_stream:
        fhe.set_streams(2)
        for _ in range(max(args.warmup, 1) + 1):   # first call allocates the second stream's workspace
            step()
        e2 = timed_steps(step, args.steps, torch.cuda.synchronize, dist, f"cuda:{dev}")
        fhe.set_streams(1)
        throughput_mode = dict(streams=2, value=round(world * batch * args.steps / e2, 1), unit="ops/s",
                               ms_per_step=round(e2 / args.steps * 1e3, 3))

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
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(dname)
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=dname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    launches=dlaunches, avg_launch_ms=round(dms / max(dlaunches, 1), 4), streams=args.streams,
                    algorithmic_bytes_per_launch=int(dbytes_total / max(dlaunches, 1)),
                    whole_op=dict(stage_model_bytes_per_op=stage_model_rows(L, K, L) * R,
                                  achieved=round(stage_model_rows(L, K, L) * R * value / world / 1e9, 1),
                                  frac=round(stage_model_rows(L, K, L) * R * value / world / 1e9 / HBM_PEAK_GBS, 4)),
                    kernels={k: dict(launches=v[0], ms=round(v[1], 3)) for k, v in sorted(prof.items())})

    result = {
        "metric": "BFV ct x ct + relinearize ops/s (n=8192, 4x60-bit moduli)",
        "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C2: BFV n=8192, 4x60-bit RNS moduli (K=9), batch={batch} ct x ct + relinearize per GPU",
                   "batch_per_gpu": batch, "global_batch": batch * world, "parallelism": f"batch-sharded x{world}"},
        "roofline": roofline,
    }
    if throughput_mode:
        result["two_stream_mode"] = throughput_mode

    if world == 1 and not args.no_cpu:
        cb, cm, (clhs, crhs, last, count, npairs) = cpu_baseline(n, MODULI_SIZES, t, SEED, args.cpu_seconds)
        result["cpu_baseline"] = cb
        result["speedup_vs_cpu_all_cores"] = round(value / cb["value"], 1)
        result["speedup_vs_cpu_single_thread"] = round(value / cb["single_thread_ops_per_s"], 1)
        # parity spot check of the timed GPU output against the same oracle
        import numpy as np
        i = (count - 1) % npairs
        assert np.array_equal(out[i].cpu().numpy().view(np.uint64), last), "GPU result differs from the oracle"
        result["parity_spot_check"] = f"ciphertext {i} bit-identical to the oracle"

    if dist is not None:
        dist.destroy_process_group()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
