"""Batch sharding across the GPUs of a node: every ct x ct + relinearise,
rotation or modulus switch touches only its own ciphertexts plus read-only tables/keys that
each rank replicates, so the global batch is split into contiguous blocks, one per rank
(one process per GPU), and NO collective runs on the data path.  torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for the barrier and
the MAX-over-ranks timing reduction, plus an optional result gather after timing."""
import time


def shard_bounds(total, rank, world):
    """[begin, end) of rank's contiguous block; the first total % world ranks get one more."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def timed_steps(step, steps, sync, dist=None, device=None):
    """Times exactly `steps` calls of `step()` bracketed by sync()+barrier on both sides and
    returns the MAX elapsed seconds over ranks."""
    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        on_gpu = device is not None and dist.get_backend() == "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
