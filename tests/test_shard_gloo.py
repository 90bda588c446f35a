"""N > 1 path on CPU: two `gloo` ranks each run the engine (host emulation of the kernel
sources) on their shard of a global batch of independent ciphertext pairs; the gathered
result equals the oracle's for the whole batch.  No collective on the data path -- only the
barrier / MAX timing reduction used by bench.py and a result gather for the check."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT, build_emu

WORKER = r'''
import os, sys, pickle
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import random
import numpy as np
import torch.distributed as dist
from helpers import load_engine, ksk_arrays, ct_arr
from fhe_oracle import bfv as obfv
fhe = load_engine("emu")
from fhe_rs_amd.shard import shard_bounds, timed_steps
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, total = 16, 5
rng = random.Random(99)                      # every rank derives the same global workload
opar = obfv.BfvParameters.default_arc(3, n)
sk = obfv.SecretKey.random(opar, rng)
ork = obfv.RelinearizationKey(sk, rng)
A = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng) for _ in range(total)]
B = [sk.encrypt([rng.randrange(opar.plaintext) for _ in range(n)], rng) for _ in range(total)]
par = fhe.BfvParameters(n, opar.plaintext, moduli=opar.moduli)
ctx = par.context_at_level(0)
c0, c0s, c1, c1s = ksk_arrays(ork.ksk)
m = fhe.Multiplicator.default(par, fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, c0, c1)), 0)
b, e = shard_bounds(total, rank, world)
lhs = np.stack([ct_arr(c) for c in A[b:e]]); rhs = np.stack([ct_arr(c) for c in B[b:e]])
res = {}
def step():
    res["out"] = m.multiply(lhs, rhs)
elapsed = timed_steps(step, 2, lambda: None, dist)
gathered = [None] * world
dist.all_gather_object(gathered, (b, e, res["out"]))
if rank == 0:
    om = obfv.Multiplicator.default(ork)
    covered = []
    for (b_, e_, out) in gathered:
        for i in range(b_, e_):
            assert np.array_equal(out[i - b_], ct_arr(om.multiply(A[i], B[i]))), i
            covered.append(i)
    assert covered == list(range(total)) and elapsed > 0
    print("SHARD_OK", world)
dist.destroy_process_group()
'''


def test_shard_bounds():
    sys.path.insert(0, ROOT)
    import fhe_rs_amd  # noqa: F401
    from fhe_rs_amd.shard import shard_bounds
    for total in (0, 1, 5, 8, 65536):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(65536, 3, 8) == (3 * 8192, 4 * 8192)


@pytest.mark.timeout(600)
def test_two_rank_sharded_multiply_gloo(tmp_path):
    build_emu()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SHARD_OK 2" in r.stdout
