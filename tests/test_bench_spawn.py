"""bench.py's multi-rank launch paths, on CPU: `python bench.py --gpus 2` must start its own two ranks (it used to
assume an external torchrun and die in init_process_group), and the torchrun form the driver uses must keep
working.  `--spawn-check` stops after the rendezvous (init_process_group over gloo, an all-reduce of the ranks and
a barrier) -- the GPU work itself needs a GPU and is covered by tests/test_gpu_parity.py."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--spawn-check"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["spawn_check"] == "ok" and line["n_gpus"] == 2 and line["dist_backend"] == "gloo"
    # more than one GPU measures BASELINE configs[3] (C4): 8192 pairs per GPU unless --batch says otherwise
    assert line["batch_per_gpu"] == 8192 and line["workload"].startswith("C4:") and "8192 per GPU" in line["workload"]


def test_bench_single_gpu_is_c2_and_batch_overrides():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--spawn-check"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["batch_per_gpu"] == 1024 and line["workload"].startswith("C2:")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--batch", "1024", "--spawn-check"], capture_output=True,
                       text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["batch_per_gpu"] == 1024 and not line["workload"].startswith("C4:")


def test_bench_under_torchrun():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--spawn-check"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_bench_rank_failure_is_reported():
    """A rank that dies must fail the whole launch (non-zero exit), not hang the other rank."""
    env = _env()
    env["BENCH_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--spawn-check"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0


def test_bench_eight_ranks_rendezvous_under_torchrun():
    """The driver's SCALE launch at its largest world size, as far as a box without GPUs can walk it: eight ranks under
    `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`, gloo rendezvous, an all-reduce of the
    ranks, C4's workload name (8192 pairs per GPU, 65536 in all)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "8",
                        "--spawn-check"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 8 and line["batch_per_gpu"] == 8192 and "65536" in line["workload"]
