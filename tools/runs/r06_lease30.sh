#!/bin/bash
# Round 6, lease 30 (LAST build): the N > 1 launch paths on the one-GPU box -- bench.py's own spawn and the driver's torchrun line,
# two ranks sharing the device (gloo fallback; per-GPU batch 1024 so that both fit the clock) -- then more random shapes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_zz
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 1024 > $OUT/bench_gpus2_spawn.out 2> $OUT/bench_gpus2_spawn.err; tail -1 $OUT/bench_gpus2_spawn.out | cut -c1-1900
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 1024 > $OUT/bench_gpus2_torchrun.out 2> $OUT/bench_gpus2_torchrun.err; tail -1 $OUT/bench_gpus2_torchrun.out | cut -c1-1900
run() { python tests/random_sweep_gpu.py "$@" 2>/dev/null | tee -a $OUT/random_sweeps.jsonl | cut -c1-220; }
run ${SWEEP_S:-240} 40000000 41000000 0 -
run ${SWEEP_S:-240} 41000000 42000000 0 f64
run ${SWEEP_S:-240} 42000000 43000000 2 f64wide
run 120 43000000 44000000 1 big
