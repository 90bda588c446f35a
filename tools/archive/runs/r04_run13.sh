#!/bin/bash
# Round 4, GPU call 14: key switch at N = 32768 on 16384-point half rows (ks_fused_kernel<14, ., ., ., ., G0 = 1>: one folded
# stage) against ks_fused_split_kernel (8192-point quarter rows, two folded stages): parity of each variant on the C5 /
# N = 32768 cases (the variant copied over the in-tree library for the test run), then the alternating A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04m
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_release.so
for v in half15_1 half15_2; do
  cp tools/_variants/libfhe_hip_$v.so fhe.rs_amd/libfhe_hip.so
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c5 or 32768 or larger_than_lds or large_rows" > gpurun_out/r04m/pytest_$v.log 2>&1
  echo $v; tail -2 gpurun_out/r04m/pytest_$v.log
done
cp /tmp/lib_release.so fhe.rs_amd/libfhe_hip.so
timeout 900 python tools/ab_mul.py 3 > gpurun_out/r04m/ab.jsonl 2> gpurun_out/r04m/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r04m/ab.jsonl"):
    d = json.loads(l)
    if "error" in d:
        print(d["build"], d["error"][-300:]); continue
    print(f'{d["build"]:28s} r{d["round"]} c5 b16 {d["c5_b16_ms"]}')
PY
python - <<'PY'
# relinearise at C5 (key switch alone) per build
import subprocess, sys, json, glob, os
code = r"""
import sys
sys.path.insert(0, '.')
import torch, fhe_rs_amd as fhe
from fhe_rs_amd import _lib
if sys.argv[1] != 'default': _lib._load_for_tests(sys.argv[1])
n, L = 32768, 16
ctx = fhe.Context(fhe.generate_moduli([60] * L, n), n)
kk = ctx.synth_uniform(5, 0, 8, 2 * L, 1)[0].reshape(L, 2, L, n)
rk = fhe.RelinearizationKey(fhe.KeySwitchingKey(ctx, ctx, kk[:, 0].contiguous(), kk[:, 1].contiguous()))
out = {}
for batch in (16, 64):
    ct3 = ctx.synth_uniform(5, 0, 0, 3, batch)
    for _ in range(3): rk.relinearizes(ct3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): rk.relinearizes(ct3)
    e1.record(); torch.cuda.synchronize()
    out[batch] = round(e0.elapsed_time(e1) / 5, 4)
print(out)
"""
libs = ["default"] + sorted(glob.glob("tools/_variants/*.so"))
for rnd in range(3):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
        print("relin C5", os.path.basename(lib), rnd, r.stdout.strip()[-80:], r.stderr.strip()[-200:] if r.returncode else "")
PY
