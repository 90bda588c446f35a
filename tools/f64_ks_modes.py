#!/usr/bin/env python3
"""Round 6: which key-switch strategy the F64 instances want.  relinearize on the reference's stock sets at batches 1 ...
1024, strategies AUTO / FUSED / UNFUSED, F64 kernels on and off, same process, alternating.  One JSON line per cell."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import fhe_rs_amd as fhe  # noqa: E402

SETS = {4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
        8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
        16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]}
timeit = bench.make_timeit(torch, 5)
# optional: FHE_MODES_GRID="8192:32,64,96;16384:16,32" restricts the sets and batches; FHE_MODES_F64_ONLY=1 skips the integer kernels
GRID = {int(a.split(":")[0]): [int(b) for b in a.split(":")[1].split(",")] for a in os.environ.get("FHE_MODES_GRID", "").split(";") if a}
F64_ONLY = os.environ.get("FHE_MODES_F64_ONLY", "0") == "1"
for n, q in SETS.items():
    if GRID and n not in GRID:
        continue
    ctx = fhe.Context(q, n)
    ksk = bench.key_for(fhe, ctx, 11)
    rk = fhe.RelinearizationKey(ksk)
    for batch in GRID.get(n, (1, 4, 16, 64, 256, 1024 if n <= 8192 else 512)):
        ct3 = ctx.synth_uniform(11, 0, 0, 3, batch)
        cell = dict(n=n, batch=batch)
        for f64 in ((True,) if F64_ONLY else (True, False)):
            fhe.set_f64(f64)
            for mode, name in ((0, "auto"), (1, "fused"), (2, "unfused")):
                ksk.set_mode(mode)
                ms = statistics.median(timeit(lambda: rk.relinearizes(ct3)) for _ in range(3))
                cell[("f64_" if f64 else "int_") + name + "_ms"] = round(ms, 4)
        fhe.set_f64(True)
        ksk.set_mode(0)
        cell["f64_best"] = min(("fused", "unfused"), key=lambda k: cell["f64_" + k + "_ms"])
        cell["f64_auto_over_best"] = round(cell["f64_auto_ms"] / min(cell["f64_fused_ms"], cell["f64_unfused_ms"]), 3)
        print(json.dumps(cell), flush=True)
        del ct3
    fhe.workspace_trim()
    torch.cuda.empty_cache()
