#!/bin/bash
# Round 4, GPU call 12: chunk plan re-check with the merged extension (batch 1024 and 8192), SQ counters of the bench's
# kernels, C3 / C5 PMC passes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04k
timeout 600 python tools/chunk_sweep.py 1024 > gpurun_out/r04k/chunk_sweep_1024.jsonl 2> gpurun_out/r04k/chunk.err
timeout 600 python tools/chunk_sweep.py 8192 > gpurun_out/r04k/chunk_sweep_8192.jsonl 2>> gpurun_out/r04k/chunk.err
python - <<'PY'
import json
for f in ("1024", "8192"):
    rows = [json.loads(l) for l in open(f"gpurun_out/r04k/chunk_sweep_{f}.jsonl")]
    for s in (1, 2):
        best = {}
        for r in rows:
            if r["streams"] == s:
                best.setdefault(r["chunk"], []).append(r["ops_per_s"])
        print(f, "streams", s, {c: v for c, v in best.items()})
PY
bash tools/collect_sq.sh r04k/sq > gpurun_out/r04k/sq.log 2>&1
python tools/sq_derive.py gpurun_out/r04k/sq > gpurun_out/r04k/sq_derived.json 2>> gpurun_out/r04k/sq.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04k/sq_derived.json"))
for k in d["kernels"]:
    print(k["kernel"][:48], k["valu_insts_per_wave"], k["valu_pipe_floor"], k["wave_wait_inst_frac"], k["wave_wait_frac"], k["waves_resident_per_simd"])
PY
bash tools/collect_configs_pmc.sh r04k/cfg > gpurun_out/r04k/cfg.log 2>&1
tail -5 gpurun_out/r04k/cfg.log
