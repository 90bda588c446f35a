#!/bin/bash
# Round 4, GPU call 5: parity of the changed streaming kernels (word-granular wire format, 16-byte mul_plain, streaming
# loads, wide-w scaler), A/B of the next-row legs against the previous build, rocprofv3 kernel stats of the legs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wire or dot_product or scaler or switch_down or expand or decrypt or rgsw or device_buffers or poly_ops" > gpurun_out/r04d/pytest_subset.log 2>&1
tail -3 gpurun_out/r04d/pytest_subset.log
for round in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then L="--lib=$R/tools/_variants/libfhe_hip_prev.so"; else L=""; fi
    timeout 600 python tools/bench_next_rows.py --no-chain $L > gpurun_out/r04d/next_${v}_$round.json 2>> gpurun_out/r04d/next.err
  done
done
python - <<'PY'
import json
rows = {}
for v in ("prev", "new"):
    for r in (1, 2):
        d = json.load(open(f"gpurun_out/r04d/next_{v}_{r}.json"))
        for k, e in d.items():
            rows.setdefault(k, {}).setdefault(v, []).append((e["ms"], e.get("frac")))
for k, e in rows.items():
    print(f"{k:32s} prev {e['prev']}  new {e['new']}")
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04d/prof -o next -- python $R/tools/bench_next_rows.py > $R/gpurun_out/r04d/next_rows_profiled.json 2> $R/gpurun_out/r04d/prof.err
f=$(find $R/gpurun_out/r04d/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-150 | head -45
find $R/gpurun_out/r04d/prof -name '*kernel_trace.csv' -size +4M -delete
