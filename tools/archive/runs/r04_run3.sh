#!/bin/bash
# Round 4, GPU call 4: the new driver-shaped bench line (next rows, C5 chain, repeats), then the same legs under
# rocprofv3 --kernel-trace --stats for profiles/r04_next_rows_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err
tail -4 gpurun_out/r04c/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c/bench.json"))
print(d["value"], d["value_all"], d["ms_per_step"], d["roofline"]["kernel_sum_ms_per_step"], d["roofline"]["kernel_sum_le_step"])
for k, v in d["other_configs"].items():
    print(k, {a: b for a, b in v.items() if a in ("ops_per_s", "polys_per_s", "ct_pt_mac_per_s", "galois_applications_per_s", "ms", "frac", "total_ms", "level_ops_per_s")})
print(d["other_configs"]["C5_chain_15_levels"]["per_level_ms"])
PY
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04c/prof -o next -- python $R/tools/bench_next_rows.py > $R/gpurun_out/r04c/next_rows_profiled.json 2> $R/gpurun_out/r04c/prof.err
find $R/gpurun_out/r04c/prof -name "*kernel_stats.csv" | head -2
f=$(find $R/gpurun_out/r04c/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-160 | head -40
