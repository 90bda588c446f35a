#!/bin/bash
# Round 4, GPU call 23: soaks on the final build (small launches under FHE_KS_AUTO against the forced fused result; the
# batch-16 two-stream soak; C3 shape), the driver-shaped bench line (--steps 20 --warmup 5) and its rocprofv3 stats.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04v
mkdir -p $O
timeout 300 python tools/soak.py 20000 1 auto_small > $O/soak_auto_small.json 2> $O/soak.err; cat $O/soak_auto_small.json
timeout 300 python tools/soak.py 20000 2 small > $O/soak_small.json 2>> $O/soak.err; cat $O/soak_small.json
timeout 300 python tools/soak.py 3000 1 c3 > $O/soak_c3.json 2>> $O/soak.err; cat $O/soak_c3.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_steps20.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04v/bench_steps20.json"))
print(d["value"], d["value_all"], d["steps"], d["ms_per_step"], d["event_free"]["value"] if "event_free" in d else None, d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"])
print({k: (v["ms"], v["frac"]) for k, v in d["roofline"]["kernels"].items()})
oc = d["other_configs"]
print({k: (v.get("ops_per_s"), v.get("frac"), v.get("total_ms")) for k, v in oc.items() if k.startswith(("C3", "C5"))})
PY
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $R/$O/stats_bench.json 2> $R/$O/stats.log
find $R/$O -name '*kernel_trace.csv' -size +8M -delete
head -12 $R/$O/stats/*kernel_stats.csv | cut -c1-160
