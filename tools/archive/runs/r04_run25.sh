#!/bin/bash
# Round 4, GPU call 28: rocprofv3 kernel stats of single-ciphertext relinearisation, fused forced vs FHE_KS_AUTO.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04z
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/tools/ks_single_ct_profile.py > $O/out.log 2> $O/err.log
tail -1 $O/out.log
find $O -name '*kernel_trace.csv' -size +8M -delete
grep -i "ks_\|ntt_kernel<true" $O/stats/*kernel_stats.csv | cut -c1-90 | head -20
python - <<PY
import csv, glob
for f in glob.glob("$O/stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void fhe::k::", "")
        if n.startswith("ks_") or "ntt_kernel<true" in n or "add" in n.lower():
            print(f'{n:50s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
PY
