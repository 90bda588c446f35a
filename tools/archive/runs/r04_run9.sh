#!/bin/bash
# Round 4, GPU call 10: HBM traffic of the two key-switch strategies at C5 (separate PMC passes, as the guide prescribes).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i
mkdir -p $O
RUN="python $R/tools/ks_pmc_c5.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $RUN > $O/stats.out 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $RUN > $O/fetch.out 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $RUN > $O/write.out 2> $O/write.log
find $O -name '*kernel_trace.csv' -size +8M -delete
python - <<PY
import csv, glob, json, collections
O = "$O"
def pmc(d, name):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0].replace("void fhe::k::", "").replace("fhe::k::", "")
                tot[k][0] += 1; tot[k][1] += float(r["Counter_Value"])
    return tot
fe, wr = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
dur = {}
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0].replace("void fhe::k::", "").replace("fhe::k::", "")] = (int(r["Calls"]), float(r["AverageNs"]))
out = {}
for k in sorted(set(fe) | set(wr)):
    if not k.startswith("ks_"): continue
    out[k] = dict(launches=fe[k][0], fetch_bytes_per_launch_x2=round(fe[k][1] * 1024 * 2 / max(fe[k][0], 1)),
                  write_bytes_per_launch=round(wr[k][1] * 1024 / max(wr[k][0], 1)), avg_ns=dur.get(k, (0, 0))[1])
json.dump(out, open(O + "/ks_c5_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
