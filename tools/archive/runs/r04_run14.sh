#!/bin/bash
# Round 4, GPU call 15: the build that ships the half-row key switch (FHE_KS_HALF15 = 2) -- GPU suite, smoke, the fused
# form against the unfused one at C5 again, its HBM counters, N = 65536 (two folded stages on 16384-point parts against
# three on 8192-point parts, lab variant half15_0), then the driver-shaped bench with its rocprofv3 passes and the C3 / C5
# configs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=$R/gpurun_out/r04n
mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python tools/ks_modes_ab.py c5 --rounds 3 > $O/ks_modes_ab_c5.jsonl 2> $O/ks_modes_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r04n/ks_modes_ab_c5.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k not in ("kernels",)})
PY
for rnd in 0 1; do
  for lib in default tools/_variants/libfhe_hip_half15_0.so; do
    timeout 200 python tools/ks_relin_time.py $lib 65536 4 8 32 >> $O/n65536_ab.jsonl 2>> $O/n65536_ab.err
    timeout 200 python tools/ks_relin_time.py $lib 65536 8 8 >> $O/n65536_ab.jsonl 2>> $O/n65536_ab.err
  done
done
cat $O/n65536_ab.jsonl
( cd /tmp && export TMPDIR=/tmp
RUN="python $R/tools/ks_pmc_c5.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_stats -o run -- $RUN > $O/ks_stats.out 2> $O/ks_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/ks_fetch -o run -- $RUN > $O/ks_fetch.out 2> $O/ks_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ks_write -o run -- $RUN > $O/ks_write.out 2> $O/ks_write.log )
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
fe, wr = pmc("ks_fetch", "FETCH_SIZE"), pmc("ks_write", "WRITE_SIZE")
dur = {}
for f in glob.glob(O + "/ks_stats/**/*kernel_stats.csv", recursive=True):
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
bash tools/collect_profiles.sh r04n/final > $O/collect.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04n/final/bench.json"))
print(d["value"], d["value_all"], d["ms_per_step"], d["default_mode"]["value"], d["roofline"]["frac"], d["roofline"]["kernel_sum_le_step"], d["parity_spot_check"])
oc = d["other_configs"]
print({k: v for k, v in oc.items() if k.startswith(("C3", "C5"))})
PY
bash tools/collect_configs_pmc.sh r04n/cfg > $O/cfg.log 2>&1
tail -3 $O/cfg.log
