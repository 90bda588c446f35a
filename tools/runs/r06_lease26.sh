#!/bin/bash
# Round 6, lease 26: the multiply's internal stream at the highest priority (its own hardware-queue pool): the stock multiply late in a
# long process, and the fresh-state A/B against the previous release.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_x
mkdir -p $OUT
cd $ROOT
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for v in before new; do
  if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_auxprio.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
  python tools/stock_after_other_legs.py 2>/dev/null | sed "s/^{/{\"build\": \"$v\", /"
done | tee $OUT/stock_after_main_pieces_ab.jsonl | cut -c1-200
for round in 1 2; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_auxprio.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/scaler_nf_ab.py 2>/dev/null)}"
  done
done > $OUT/aux_priority_fresh_ab.jsonl
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r06_x/aux_priority_fresh_ab.jsonl")]
for k in rows[0]["t"]:
    if k.endswith("_ms") and not k.endswith("per_level_ms"):
        a = statistics.median(r["t"][k] for r in rows if r["build"] == "before"); b = statistics.median(r["t"][k] for r in rows if r["build"] == "new")
        print(k.ljust(24), round(a, 4), round(b, 4), "new/before %.3f" % (b / a))
print({k: len({r["t"][k] for r in rows}) for k in rows[0]["t"] if k.endswith("digest")})
PY
python bench.py > $OUT/bench_default.out 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['configs'])"
grep '^DETAIL ' $OUT/bench_default.out | sed 's/^DETAIL //' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default_mode', d.get('default_mode',{}).get('value'), 'event_free', d.get('event_free',{}).get('value'))"
