# same-box A/B of the whole bench line: tools/_variants/$1 against the in-tree build, alternating, two rounds
ALT=tools/_variants/$1
mkdir -p gpurun_out/r02e
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_main.so
for round in 1 2; do
for v in alt main; do
  if [ $v = alt ]; then cp $ALT fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so; fi
  python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['event_free']['value'], d['default_mode']['value'], {k:v['ops_per_s'] for k,v in d['other_configs'].items()})"
done
done 2>&1 | tee gpurun_out/r02e/ab_bench_$1.txt
cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so
