# Same-box alternating A/B of two builds of the library: tools/_variants/$1 ("before") against the in-tree build ("new"),
# running the python tool $2 (one JSON line per run).  usage: bash tools/ab_two_libs.sh libfhe_hip_before_x.so tools/foo.py
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/$1 fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python $2 2>/dev/null)}"
  done
done
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
