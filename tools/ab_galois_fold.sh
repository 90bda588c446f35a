cp fhe.rs_amd/libfhe_hip.so /tmp/lib_new.so
for round in 1 2; do
  for v in before new; do
    if [ $v = before ]; then cp tools/_variants/libfhe_hip_before_fold.so fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so; fi
    echo "{\"build\": \"$v\", \"round\": $round, \"t\": $(python tools/galois_ab.py 2>/dev/null)}"
  done
done
cp /tmp/lib_new.so fhe.rs_amd/libfhe_hip.so
