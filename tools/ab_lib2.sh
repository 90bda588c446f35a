# A/B of two library builds, order ABBAABBA to cancel clock / thermal drift: tools/_variants/$1 (alt) against the
# in-tree build (main); prints per-kernel means.
ALT=tools/_variants/$1
mkdir -p gpurun_out/r02e
cp fhe.rs_amd/libfhe_hip.so /tmp/lib_main.so
python bench.py --no-cpu --no-extras --steps 10 > /dev/null 2>&1   # warm the box
for v in main alt alt main main alt alt main; do
  if [ $v = alt ]; then cp $ALT fhe.rs_amd/libfhe_hip.so; else cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so; fi
  python bench.py --no-cpu --no-extras --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], json.dumps({k:v['ms'] for k,v in d['roofline']['kernels'].items()}))"
done 2>&1 | tee gpurun_out/r02e/ab2_$1.txt
cp /tmp/lib_main.so fhe.rs_amd/libfhe_hip.so
python -c "
import json,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('gpurun_out/r02e/ab2_$1.txt'):
    p=l.split(' ',2)
    if p[0] not in ('main','alt'): continue
    acc[p[0]]['value'].append(float(p[1]))
    for k,v in json.loads(p[2]).items(): acc[p[0]][k].append(v)
for k in acc['main']:
    m=sum(acc['main'][k])/len(acc['main'][k]); a=sum(acc['alt'][k])/len(acc['alt'][k])
    print('%-18s main %.3f  alt %.3f  main/alt %.4f' % (k,m,a,m/a))
" | tee -a gpurun_out/r02e/ab2_$1.txt
