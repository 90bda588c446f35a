import csv, glob, sys, collections
d=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        if 'fhe' not in r['Kernel_Name']: continue
        e=d[k][r['Counter_Name']]; e[0]+=1; e[1]+=float(r['Counter_Value'])
for k,v in d.items():
    print(k)
    for c,(n,s) in sorted(v.items()): print('   %-28s %14.0f (n=%d)'%(c, s/n, n))
