#!/usr/bin/env python3
"""Bank-conflict degree of every LDS access pattern of the radix passes and the chunk I/O (32-lane halves, 32 bank pairs
of 8 bytes) for padding functions i + a (i >> 3) + b (i >> 4) + c (i >> 5) + d (i >> 6); prints the best candidates."""
import itertools, sys
from collections import Counter
def degree(addrs):
    c = Counter(a % 32 for a in set(addrs)); return max(c.values())
def patterns(LOGM, T, plan_f, plan_i, CPT):
    pats=[]   # list of (name, list of half-wave index lists)
    s0=0
    for G in plan_f:
        lo_bits=LOGM-s0-G; R=1<<G; ngroups=1<<(LOGM-G); hw=[]
        for g0 in range(0,ngroups,T):
            for half in range(0,min(T,ngroups),32):
                for e in range(R):
                    idx=[]
                    for tid in range(half,half+32):
                        grp=g0+tid
                        if grp>=ngroups: continue
                        lo=grp&((1<<lo_bits)-1); base=((grp>>lo_bits)<<(LOGM-s0))+lo
                        idx.append(base+(e<<lo_bits))
                    hw.append(idx)
        pats.append(('f S0=%d G=%d'%(s0,G),hw)); s0+=G
    v0=0
    for G in plan_i:
        R=1<<G; ngroups=1<<(LOGM-G); hw=[]
        for g0 in range(0,ngroups,T):
            for half in range(0,min(T,ngroups),32):
                for e in range(R):
                    idx=[]
                    for tid in range(half,half+32):
                        grp=g0+tid
                        if grp>=ngroups: continue
                        lo=grp&((1<<v0)-1); base=((grp>>v0)<<(v0+G))+lo
                        idx.append(base+(e<<v0))
                    hw.append(idx)
        pats.append(('i V0=%d G=%d'%(v0,G),hw)); v0+=G
    hw=[]
    CH=(1<<LOGM)//(2*T)
    for c in range(CH):
        for half in range(0,T,32):
            for k in (0,1):
                hw.append([2*(c*T+t)+k for t in range(half,half+32)])
    pats.append(('chunk io',hw))
    return pats
P13=patterns(13,512,(4,3,3,3),(3,3,3,4),16)
PKS=patterns(13,1024,(2,2,3,3,3),(),8)
def evalpad(pad, pats):
    res=[]
    for name,hw in pats:
        # sample to keep it fast
        ds=[degree([pad(i) for i in idx]) for idx in hw[::7]]
        res.append((name,max(ds),sum(ds)/len(ds)))
    return res
cands={}
for a,b,c,d in itertools.product(range(0,4),range(0,4),range(0,4),range(0,4)):
    if a+b+c+d==0: continue
    pad=lambda i,a=a,b=b,c=c,d=d: i + a*(i>>3) + b*(i>>4) + c*(i>>5) + d*(i>>6)
    ov = a/8+b/16+c/32+d/64
    if ov>0.13: continue
    r=evalpad(pad,P13)+evalpad(pad,PKS)
    score=sum(x[2] for x in r)
    cands[(a,b,c,d)]=(score,ov,max(x[1] for x in r))
best=sorted(cands.items(), key=lambda kv: kv[1][0])[:12]
for k,v in best: print(k, 'sum of avg degrees %.2f'%v[0], 'overhead %.3f'%v[1], 'worst', v[2])
print('current (0,1,0,0):', cands[(0,1,0,0)])
k=best[0][0]
pad=lambda i,a=k[0],b=k[1],c=k[2],d=k[3]: i + a*(i>>3) + b*(i>>4) + c*(i>>5) + d*(i>>6)
for r in evalpad(pad,P13)+evalpad(pad,PKS): print('  ',r)
print('---- details')
P14=patterns(14,1024,(4,4,3,3),(3,3,4,4),16)
PK14=patterns(14,1024,(3,3,2,2,2,2),(),16)
P12=patterns(12,256,(3,3,3,3),(3,3,3,3),16)
for k in [(0,1,0,0),(0,0,3,0),(0,0,1,2)]:
    pad=lambda i,a=k[0],b=k[1],c=k[2],d=k[3]: i + a*(i>>3) + b*(i>>4) + c*(i>>5) + d*(i>>6)
    print(k)
    for nm,P in (('13',P13),('ks13',PKS),('14',P14),('ks14',PK14),('12',P12)):
        r=evalpad(pad,P)
        print('   ',nm,' '.join('%d'%x[1] for x in r), ' avg sum %.1f'%sum(x[2] for x in r))
