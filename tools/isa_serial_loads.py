#!/usr/bin/env python3
"""Which kernels wait for single global loads?  For every kernel of a device assembly dump (hipcc -S --cuda-device-only, see
tools/isa_mix.py) counts the loads that are followed by `s_waitcnt vmcnt(0)` before any other load is issued -- a load the
wave sits out alone -- next to the kernel's loads, branches and instructions.  High counts with many branches are loaders whose
per-element control flow keeps the compiler from batching their loads (round 5: ks_ntt_kernel's lift mode and folded stages).
usage: python tools/isa_serial_loads.py /tmp/fhe_dev.s [regex on the kernel name]   (TOP=n lines, default 60)"""
import re,sys,subprocess,collections
lines=open(sys.argv[1]).read().split("\n")
kern=None; res={}
seq=[]
def flush():
    global seq
    if kern is None: return
    # count loads that are directly followed (before any other vmem load) by s_waitcnt vmcnt(0)
    n=0; loads=0; br=0
    i=0
    ops=seq
    for idx,(op,arg) in enumerate(ops):
        if op.startswith(("global_load","buffer_load","flat_load")):
            loads+=1
            # look ahead
            for op2,arg2 in ops[idx+1: idx+40]:
                if op2.startswith(("global_load","buffer_load","flat_load")): break
                if op2=="s_waitcnt" and "vmcnt(0)" in arg2: n+=1; break
        if op.startswith("s_cbranch"): br+=1
    res[kern]=(n,loads,br,len(ops))
for l in lines:
    m=re.match(r"^(_Z\S+):\s*;\s*@",l)
    if m:
        flush(); kern=m.group(1); seq=[]; continue
    if l.startswith(".Lfunc_end"):
        flush(); kern=None; seq=[]; continue
    m=re.match(r"^\t([a-z_0-9]+)\s*(.*)",l)
    if m and kern: seq.append((m.group(1),m.group(2)))
names=subprocess.run(["c++filt"],input="\n".join(res),capture_output=True,text=True).stdout.splitlines()
out=[]
for (k,v),nm in zip(res.items(),names):
    short=re.sub(r"\(.*","",nm).replace("void fhe::k::","")
    out.append((v[0],v[1],v[2],v[3],short))
out.sort(reverse=True)
import os
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for o in [x for x in out if flt is None or flt.search(x[4])][:int(os.environ.get("TOP", "60"))]: print("%4d isolated-wait loads of %4d loads, %4d branches, %6d instrs  %s"%o)
