#!/usr/bin/env python3
"""Compact register / scratch / occupancy table of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [regex]   -- one line per kernel whose demangled name matches"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
extra = sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra,
                    os.path.join(ROOT, "fhe.rs_amd", "csrc", "fhe_hip.cpp"), "-o", "/tmp/_fhe_ru.so"],
                   capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    short = re.sub(r"\(.*", "", name).replace("void fhe::k::", "")
    if pat.search(short):
        d = rows[mangled]
        print(f"{short:60s} vgpr {d.get('VGPRs', -1):4d} scratch {d.get('ScratchSize', -1):4d} occ {d.get('Occupancy', -1):2d} lds {d.get('LDS Size', -1)}")
