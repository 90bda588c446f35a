#!/usr/bin/env python3
"""Instruction mix of one kernel out of a device assembly dump (hipcc -S --cuda-device-only):
   python tools/isa_mix.py /tmp/fhe_dev.s '<substring of the mangled name>'
Counts static instructions by mnemonic class (the kernels' hot code is straight-line, fully unrolled, so the static mix
is the dynamic mix up to the few loops) -- what the round-4 look for instructions around the butterflies used."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.endswith(":") is False and re.match(r"^_Z\S*: ", l) and pat in l.split(":")[0]:
        start = i
        break
    if re.match(r"^_Z\S*:\s*;", l) and pat in l:
        start = i
        break
if start is None:
    sys.exit("kernel not found")
mix = collections.Counter()
n = 0
for l in lines[start + 1:]:
    if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^\t([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith(("s_", "v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        mix[op] += 1
        n += 1
print(lines[start].split(":")[0])
groups = collections.Counter()
for op, c in mix.items():
    g = ("valu_mul" if op in ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32") else
         "valu_other" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu_nop_wait" if op in ("s_nop", "s_waitcnt", "s_barrier") else "salu")
    groups[g] += c
print("total", n, dict(groups))
for op, c in mix.most_common(28):
    print(f"  {op:24s} {c}")
