#!/usr/bin/env python3
"""Derived per-kernel metrics from the SQ counter passes of tools/collect_sq.sh (summary via the raw CSVs):
instructions per wave, valu_pipe_floor (SQ_INSTS_VALU x 2 cycles / (kernel cycles x 1024 SIMDs); kernel cycles =
SQ_BUSY_CYCLES / 32), wave_wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, wave_wait_frac = SQ_WAIT_ANY /
SQ_WAVE_CYCLES, waves resident per SIMD = SQ_WAVE_CYCLES x 4 / (kernel cycles x 1024)  (SQ_WAVE_CYCLES in quad-cycles).  Kernels with resident workgroups
(ks_fused_kernel at N = 8192 since round 2: one workgroup per CU walks 8 items at C2) have fewer, longer waves:
their instructions per wave are per 8 items, not per item.
usage: sq_derive.py <dir> > profiles/<round>_sq_derived.json"""
import collections, csv, glob, json, os, sys

d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fhe" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fhe::k::", "")
        e = d[k][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
out = []
for k, v in sorted(d.items()):
    a = {c: s / n for c, (n, s) in v.items()}
    if "SQ_WAVES" not in a or a["SQ_WAVES"] == 0 or "synth" in k:
        continue
    waves, cyc = a["SQ_WAVES"], a["SQ_BUSY_CYCLES"] / 32.0
    out.append(dict(kernel=k, waves=int(waves), kernel_cycles=int(cyc),
                    valu_insts_per_wave=round(a["SQ_INSTS_VALU"] / waves),
                    salu_insts_per_wave=round(a["SQ_INSTS_SALU"] / waves),
                    smem_insts_per_wave=round(a["SQ_INSTS_SMEM"] / waves, 1),
                    lds_insts_per_wave=round(a["SQ_INSTS_LDS"] / waves, 1),
                    vmem_insts_per_wave=round((a["SQ_INSTS_VMEM_RD"] + a["SQ_INSTS_VMEM_WR"]) / waves, 1),
                    valu_pipe_floor=round(a["SQ_INSTS_VALU"] * 2 / (cyc * 1024), 3),
                    wave_wait_inst_frac=round(a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3),
                    wave_wait_frac=round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
                    waves_resident_per_simd=round(a["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024), 2)))
print(json.dumps(dict(note=__doc__.split("usage")[0].strip(), kernels=out), indent=1))
