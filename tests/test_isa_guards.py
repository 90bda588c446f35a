"""Static guards on the gfx950 code of the unfused key switch's stage A (CPU-only: hipcc cross-compiles, nothing runs).

Round 5 found `ks_ntt_kernel`'s loader issuing its sixteen first-pass global loads ONE AT A TIME
(`global_load -> s_waitcnt vmcnt(0) -> s_cbranch`, sixteen times per thread) wherever per-element control flow sat behind
the loads: the run-time lift-mode test (bases whose moduli differ in width, e.g. the reference's stock parameter sets) and
the `minus` test of the folded first stages (rows larger than LDS).  That cost 4-5 us of a 12 us single-workgroup transform
(profiles/r05_final6_latency_breakdown.json).  Nothing in a parity test notices such a regression, so this test compiles
the kernel's instances (tests/isa/ks_ntt_probe.cpp, a few seconds) and counts, with tools/isa_serial_loads.py, the loads
a wave waits for alone and the branches."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_unfused_stage_a_issues_its_loads_in_batches(tmp_path):
    asm = tmp_path / "ks_ntt_probe.s"
    r = subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Wno-unused-function", "-I", os.path.join(ROOT, "fhe.rs_amd", "csrc"),
                        os.path.join(ROOT, "tests", "isa", "ks_ntt_probe.cpp"), "-o", str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_serial_loads.py"), str(asm), "ks_ntt_kernel"],
                         capture_output=True, text=True, check=True).stdout
    rows = re.findall(r"(\d+) isolated-wait loads of\s+(\d+) loads,\s+(\d+) branches,\s+(\d+) instrs\s+(ks_ntt_kernel<[^>]*>)", out)
    assert len(rows) == 4, out
    for isolated, loads, branches, _, name in rows:
        # (the closing wait of a batch counts its last load: a handful per kernel is the floor; the broken loaders had
        # 16-64 of them and 100-320 branches)
        assert int(isolated) <= 6, (name, out)
        assert int(branches) <= 24, (name, out)
        assert int(loads) >= 32, (name, out)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_fused_key_switch_instances_keep_their_registers(tmp_path):
    """The N = 16384 tile of the fused key switch sits at 128 of 128 VGPRs; round 6 found that restructuring its loader
    (the lift mode as a compile-time constant) makes EVERY LOGN = 14 instance spill 244-340 B per lane --
    including the RNS instances BASELINE configs C3 / C5 run on -- although the source of those was semantically unchanged
    (profiles/r06_ks_lift_compile_time_rejected.txt).  No parity test notices a spill: this one reads it from the ISA.
    Also: the loaders' global loads go out in batches (isolated waits at the floor of ~20 that closes batches)."""
    asm = tmp_path / "ks_fused_probe.s"
    r = subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Wno-unused-function", "-I", os.path.join(ROOT, "fhe.rs_amd", "csrc"),
                        os.path.join(ROOT, "tests", "isa", "ks_fused_probe.cpp"), "-o", str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    scratch = dict(re.findall(r"\.set (_ZN3fhe1k\d+ks_fused\w+)\.private_seg_size, (\d+)", text))
    vgprs = dict(re.findall(r"\.set (_ZN3fhe1k\d+ks_fused\w+)\.num_vgpr, (\d+)", text))
    assert len(scratch) == 11 and len(vgprs) == 11, (scratch, vgprs)     # 7 integer instances + 4 F64 instances (round 6)
    for name, b in scratch.items():
        assert int(b) == 0, (name, b)
        assert int(vgprs[name]) <= 128, (name, vgprs[name])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_serial_loads.py"), str(asm), "ks_fused"],
                         capture_output=True, text=True, check=True).stdout
    rows = re.findall(r"(\d+) isolated-wait loads of\s+(\d+) loads,\s+(\d+) branches,\s+(\d+) instrs\s+(ks_fused\w*<[^>]*>)", out)
    assert len(rows) == 11, out
    for isolated, loads, branches, _, name in rows:
        assert int(isolated) <= 24 and int(loads) >= 40, (name, out)
        if "true, 0, false, 0>" in name or "true, 1, false, 0>" in name:      # the integer RNS instances: no per-element lift branches
            assert int(branches) <= 60, (name, out)
