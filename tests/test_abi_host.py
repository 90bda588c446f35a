"""CPU checks on the REAL HIP build (no compute, no GPU): the shared library loads, exports
every symbol include/fhe_hip.h declares, and its host-only setup logic (device = -1: prime
generation, NTT tables, BigUint scaler constants, parameter levels, error codes) matches the
oracle.  The product path must fail loudly without the extension: also checked here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

import cases
from helpers import HIP_LIB, ROOT, load_engine


@pytest.fixture(scope="module")
def fhe():
    import __graft_entry__ as g
    g.build()
    return load_engine("hip")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(fhe):
    lib = ctypes.CDLL(HIP_LIB)
    names = declared_symbols()
    assert len(names) >= 60
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/fhe_hip.h but not exported"
    from fhe_rs_amd import _lib
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)


def test_loaded_library_is_the_hip_build(fhe):
    from fhe_rs_amd import _lib
    assert _lib.loaded_path() == HIP_LIB
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-n", HIP_LIB], text=True) \
        if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") else ""
    raw = open(HIP_LIB, "rb").read()
    assert b"gfx950" in raw, "no gfx950 code object embedded"
    del out


def test_release_build_reads_no_environment(fhe):
    """No wrong-result or kernel-selection switch can reach the shipped library: it does not import getenv and holds
    none of the FHE_* variable names (those exist only in -DFHE_LAB builds, which the package never loads)."""
    raw = open(HIP_LIB, "rb").read()
    for name in (b"FHE_DEBUG", b"FHE_LAB_", b"FHE_NO_", b"FHE_KS_", b"FHE_NTT_", b"FHE_SENS"):
        assert name not in raw, name
    nm = subprocess.check_output(["nm", "-D", "--undefined-only", HIP_LIB], text=True)
    assert "getenv" not in nm
    # the build recipe itself never defines FHE_LAB, and the product sources keep the lab kernels out of reach
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "FHE_LAB" not in entry
    csrc = os.path.join(ROOT, "fhe.rs_amd", "csrc")
    shipped = [f for f in os.listdir(csrc) if os.path.isfile(os.path.join(csrc, f))]   # (lab-only sources live in tools/lab/)
    assert {"kernels.hpp", "kernels_ntt.hpp", "kernels_ks.hpp", "kernels_scaler.hpp"} <= set(shipped)
    for f in shipped:
        text = open(os.path.join(csrc, f)).read()
        for rejected in ("ntt_fwd_swap_kernel<", "ntt_fwd8_kernel<", "ks_pair_kernel<", "FHE_SENS &", "getenv(\"FHE_DEBUG"):
            assert rejected not in text, (f, rejected)
        assert not os.path.isdir(os.path.join(csrc, "lab")), "lab sources belong in tools/lab/, not in the product tree"
        if "lab/" in text:   # every include of a lab file (tools/lab/, found through -I tools in lab builds) sits behind the FHE_LAB guard
            for m in re.finditer(r'#include "lab/', text):
                assert "#if defined(FHE_LAB)" in text[max(0, m.start() - 200):m.start()], f


def test_missing_extension_fails_loudly(tmp_path):
    """No silent CPU fallback: importing the package from a copy without the .so raises."""
    import shutil
    dst = tmp_path / "fhe.rs_amd"
    shutil.copytree(os.path.join(ROOT, "fhe.rs_amd"), dst, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    code = ("import importlib.util,sys;"
            f"s=importlib.util.spec_from_file_location('x', r'{dst}/__init__.py', submodule_search_locations=[r'{dst}']);"
            "m=importlib.util.module_from_spec(s);sys.modules['x']=m;s.loader.exec_module(m);"
            "m.Context([1153],8,device=-1)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_product_never_imports_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "fhe.rs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(base, f)).read()
                assert "fhe_oracle" not in text and "oracle/" not in text.replace("touches `oracle/`", ""), f


def test_host_context_tables(fhe):
    cases.case_context_tables(fhe)


def test_host_errors_and_primes(fhe):
    cases.case_errors(fhe)


def test_host_primes_match_reference_kats(fhe):
    """primes.rs:67-101 and parameters.rs:840-856 through the C ABI."""
    upper, got = ((1 << 64) - 1) >> 2, []
    while len(got) != 5:
        upper = fhe.generate_prime(62, 2 * 1048576, upper)
        got.append(upper)
    assert got == [4611686018326724609, 4611686018309947393, 4611686018282684417, 4611686018257518593,
                   4611686018232352769]
    assert fhe.generate_moduli([62, 62, 62, 61, 60, 11], 16) == [
        4611686018427387617, 4611686018427387329, 4611686018427387073, 2305843009213693921,
        1152921504606845473, 2017]
    assert fhe.generate_moduli([60] * 4, 8192) == [1152921504606830593, 1152921504606748673,
                                                    1152921504606683137, 1152921504606601217]


def test_host_scaler_constants_c2(fhe):
    """BigUint precompute of the three C2 scalers (N=8192 moduli; host-only contexts use a
    small degree to keep the oracle side fast: the constants do not depend on N)."""
    from fhe_oracle.rq import Context as OCtx, Scaler as OScaler
    from fhe_oracle.rns import ScalingFactor
    q = [1152921504606830593, 1152921504606748673, 1152921504606683137, 1152921504606601217]
    ext = [4611686018427322369, 4611686018427289601, 4611686018426454017, 4611686018426257409, 4611686018425815041]
    n, t = 8192, 1032193
    ob, om = OCtx(q, n), OCtx(q + ext, n)
    cb, cm = fhe.Context(q, n, device=-1), fhe.Context(q + ext, n, device=-1)
    for (of, ot, cf, ct, num, den) in ((ob, om, cb, cm, 1, 1), (om, ob, cm, cb, t, ob.modulus())):
        osc = OScaler(of, ot, ScalingFactor(num, den)).scaler
        sc = fhe.Scaler(cf, ct, num, den)
        assert sc.constants(0).tolist() == osc.gamma
        assert sc.constants(2).tolist() == [v for r in osc.omega for v in r]
        assert sc.constants(3).tolist() == [v for r in osc.omega_shoup for v in r]
        assert sc.constants(4).tolist() == osc.theta_omega_lo and sc.constants(5).tolist() == osc.theta_omega_hi
        assert sc.constants(7).tolist() == osc.theta_garner_lo and sc.constants(8).tolist() == osc.theta_garner_hi
        assert sc.constants(9).tolist()[:4] == [osc.theta_gamma_lo, osc.theta_gamma_hi,
                                               1 if osc.theta_gamma_sign else 0, osc.theta_garner_shift]


def test_header_is_plain_c_and_links(tmp_path):
    """include/fhe_hip.h is the boundary a C / cgo / JNI / Rust-bindgen host consumes: it must
    compile as C99 on its own, and a C program linked against the library can call the host-only
    entry points (no GPU needed: prime search and a device = -1 context)."""
    import subprocess
    src = tmp_path / "abi_smoke.c"
    src.write_text(r'''
#include "fhe_hip.h"
#include <stdio.h>
int main(void) {
    uint64_t p = fhe_generate_prime(62, 2 * 8192, (uint64_t)1 << 62);
    uint64_t moduli[2] = {4611686018427322369ull, 4611686018427289601ull};
    fhe_ctx *ctx = NULL;
    fhe_status st = fhe_ctx_create(-1, 8192, 2, moduli, NULL, NULL, NULL, NULL, NULL, NULL, &ctx);
    printf("%llu %d %zu %zu\n", (unsigned long long)p, (int)st, fhe_ctx_degree(ctx), fhe_poly_serialized_size(ctx));
    /* compute calls on a host-only handle fail cleanly with FHE_E_NO_DEVICE */
    uint64_t dummy[2 * 8192] = {0};
    st = fhe_ntt_forward(ctx, dummy, 1);
    printf("%d %s\n", (int)st, st == FHE_E_NO_DEVICE ? "no-device" : "unexpected");
    fhe_ctx_destroy(ctx);
    return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "fhe.rs_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I", inc,
                           os.path.join(inc, "fhe_hip.h")])
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", inc, str(src), "-o", str(exe), "-L", libdir,
                           "-lfhe_hip", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    assert out[0] == "4611686018427322369 0 8192 126976", out   # 2 x 8192 x 62 bits / 8
    assert out[1] == "-18 no-device", out
