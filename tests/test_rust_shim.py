"""The Rust side of the boundary (rust/) is source that cannot be compiled here (no cargo / rustc in the image), so
it is checked structurally: the generated `extern "C"` block must mirror include/fhe_hip.h symbol for symbol and type
for type, the generator must reproduce the committed file, the call-site patches must apply to the reference checkout,
and the status constants of the safe wrappers must equal the header's enum."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as gen  # noqa: E402

FFI = os.path.join(ROOT, "rust", "fhe-math-hip", "src", "ffi.rs")
LIB = os.path.join(ROOT, "rust", "fhe-math-hip", "src", "lib.rs")
HEADER = os.path.join(ROOT, "include", "fhe_hip.h")


def rust_fns(text):
    """{name: ([arg types], return type or None)} of the extern block."""
    body = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (fhe_[a-z0-9_]+)\((.*?)\)(?: -> ([^;]+))?;", body, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in re.split(r",\s*(?![^<]*>)", m.group(2)) if a.strip()]
        out[m.group(1)] = (args, m.group(3).strip() if m.group(3) else None)
    return out


def test_ffi_matches_header_symbol_for_symbol():
    protos = gen.prototypes(open(HEADER).read())
    rs = rust_fns(open(FFI).read())
    assert len(protos) >= 95 and len({n for _, n, _ in protos}) == len(protos)
    assert set(rs) == {n for _, n, _ in protos}, "extern block and header export different symbols"
    for ret, name, params in protos:
        args, rret = rs[name]
        assert len(args) == len(params), f"{name}: arity {len(args)} != {len(params)}"
        assert args == [gen.rust_type(t) for t, _ in params], name
        assert rret == (None if ret == "void" else gen.rust_type(ret)), name


def test_ffi_is_what_the_generator_produces():
    assert open(FFI).read() == gen.render(gen.prototypes(open(HEADER).read())), "run tools/gen_rust_ffi.py"


def test_header_symbols_match_the_ctypes_table():
    """Three mirrors of one ABI: the header, the Rust extern block and fhe.rs_amd/_lib.py."""
    sys.path.insert(0, ROOT)
    from fhe_rs_amd import _lib
    names = {n for _, n, _ in gen.prototypes(open(HEADER).read())}
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    for _, name, params in gen.prototypes(open(HEADER).read()):
        assert len(_lib.SIGNATURES[name][1]) == len(params), name


def test_status_constants_match_header_enum():
    hdr = dict(re.findall(r"FHE_E_([A-Z_]+) = (-\d+)", open(HEADER).read()))
    lib = dict(re.findall(r"pub const ([A-Z_]+): i32 = (-\d+);", open(LIB).read()))
    assert lib and lib == hdr


def test_wrappers_only_call_declared_functions():
    declared = set(rust_fns(open(FFI).read()))
    used = set(re.findall(r"ffi::(fhe_[a-z0-9_]+)\(", open(LIB).read()))
    assert used and used <= declared, used - declared


@pytest.mark.skipif(not os.path.isdir("/root/reference/crates"), reason="reference checkout not present")
def test_patches_apply_to_the_reference():
    pdir = os.path.join(ROOT, "rust", "patches")
    patches = sorted(f for f in os.listdir(pdir) if f.endswith(".patch"))
    assert len(patches) >= 10
    for f in patches:
        r = subprocess.run(["patch", "--dry-run", "-p1", "-s", "-i", os.path.join(pdir, f)], cwd="/root/reference",
                           capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stdout, r.stderr)
        added = [l[1:] for l in open(os.path.join(pdir, f)) if l.startswith("+") and not l.startswith("+++")]
        removed = [l for l in open(os.path.join(pdir, f)) if l.startswith("-") and not l.startswith("---")]
        assert added and not removed, f"{f}: the patches only insert code behind cfg(feature = \"hip\")"
