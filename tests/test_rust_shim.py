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
    assert len(patches) >= 13
    for f in patches:
        r = subprocess.run(["patch", "--dry-run", "-p1", "-s", "-i", os.path.join(pdir, f)], cwd="/root/reference",
                           capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stdout, r.stderr)
        added = [l[1:] for l in open(os.path.join(pdir, f)) if l.startswith("+") and not l.startswith("+++")]
        removed = [l for l in open(os.path.join(pdir, f)) if l.startswith("-") and not l.startswith("---")]
        assert added and not removed, f"{f}: the patches only insert code behind cfg(feature = \"hip\")"


# ---- every identifier a patch introduces must be defined: by rust/fhe-math-hip/src/lib.rs, by a patch of the same
# ---- crate, or by the reference itself (round 2's patches called four helpers nobody had written)
PATCH_DIR = os.path.join(ROOT, "rust", "patches")
RUST_STD = {  # methods / functions of std, ndarray and itertools the inserted code calls
    "iter", "map", "collect", "flat_map", "copied", "to_vec", "len", "unwrap", "expect", "clone", "as_slice",
    "as_slice_mut", "chunks_exact", "first", "ok_or", "map_err", "any", "zip", "into", "max", "as_ref", "from", "default",
    "ptr_eq", "get_or_try_init", "reset", "is_none", "new", "with_capacity", "extend_from_slice", "push",
}


def _added_lines():
    out = {}
    for f in sorted(os.listdir(PATCH_DIR)):
        if f.endswith(".patch"):
            lines = open(os.path.join(PATCH_DIR, f)).read().split("\n")
            target = [l[6:] for l in lines if l.startswith("+++ b/")][0]
            out[f] = (target, [l[1:] for l in lines if l.startswith("+") and not l.startswith("+++")])
    return out


def _strip_comments(lines):
    return "\n".join(re.sub(r"//.*$", "", l) for l in lines)


def test_patches_use_only_items_the_shim_defines():
    lib = open(LIB).read()
    pub_items = set(re.findall(r"pub (?:struct|enum|type|mod|fn) ([A-Za-z_][A-Za-z0-9_]*)", lib))
    methods = set(re.findall(r"\bfn ([a-z_][a-z0-9_]*)", lib))
    consts = set(re.findall(r"pub const ([A-Z_]+):", lib))
    fields = set(re.findall(r"pub ([a-z_][a-z0-9_]*):", lib))
    used_any = False
    for f, (_, lines) in _added_lines().items():
        code = _strip_comments(lines)
        for path in re.findall(r"fhe_math_hip::((?:[A-Za-z_][A-Za-z0-9_]*)(?:::[A-Za-z_][A-Za-z0-9_]*)*)", code):
            used_any = True
            parts = path.split("::")
            assert parts[0] in pub_items, f"{f}: fhe_math_hip::{path}: `{parts[0]}` is not a public item of lib.rs"
            for seg in parts[1:]:
                assert seg in methods | consts | pub_items, f"{f}: fhe_math_hip::{path}: `{seg}` is not defined in lib.rs"
        for const in re.findall(r"\bst::([A-Z_]+)", code):
            assert const in consts, f"{f}: status constant {const}"
        for fld in re.findall(r"\bk\.([a-z_]+)\b", code):   # RnsScalerConstantsBuf accesses in 06-rq-scaler
            assert fld in fields | methods, f"{f}: k.{fld}"
    assert used_any


def test_patches_call_only_defined_helpers():
    """`crate::hip_error`, `.hip_handle()`, `Ciphertext::from_ntt_coefficients` ...: every function the inserted code
    calls is defined by lib.rs, by a patch that touches the SAME crate, by the reference, or is a std / ndarray method."""
    lib_fns = set(re.findall(r"\bfn ([a-z_][a-z0-9_]*)", open(LIB).read()))
    added = _added_lines()
    crate_of = lambda target: target.split("/")[1]            # crates/<crate>/...
    defined = {}                                               # crate -> functions defined by patches
    for f, (target, lines) in added.items():
        defined.setdefault(crate_of(target), set()).update(re.findall(r"\bfn ([a-z_][a-z0-9_]*)", _strip_comments(lines)))
    have_ref = os.path.isdir("/root/reference/crates")
    ref_fns = set()
    if have_ref:
        for base, _, files in os.walk("/root/reference/crates"):
            for fn in files:
                if fn.endswith(".rs"):
                    ref_fns.update(re.findall(r"\bfn ([a-z_][a-z0-9_]*)", open(os.path.join(base, fn)).read()))
    for f, (target, lines) in added.items():
        if not target.endswith(".rs"):
            continue
        code = _strip_comments(lines)
        crate = crate_of(target)
        # crate-local paths must resolve inside THIS crate's patches (round 2: fhe called fhe-math's pub(crate) fn)
        for name in re.findall(r"\bcrate::([a-z_][a-z0-9_]*)\(|\bcrate::([a-z_][a-z0-9_]*)\)", code):
            name = name[0] or name[1]
            assert name in defined[crate] | (ref_fns if have_ref else {name}), f"{f}: crate::{name} is not defined in crate {crate}"
        for name in re.findall(r"(?:\.|::)([a-z_][a-z0-9_]*)\(", code):
            ok = (name in lib_fns or name in RUST_STD or any(name in d for d in defined.values())
                  or (not have_ref) or name in ref_fns)
            assert ok, f"{f}: `{name}(...)` is defined nowhere (lib.rs, the patches, the reference)"


def test_hip_fields_are_declared_and_filled():
    """A patch that reads `self.hip` must also add the field to the struct and to every struct literal of the file."""
    for f, (target, lines) in _added_lines().items():
        code = _strip_comments(lines)
        if "self.hip." not in code and "self.hip)" not in code:
            continue
        assert re.search(r"hip: fhe_math_hip::LazyHandle<", code), f"{f}: uses self.hip without declaring the field"
        assert "hip: Default::default()," in code, f"{f}: no struct literal is given the new field"
        if os.path.isdir("/root/reference/crates"):
            src = open(os.path.join("/root/reference", target)).read()
            struct = re.search(r"hip: fhe_math_hip::LazyHandle<fhe_math_hip::(\w+)>", code).group(1)
            owner = {"HipCtx": "Context", "HipScaler": "Scaler", "HipKsk": "KeySwitchingKey", "HipMul": "Multiplicator"}[struct]
            body = src.split("#[cfg(test)]")[0]
            literals = len(re.findall(r"(?<!-> )\b(?:Self|%s) \{\n" % owner, body)) - len(re.findall(r"pub struct %s \{\n" % owner, body)) \
                - len(re.findall(r"impl(?:<[^>]*>)? (?:\w+(?:<[^>]*>)? for )?%s \{\n" % owner, body))
            assert code.count("hip: Default::default(),") >= max(literals, 1), (f, literals, code.count("hip: Default::default(),"))


def test_shim_wrappers_validate_lengths():
    """safe functions forwarded caller-supplied `batch` values to the C ABI.  Now no public safe method
    takes a `batch`; every slice-taking method derives it (whole_batch) and checks the other slices (expect_len)."""
    lib = open(LIB).read()
    for m in re.finditer(r"pub fn (\w+)\(&self,([^)]*)\)[^{]*\{(.*?)\n    \}", lib, flags=re.S):
        name, args, body = m.groups()
        if "&[u64]" in args or "&mut [u64]" in args:
            assert "batch: usize" not in args, name
            assert "whole_batch(" in body or "expect_len(" in body or "switch_to_level(" in body, name
