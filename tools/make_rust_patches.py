#!/usr/bin/env python3
"""Writes rust/patches/*.patch: the call-site changes that put fhe-math / fhe behind the `hip` feature
(SURVEY.md §8b's list of redirect points).  Each edit inserts NEW code at an anchor line of the reference file; the
patches are produced by `diff -U1` against /root/reference, so they carry one line of context and none of the
reference's code beyond it.  tests/test_rust_shim.py applies them with `patch --dry-run` when the reference is
present.  No Rust toolchain exists in the build image: the patches are reviewed source, not compiled here."""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FHE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "rust", "patches")

# (patch name, file, [(anchor line (exact, stripped), "after" | "before", inserted text)])
EDITS = [
    ("01-fhe-math-cargo", "crates/fhe-math/Cargo.toml", [
        ('tfhe-ntt = ["dep:tfhe-ntt"]', "after", 'hip = ["dep:fhe-math-hip"]  # MI355X engine (libfhe_hip.so) behind Poly / Scaler\n'),
        ("tfhe-ntt = { workspace = true, optional = true }", "after",
         'fhe-math-hip = { path = "../../rust/fhe-math-hip", optional = true }\n'),
    ]),
    ("02-ntt-operator-tables", "crates/fhe-math/src/ntt/native.rs", [
        ("impl NttOperator {", "after", '''    /// The operator's tables, which the `hip` backend uploads as they are (`fhe_ctx_create`): psi stays the host's.
    #[cfg(feature = "hip")]
    pub(crate) fn hip_tables(&self) -> (&[u64], &[u64], &[u64], &[u64], u64, u64) {
        (&self.omegas, &self.omegas_shoup, &self.zetas_inv, &self.zetas_inv_shoup, self.size_inv, self.size_inv_shoup)
    }

'''),
    ]),
    ("03-rq-context", "crates/fhe-math/src/rq/context.rs", [
        ("pub(crate) next_context: Option<Arc<Context>>,", "after", '''    /// Device twin of this context (tables uploaded once; immutable, shared by every Poly over the context).
    #[cfg(feature = "hip")]
    pub(crate) hip: fhe_math_hip::Handle<fhe_math_hip::HipCtx>,
'''),
        ("Ok(Self {", "before", '''            #[cfg(feature = "hip")]
            let hip = {
                let cat = |f: fn(&NttOperator) -> &[u64]| ops.iter().flat_map(|o| f(o).iter().copied()).collect::<Vec<u64>>();
                let (om, oms) = (cat(|o| o.hip_tables().0), cat(|o| o.hip_tables().1));
                let (zi, zis) = (cat(|o| o.hip_tables().2), cat(|o| o.hip_tables().3));
                let si = ops.iter().map(|o| o.hip_tables().4).collect::<Vec<u64>>();
                let sis = ops.iter().map(|o| o.hip_tables().5).collect::<Vec<u64>>();
                let tables = fhe_math_hip::NttTables {
                    omegas: &om, omegas_shoup: &oms, zetas_inv: &zi, zetas_inv_shoup: &zis, size_inv: &si, size_inv_shoup: &sis,
                };
                // device 0; FHE_HIP_DEVICE selects another one (one process per GPU when a batch is sharded)
                let dev = std::env::var("FHE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
                fhe_math_hip::Handle::new(
                    fhe_math_hip::HipCtx::new(dev, degree, moduli, Some(&tables)).map_err(crate::hip_error)?,
                )
            };
'''),
        ("next_context,", "after", '''                #[cfg(feature = "hip")]
                hip,
'''),
    ]),
    ("04-rq-poly", "crates/fhe-math/src/rq/mod.rs", [
        ("fn ntt_forward(&mut self) {", "after", '''        #[cfg(feature = "hip")]
        if let Some(h) = self.ctx.hip.get() {
            // [L][N] standard layout: exactly the buffer the C ABI takes
            h.ntt_forward(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: ntt_forward");
            return;
        }
'''),
        ("fn ntt_backward(&mut self) {", "after", '''        #[cfg(feature = "hip")]
        if let Some(h) = self.ctx.hip.get() {
            h.ntt_backward(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: ntt_backward");
            return;
        }
'''),
    ]),
    ("05-rq-ops", "crates/fhe-math/src/rq/ops.rs", [
        ("impl AddAssign<&Poly<PowerBasis>> for Poly<PowerBasis> {", "before", '''/// `hip` backend of the element-wise assignments below: one call on the `[L][N]` buffers, values identical.
#[cfg(feature = "hip")]
macro_rules! hip_elementwise {
    ($self:ident, $p:ident, $method:ident) => {
        if let Some(h) = $self.ctx.hip.get() {
            h.$method($self.coefficients.as_slice_mut().unwrap(), $p.coefficients.as_slice().unwrap())
                .expect(concat!("fhe_hip: ", stringify!($method)));
            return;
        }
    };
}

'''),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", '''        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, add_assign);
'''),
    ]),
    ("06-rq-scaler", "crates/fhe-math/src/rq/scaler.rs", [
        ("scaler: RnsScaler,", "after", '''    /// Device twin (constants of `scaler` uploaded once).
    #[cfg(feature = "hip")]
    hip: fhe_math_hip::Handle<fhe_math_hip::HipScaler>,
'''),
        ("let mut new_coefficients = Array2::<u64>::zeros((self.to.q.len(), self.to.degree));", "after", '''
            #[cfg(feature = "hip")]
            if let Some(h) = self.hip.get() {
                // copy of the common rows, inverse NTT, per-column RnsScaler::scale and the forward NTT of the
                // new rows: one device call (fhe_poly_scale), same canonical residues
                h.scale(
                    p.coefficients.as_slice().unwrap(),
                    new_coefficients.as_slice_mut().unwrap(),
                    R::REPRESENTATION != Representation::PowerBasis,
                )
                .map_err(crate::hip_error)?;
                return Ok(Poly {
                    ctx: self.to.clone(),
                    allow_variable_time_computations: p.allow_variable_time_computations,
                    coefficients: new_coefficients,
                    coefficients_shoup: None,
                    has_lazy_coefficients: false,
                    _repr: PhantomData,
                });
            }
'''),
    ]),
    ("10-fhe-math-errors", "crates/fhe-math/src/lib.rs", [
        ("pub use errors::{Error, PolynomialSerializationError, Result};", "after", '''
/// `fhe_status` of the `hip` backend -> the `Error` variant the native path returns in the same situation
/// (table in include/fhe_hip.h); HIP runtime failures and argument errors have no native counterpart and panic,
/// as the reference does on programming errors.
#[cfg(feature = "hip")]
pub(crate) fn hip_error(e: fhe_math_hip::HipError) -> Error {
    use fhe_math_hip::status as st;
    match e.status {
        st::CONTEXT_MISMATCH => Error::PolynomialContextMismatch,
        st::NO_MORE_CONTEXT => Error::NoMoreContext,
        st::CONTEXT_NOT_REACHABLE => Error::ContextNotReachable,
        st::EMPTY_DOT_PRODUCT => Error::EmptyDotProduct,
        _ => panic!("{e}"),
    }
}
'''),
    ]),
    ("07-bfv-multiplicator", "crates/fhe/src/bfv/ops/mul.rs", [
        ("level: usize,\n}", "before_last_line", '''    /// Device twin: extenders, down scaler, relinearisation key (`fhe_mul_create`).
    #[cfg(feature = "hip")]
    hip: fhe_math_hip::Handle<fhe_math_hip::HipMul>,
'''),
        ("// Extend", "before", '''        #[cfg(feature = "hip")]
        if let Some(h) = self.hip.get() {
            // extend, tensor, scale, relinearise (and switch down) of one ciphertext pair on the device; the batched
            // form (`multiply_batch`) amortises the PCIe copies and is what the throughput numbers use
            let flat = |ct: &Ciphertext| ct.iter().flat_map(|p| p.coefficients().iter().copied().collect::<Vec<u64>>()).collect::<Vec<u64>>();
            let (parts, rows) = h.out_shape().map_err(crate::hip_error)?;
            let mut out = vec![0u64; parts * rows * self.par.degree()];
            h.multiply(&flat(lhs), &flat(rhs), &mut out, 1).map_err(crate::hip_error)?;
            return Ciphertext::from_ntt_coefficients(&self.par, &out, parts, self.level + usize::from(self.mod_switch));
        }

'''),
    ]),
    ("08-bfv-key-switch", "crates/fhe/src/bfv/keys/key_switching_key.rs", [
        ("let mut c0 = Poly::<Ntt>::zero(&self.ctx_ksk);", "before", '''        #[cfg(feature = "hip")]
        if let Some(h) = self.hip.get() {
            // lazy lift + NTT per (digit, key modulus) + Shoup MAC, fused on the device (fhe_key_switch)
            let n = self.ctx_ksk.moduli().len() * self.par.degree();
            let (mut o0, mut o1) = (vec![0u64; n], vec![0u64; n]);
            h.key_switch(p.coefficients().as_slice().unwrap(), &mut o0, &mut o1, 1).map_err(crate::hip_error)?;
            return Ok((Poly::<Ntt>::from_canonical_ntt(&self.ctx_ksk, o0)?, Poly::<Ntt>::from_canonical_ntt(&self.ctx_ksk, o1)?));
        }
'''),
    ]),
    ("09-bfv-ciphertext-switch-down", "crates/fhe/src/bfv/ciphertext.rs", [
        ("self.seed = None;\n        for ci in self.c.iter_mut() {", "before_hip_switch_down", '''        #[cfg(feature = "hip")]
        if let Some(h) = self.c[0].ctx().hip_handle() {
            // all parts at once: inverse NTT, divide-and-round by the last modulus, forward NTT (fhe_bfv_switch_down)
            let flat = self.c.iter().flat_map(|p| p.coefficients().iter().copied().collect::<Vec<u64>>()).collect::<Vec<u64>>();
            let rows = self.c[0].ctx().moduli().len() - 1;
            let mut out = vec![0u64; self.c.len() * rows * self.par.degree()];
            h.ciphertext_switch_down(self.c.len(), &flat, &mut out).map_err(crate::hip_error)?;
            *self = Ciphertext::from_ntt_coefficients(&self.par, &out, self.c.len(), self.level + 1)?;
            return Ok(());
        }
'''),
    ]),
]


def apply(text, edits, path):
    lines = text.split("\n")
    for anchor, where, ins in edits:
        first = anchor.split("\n")[0].strip()
        idx = [i for i, l in enumerate(lines) if l.strip() == first]
        if "\n" in anchor:   # multi-line anchor: the following lines must match too
            rest = [a.strip() for a in anchor.split("\n")[1:]]
            idx = [i for i in idx if [l.strip() for l in lines[i + 1:i + 1 + len(rest)]] == rest]
        if not idx:
            raise SystemExit(f"{path}: anchor not found: {first!r}")
        i = idx[0]
        new = ins.rstrip("\n").split("\n")
        if where == "after":
            lines[i + 1:i + 1] = new
        elif where in ("before", "before_last_line", "before_hip_switch_down"):
            at = i + 1 if where == "before_hip_switch_down" else i
            lines[at:at] = new
        else:
            raise SystemExit(where)
    return "\n".join(lines)


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not present: the committed patches stay as they are")
    os.makedirs(OUT, exist_ok=True)
    for name, rel, edits in EDITS:
        src = open(os.path.join(REF, rel)).read()
        dst = apply(src, edits, rel)
        diff = difflib.unified_diff(src.split("\n"), dst.split("\n"), "a/" + rel, "b/" + rel, n=1, lineterm="")
        open(os.path.join(OUT, name + ".patch"), "w").write("\n".join(diff) + "\n")
        print(name, "ok")


if __name__ == "__main__":
    main()
