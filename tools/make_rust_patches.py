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

# (patch name, file, [(anchor line(s) (exact, stripped), "after" | "before", inserted text)])
# A multi-line anchor matches consecutive lines; the text goes after / before its FIRST line.
EDITS = [
    # ------------------------------------------------------------------------------------------- fhe-math
    ("01-fhe-math-cargo", "crates/fhe-math/Cargo.toml", [
        ('tfhe-ntt = ["dep:tfhe-ntt"]', "after", 'hip = ["dep:fhe-math-hip"]  # MI355X engine (libfhe_hip.so) behind Poly / Scaler\n'),
        ("tfhe-ntt = { workspace = true, optional = true }", "after",
         'fhe-math-hip = { path = "../../rust/fhe-math-hip", optional = true }\n'),
    ]),
    ("02-ntt-operator-tables", "crates/fhe-math/src/ntt/native.rs", [
        ("impl NttOperator {", "after", """    /// The operator's tables, which the `hip` backend uploads as they are (`fhe_ctx_create`): psi stays the host's.
    #[cfg(feature = "hip")]
    pub(crate) fn hip_tables(&self) -> (&[u64], &[u64], &[u64], &[u64], u64, u64) {
        (&self.omegas, &self.omegas_shoup, &self.zetas_inv, &self.zetas_inv_shoup, self.size_inv, self.size_inv_shoup)
    }

"""),
    ]),
    ("03-rq-context", "crates/fhe-math/src/rq/context.rs", [
        ("pub(crate) next_context: Option<Arc<Context>>,", "after", """    /// Device twin of this context, made on first use (tables uploaded once; immutable, shared by every Poly over
    /// the context and by every clone made afterwards).  Takes no part in `PartialEq`.
    #[cfg(feature = "hip")]
    pub(crate) hip: fhe_math_hip::LazyHandle<fhe_math_hip::HipCtx>,
"""),
        ("impl Context {", "after", """    /// The device twin (`fhe_ctx_create` with the tables this context's `NttOperator`s already hold, so psi -- and
    /// with it every Ntt-form value -- is the host's).  Each level of a chain has its own twin; equal tables make
    /// them interchangeable on the device side (the engine compares table fingerprints, not pointers).
    #[cfg(feature = "hip")]
    pub fn hip_handle(&self) -> Result<&Arc<fhe_math_hip::HipCtx>> {
        self.hip.get_or_try_init(|| {
            let full = self.ops.len() * self.degree;
            let (mut om, mut oms, mut zi, mut zis) = (Vec::with_capacity(full), Vec::with_capacity(full), Vec::with_capacity(full), Vec::with_capacity(full));
            let (mut si, mut sis) = (Vec::with_capacity(self.ops.len()), Vec::with_capacity(self.ops.len()));
            for op in self.ops.iter() {
                let t = op.hip_tables();
                om.extend_from_slice(t.0);
                oms.extend_from_slice(t.1);
                zi.extend_from_slice(t.2);
                zis.extend_from_slice(t.3);
                si.push(t.4);
                sis.push(t.5);
            }
            let tables = fhe_math_hip::NttTables {
                omegas: &om, omegas_shoup: &oms, zetas_inv: &zi, zetas_inv_shoup: &zis, size_inv: &si, size_inv_shoup: &sis,
            };
            fhe_math_hip::HipCtx::new(fhe_math_hip::default_device(), self.degree, &self.moduli, Some(&tables))
                .map_err(crate::hip_error)
        })
    }

"""),
        ("next_context,", "after", """                #[cfg(feature = "hip")]
                hip: Default::default(),
"""),
    ]),
    ("04-rq-poly", "crates/fhe-math/src/rq/mod.rs", [
        ("fn ntt_forward(&mut self) {", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // [L][N] standard layout: exactly the buffer the C ABI takes
            let h = self.ctx.hip_handle().expect("fhe_hip: device context");
            h.ntt_forward(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: ntt_forward");
            return;
        }
"""),
        ("fn ntt_backward(&mut self) {", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            let h = self.ctx.hip_handle().expect("fhe_hip: device context");
            h.ntt_backward(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: ntt_backward");
            return;
        }
"""),
        ("match R::REPRESENTATION {", "before", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() && R::REPRESENTATION != Representation::NttShoup {
            // x -> x^i as one gather on the device (fhe_poly_substitute); NttShoup polynomials carry their twins
            // along and stay on the native path
            self.ctx
                .hip_handle()?
                .substitute(
                    i.exponent,
                    self.coefficients.as_slice().unwrap(),
                    q.coefficients.as_slice_mut().unwrap(),
                    R::REPRESENTATION == Representation::Ntt,
                )
                .map_err(crate::hip_error)?;
            return Ok(q);
        }
""", 0),
        ("let next_context = self.ctx.next_context.as_ref().unwrap();", "after", """
        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // divide-and-round by the last modulus, one lane per coefficient column (fhe_poly_switch_down)
            let mut lower = Array2::<u64>::zeros((self.ctx.q.len() - 1, self.ctx.degree));
            self.ctx
                .hip_handle()?
                .switch_down(self.coefficients.as_slice().unwrap(), lower.as_slice_mut().unwrap())
                .map_err(crate::hip_error)?;
            if !self.allow_variable_time_computations {
                self.coefficients.as_slice_mut().unwrap().zeroize();
            }
            self.coefficients = lower;
            self.ctx = next_context.clone();
            return Ok(());
        }
"""),
    ]),
    ("05-rq-ops", "crates/fhe-math/src/rq/ops.rs", [
        ("impl AddAssign<&Poly<PowerBasis>> for Poly<PowerBasis> {", "before", """/// `hip` backend of the element-wise assignments below: one call on the `[L][N]` buffers, values identical.
#[cfg(feature = "hip")]
macro_rules! hip_elementwise {
    ($self:ident, $p:ident, $method:ident) => {
        if fhe_math_hip::enabled() {
            let h = $self.ctx.hip_handle().expect("fhe_hip: device context");
            h.$method($self.coefficients.as_slice_mut().unwrap(), $p.coefficients.as_slice().unwrap())
                .expect(concat!("fhe_hip: ", stringify!($method)));
            return;
        }
    };
}

"""),
        # `self.allow_variable_time_computations &= p...;` occurs once per assignment impl, in source order:
        # AddAssign<PowerBasis> :15, SubAssign<PowerBasis> :60, AddAssign<Ntt> :97, SubAssign<Ntt> :142,
        # MulAssign<&Poly<Ntt>> :182, MulAssign<&Poly<NttShoup>> :212 (M/rq/ops.rs:10-418)
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, add_assign);
""", 0),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, sub_assign);
""", 1),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, add_assign);
""", 2),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, sub_assign);
""", 3),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        hip_elementwise!(self, p, mul_assign);
""", 4),
        ("self.allow_variable_time_computations &= p.allow_variable_time_computations;", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // (a lazy left operand is fine: the Shoup product takes any 64-bit value and returns a canonical one)
            let h = self.ctx.hip_handle().expect("fhe_hip: device context");
            h.mul_shoup_assign(
                self.coefficients.as_slice_mut().unwrap(),
                p.coefficients.as_slice().unwrap(),
                p.coefficients_shoup.as_ref().unwrap().as_slice().unwrap(),
            )
            .expect("fhe_hip: mul_shoup_assign");
            self.has_lazy_coefficients = false;
            return;
        }
""", 5),
        # `assert!(!self.has_lazy_coefficients);` opens the four Neg impls: &Poly<Ntt> :358, &Poly<PowerBasis> :375
        # (they negate a clone, `out`), Poly<Ntt> :392, Poly<PowerBasis> :408 (they negate `self`)
        ("let mut out = self.clone();", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            let h = out.ctx.hip_handle().expect("fhe_hip: device context").clone();
            h.neg_assign(out.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: neg_assign");
            return out;
        }
""", 0),
        ("let mut out = self.clone();", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            let h = out.ctx.hip_handle().expect("fhe_hip: device context").clone();
            h.neg_assign(out.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: neg_assign");
            return out;
        }
""", 1),
        ("assert!(!self.has_lazy_coefficients);", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            let h = self.ctx.hip_handle().expect("fhe_hip: device context").clone();
            h.neg_assign(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: neg_assign");
            return self;
        }
""", 2),
        ("assert!(!self.has_lazy_coefficients);", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            let h = self.ctx.hip_handle().expect("fhe_hip: device context").clone();
            h.neg_assign(self.coefficients.as_slice_mut().unwrap()).expect("fhe_hip: neg_assign");
            return self;
        }
""", 3),
    ]),
    ("06-rq-scaler", "crates/fhe-math/src/rq/scaler.rs", [
        ("scaler: RnsScaler,", "after", """    /// Device twin (constants of `scaler` uploaded on first use).
    #[cfg(feature = "hip")]
    hip: fhe_math_hip::LazyHandle<fhe_math_hip::HipScaler>,
"""),
        ("impl Scaler {", "after", """    /// The device twin (`fhe_scaler_create_from_constants` with every field `RnsScaler::new` computed: nothing is
    /// re-derived on the device side, so the rounding constants are the host's bit for bit).
    #[cfg(feature = "hip")]
    pub fn hip_handle(&self) -> Result<&Arc<fhe_math_hip::HipScaler>> {
        self.hip.get_or_try_init(|| {
            let (from, to) = (self.from.hip_handle()?, self.to.hip_handle()?);
            let k = self.scaler.hip_constants();
            fhe_math_hip::HipScaler::from_constants(from, to, self.number_common_moduli, k.is_one, &k.view())
                .map_err(crate::hip_error)
        })
    }

"""),
        ("number_common_moduli,\n            scaler,", "after_second", """            #[cfg(feature = "hip")]
            hip: Default::default(),
"""),
        ("let mut new_coefficients = Array2::<u64>::zeros((self.to.q.len(), self.to.degree));", "after", """
            #[cfg(feature = "hip")]
            if fhe_math_hip::enabled() {
                // copy of the common rows, inverse NTT, per-column RnsScaler::scale and the forward NTT of the
                // new rows: one device call (fhe_poly_scale), same canonical residues
                self.hip_handle()?
                    .scale(
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
"""),
    ]),
    ("10-fhe-math-errors", "crates/fhe-math/src/lib.rs", [
        ("pub use errors::{Error, PolynomialSerializationError, Result};", "after", """
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
"""),
    ]),
    ("11-rns-scaler-constants", "crates/fhe-math/src/rns/scaler.rs", [
        ("impl RnsScaler {", "after", """    /// Every constant `new` computed, flattened the way `fhe_scaler_create_from_constants` takes them
    /// (`omega*` row-major `[to][from]`, signs as bytes).
    #[cfg(feature = "hip")]
    pub(crate) fn hip_constants(&self) -> fhe_math_hip::RnsScalerConstantsBuf {
        fhe_math_hip::RnsScalerConstantsBuf {
            is_one: self.scaling_factor.is_one,
            gamma: self.gamma.to_vec(),
            gamma_shoup: self.gamma_shoup.to_vec(),
            omega: self.omega.iter().flat_map(|row| row.iter().copied()).collect(),
            omega_shoup: self.omega_shoup.iter().flat_map(|row| row.iter().copied()).collect(),
            theta_gamma_lo: self.theta_gamma_lo,
            theta_gamma_hi: self.theta_gamma_hi,
            theta_gamma_sign: self.theta_gamma_sign,
            theta_omega_lo: self.theta_omega_lo.to_vec(),
            theta_omega_hi: self.theta_omega_hi.to_vec(),
            theta_omega_sign: self.theta_omega_sign.iter().map(|&s| s as u8).collect(),
            theta_garner_lo: self.theta_garner_lo.to_vec(),
            theta_garner_hi: self.theta_garner_hi.to_vec(),
            theta_garner_shift: self.theta_garner_shift,
        }
    }

"""),
    ]),
    # ------------------------------------------------------------------------------------------------ fhe
    ("12-fhe-cargo", "crates/fhe/Cargo.toml", [
        ('tfhe-ntt = ["fhe-math/tfhe-ntt"]', "after",
         'hip = ["dep:fhe-math-hip", "fhe-math/hip"]  # MI355X engine behind Multiplicator / KeySwitchingKey / Ciphertext\n'),
        ('fhe-util = { version = "=0.1.1", path = "../fhe-util" }', "after",
         'fhe-math-hip = { path = "../../rust/fhe-math-hip", optional = true }\n'),
    ]),
    ("13-fhe-errors", "crates/fhe/src/lib.rs", [
        ("pub mod proto;", "after", """
/// `fhe_status` of the `hip` backend -> the `Error` this crate's native path returns in the same situation
/// (table in include/fhe_hip.h).  HIP runtime failures and argument errors have no native counterpart and panic, as
/// the reference does on programming errors.
#[cfg(feature = "hip")]
pub(crate) fn hip_error(e: fhe_math_hip::HipError) -> Error {
    use fhe_math_hip::status as st;
    match e.status {
        st::CONTEXT_MISMATCH => fhe_math::Error::PolynomialContextMismatch.into(),
        st::NO_MORE_CONTEXT => fhe_math::Error::NoMoreContext.into(),
        st::CONTEXT_NOT_REACHABLE => fhe_math::Error::ContextNotReachable.into(),
        st::PARAMETER_MISMATCH => Error::ParameterMismatch {
            left: ParameterSource::Ciphertext,
            right: ParameterSource::Parameters,
        },
        st::MUL_POLY_COUNT => CiphertextError::MultiplicationPolynomialCount { left: 0, right: 0, expected: 2 }.into(),
        _ => panic!("{e}"),
    }
}
"""),
    ]),
    ("09-bfv-ciphertext", "crates/fhe/src/bfv/ciphertext.rs", [
        ("/// Truncate the underlying vector of polynomials.", "before", """    /// The polynomials' coefficients, concatenated `[parts][L][N]`: the layout every ciphertext entry point of
    /// the `hip` backend takes (a batch is the concatenation of these).
    #[cfg(feature = "hip")]
    pub(crate) fn to_flat_coefficients(&self) -> Vec<u64> {
        self.c.iter().flat_map(|p| p.coefficients().iter().copied().collect::<Vec<u64>>()).collect()
    }

    /// Rebuilds a ciphertext at `level` from `parts` Ntt-form polynomials laid out `[parts][L][N]` (what the
    /// `hip` backend returns; canonical residues, so `TryConvertFrom<Vec<u64>>` takes them verbatim).
    #[cfg(feature = "hip")]
    pub(crate) fn from_ntt_coefficients(par: &Arc<BfvParameters>, flat: &[u64], parts: usize, level: usize) -> Result<Self> {
        use fhe_math::rq::traits::TryConvertFrom as PolyTryConvertFrom;
        let ctx = par.context_at_level(level)?;
        let per = ctx.moduli().len() * par.degree();
        if parts < 2 || flat.len() != parts * per {
            return Err(crate::CiphertextError::TooFewPolynomials { actual: flat.len() / per.max(1), minimum: 2 }.into());
        }
        let c = flat
            .chunks_exact(per)
            .map(|chunk| Poly::<Ntt>::try_convert_from(chunk.to_vec(), ctx, false).map_err(Error::MathError))
            .collect::<Result<Vec<Poly<Ntt>>>>()?;
        Ok(Self { par: par.clone(), seed: None, c, level })
    }

    /// Uploads a batch of ciphertexts (same level, same number of parts) once; the result stays on the GPU across
    /// `Multiplicator::multiply_dev`, `RelinearizationKey` / rotation calls on `DeviceCiphertexts`, and comes back
    /// through [`Ciphertext::from_device`].
    #[cfg(feature = "hip")]
    pub fn to_device(cts: &[Ciphertext], stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        let first = cts.first().ok_or(crate::CiphertextError::TooFewPolynomials { actual: 0, minimum: 2 })?;
        if cts.iter().any(|c| c.level != first.level || c.c.len() != first.c.len() || !Arc::ptr_eq(&c.par, &first.par)) {
            return Err(Error::ParameterMismatch { left: crate::ParameterSource::Ciphertext, right: crate::ParameterSource::Ciphertext });
        }
        let ctx = first.par.context_at_level(first.level)?;
        let flat = cts.iter().flat_map(|c| c.to_flat_coefficients()).collect::<Vec<u64>>();
        fhe_math_hip::DeviceCiphertexts::upload(fhe_math_hip::default_device(), &flat, first.c.len(), ctx.moduli().len(),
            first.par.degree(), first.level, stream)
            .map_err(crate::hip_error)
    }

    /// [`Ciphertext::switch_to_level`] on a device-resident batch over `par` (all of it at `d.level`): one inverse /
    /// forward transform pair and `target_level - d.level` divide-and-round steps on the GPU, stream-ordered; the
    /// same level checks and errors as the host method.
    #[cfg(feature = "hip")]
    pub fn switch_to_level_dev(par: &Arc<BfvParameters>, d: &fhe_math_hip::DeviceCiphertexts, target_level: usize,
                               stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        if target_level < d.level || target_level > par.max_level() {
            return Err(Error::InvalidLevel { level: target_level, min_level: d.level, max_level: par.max_level() });
        }
        par.context_at_level(d.level)?
            .hip_handle()?
            .ciphertexts_switch_to_level_dev(target_level - d.level, d, stream)
            .map_err(crate::hip_error)
    }

    /// Downloads a device-resident batch (waits for `stream`) into ciphertexts over `par`.
    #[cfg(feature = "hip")]
    pub fn from_device(par: &Arc<BfvParameters>, d: &fhe_math_hip::DeviceCiphertexts, stream: &fhe_math_hip::Stream) -> Result<Vec<Ciphertext>> {
        let flat = d.download(stream).map_err(crate::hip_error)?;
        flat.chunks_exact(d.words_per_ct()).map(|one| Self::from_ntt_coefficients(par, one, d.parts, d.level)).collect()
    }

"""),
        ("self.seed = None;\n        for ci in self.c.iter_mut() {", "after", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // all parts at once: inverse NTT, divide-and-round by the last modulus, forward NTT (fhe_bfv_switch_to_level)
            let h = self.c[0].ctx().hip_handle()?;
            let rows = self.c[0].ctx().moduli().len() - 1;
            let mut out = vec![0u64; self.c.len() * rows * self.par.degree()];
            h.ciphertext_switch_down(self.c.len(), &self.to_flat_coefficients(), &mut out).map_err(crate::hip_error)?;
            *self = Ciphertext::from_ntt_coefficients(&self.par, &out, self.c.len(), self.level + 1)?;
            return Ok(());
        }
"""),
    ]),
    ("14-bfv-relinearization-key", "crates/fhe/src/bfv/keys/relinearization_key.rs", [
        ("/// Relinearize using polynomials.", "before", """    /// [`RelinearizationKey::relinearizes`] on a device-resident batch of three-part ciphertexts
    /// (`Ciphertext::to_device`, `Multiplicator::multiply_dev` without a key): stream-ordered, nothing crosses PCIe;
    /// the two-part result stays on the GPU.  Same checks, same errors, same values as `relinearizes`.
    #[cfg(feature = "hip")]
    pub fn relinearizes_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        if ct.parts != 3 {
            return Err(crate::CiphertextError::InvalidPolynomialCount {
                operation: crate::CiphertextOperation::Relinearization,
                actual: ct.parts,
                expected: 3,
            }
            .into());
        }
        if ct.level != self.ksk.ciphertext_level {
            return Err(Error::InvalidLevel {
                level: ct.level,
                min_level: self.ksk.ciphertext_level,
                max_level: self.ksk.ciphertext_level,
            });
        }
        self.ksk.hip_handle()?.relinearize_dev(ct, stream).map_err(crate::hip_error)
    }

"""),
    ]),
    ("15-bfv-galois-key", "crates/fhe/src/bfv/keys/galois_key.rs", [
        ("/// Relinearize a [`Ciphertext`] writing the result into `out`.", "before", """    /// [`GaloisKey::relinearize`] on a device-resident batch of two-part ciphertexts: substitution, key switch and
    /// the level fix-up in one stream-ordered call (`fhe_bfv_galois_dev`); the result stays on the GPU.
    #[cfg(feature = "hip")]
    pub fn relinearize_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        if ct.parts != 2 {
            return Err(crate::CiphertextError::InvalidPolynomialCount {
                operation: crate::CiphertextOperation::Galois,
                actual: ct.parts,
                expected: 2,
            }
            .into());
        }
        if ct.level != self.ksk.ciphertext_level {
            return Err(Error::InvalidLevel {
                level: ct.level,
                min_level: self.ksk.ciphertext_level,
                max_level: self.ksk.ciphertext_level,
            });
        }
        self.ksk.hip_handle()?.galois_dev(self.element.exponent, ct, stream).map_err(crate::hip_error)
    }

"""),
    ]),
    ("16-bfv-evaluation-key", "crates/fhe/src/bfv/keys/evaluation_key.rs", [
        ("fn validate_ciphertext(&self, ct: &Ciphertext) -> Result<()> {", "before", """    #[cfg(feature = "hip")]
    fn validate_device_ciphertexts(&self, ct: &fhe_math_hip::DeviceCiphertexts) -> Result<()> {
        if ct.parts != 2 {
            return Err(crate::CiphertextError::InvalidPolynomialCount {
                operation: crate::CiphertextOperation::EvaluationKey,
                actual: ct.parts,
                expected: 2,
            }
            .into());
        }
        if ct.level != self.ciphertext_level {
            return Err(Error::InvalidLevel {
                level: ct.level,
                min_level: self.ciphertext_level,
                max_level: self.ciphertext_level,
            });
        }
        Ok(())
    }

    #[cfg(feature = "hip")]
    fn galois_key_for(&self, element: usize) -> Result<&GaloisKey> {
        self.gk.get(&element).ok_or_else(|| {
            crate::EvaluationKeyError::Missing { component: crate::EvaluationKeyComponent::GaloisKey { element } }.into()
        })
    }

    /// [`EvaluationKey::rotates_rows`] on a device-resident batch (the exponent lookup stays here; the Galois key
    /// switch runs on the GPU, stream-ordered, and its result stays there).
    #[cfg(feature = "hip")]
    pub fn rotates_rows_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        self.validate_device_ciphertexts(ct)?;
        if !self.supports_row_rotation() {
            return Err(crate::EvaluationKeyError::Unsupported { operation: crate::EvaluationOperation::RowRotation }.into());
        }
        self.galois_key_for(self.par.degree() * 2 - 1)?.relinearize_dev(ct, stream)
    }

    /// [`EvaluationKey::rotates_columns_by`] on a device-resident batch.
    #[cfg(feature = "hip")]
    pub fn rotates_columns_by_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, i: usize, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        self.validate_device_ciphertexts(ct)?;
        if !self.supports_column_rotation_by(i) {
            return Err(crate::EvaluationKeyError::Unsupported { operation: crate::EvaluationOperation::ColumnRotation { step: i } }.into());
        }
        let exponent = *self.rot_to_gk_exponent.get(&i).ok_or_else(|| crate::EvaluationKeyError::InvalidRotationStep {
            step: i,
            min: 1,
            max: self.par.degree() / 2 - 1,
        })?;
        self.galois_key_for(exponent)?.relinearize_dev(ct, stream)
    }

    /// [`EvaluationKey::computes_inner_sum`] on a device-resident batch: the reference's sequence of Galois keys
    /// (3^(2^j) mod 2N for j = 0 .. log2(N/2) - 1, then 2N - 1) handed to one engine call (`fhe_bfv_inner_sum_dev`).
    #[cfg(feature = "hip")]
    pub fn computes_inner_sum_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        self.validate_device_ciphertexts(ct)?;
        if !self.supports_inner_sum() {
            return Err(crate::EvaluationKeyError::Unsupported { operation: crate::EvaluationOperation::InnerSum }.into());
        }
        let mut elements = Vec::new();
        let mut i = 1;
        while i < self.par.degree() / 2 {
            elements.push(*self.rot_to_gk_exponent.get(&i).ok_or(crate::EvaluationKeyError::Missing {
                component: crate::EvaluationKeyComponent::GaloisExponent { step: i },
            })?);
            i *= 2
        }
        elements.push(self.par.degree() * 2 - 1);
        let mut keys = Vec::with_capacity(elements.len());
        for e in elements.iter() {
            keys.push(self.galois_key_for(*e)?.ksk.hip_handle()?.clone());
        }
        fhe_math_hip::HipKsk::inner_sum_dev(&keys, &elements, ct, stream).map_err(crate::hip_error)
    }

    /// [`EvaluationKey::expands`] on a device-resident batch.  The result holds `size * ct.batch` ciphertexts laid out
    /// `[size][batch]` (output `k` of input `b` is ciphertext `k * ct.batch + b`); the expansion monomials are derived
    /// on the device from the context's tables.
    #[cfg(feature = "hip")]
    pub fn expands_dev(&self, ct: &fhe_math_hip::DeviceCiphertexts, size: usize, stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        self.validate_device_ciphertexts(ct)?;
        if size == 0 || size > self.par.degree() {
            return Err(crate::EvaluationKeyError::InvalidExpansionSize { size, degree: self.par.degree() }.into());
        }
        let level = size.next_power_of_two().ilog2() as usize;
        if !self.supports_expansion(level) {
            return Err(crate::EvaluationKeyError::Unsupported { operation: crate::EvaluationOperation::Expansion { level } }.into());
        }
        let mut keys = Vec::with_capacity(level);
        for l in 0..level {
            keys.push(self.galois_key_for((self.par.degree() >> l) + 1)?.ksk.hip_handle()?.clone());
        }
        fhe_math_hip::HipKsk::expand_dev(&keys, size, ct, stream).map_err(crate::hip_error)
    }

"""),
    ]),
    ("08-bfv-key-switch", "crates/fhe/src/bfv/keys/key_switching_key.rs", [
        ("pub(crate) log_base: usize,", "after", """
    /// Device twin: the key polynomials uploaded on first use (`fhe_ksk_create`).
    #[cfg(feature = "hip")]
    pub(crate) hip: fhe_math_hip::LazyHandle<fhe_math_hip::HipKsk>,
"""),
        ("impl KeySwitchingKey {", "after", """    /// The device twin of this key: `c0`, `c1` as `[digits][Lk][N]` (their Shoup twins are recomputed by the engine
    /// as floor(c * 2^64 / q), the definition `Poly::<NttShoup>` uses, zq/mod.rs:195-199).
    #[cfg(feature = "hip")]
    pub fn hip_handle(&self) -> Result<&Arc<fhe_math_hip::HipKsk>> {
        self.hip.get_or_try_init(|| {
            let flat = |ps: &[Poly<NttShoup>]| ps.iter().flat_map(|p| p.coefficients().iter().copied().collect::<Vec<u64>>()).collect::<Vec<u64>>();
            let (ct, ksk) = (self.ctx_ciphertext.hip_handle()?, self.ctx_ksk.hip_handle()?);
            fhe_math_hip::HipKsk::new(ct, ksk, self.c0.len(), &flat(&self.c0), None, &flat(&self.c1), None, self.log_base)
                .map_err(crate::hip_error)
        })
    }

"""),
        ("log_base,\n            })", "after", """                #[cfg(feature = "hip")]
                hip: Default::default(),
"""),
        ("log_base: 0,", "after", """                #[cfg(feature = "hip")]
                hip: Default::default(),
"""),
        ("log_base: value.log_base as usize,", "after", """            #[cfg(feature = "hip")]
            hip: Default::default(),
"""),
        ("let mut c0 = Poly::<Ntt>::zero(&self.ctx_ksk);\n        let mut c1 = Poly::<Ntt>::zero(&self.ctx_ksk);\n        self.configure_accumulators(p, &mut c0, &mut c1);\n        let p_coefficients = p.coefficients();", "before", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // lazy lift + NTT per (digit, key modulus) + Shoup MAC, fused on the device (fhe_key_switch)
            let n = self.ctx_ksk.moduli().len() * self.par.degree();
            let (mut o0, mut o1) = (vec![0u64; n], vec![0u64; n]);
            self.hip_handle()?
                .key_switch(p.coefficients().as_slice().unwrap(), &mut o0, &mut o1)
                .map_err(crate::hip_error)?;
            let vt = self.permits_variable_time_with(p);
            return Ok((
                Poly::<Ntt>::try_convert_from(o0, &self.ctx_ksk, vt)?,
                Poly::<Ntt>::try_convert_from(o1, &self.ctx_ksk, vt)?,
            ));
        }
"""),
    ]),
    ("07-bfv-multiplicator", "crates/fhe/src/bfv/ops/mul.rs", [
        ("level: usize,\n}", "after", """    /// Device twin: extenders, down scaler, relinearisation key and the mod-switch flag (`fhe_mul_create`), built on
    /// first use and rebuilt after `enable_relinearization` / `enable_mod_switching`.
    #[cfg(feature = "hip")]
    hip: fhe_math_hip::LazyHandle<fhe_math_hip::HipMul>,
"""),
        ("impl Multiplicator {", "after", """    #[cfg(feature = "hip")]
    fn hip_handle(&self) -> Result<&Arc<fhe_math_hip::HipMul>> {
        self.hip.get_or_try_init(|| {
            let rk = match self.rk.as_ref() {
                Some(rk) => Some(rk.ksk.hip_handle()?),
                None => None,
            };
            fhe_math_hip::HipMul::new(self.extender_lhs.hip_handle()?, self.extender_rhs.hip_handle()?,
                self.down_scaler.hip_handle()?, rk, self.mod_switch)
                .map_err(crate::hip_error)
        })
    }

    #[cfg(feature = "hip")]
    fn check_operands(&self, lhs: &Ciphertext, rhs: &Ciphertext) -> Result<()> {
        lhs.validate_for(&self.par)?;
        rhs.validate_for(&self.par)?;
        if lhs.level != self.level || rhs.level != self.level {
            let level = if lhs.level != self.level { lhs.level } else { rhs.level };
            return Err(Error::InvalidLevel { level, min_level: self.level, max_level: self.level });
        }
        if lhs.len() != 2 || rhs.len() != 2 {
            return Err(crate::CiphertextError::MultiplicationPolynomialCount { left: lhs.len(), right: rhs.len(), expected: 2 }.into());
        }
        Ok(())
    }

    /// `multiply` on many independent pairs in ONE device call (the PCIe copies and the launch chain are paid once
    /// per batch, not once per pair): what a throughput-bound host uses.  Same values as `multiply` pair by pair.
    #[cfg(feature = "hip")]
    pub fn multiply_batch(&self, lhs: &[Ciphertext], rhs: &[Ciphertext]) -> Result<Vec<Ciphertext>> {
        if lhs.len() != rhs.len() {
            return Err(crate::CiphertextError::MultiplicationPolynomialCount { left: lhs.len(), right: rhs.len(), expected: 2 }.into());
        }
        for (l, r) in lhs.iter().zip(rhs.iter()) {
            self.check_operands(l, r)?;
        }
        let h = self.hip_handle()?;
        let (parts, rows) = h.out_shape();
        let flat = |cts: &[Ciphertext]| cts.iter().flat_map(|c| c.to_flat_coefficients()).collect::<Vec<u64>>();
        let per = parts * rows * self.par.degree();
        let mut out = vec![0u64; lhs.len() * per];
        h.multiply(&flat(lhs), &flat(rhs), &mut out).map_err(crate::hip_error)?;
        let level = self.level + usize::from(self.mod_switch);
        out.chunks_exact(per).map(|one| Ciphertext::from_ntt_coefficients(&self.par, one, parts, level)).collect()
    }

    /// `multiply` on device-resident batches (`Ciphertext::to_device`): stream-ordered, nothing crosses PCIe.
    #[cfg(feature = "hip")]
    pub fn multiply_dev(&self, lhs: &fhe_math_hip::DeviceCiphertexts, rhs: &fhe_math_hip::DeviceCiphertexts,
                        stream: &fhe_math_hip::Stream) -> Result<fhe_math_hip::DeviceCiphertexts> {
        if lhs.level != self.level || rhs.level != self.level {
            let level = if lhs.level != self.level { lhs.level } else { rhs.level };
            return Err(Error::InvalidLevel { level, min_level: self.level, max_level: self.level });
        }
        self.hip_handle()?.multiply_dev(lhs, rhs, stream).map_err(crate::hip_error)
    }

"""),
        ("mod_switch: false,\n            level,", "after_second", """            #[cfg(feature = "hip")]
            hip: Default::default(),
"""),
        ("self.rk = Some(rk.clone());", "after", """        #[cfg(feature = "hip")]
        self.hip.reset();
"""),
        ("self.mod_switch = true;", "after", """            #[cfg(feature = "hip")]
            self.hip.reset();
"""),
        ("// Extend", "before", """        #[cfg(feature = "hip")]
        if fhe_math_hip::enabled() {
            // extend, tensor, scale, relinearise (and switch down) of one ciphertext pair on the device; the batched
            // form (`multiply_batch`) amortises the PCIe copies and `multiply_dev` removes them
            let h = self.hip_handle()?;
            let (parts, rows) = h.out_shape();
            let mut out = vec![0u64; parts * rows * self.par.degree()];
            h.multiply(&lhs.to_flat_coefficients(), &rhs.to_flat_coefficients(), &mut out).map_err(crate::hip_error)?;
            return Ciphertext::from_ntt_coefficients(&self.par, &out, parts, self.level + usize::from(self.mod_switch));
        }

"""),
    ]),
]


# New files (patches 17 / 18, round 6): integration tests that run the native path and the engine in ONE process
# (fhe_math_hip::with_native) and compare them bit for bit -- public API only, behind the `hip` feature.
FHE_MATH_PARITY_TEST = r'''//! Native-versus-engine parity of `fhe-math`, inside ONE process: every value is computed twice -- once through this
//! crate's own CPU code (`fhe_math_hip::with_native`), once through the `hip` feature's forwarding to libfhe_hip.so --
//! and compared bit for bit.  This is the test that pins, against fhe.rs itself, what no other check in the engine's
//! repository can: the NTT tables (psi is drawn from a seeded ChaCha8 stream here and re-derived by the engine when a
//! context is built WITHOUT host tables), the seeded sampler behind `Poly::random_from_seed`, and the RNS scaler.
//!
//!     cargo test -p fhe-math --features hip --test hip_parity -- --test-threads 1
//!
//! Needs an AMD GPU and libfhe_hip.so (FHE_HIP_LIB_DIR).  Skips (passes vacuously, saying so) when no device is visible.
#![cfg(feature = "hip")]

use fhe_math::rns::ScalingFactor;
use fhe_math::rq::{scaler::Scaler, Context, Ntt, Poly, PowerBasis, SubstitutionExponent};
use num_bigint::BigUint;
use rand::rng;
use std::sync::Arc;

/// The moduli of `BfvParameters::default_parameters_128(20)` (36 ... 49 bits: the engine's FP64 kernels) and a
/// 62-bit basis (its integer kernels).
const BASES: &[(usize, &[u64])] = &[
    (4096, &[0xffffee001, 0xffffc4001, 0x1ffffe0001]),
    (8192, &[0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001]),
    (16384, &[0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
              0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001]),
    (8192, &[4611686018326724609, 4611686018309947393, 4611686018282684417, 4611686018257518593]),
];

fn device_present() -> bool {
    if fhe_math_hip::enabled() {
        return true;
    }
    eprintln!("hip_parity: no HIP device visible (or FHE_HIP_DISABLE set) -- nothing compared");
    false
}

/// `f` on the native path and on the engine, with the FP64 kernels on and off: three results that must be equal.
fn both<T: PartialEq + std::fmt::Debug>(what: &str, f: impl Fn() -> T) {
    let native = fhe_math_hip::with_native(&f);
    for f64_kernels in [true, false] {
        fhe_math_hip::set_f64_kernels(f64_kernels);
        assert!(!fhe_math_hip::native_forced());
        assert_eq!(f(), native, "{what}: engine (f64 kernels: {f64_kernels}) differs from the native path");
    }
    fhe_math_hip::set_f64_kernels(true);
}

#[test]
fn ntt_round_trips_and_psi() {
    if !device_present() {
        return;
    }
    for (degree, moduli) in BASES {
        let ctx = Context::new_arc(moduli, *degree).unwrap();
        let p = Poly::<PowerBasis>::random(&ctx, &mut rng());
        // forward transform: the engine uses THIS context's tables (uploaded by `hip_handle`), so equality here pins the
        // butterflies; the table-free context below pins psi itself
        both("PowerBasis -> Ntt", || p.clone().into_ntt());
        let q = p.clone().into_ntt();
        both("Ntt -> PowerBasis", || q.clone().into_power_basis());
        // psi: a device context built WITHOUT host tables derives its own psi (the engine's restatement of
        // NttOperator::new); its forward transform must equal this crate's
        let own = fhe_math_hip::HipCtx::new(fhe_math_hip::default_device(), *degree, moduli, None).unwrap();
        let mut words = p.coefficients().as_slice().unwrap().to_vec();
        own.ntt_forward(&mut words).unwrap();
        let native = fhe_math_hip::with_native(|| p.clone().into_ntt());
        assert_eq!(words.as_slice(), native.coefficients().as_slice().unwrap(), "psi / table derivation, degree {degree}");
    }
}

#[test]
fn element_wise_operators() {
    if !device_present() {
        return;
    }
    for (degree, moduli) in BASES {
        let ctx = Context::new_arc(moduli, *degree).unwrap();
        let a = Poly::<Ntt>::random(&ctx, &mut rng());
        let b = Poly::<Ntt>::random(&ctx, &mut rng());
        both("+=", || { let mut x = a.clone(); x += &b; x });
        both("-=", || { let mut x = a.clone(); x -= &b; x });
        both("*=", || { let mut x = a.clone(); x *= &b; x });
        both("neg", || -&a);
    }
}

#[test]
fn substitute_and_switch_down() {
    if !device_present() {
        return;
    }
    for (degree, moduli) in BASES {
        let ctx = Context::new_arc(moduli, *degree).unwrap();
        let p = Poly::<PowerBasis>::random(&ctx, &mut rng());
        let q = p.clone().into_ntt();
        for e in [3usize, 2 * degree - 1, degree + 1] {
            let exp = SubstitutionExponent::new(&ctx, e).unwrap();
            both("substitute (PowerBasis)", || p.substitute(&exp).unwrap());
            both("substitute (Ntt)", || q.substitute(&exp).unwrap());
        }
        both("switch_down", || { let mut x = p.clone(); x.switch_down().unwrap(); x });
    }
}

#[test]
fn scaler_scale() {
    if !device_present() {
        return;
    }
    for (degree, moduli) in BASES {
        let from = Context::new_arc(moduli, *degree).unwrap();
        let to = Context::new_arc(&moduli[..moduli.len() - 1], *degree).unwrap();
        let p = Poly::<PowerBasis>::random(&from, &mut rng());
        for (num, den) in [(1u64, 1u64), (1153, 46116860181065), (3, 7)] {
            let scaler = Scaler::new(&from, &to, ScalingFactor::new(&BigUint::from(num), &BigUint::from(den))).unwrap();
            both("Scaler::scale (PowerBasis)", || p.scale(&scaler).unwrap());
            let q = p.clone().into_ntt();
            both("Scaler::scale (Ntt)", || q.scale(&scaler).unwrap());
        }
    }
}

#[test]
fn random_from_seed_is_the_reference_sampler() {
    // The engine expands seeded polynomials on the device (SHA-256 -> ChaCha8 -> rejection sampling); this compares its
    // stream with rand_chacha's, i.e. with `Poly::random_from_seed` on the native path.
    if !device_present() {
        return;
    }
    for (degree, moduli) in BASES {
        let ctx = Context::new_arc(moduli, *degree).unwrap();
        for s in 0u8..4 {
            let seed = [s.wrapping_mul(37).wrapping_add(1); 32];
            both("random_from_seed", || Poly::<Ntt>::random_from_seed(&ctx, seed));
        }
    }
    let _ = Arc::new(());
}
'''

FHE_PARITY_TEST = r'''//! Native-versus-engine parity of `fhe` (BFV), inside ONE process: every hot-path Criterion ID of benches/bfv.rs
//! (relinearize, rotate_rows, rotate_columns, inner_sum, expand_*, mul, square, mul_then_relinearize, mul_and_relin,
//! mul_and_relin_2), `Ciphertext::switch_down` / `switch_to_level`, on `default_parameters_128(20)` and on a six-modulus
//! 62-bit set -- computed once through this crate's own CPU code (`fhe_math_hip::with_native`) and once through the `hip`
//! feature's forwarding to libfhe_hip.so (FP64 kernels on and off), compared bit for bit.
//!
//!     cargo test -p fhe --features hip --test hip_parity -- --test-threads 1
//!
//! Needs an AMD GPU and libfhe_hip.so (FHE_HIP_LIB_DIR).  Skips (passes vacuously, saying so) when no device is visible.
#![cfg(feature = "hip")]

use fhe::bfv::{
    BfvParameters, BfvParametersBuilder, Ciphertext, Encoding, EvaluationKeyBuilder, Multiplicator, Plaintext,
    RelinearizationKey, SecretKey,
};
use fhe_math::rns::{RnsContext, ScalingFactor};
use fhe_math::zq::primes::generate_prime;
use fhe_traits::{FheEncoder, FheEncrypter};
use num_bigint::BigUint;
use rand::rng;
use std::sync::Arc;

fn device_present() -> bool {
    if fhe_math_hip::enabled() {
        return true;
    }
    eprintln!("hip_parity: no HIP device visible (or FHE_HIP_DISABLE set) -- nothing compared");
    false
}

fn parameter_sets() -> Vec<Arc<BfvParameters>> {
    let mut sets: Vec<Arc<BfvParameters>> = BfvParameters::default_parameters_128(20).unwrap().collect();
    sets.retain(|p| p.moduli().len() > 1 && p.degree() <= 16384);
    sets.push(BfvParametersBuilder::new().set_degree(16).set_plaintext_modulus(1153).set_moduli_sizes(&[62usize; 6]).build_arc().unwrap());
    sets
}

/// `f` on the native path and on the engine, with the FP64 kernels on and off: three results that must be equal.
fn both<T: PartialEq + std::fmt::Debug>(what: &str, f: impl Fn() -> T) -> T {
    let native = fhe_math_hip::with_native(&f);
    for f64_kernels in [true, false] {
        fhe_math_hip::set_f64_kernels(f64_kernels);
        assert_eq!(f(), native, "{what}: engine (f64 kernels: {f64_kernels}) differs from the native path");
    }
    fhe_math_hip::set_f64_kernels(true);
    native
}

#[test]
fn every_bench_id_matches_the_native_path() {
    if !device_present() {
        return;
    }
    let mut rng = rng();
    for par in parameter_sets() {
        let tag = format!("n={}/moduli={}", par.degree(), par.moduli().len());
        // keys and inputs are made ONCE (on the native path: key generation is out of the engine's scope)
        let (sk, rk, ek, c1, c2) = fhe_math_hip::with_native(|| {
            let sk = SecretKey::random(&par, &mut rng);
            let rk = RelinearizationKey::new(&sk, &mut rng).unwrap();
            let ek = EvaluationKeyBuilder::new(&sk).unwrap()
                .enable_inner_sum().unwrap()
                .enable_row_rotation().unwrap()
                .enable_column_rotation(1).unwrap()
                .enable_expansion(4usize.min(par.degree().ilog2() as usize)).unwrap()
                .build(&mut rng).unwrap();
            let pt1 = Plaintext::try_encode(&(1..16u64).collect::<Vec<u64>>(), Encoding::simd(), &par).unwrap();
            let pt2 = Plaintext::try_encode(&(3..39u64).map(|v| v % 16).collect::<Vec<u64>>(), Encoding::simd(), &par).unwrap();
            let c1: Ciphertext = sk.try_encrypt(&pt1, &mut rng).unwrap();
            let c2: Ciphertext = sk.try_encrypt(&pt2, &mut rng).unwrap();
            (sk, rk, ek, c1, c2)
        });
        let _ = &sk;
        both(&format!("{tag} add_ct"), || &c1 + &c2);
        both(&format!("{tag} sub_ct"), || &c1 - &c2);
        both(&format!("{tag} neg"), || -&c1);
        let c3 = both(&format!("{tag} mul"), || &c1 * &c2);
        both(&format!("{tag} square"), || &c1 * &c1);
        both(&format!("{tag} relinearize"), || { let mut x = c3.clone(); rk.relinearizes(&mut x).unwrap(); x });
        both(&format!("{tag} mul_then_relinearize"), || { let mut x = &c1 * &c2; rk.relinearizes(&mut x).unwrap(); x });
        both(&format!("{tag} rotate_rows"), || ek.rotates_rows(&c1).unwrap());
        both(&format!("{tag} rotate_columns"), || ek.rotates_columns_by(&c1, 1).unwrap());
        both(&format!("{tag} inner_sum"), || ek.computes_inner_sum(&c1).unwrap());
        for i in 1..=4usize.min(par.degree().ilog2() as usize) {
            both(&format!("{tag} expand_{i}"), || ek.expands(&c1, 1 << i).unwrap());
        }
        let multiplicator = Multiplicator::default(&rk).unwrap();
        both(&format!("{tag} mul_and_relin"), || multiplicator.multiply(&c1, &c2).unwrap());
        // benches/bfv.rs:257-286: the second strategy
        let q = par.moduli_sizes().iter().sum::<usize>();
        let nmoduli = q.div_ceil(62);
        let mut extended_basis = par.moduli().to_vec();
        let mut upper_bound = u64::MAX >> 2;
        while extended_basis.len() != nmoduli + par.moduli().len() {
            upper_bound = generate_prime(62, 2 * par.degree() as u64, upper_bound).unwrap();
            if !extended_basis.contains(&upper_bound) {
                extended_basis.push(upper_bound)
            }
        }
        let rns_q = RnsContext::new(&extended_basis[..par.moduli().len()]).unwrap();
        let rns_p = RnsContext::new(&extended_basis[par.moduli().len()..]).unwrap();
        let mut second = Multiplicator::new(
            ScalingFactor::one(),
            ScalingFactor::new(rns_p.modulus(), rns_q.modulus()),
            &extended_basis,
            ScalingFactor::new(&BigUint::from(par.plaintext()), rns_p.modulus()),
            &par,
        ).unwrap();
        second.enable_relinearization(&rk).unwrap();
        both(&format!("{tag} mul_and_relin_2"), || second.multiply(&c1, &c2).unwrap());
        // modulus switching (Ciphertext::switch_down, switch_to_level) and the multiplicator that ends with it
        both(&format!("{tag} switch_down"), || { let mut x = c1.clone(); x.switch_down().unwrap(); x });
        both(&format!("{tag} switch_to_level"), || { let mut x = c1.clone(); x.switch_to_level(x.max_switchable_level()).unwrap(); x });
        let mut switching = Multiplicator::default(&rk).unwrap();
        switching.enable_mod_switching().unwrap();
        both(&format!("{tag} mul_and_relin + mod switch"), || switching.multiply(&c1, &c2).unwrap());
    }
}
'''

NEW_FILES = [
    ("17-fhe-math-hip-parity-tests", "crates/fhe-math/tests/hip_parity.rs", FHE_MATH_PARITY_TEST),
    ("18-fhe-hip-parity-tests", "crates/fhe/tests/hip_parity.rs", FHE_PARITY_TEST),
]


def apply(text, edits, path):
    lines = text.split("\n")
    for edit in edits:
        anchor, where, ins = edit[:3]
        occ = edit[3] if len(edit) > 3 else None     # which match of an anchor that occurs several times (0-based,
        first = anchor.split("\n")[0].strip()        # counted in the file as the EARLIER edits of this list left it)
        idx = [i for i, l in enumerate(lines) if l.strip() == first]
        if "\n" in anchor:   # multi-line anchor: the following lines must match too
            rest = [a.strip() for a in anchor.split("\n")[1:]]
            idx = [i for i in idx if [l.strip() for l in lines[i + 1:i + 1 + len(rest)]] == rest]
        if not idx:
            raise SystemExit(f"{path}: anchor not found: {first!r}")
        if len(idx) > 1 and "\n" not in anchor and occ is None:
            raise SystemExit(f"{path}: anchor is ambiguous ({len(idx)} matches): {first!r}")
        if occ is not None and occ >= len(idx):
            raise SystemExit(f"{path}: anchor has {len(idx)} matches, occurrence {occ} asked for: {first!r}")
        i = idx[occ or 0]
        new = ins.rstrip("\n").split("\n")
        if where == "after":
            lines[i + 1:i + 1] = new
        elif where == "after_second":   # after the second line of a multi-line anchor
            lines[i + 2:i + 2] = new
        elif where == "after_third":
            lines[i + 3:i + 3] = new
        elif where == "before":
            lines[i:i] = new
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
    for name, rel, content in NEW_FILES:
        if os.path.exists(os.path.join(REF, rel)):
            raise SystemExit(f"{rel} exists in the reference: a new-file patch would overwrite it")
        body = content.rstrip("\n").split("\n")
        diff = ["--- /dev/null", "+++ b/" + rel, "@@ -0,0 +1,%d @@" % len(body)] + ["+" + l for l in body]
        open(os.path.join(OUT, name + ".patch"), "w").write("\n".join(diff) + "\n")
        print(name, "ok (new file)")


if __name__ == "__main__":
    main()
