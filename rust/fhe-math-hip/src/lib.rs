//! Safe handles over the C ABI of the MI355X engine (`include/fhe_hip.h`, `libfhe_hip.so`).
//!
//! One RAII type per opaque C handle, named after the reference type it shadows:
//!
//! | here               | C handle / object        | reference type (fhe.rs)                                   |
//! |--------------------|--------------------------|-----------------------------------------------------------|
//! | [`HipCtx`]         | `fhe_ctx`                | `fhe_math::rq::Context`            (rq/context.rs:9-19)    |
//! | [`HipScaler`]      | `fhe_scaler`             | `fhe_math::rq::scaler::Scaler`     (rq/scaler.rs:18-23)    |
//! | [`HipKsk`]         | `fhe_ksk`                | `fhe::bfv::KeySwitchingKey`        (keys/key_switching_key.rs:22-46) |
//! | [`HipMul`]         | `fhe_mul`                | `fhe::bfv::Multiplicator`          (ops/mul.rs:21-32)      |
//! | [`HipParams`]      | `fhe_params`             | level tables of `BfvParameters`    (parameters.rs:83-117)  |
//! | [`Stream`]         | `hipStream_t` (ABI-made) | --                                                         |
//! | [`DeviceBuffer`]   | `fhe_buf_alloc` memory   | the device shadow of `Poly.coefficients: Array2<u64>`      |
//! | [`DeviceCiphertexts`] | a `DeviceBuffer` + shape | a batch of `bfv::Ciphertext` kept on the GPU between calls |
//!
//! Two ways to call the engine:
//! * **host slices** -- `Poly`'s own `[L][N]` row-major `u64` slices (`coefficients.as_slice()`), batches are
//!   concatenations; synchronous, H2D + compute + D2H inside every call (the drop-in path of the call-site patches);
//! * **device resident** -- [`DeviceCiphertexts`] on a [`Stream`]: upload once, chain `multiply` -> `relinearize` ->
//!   `rotate` -> `switch_to_level` without leaving the GPU, download lazily.  This is the path the throughput numbers
//!   are quoted on.
//!
//! Safety contract of the safe API: every slice length is checked against the handle's own geometry before the
//! call (the C side trusts `batch`); borrowed level contexts carry a lifetime ([`CtxView`]); handles that the C side
//! keeps pointers to (`fhe_scaler` -> its two contexts, `fhe_ksk` -> its two contexts, `fhe_mul` -> scalers and key)
//! are kept alive by `Arc`s inside the dependent wrapper.  Handles are immutable after creation: `Send + Sync`.
//! `ffi.rs` is generated from the header (`tools/gen_rust_ffi.py`); `tests/test_rust_shim.py` keeps the two equal and
//! checks that every identifier the call-site patches use is defined here or by another patch.
//! (No Rust toolchain exists in the build image: this crate is reviewed source, not compiled there.)
pub mod ffi;

use std::ffi::CStr;
use std::marker::PhantomData;
use std::ops::Deref;
use std::os::raw::{c_int, c_void};
use std::ptr;
use std::sync::{Arc, OnceLock};

/// Non-zero `fhe_status` with the engine's thread-local message.  The codes map 1:1 onto the `Result` variants
/// of the path (table in `fhe_hip.h`); `fhe-math` / `fhe` translate them back (`hip_error` in each crate,
/// rust/patches/10-*.patch and 13-*.patch).
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct HipError {
    pub status: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "fhe_hip status {}: {}", self.status, self.message)
    }
}
impl std::error::Error for HipError {}
pub type Result<T> = std::result::Result<T, HipError>;

pub mod status {
    pub const ARG: i32 = -1;
    pub const HIP: i32 = -2;
    pub const INVALID_MODULUS: i32 = -3;
    pub const INVALID_DEGREE: i32 = -4;
    pub const NTT_UNAVAILABLE: i32 = -5;
    pub const CONTEXT_MISMATCH: i32 = -6;
    pub const DEGREE_MISMATCH: i32 = -7;
    pub const NO_MORE_CONTEXT: i32 = -8;
    pub const CONTEXT_NOT_REACHABLE: i32 = -9;
    pub const INVALID_SUBSTITUTION_EXPONENT: i32 = -10;
    pub const PARAMETER_MISMATCH: i32 = -11;
    pub const INVALID_LEVEL: i32 = -12;
    pub const MUL_POLY_COUNT: i32 = -13;
    pub const EMPTY_MODULI: i32 = -14;
    pub const NON_COPRIME: i32 = -15;
    pub const NOT_ENOUGH_PRIMES: i32 = -16;
    pub const KEYSWITCH_UNSUPPORTED: i32 = -17;
    pub const NO_DEVICE: i32 = -18;
    pub const EMPTY_DOT_PRODUCT: i32 = -19;
    pub const INVALID_EXPANSION_SIZE: i32 = -20;
    pub const EXPANSION_UNSUPPORTED: i32 = -21;
}

pub fn check(status: ffi::FheStatus) -> Result<()> {
    if status == 0 {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(ffi::fhe_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { status, message })
}
/// A shape the caller got wrong on the Rust side (never reaches the C ABI).
fn shape_error(what: &str) -> HipError {
    HipError { status: status::ARG, message: format!("fhe-math-hip: {what}") }
}
fn expect_len(what: &str, got: usize, want: usize) -> Result<()> {
    if got == want { Ok(()) } else { Err(shape_error(&format!("{what}: {got} u64 words, expected {want}"))) }
}
/// `len / per` when `len` is a whole number of `per`-word items.
fn whole_batch(what: &str, len: usize, per: usize) -> Result<usize> {
    if per == 0 || len % per != 0 {
        return Err(shape_error(&format!("{what}: {len} u64 words is not a whole number of {per}-word items")));
    }
    Ok(len / per)
}
/// The device the patched crates put their handles on: `FHE_HIP_DEVICE` (one process per GPU when a batch is
/// sharded across the GPUs of a node), default 0.
pub fn default_device() -> i32 {
    std::env::var("FHE_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}

thread_local! {
    /// Depth of `with_native` scopes on this thread (see `enabled`).
    static FORCE_NATIVE: std::cell::Cell<u32> = const { std::cell::Cell::new(0) };
}

/// Runs `f` with the patched crates on their NATIVE (CPU) path on the calling thread, whatever the process-wide
/// switch says, and restores the previous state afterwards (also when `f` panics).  This is what lets ONE test
/// process compute a value twice -- once through fhe.rs's own code, once through the engine -- and compare the two
/// bit for bit (`crates/*/tests/hip_parity.rs`, patches 17 / 18): the only way psi, the seeded sampler and every
/// composite operation can be pinned against the reference itself.  Nests; affects this thread only.
pub fn with_native<R>(f: impl FnOnce() -> R) -> R {
    struct Restore;
    impl Drop for Restore {
        fn drop(&mut self) { FORCE_NATIVE.with(|c| c.set(c.get() - 1)); }
    }
    FORCE_NATIVE.with(|c| c.set(c.get() + 1));
    let _restore = Restore;
    f()
}

/// Is the calling thread inside a `with_native` scope?
pub fn native_forced() -> bool { FORCE_NATIVE.with(|c| c.get() > 0) }

/// Whether the patched crates forward to the engine at all: the `hip` cargo feature compiles the forwarding in, this
/// decides at run time.  False when `FHE_HIP_DISABLE` is set or no HIP device is visible (both read once per process),
/// so a binary built with the feature still runs -- on the native CPU path -- on a machine without a GPU; and false
/// inside `with_native` on the calling thread.
pub fn enabled() -> bool {
    static ON: OnceLock<bool> = OnceLock::new();
    !native_forced() && *ON.get_or_init(|| std::env::var_os("FHE_HIP_DISABLE").is_none() && unsafe { ffi::fhe_device_count() } > 0)
}

/// Execution option of the engine (`fhe_engine_set_f64`): rows whose moduli are all below 2^50 -- every modulus of
/// `BfvParameters::default_parameters_128` -- run on the FP64-FMA kernels (default) or on the integer kernels.  Results are
/// bit-identical either way; the parity tests run both.
pub fn set_f64_kernels(on: bool) { unsafe { ffi::fhe_engine_set_f64(on as std::os::raw::c_int) } }
pub fn f64_kernels() -> bool { unsafe { ffi::fhe_engine_get_f64() != 0 } }

/// The six tables of one `NttOperator` per modulus (ntt/native.rs:16-26), flattened `[nmoduli][degree]`.
/// The host passes its own so that psi -- drawn from ChaCha8 in the reference -- is the same on both sides.
pub struct NttTables<'a> {
    pub omegas: &'a [u64],
    pub omegas_shoup: &'a [u64],
    pub zetas_inv: &'a [u64],
    pub zetas_inv_shoup: &'a [u64],
    pub size_inv: &'a [u64],
    pub size_inv_shoup: &'a [u64],
}

/// How the reference's structs carry a device twin: `Context`, `Scaler`, `KeySwitchingKey`, `Multiplicator` derive
/// `Clone + PartialEq + Eq` (some `Default`), and a device handle must not take part in equality (two equal contexts
/// are equal whichever device object backs them).  The twin is built on FIRST USE (`get_or_try_init`), so
/// constructors and deserialisers only add `hip: Default::default()`, objects that never compute never touch the GPU,
/// and a clone made after the first use shares the same device object.
pub struct LazyHandle<T>(OnceLock<Arc<T>>);
impl<T> LazyHandle<T> {
    pub fn get(&self) -> Option<&Arc<T>> { self.0.get() }
    /// The device twin, built by `init` the first time (concurrent first users may both build; one result is kept).
    pub fn get_or_try_init<E>(&self, init: impl FnOnce() -> std::result::Result<T, E>) -> std::result::Result<&Arc<T>, E> {
        if let Some(v) = self.0.get() {
            return Ok(v);
        }
        let built = Arc::new(init()?);
        Ok(self.0.get_or_init(|| built))
    }
    /// Forgets the twin (the owner changed something the twin was built from); the next use rebuilds it.
    pub fn reset(&mut self) { self.0 = OnceLock::new(); }
}
impl<T> Default for LazyHandle<T> {
    fn default() -> Self { Self(OnceLock::new()) }
}
impl<T> Clone for LazyHandle<T> {
    fn clone(&self) -> Self {
        let c = OnceLock::new();
        if let Some(v) = self.0.get() {
            let _ = c.set(v.clone());
        }
        Self(c)
    }
}
impl<T> PartialEq for LazyHandle<T> {
    fn eq(&self, _: &Self) -> bool { true }
}
impl<T> Eq for LazyHandle<T> {}
impl<T> std::fmt::Debug for LazyHandle<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(if self.0.get().is_some() { "LazyHandle(device)" } else { "LazyHandle(unset)" })
    }
}

// ------------------------------------------------------------------------------------------ rq::Context
/// A context handle that does not own the C object: either the view inside a [`HipCtx`] or a level handle borrowed
/// from one (`'a` ties it to its root, which frees the whole chain).  Deliberately not `Clone`.
pub struct CtxView<'a> {
    ptr: *const ffi::FheCtx,
    _root: PhantomData<&'a ()>,
}
unsafe impl Send for CtxView<'_> {}
unsafe impl Sync for CtxView<'_> {}
impl CtxView<'_> {
    pub fn as_ptr(&self) -> *const ffi::FheCtx { self.ptr }
    pub fn degree(&self) -> usize { unsafe { ffi::fhe_ctx_degree(self.ptr) } }
    pub fn nmoduli(&self) -> usize { unsafe { ffi::fhe_ctx_nmoduli(self.ptr) } }
    pub fn device(&self) -> i32 { unsafe { ffi::fhe_ctx_device(self.ptr) as i32 } }
    /// u64 words of one polynomial, `L * N`.
    pub fn poly_words(&self) -> usize { self.nmoduli() * self.degree() }
    /// `Poly::ntt_forward` (rq/mod.rs:335-343) on `polys.len() / (L*N)` polynomials, in place.
    pub fn ntt_forward(&self, polys: &mut [u64]) -> Result<()> {
        let b = whole_batch("ntt_forward", polys.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_ntt_forward(self.ptr, polys.as_mut_ptr(), b) })
    }
    /// `Poly::ntt_backward` (rq/mod.rs:346-354).
    pub fn ntt_backward(&self, polys: &mut [u64]) -> Result<()> {
        let b = whole_batch("ntt_backward", polys.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_ntt_backward(self.ptr, polys.as_mut_ptr(), b) })
    }
    /// `AddAssign` / `SubAssign` / `MulAssign<&Poly<Ntt>>` (rq/ops.rs:10-206): `a op= b`.
    pub fn add_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        expect_len("add_assign rhs", b.len(), a.len())?;
        let n = whole_batch("add_assign", a.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_poly_add(self.ptr, a.as_mut_ptr(), b.as_ptr(), n) })
    }
    pub fn sub_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        expect_len("sub_assign rhs", b.len(), a.len())?;
        let n = whole_batch("sub_assign", a.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_poly_sub(self.ptr, a.as_mut_ptr(), b.as_ptr(), n) })
    }
    pub fn mul_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        expect_len("mul_assign rhs", b.len(), a.len())?;
        let n = whole_batch("mul_assign", a.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_poly_mul(self.ptr, a.as_mut_ptr(), b.as_ptr(), n) })
    }
    /// `MulAssign<&Poly<NttShoup>>` (rq/ops.rs:208-245).
    pub fn mul_shoup_assign(&self, a: &mut [u64], b: &[u64], b_shoup: &[u64]) -> Result<()> {
        expect_len("mul_shoup_assign rhs", b.len(), a.len())?;
        expect_len("mul_shoup_assign rhs twins", b_shoup.len(), a.len())?;
        let n = whole_batch("mul_shoup_assign", a.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_poly_mul_shoup(self.ptr, a.as_mut_ptr(), b.as_ptr(), b_shoup.as_ptr(), n) })
    }
    pub fn neg_assign(&self, a: &mut [u64]) -> Result<()> {
        let n = whole_batch("neg_assign", a.len(), self.poly_words())?;
        check(unsafe { ffi::fhe_poly_neg(self.ptr, a.as_mut_ptr(), n) })
    }
    /// `Poly::substitute` (rq/mod.rs:360-412); `exponent` as in `SubstitutionExponent::new`.
    pub fn substitute(&self, exponent: usize, input: &[u64], out: &mut [u64], repr_is_ntt: bool) -> Result<()> {
        expect_len("substitute output", out.len(), input.len())?;
        let n = whole_batch("substitute", input.len(), self.poly_words())?;
        check(unsafe {
            ffi::fhe_poly_substitute(self.ptr, exponent, input.as_ptr(), out.as_mut_ptr(), n, repr_is_ntt as c_int)
        })
    }
    /// `Poly::<PowerBasis>::switch_down` (rq/mod.rs:433-492): `[batch][L][N]` -> `[batch][L-1][N]`.
    pub fn switch_down(&self, input: &[u64], out: &mut [u64]) -> Result<()> {
        let b = whole_batch("switch_down", input.len(), self.poly_words())?;
        expect_len("switch_down output", out.len(), b * (self.nmoduli().saturating_sub(1)) * self.degree())?;
        check(unsafe { ffi::fhe_poly_switch_down(self.ptr, input.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// `Poly::<PowerBasis>::switch_down_to` (rq/mod.rs:498-507) in one call: `[batch][L][N]` -> `[batch][to.L][N]`.
    pub fn switch_down_to(&self, to: &CtxView<'_>, input: &[u64], out: &mut [u64]) -> Result<()> {
        let b = whole_batch("switch_down_to", input.len(), self.poly_words())?;
        expect_len("switch_down_to output", out.len(), b * to.poly_words())?;
        check(unsafe { ffi::fhe_poly_switch_down_to(self.ptr, to.ptr, input.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// `Ciphertext::switch_down` (bfv/ciphertext.rs:148-161) on `nparts`-part ciphertexts in Ntt form.
    pub fn ciphertext_switch_down(&self, nparts: usize, ct: &[u64], out: &mut [u64]) -> Result<()> {
        self.ciphertext_switch_to_level(1, nparts, ct, out)
    }
    /// `Ciphertext::switch_to_level` (bfv/ciphertext.rs:164-183), `levels` = target level - current level.
    pub fn ciphertext_switch_to_level(&self, levels: usize, nparts: usize, ct: &[u64], out: &mut [u64]) -> Result<()> {
        if nparts == 0 || levels >= self.nmoduli() {
            return Err(shape_error("switch_to_level: no parts, or more levels than the context chain has"));
        }
        let b = whole_batch("switch_to_level", ct.len(), nparts * self.poly_words())?;
        expect_len("switch_to_level output", out.len(), b * nparts * (self.nmoduli() - levels) * self.degree())?;
        check(unsafe { ffi::fhe_bfv_switch_to_level(self.ptr, levels, nparts, ct.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// The same on a device-resident batch (`fhe_bfv_switch_to_level_dev`), stream-ordered.
    pub fn ciphertexts_switch_to_level_dev(&self, levels: usize, ct: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        if ct.rows != self.nmoduli() || ct.degree != self.degree() || levels >= ct.rows {
            return Err(shape_error("switch_to_level_dev: ciphertexts do not live over this context"));
        }
        let out = DeviceCiphertexts::alloc_on(self.device(), ct.batch, ct.parts, ct.rows - levels, ct.degree, ct.level + levels, stream)?;
        check(unsafe {
            ffi::fhe_bfv_switch_to_level_dev(self.ptr, levels, ct.parts, ct.buf.as_ptr(), out.buf.as_mut_ptr(), ct.batch, stream.as_ptr())
        })?;
        Ok(out)
    }
}

pub struct HipCtx {
    view: CtxView<'static>, // (never handed out with the 'static lifetime: only through `Deref`, i.e. bounded by `&self`)
}
impl HipCtx {
    /// `Context::new` (rq/context.rs:42-92) on `device` (-1: host-only handle).
    pub fn new(device: i32, degree: usize, moduli: &[u64], tables: Option<&NttTables<'_>>) -> Result<Self> {
        let mut out: *mut ffi::FheCtx = ptr::null_mut();
        // (all six table pointers, or all NULL: the engine then derives its own primitive roots)
        let null = ptr::null::<u64>();
        let (om, oms, zi, zis, si, sis) = match tables {
            Some(t) => (t.omegas.as_ptr(), t.omegas_shoup.as_ptr(), t.zetas_inv.as_ptr(), t.zetas_inv_shoup.as_ptr(),
                        t.size_inv.as_ptr(), t.size_inv_shoup.as_ptr()),
            None => (null, null, null, null, null, null),
        };
        if let Some(t) = tables {
            let full = moduli.len() * degree;
            expect_len("NttTables.omegas", t.omegas.len(), full)?;
            expect_len("NttTables.omegas_shoup", t.omegas_shoup.len(), full)?;
            expect_len("NttTables.zetas_inv", t.zetas_inv.len(), full)?;
            expect_len("NttTables.zetas_inv_shoup", t.zetas_inv_shoup.len(), full)?;
            expect_len("NttTables.size_inv", t.size_inv.len(), moduli.len())?;
            expect_len("NttTables.size_inv_shoup", t.size_inv_shoup.len(), moduli.len())?;
        }
        check(unsafe {
            ffi::fhe_ctx_create(device as c_int, degree, moduli.len(), moduli.as_ptr(), om, oms, zi, zis, si, sis, &mut out)
        })?;
        Ok(Self { view: CtxView { ptr: out, _root: PhantomData } })
    }
    /// `Context::context_at_level`: a handle that lives as long as `self` (the root frees the chain).
    pub fn at_level(&self, level: usize) -> Result<CtxView<'_>> {
        let mut out: *const ffi::FheCtx = ptr::null();
        check(unsafe { ffi::fhe_ctx_at_level(self.view.ptr, level, &mut out) })?;
        Ok(CtxView { ptr: out, _root: PhantomData })
    }
}
impl Deref for HipCtx {
    type Target = CtxView<'static>;
    fn deref(&self) -> &CtxView<'static> { &self.view }
}
impl Drop for HipCtx {
    fn drop(&mut self) { unsafe { ffi::fhe_ctx_destroy(self.view.ptr as *mut ffi::FheCtx) } }
}

// ------------------------------------------------------------------------------------------ rq::Scaler
/// Every field of `RnsScaler` (rns/scaler.rs:52-72), `omega*` flattened `[to][from]`.
pub struct RnsScalerConstants<'a> {
    pub gamma: &'a [u64],
    pub gamma_shoup: &'a [u64],
    pub omega: &'a [u64],
    pub omega_shoup: &'a [u64],
    pub theta_gamma_lo: u64,
    pub theta_gamma_hi: u64,
    pub theta_gamma_sign: bool,
    pub theta_omega_lo: &'a [u64],
    pub theta_omega_hi: &'a [u64],
    pub theta_omega_sign: &'a [u8],
    pub theta_garner_lo: &'a [u64],
    pub theta_garner_hi: &'a [u64],
    pub theta_garner_shift: usize,
}
/// The same, owned (what `RnsScaler::hip_constants` of rust/patches/11-*.patch returns), plus `is_one`.
pub struct RnsScalerConstantsBuf {
    pub is_one: bool,
    pub gamma: Vec<u64>,
    pub gamma_shoup: Vec<u64>,
    pub omega: Vec<u64>,
    pub omega_shoup: Vec<u64>,
    pub theta_gamma_lo: u64,
    pub theta_gamma_hi: u64,
    pub theta_gamma_sign: bool,
    pub theta_omega_lo: Vec<u64>,
    pub theta_omega_hi: Vec<u64>,
    pub theta_omega_sign: Vec<u8>,
    pub theta_garner_lo: Vec<u64>,
    pub theta_garner_hi: Vec<u64>,
    pub theta_garner_shift: usize,
}
impl RnsScalerConstantsBuf {
    pub fn view(&self) -> RnsScalerConstants<'_> {
        RnsScalerConstants {
            gamma: &self.gamma, gamma_shoup: &self.gamma_shoup, omega: &self.omega, omega_shoup: &self.omega_shoup,
            theta_gamma_lo: self.theta_gamma_lo, theta_gamma_hi: self.theta_gamma_hi, theta_gamma_sign: self.theta_gamma_sign,
            theta_omega_lo: &self.theta_omega_lo, theta_omega_hi: &self.theta_omega_hi, theta_omega_sign: &self.theta_omega_sign,
            theta_garner_lo: &self.theta_garner_lo, theta_garner_hi: &self.theta_garner_hi,
            theta_garner_shift: self.theta_garner_shift,
        }
    }
}
pub struct HipScaler {
    ptr: *mut ffi::FheScaler,
    from: Arc<HipCtx>, // the C object points at both contexts
    to: Arc<HipCtx>,
}
unsafe impl Send for HipScaler {}
unsafe impl Sync for HipScaler {}
impl HipScaler {
    /// `Scaler::new` (rq/scaler.rs:27-52) with the constants the host's `RnsScaler::new` already computed.
    pub fn from_constants(from: &Arc<HipCtx>, to: &Arc<HipCtx>, number_common_moduli: usize, is_one: bool,
                          k: &RnsScalerConstants<'_>) -> Result<Self> {
        let (nf, nt) = (from.nmoduli(), to.nmoduli());
        expect_len("gamma", k.gamma.len(), nt)?;
        expect_len("gamma_shoup", k.gamma_shoup.len(), nt)?;
        expect_len("omega", k.omega.len(), nt * nf)?;
        expect_len("omega_shoup", k.omega_shoup.len(), nt * nf)?;
        expect_len("theta_omega_lo", k.theta_omega_lo.len(), nf)?;
        expect_len("theta_omega_hi", k.theta_omega_hi.len(), nf)?;
        expect_len("theta_omega_sign", k.theta_omega_sign.len(), nf)?;
        expect_len("theta_garner_lo", k.theta_garner_lo.len(), nf)?;
        expect_len("theta_garner_hi", k.theta_garner_hi.len(), nf)?;
        let mut out: *mut ffi::FheScaler = ptr::null_mut();
        check(unsafe {
            ffi::fhe_scaler_create_from_constants(from.as_ptr(), to.as_ptr(), number_common_moduli, is_one as c_int,
                k.gamma.as_ptr(), k.gamma_shoup.as_ptr(), k.omega.as_ptr(), k.omega_shoup.as_ptr(), k.theta_gamma_lo,
                k.theta_gamma_hi, k.theta_gamma_sign as c_int, k.theta_omega_lo.as_ptr(), k.theta_omega_hi.as_ptr(),
                k.theta_omega_sign.as_ptr(), k.theta_garner_lo.as_ptr(), k.theta_garner_hi.as_ptr(),
                k.theta_garner_shift, &mut out)
        })?;
        Ok(Self { ptr: out, from: from.clone(), to: to.clone() })
    }
    pub fn as_ptr(&self) -> *const ffi::FheScaler { self.ptr }
    pub fn from_ctx(&self) -> &Arc<HipCtx> { &self.from }
    pub fn to_ctx(&self) -> &Arc<HipCtx> { &self.to }
    /// `Scaler::scale` (rq/scaler.rs:55-127): `[batch][from.L][N]` -> `[batch][to.L][N]`.
    pub fn scale(&self, input: &[u64], out: &mut [u64], repr_is_ntt: bool) -> Result<()> {
        let b = whole_batch("scale", input.len(), self.from.poly_words())?;
        expect_len("scale output", out.len(), b * self.to.poly_words())?;
        check(unsafe { ffi::fhe_poly_scale(self.ptr, input.as_ptr(), out.as_mut_ptr(), b, repr_is_ntt as c_int) })
    }
}
impl Drop for HipScaler {
    fn drop(&mut self) { unsafe { ffi::fhe_scaler_destroy(self.ptr) } }
}

// ------------------------------------------------------------------------------------- KeySwitchingKey
pub struct HipKsk {
    ptr: *mut ffi::FheKsk,
    ct_ctx: Arc<HipCtx>, // the C object points at both contexts
    ksk_ctx: Arc<HipCtx>,
}
unsafe impl Send for HipKsk {}
unsafe impl Sync for HipKsk {}
impl HipKsk {
    /// Uploads `c0`, `c1` (`[ndigits][Lk][N]`, the `Poly<NttShoup>`s' coefficients).  The Shoup twins are optional:
    /// `None` lets the engine compute floor(c * 2^64 / q) itself -- the very definition `Poly<NttShoup>` uses
    /// (zq/mod.rs:195-199), and `Poly` has no public accessor for its twins.
    pub fn new(ct_ctx: &Arc<HipCtx>, ksk_ctx: &Arc<HipCtx>, ndigits: usize, c0: &[u64], c0_shoup: Option<&[u64]>,
               c1: &[u64], c1_shoup: Option<&[u64]>, log_base: usize) -> Result<Self> {
        let n = ndigits * ksk_ctx.poly_words();
        expect_len("ksk c0", c0.len(), n)?;
        expect_len("ksk c1", c1.len(), n)?;
        if let Some(t) = c0_shoup { expect_len("ksk c0 twins", t.len(), n)?; }
        if let Some(t) = c1_shoup { expect_len("ksk c1 twins", t.len(), n)?; }
        let mut out: *mut ffi::FheKsk = ptr::null_mut();
        check(unsafe {
            ffi::fhe_ksk_create(ct_ctx.as_ptr(), ksk_ctx.as_ptr(), ndigits, c0.as_ptr(),
                c0_shoup.map_or(ptr::null(), |t| t.as_ptr()), c1.as_ptr(), c1_shoup.map_or(ptr::null(), |t| t.as_ptr()),
                log_base, &mut out)
        })?;
        Ok(Self { ptr: out, ct_ctx: ct_ctx.clone(), ksk_ctx: ksk_ctx.clone() })
    }
    pub fn as_ptr(&self) -> *const ffi::FheKsk { self.ptr }
    pub fn ct_ctx(&self) -> &Arc<HipCtx> { &self.ct_ctx }
    pub fn ksk_ctx(&self) -> &Arc<HipCtx> { &self.ksk_ctx }
    /// `KeySwitchingKey::key_switch` (keys/key_switching_key.rs:241-270): `p` `[batch][L][N]` PowerBasis over the
    /// ciphertext context -> `c0`, `c1` `[batch][Lk][N]` Ntt over the key context.
    pub fn key_switch(&self, p: &[u64], c0: &mut [u64], c1: &mut [u64]) -> Result<()> {
        let b = whole_batch("key_switch input", p.len(), self.ct_ctx.poly_words())?;
        expect_len("key_switch c0", c0.len(), b * self.ksk_ctx.poly_words())?;
        expect_len("key_switch c1", c1.len(), b * self.ksk_ctx.poly_words())?;
        check(unsafe { ffi::fhe_key_switch(self.ptr, p.as_ptr(), c0.as_mut_ptr(), c1.as_mut_ptr(), b) })
    }
    /// `RelinearizationKey::relinearizes` (keys/relinearization_key.rs:69-102): `[batch][3][L][N]` -> `[batch][2][L][N]`.
    pub fn relinearize(&self, ct3: &[u64], out: &mut [u64]) -> Result<()> {
        let b = whole_batch("relinearize input", ct3.len(), 3 * self.ct_ctx.poly_words())?;
        expect_len("relinearize output", out.len(), b * 2 * self.ct_ctx.poly_words())?;
        check(unsafe { ffi::fhe_bfv_relinearize(self.ptr, ct3.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// `GaloisKey::relinearize` (keys/galois_key.rs:63-86): `[batch][2][L][N]` -> the same shape.
    pub fn galois(&self, exponent: usize, ct: &[u64], out: &mut [u64]) -> Result<()> {
        let b = whole_batch("galois input", ct.len(), 2 * self.ct_ctx.poly_words())?;
        expect_len("galois output", out.len(), ct.len())?;
        check(unsafe { ffi::fhe_bfv_galois(self.ptr, exponent, ct.as_ptr(), out.as_mut_ptr(), b) })
    }
    fn check_resident(&self, what: &str, ct: &DeviceCiphertexts, parts: usize) -> Result<()> {
        if ct.parts != parts || ct.rows != self.ct_ctx.nmoduli() || ct.degree != self.ct_ctx.degree() {
            return Err(shape_error(&format!("{what}: expected {parts}-part ciphertexts over the key's ciphertext context")));
        }
        Ok(())
    }
    /// `relinearizes` on a device-resident batch, stream-ordered (`fhe_bfv_relinearize_dev`).
    pub fn relinearize_dev(&self, ct3: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        self.check_resident("relinearize_dev", ct3, 3)?;
        let out = DeviceCiphertexts::alloc_on(self.ct_ctx.device(), ct3.batch, 2, ct3.rows, ct3.degree, ct3.level, stream)?;
        check(unsafe { ffi::fhe_bfv_relinearize_dev(self.ptr, ct3.buf.as_ptr(), out.buf.as_mut_ptr(), ct3.batch, stream.as_ptr()) })?;
        Ok(out)
    }
    /// `EvaluationKey::rotates_columns_by` / `rotates_rows` on a device-resident batch (`fhe_bfv_galois_dev`).
    pub fn galois_dev(&self, exponent: usize, ct: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        self.check_resident("galois_dev", ct, 2)?;
        let out = DeviceCiphertexts::alloc_on(self.ct_ctx.device(), ct.batch, 2, ct.rows, ct.degree, ct.level, stream)?;
        check(unsafe {
            ffi::fhe_bfv_galois_dev(self.ptr, exponent, ct.buf.as_ptr(), out.buf.as_mut_ptr(), ct.batch, stream.as_ptr())
        })?;
        Ok(out)
    }
    /// `EvaluationKey::computes_inner_sum` on a device-resident batch (`fhe_bfv_inner_sum_dev`): `keys[i]` is the
    /// Galois key of element `exponents[i]`, in the reference's order (evaluation_key.rs:56-100).
    pub fn inner_sum_dev(keys: &[Arc<HipKsk>], exponents: &[usize], ct: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        let first = keys.first().ok_or_else(|| shape_error("inner_sum_dev: no Galois keys"))?;
        if keys.len() != exponents.len() {
            return Err(shape_error("inner_sum_dev: one exponent per Galois key"));
        }
        first.check_resident("inner_sum_dev", ct, 2)?;
        let ptrs: Vec<*const ffi::FheKsk> = keys.iter().map(|k| k.as_ptr()).collect();
        let out = DeviceCiphertexts::alloc_on(first.ct_ctx.device(), ct.batch, 2, ct.rows, ct.degree, ct.level, stream)?;
        check(unsafe {
            ffi::fhe_bfv_inner_sum_dev(ptrs.as_ptr(), exponents.as_ptr(), keys.len(), ct.buf.as_ptr(), out.buf.as_mut_ptr(),
                                       ct.batch, stream.as_ptr())
        })?;
        Ok(out)
    }
    /// `EvaluationKey::expands` on a device-resident batch (`fhe_bfv_expand_dev`): `keys[l]` is the Galois key of
    /// element `(N >> l) + 1`.  The result has `size * ct.batch` ciphertexts laid out `[size][batch]`.
    pub fn expand_dev(keys: &[Arc<HipKsk>], size: usize, ct: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        if size == 0 || size > ct.degree {
            return Err(HipError { status: status::INVALID_EXPANSION_SIZE, message: format!("expand_dev: size {size}, degree {}", ct.degree) });
        }
        let device = match keys.first() {
            Some(k) => {
                k.check_resident("expand_dev", ct, 2)?;
                k.ct_ctx.device()
            }
            None if size == 1 => default_device(),
            None => return Err(HipError { status: status::EXPANSION_UNSUPPORTED, message: "expand_dev: no Galois keys".into() }),
        };
        let out = DeviceCiphertexts::alloc_on(device, size * ct.batch, 2, ct.rows, ct.degree, ct.level, stream)?;
        if keys.is_empty() {   // size 1: the expansion is the ciphertext itself
            check(unsafe {
                ffi::fhe_buf_copy_async(out.buf.as_mut_ptr() as *mut c_void, ct.buf.as_ptr() as *const c_void, ct.buf.len() * 8, stream.as_ptr())
            })?;
            return Ok(out);
        }
        let ptrs: Vec<*const ffi::FheKsk> = keys.iter().map(|k| k.as_ptr()).collect();
        check(unsafe {
            ffi::fhe_bfv_expand_dev(ptrs.as_ptr(), keys.len(), ct.buf.as_ptr(), out.buf.as_mut_ptr(), size, ct.batch, stream.as_ptr())
        })?;
        Ok(out)
    }
    /// How the engine evaluates every key switch through this key (`fhe_ksk_set_mode`): [`KsMode`]; `w_budget` bytes of
    /// transformed digit rows per launch pair of the unfused strategy (0 = default).  Values never change, only speed.
    pub fn set_mode(&self, mode: KsMode, w_budget: usize) -> Result<()> {
        check(unsafe { ffi::fhe_ksk_set_mode(self.ptr, mode as c_int, w_budget) })
    }
}
/// Key-switch evaluation strategies (`FHE_KS_*` of the header).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum KsMode {
    Auto = 0,
    Fused = 1,
    Unfused = 2,
    UnfusedSub = 3,
    /// rows larger than LDS (N >= 32768) on 8192-point sub-blocks instead of 16384-point parts
    FusedSub = 4,
}
impl Drop for HipKsk {
    fn drop(&mut self) { unsafe { ffi::fhe_ksk_destroy(self.ptr) } }
}

// ----------------------------------------------------------------------------------------- Multiplicator
/// What a `fhe_mul` points at and therefore has to outlive.
enum MulKeep {
    Parts { _ext_lhs: Arc<HipScaler>, _ext_rhs: Arc<HipScaler>, _down: Arc<HipScaler>, _rk: Option<Arc<HipKsk>> },
    Params { _params: Arc<HipParams>, _rk: Option<Arc<HipKsk>> },
}
pub struct HipMul {
    ptr: *mut ffi::FheMul,
    device: i32,
    in_words: usize,  // one input ciphertext: 2 * L * N
    out_words: usize, // one output ciphertext: parts * rows * N
    out_parts: usize,
    out_rows: usize,
    degree: usize,
    _keep: MulKeep,
}
unsafe impl Send for HipMul {}
unsafe impl Sync for HipMul {}
impl HipMul {
    fn finish(ptr: *mut ffi::FheMul, base: &CtxView<'_>, keep: MulKeep) -> Result<Self> {
        let (mut p, mut r) = (0usize, 0usize);
        let st = unsafe { ffi::fhe_mul_out_shape(ptr, &mut p, &mut r) };
        if st != 0 {
            unsafe { ffi::fhe_mul_destroy(ptr) };
            check(st)?;
        }
        let n = base.degree();
        Ok(Self { ptr, device: base.device(), in_words: 2 * base.poly_words(), out_words: p * r * n, out_parts: p,
                  out_rows: r, degree: n, _keep: keep })
    }
    /// `Multiplicator::new_leveled_internal` (+ `enable_relinearization`, `enable_mod_switching`), ops/mul.rs:74-163.
    pub fn new(extender_lhs: &Arc<HipScaler>, extender_rhs: &Arc<HipScaler>, down_scaler: &Arc<HipScaler>,
               rk: Option<&Arc<HipKsk>>, mod_switch: bool) -> Result<Self> {
        let mut out: *mut ffi::FheMul = ptr::null_mut();
        check(unsafe {
            ffi::fhe_mul_create(extender_lhs.as_ptr(), extender_rhs.as_ptr(), down_scaler.as_ptr(),
                rk.map_or(ptr::null(), |k| k.as_ptr()), mod_switch as c_int, &mut out)
        })?;
        let keep = MulKeep::Parts { _ext_lhs: extender_lhs.clone(), _ext_rhs: extender_rhs.clone(),
                                    _down: down_scaler.clone(), _rk: rk.cloned() };
        Self::finish(out, extender_lhs.from_ctx(), keep)
    }
    /// (parts, rows) of one output ciphertext.
    pub fn out_shape(&self) -> (usize, usize) { (self.out_parts, self.out_rows) }
    /// `Multiplicator::multiply` (ops/mul.rs:165-243) on `lhs.len() / (2*L*N)` ciphertext pairs `[batch][2][L][N]`
    /// in Ntt form -> `out` `[batch][parts][rows][N]`.  One pair is the reference's call; a whole `&[Ciphertext]`
    /// flattened into one slice amortises the PCIe copies.
    pub fn multiply(&self, lhs: &[u64], rhs: &[u64], out: &mut [u64]) -> Result<()> {
        expect_len("multiply rhs", rhs.len(), lhs.len())?;
        let b = whole_batch("multiply lhs", lhs.len(), self.in_words)?;
        expect_len("multiply output", out.len(), b * self.out_words)?;
        check(unsafe { ffi::fhe_bfv_mul(self.ptr, lhs.as_ptr(), rhs.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// The same on device-resident batches (`fhe_bfv_mul_dev`), stream-ordered: the result stays on the GPU.
    pub fn multiply_dev(&self, lhs: &DeviceCiphertexts, rhs: &DeviceCiphertexts, stream: &Stream) -> Result<DeviceCiphertexts> {
        if lhs.parts != 2 || rhs.parts != 2 || lhs.batch != rhs.batch || lhs.words_per_ct() != self.in_words
            || rhs.words_per_ct() != self.in_words || lhs.level != rhs.level {
            return Err(HipError { status: status::MUL_POLY_COUNT,
                                  message: "multiply_dev: operands must be equal batches of 2-part ciphertexts at the multiplicator's level".into() });
        }
        let level = lhs.level + (lhs.rows - self.out_rows); // (one level deeper when the handle switches the modulus)
        let out = DeviceCiphertexts::alloc_on(self.device, lhs.batch, self.out_parts, self.out_rows, self.degree, level, stream)?;
        check(unsafe {
            ffi::fhe_bfv_mul_dev(self.ptr, lhs.buf.as_ptr(), rhs.buf.as_ptr(), out.buf.as_mut_ptr(), lhs.batch, stream.as_ptr())
        })?;
        Ok(out)
    }
    /// 1: caller's stream only; 2 (default): chunks alternate with an internal stream.
    pub fn set_streams(&self, n: usize) -> Result<()> { check(unsafe { ffi::fhe_mul_set_streams(self.ptr, n) }) }
    pub fn set_chunk(&self, pairs: usize) -> Result<()> { check(unsafe { ffi::fhe_mul_set_chunk(self.ptr, pairs) }) }
}
impl Drop for HipMul {
    fn drop(&mut self) { unsafe { ffi::fhe_mul_destroy(self.ptr) } }
}

// ---------------------------------------------------------------------------------------- BfvParameters
/// `FnMut(modulus, degree) -> NttOperator tables` used by [`HipParams::with_tables`]: the engine asks for every
/// modulus it builds a context over (ciphertext moduli and the 62-bit extension primes).
pub type TablesFn<'a> = dyn FnMut(u64, usize, &mut [u64], &mut [u64], &mut [u64], &mut [u64]) -> Option<(u64, u64)> + 'a;

unsafe extern "C" fn tables_trampoline(user: *mut c_void, modulus: u64, degree: usize, omegas: *mut u64,
                                       omegas_shoup: *mut u64, zetas_inv: *mut u64, zetas_inv_shoup: *mut u64,
                                       size_inv: *mut u64, size_inv_shoup: *mut u64) -> c_int {
    let f = unsafe { &mut *(user as *mut &mut TablesFn<'_>) };
    let s = |p: *mut u64| unsafe { std::slice::from_raw_parts_mut(p, degree) };
    match f(modulus, degree, s(omegas), s(omegas_shoup), s(zetas_inv), s(zetas_inv_shoup)) {
        Some((inv, inv_shoup)) => {
            unsafe {
                *size_inv = inv;
                *size_inv_shoup = inv_shoup;
            }
            0
        }
        None => 1,
    }
}

pub struct HipParams {
    ptr: *mut ffi::FheParams,
}
unsafe impl Send for HipParams {}
unsafe impl Sync for HipParams {}
impl HipParams {
    /// `BfvParametersBuilder::build`'s level tables with the HOST's NTT tables (parameters.rs:560-738): `tables`
    /// writes `NttOperator::new(modulus, degree)`'s four arrays and returns `(size_inv, size_inv_shoup)`.
    pub fn with_tables(device: i32, degree: usize, moduli: &[u64], plaintext_modulus: u64,
                       tables: &mut TablesFn<'_>) -> Result<Arc<Self>> {
        let mut out: *mut ffi::FheParams = ptr::null_mut();
        let mut fat: &mut TablesFn<'_> = tables;
        check(unsafe {
            ffi::fhe_params_create_with_tables(device as c_int, degree, moduli.len(), moduli.as_ptr(), plaintext_modulus,
                Some(tables_trampoline), &mut fat as *mut &mut TablesFn<'_> as *mut c_void, &mut out)
        })?;
        Ok(Arc::new(Self { ptr: out }))
    }
    /// `Multiplicator::default(rk)` (+ `enable_mod_switching`) at `level`; `rk = None`: `&ct * &ct`.
    /// (The engine cached every table `with_tables` supplied; the closure is not called again.)  The handle keeps
    /// the parameter set (and the key) alive.
    pub fn multiplicator(self: &Arc<Self>, level: usize, rk: Option<&Arc<HipKsk>>, mod_switch: bool) -> Result<HipMul> {
        let mut out: *mut ffi::FheMul = ptr::null_mut();
        check(unsafe {
            ffi::fhe_mul_create_default(self.ptr, level, rk.map_or(ptr::null(), |k| k.as_ptr()), mod_switch as c_int, &mut out)
        })?;
        let base = self.context_at_level(level)?;
        HipMul::finish(out, &base, MulKeep::Params { _params: self.clone(), _rk: rk.cloned() })
    }
    /// `BfvParameters::context_at_level`: borrowed from `self`.
    pub fn context_at_level(&self, level: usize) -> Result<CtxView<'_>> {
        let mut out: *const ffi::FheCtx = ptr::null();
        check(unsafe { ffi::fhe_params_ctx(self.ptr, level, &mut out) })?;
        Ok(CtxView { ptr: out, _root: PhantomData })
    }
}
impl Drop for HipParams {
    fn drop(&mut self) { unsafe { ffi::fhe_params_destroy(self.ptr) } }
}

// ------------------------------------------------------------------------------------ engine-wide state
/// Bounds on the scratch memory the engine retains between calls (`fhe_workspace_set_limit`; 0 = none, `usize::MAX` =
/// the default: a quarter of the device's memory in total): bytes per (device, stream) and in total.  Idle blocks beyond
/// a bound are evicted least-recently-used first; calls that need more still run.  A host that destroys HIP streams of
/// its own (not [`Stream`]s, whose `Drop` tells the engine) sets `total_bytes` to a few streams' footprint.
pub fn workspace_set_limit(per_stream_bytes: usize, total_bytes: usize) -> Result<()> {
    check(unsafe { ffi::fhe_workspace_set_limit(per_stream_bytes, total_bytes) })
}
/// (bytes held, bytes in use, blocks, distinct (device, stream) owners, internal second streams) of the engine's
/// scratch pool.
pub fn workspace_stats() -> Result<(usize, usize, usize, usize, usize)> {
    let (mut h, mut u, mut b, mut o, mut a) = (0usize, 0usize, 0usize, 0usize, 0usize);
    check(unsafe { ffi::fhe_workspace_stats(&mut h, &mut u, &mut b, &mut o, &mut a) })?;
    Ok((h, u, b, o, a))
}
/// What the driver says the library's two private stream-ordered pools on `device` hold: (scratch reserved, scratch used,
/// buffers reserved, buffers used) bytes (`fhe_workspace_pool_stats`).  The scratch pool's reserved bytes respect the
/// `total_bytes` bound of [`workspace_set_limit`] once evicted blocks have retired.
pub fn workspace_pool_stats(device: i32) -> Result<(usize, usize, usize, usize)> {
    let (mut sr, mut su, mut br, mut bu) = (0usize, 0usize, 0usize, 0usize);
    check(unsafe { ffi::fhe_workspace_pool_stats(device as c_int, &mut sr, &mut su, &mut br, &mut bu) })?;
    Ok((sr, su, br, bu))
}
/// Frees every idle scratch block, internal stream and pooled event; returns the bytes released.
pub fn workspace_trim() -> usize { unsafe { ffi::fhe_workspace_trim() } }
/// Number of HIP devices the engine sees (a host that shards a batch over the GPUs of a node makes one set of handles
/// per device and drives each from its own thread -- `tests/c_host/c4_sharded.c` is that host in C).
pub fn device_count() -> i32 { unsafe { ffi::fhe_device_count() as i32 } }

// -------------------------------------------------------------------- device memory, streams, residency
/// A HIP stream made by the C ABI (`fhe_stream_create`); `_dev` calls on one stream run in order.
pub struct Stream {
    ptr: *mut c_void,
}
unsafe impl Send for Stream {}
unsafe impl Sync for Stream {}
impl Stream {
    pub fn new(device: i32) -> Result<Self> {
        let mut out: *mut c_void = ptr::null_mut();
        check(unsafe { ffi::fhe_stream_create(device as c_int, &mut out) })?;
        Ok(Self { ptr: out })
    }
    pub fn as_ptr(&self) -> *mut c_void { self.ptr }
    /// Blocks until everything enqueued on the stream has finished.
    pub fn synchronize(&self) -> Result<()> { check(unsafe { ffi::fhe_stream_sync(self.ptr) }) }
}
impl Drop for Stream {
    fn drop(&mut self) { unsafe { ffi::fhe_stream_destroy(self.ptr); } }
}

/// `len` u64 words of device memory (`fhe_buf_alloc`).  Freed on drop; `hipFree` waits for work that still uses
/// the allocation, so dropping a buffer a stream is reading is slow, not unsound.
pub struct DeviceBuffer {
    ptr: *mut u64,
    len: usize,
}
unsafe impl Send for DeviceBuffer {}
unsafe impl Sync for DeviceBuffer {}
impl DeviceBuffer {
    pub fn alloc(device: i32, len: usize) -> Result<Self> {
        let mut out: *mut c_void = ptr::null_mut();
        check(unsafe { ffi::fhe_buf_alloc(device as c_int, len * 8, &mut out) })?;
        Ok(Self { ptr: out as *mut u64, len })
    }
    /// Stream-ordered allocation (`fhe_buf_alloc_async`): usable by work enqueued on `stream` from here on, and no
    /// device synchronisation -- what the `_dev` operations allocate their results with.
    pub fn alloc_on(device: i32, len: usize, stream: &Stream) -> Result<Self> {
        let mut out: *mut c_void = ptr::null_mut();
        check(unsafe { ffi::fhe_buf_alloc_async(device as c_int, len * 8, stream.as_ptr(), &mut out) })?;
        Ok(Self { ptr: out as *mut u64, len })
    }
    /// Gives the block back behind the work already enqueued on `stream` (`fhe_buf_free_async`) instead of waiting for
    /// the device as `drop` does.  `stream` must be the stream of the buffer's LAST USE (every operation that read or
    /// wrote it was enqueued there, or is ordered before it by the caller's own events): a free that is ordered on
    /// another stream can hand the memory out again while the last reader still runs.  When in doubt,
    /// drop the buffer: `Drop` waits for the device.
    pub fn release_on(self, stream: &Stream) -> Result<()> {
        let p = self.ptr as *mut c_void;
        std::mem::forget(self);
        check(unsafe { ffi::fhe_buf_free_async(p, stream.as_ptr()) })
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_ptr(&self) -> *const u64 { self.ptr }
    pub fn as_mut_ptr(&self) -> *mut u64 { self.ptr }
    /// Host -> device, stream-ordered; returns when the bytes have left `src` (the copy is waited for).
    pub fn upload(&self, src: &[u64], stream: &Stream) -> Result<()> {
        expect_len("upload", src.len(), self.len)?;
        check(unsafe { ffi::fhe_buf_upload(self.ptr as *mut c_void, src.as_ptr() as *const c_void, self.len * 8, stream.as_ptr()) })
    }
    /// Device -> host after everything enqueued on `stream` so far (waits for the stream).
    pub fn download(&self, dst: &mut [u64], stream: &Stream) -> Result<()> {
        expect_len("download", dst.len(), self.len)?;
        check(unsafe { ffi::fhe_buf_download(dst.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * 8, stream.as_ptr()) })
    }
}
impl Drop for DeviceBuffer {
    fn drop(&mut self) { unsafe { ffi::fhe_buf_free(self.ptr as *mut c_void); } }
}

/// A batch of `bfv::Ciphertext`s that stays on the GPU between operations: `[batch][parts][rows][N]` u64, Ntt form,
/// plus the `level` bookkeeping of `Ciphertext` (ciphertext.rs:18-30).  Made by `upload` (or by an operation), consumed
/// by `HipMul::multiply_dev`, `HipKsk::{relinearize_dev, galois_dev}`, `CtxView::ciphertexts_switch_to_level_dev`;
/// `download` is the only point that waits.  (`seed` is dropped on upload exactly as every evaluation drops it,
/// ops/mod.rs:66.)
pub struct DeviceCiphertexts {
    buf: DeviceBuffer,
    pub batch: usize,
    pub parts: usize,
    pub rows: usize,
    pub degree: usize,
    pub level: usize,
}
impl DeviceCiphertexts {
    pub fn alloc(device: i32, batch: usize, parts: usize, rows: usize, degree: usize, level: usize) -> Result<Self> {
        Ok(Self { buf: DeviceBuffer::alloc(device, batch * parts * rows * degree)?, batch, parts, rows, degree, level })
    }
    /// The same on `stream`'s allocation order (results of the `_dev` operations).
    pub fn alloc_on(device: i32, batch: usize, parts: usize, rows: usize, degree: usize, level: usize, stream: &Stream) -> Result<Self> {
        Ok(Self { buf: DeviceBuffer::alloc_on(device, batch * parts * rows * degree, stream)?, batch, parts, rows, degree, level })
    }
    /// Frees the batch behind the work enqueued on `stream` (no device synchronisation).
    pub fn release_on(self, stream: &Stream) -> Result<()> { self.buf.release_on(stream) }
    pub fn words_per_ct(&self) -> usize { self.parts * self.rows * self.degree }
    pub fn buffer(&self) -> &DeviceBuffer { &self.buf }
    /// `flat`: the ciphertexts' polynomials' coefficients, concatenated `[batch][parts][rows][N]`.
    pub fn upload(device: i32, flat: &[u64], parts: usize, rows: usize, degree: usize, level: usize, stream: &Stream) -> Result<Self> {
        let batch = whole_batch("DeviceCiphertexts::upload", flat.len(), parts * rows * degree)?;
        let d = Self::alloc(device, batch, parts, rows, degree, level)?;
        d.buf.upload(flat, stream)?;
        Ok(d)
    }
    pub fn download(&self, stream: &Stream) -> Result<Vec<u64>> {
        let mut v = vec![0u64; self.buf.len()];
        self.buf.download(&mut v, stream)?;
        Ok(v)
    }
}
