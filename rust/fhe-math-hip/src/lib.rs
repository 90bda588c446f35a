//! Safe handles over the C ABI of the MI355X engine (`include/fhe_hip.h`, `libfhe_hip.so`).
//!
//! One RAII type per opaque C handle, named after the reference type it shadows:
//!
//! | here          | C handle     | reference type (fhe.rs)                                   |
//! |---------------|--------------|-----------------------------------------------------------|
//! | [`HipCtx`]    | `fhe_ctx`    | `fhe_math::rq::Context`            (rq/context.rs:9-19)    |
//! | [`HipScaler`] | `fhe_scaler` | `fhe_math::rq::scaler::Scaler`     (rq/scaler.rs:18-23)    |
//! | [`HipKsk`]    | `fhe_ksk`    | `fhe::bfv::KeySwitchingKey`        (keys/key_switching_key.rs:22-46) |
//! | [`HipMul`]    | `fhe_mul`    | `fhe::bfv::Multiplicator`          (ops/mul.rs:21-32)      |
//! | [`HipParams`] | `fhe_params` | level tables of `BfvParameters`    (parameters.rs:83-117)  |
//!
//! Buffers are `Poly`'s own `[L][N]` row-major `u64` slices (`coefficients.as_slice()`), batches are
//! concatenations; every call is synchronous (host pointers).  The `_dev` twins of `ffi` take device pointers
//! and a `hipStream_t` for resident batches.  Handles are immutable after creation: `Send + Sync`.
//! `ffi.rs` is generated from the header (`tools/gen_rust_ffi.py`); `tests/test_rust_shim.py` keeps the two equal.
pub mod ffi;

use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;

/// Non-zero `fhe_status` with the engine's thread-local message.  The codes map 1:1 onto the `Result` variants
/// of the path (table in `fhe_hip.h`); `fhe-math` / `fhe` translate them back (rust/patches/errors.patch).
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

/// How the reference's structs carry a device twin: `Context`, `Scaler`, `Multiplicator` derive
/// `Default + Clone + PartialEq + Eq`, and a device handle must not take part in equality (two equal contexts are
/// equal whichever device object backs them).  `None` (the `Default`) leaves the native CPU path in place.
pub struct Handle<T>(Option<std::sync::Arc<T>>);
impl<T> Handle<T> {
    pub fn new(v: T) -> Self { Self(Some(std::sync::Arc::new(v))) }
    pub fn get(&self) -> Option<&T> { self.0.as_deref() }
}
impl<T> Default for Handle<T> {
    fn default() -> Self { Self(None) }
}
impl<T> Clone for Handle<T> {
    fn clone(&self) -> Self { Self(self.0.clone()) }
}
impl<T> PartialEq for Handle<T> {
    fn eq(&self, _: &Self) -> bool { true }
}
impl<T> Eq for Handle<T> {}
impl<T> std::fmt::Debug for Handle<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(if self.0.is_some() { "Handle(device)" } else { "Handle(none)" })
    }
}

// ------------------------------------------------------------------------------------------ rq::Context
pub struct HipCtx {
    ptr: *const ffi::FheCtx,
    owned: bool, // borrowed level handles are freed with their root
}
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}
impl HipCtx {
    /// `Context::new` (rq/context.rs:42-92) on `device` (-1: host-only handle).
    pub fn new(device: i32, degree: usize, moduli: &[u64], tables: Option<&NttTables<'_>>) -> Result<Self> {
        let mut out: *mut ffi::FheCtx = ptr::null_mut();
        let p = |f: fn(&NttTables<'_>) -> &[u64]| tables.map_or(ptr::null(), |t| f(t).as_ptr());
        if let Some(t) = tables {
            assert!(t.omegas.len() == moduli.len() * degree && t.size_inv.len() == moduli.len());
        }
        check(unsafe {
            ffi::fhe_ctx_create(device as c_int, degree, moduli.len(), moduli.as_ptr(), p(|t| t.omegas),
                p(|t| t.omegas_shoup), p(|t| t.zetas_inv), p(|t| t.zetas_inv_shoup), p(|t| t.size_inv),
                p(|t| t.size_inv_shoup), &mut out)
        })?;
        Ok(Self { ptr: out, owned: true })
    }
    /// `Context::context_at_level`: a handle borrowed from (and valid as long as) `self`'s root.
    pub fn at_level(&self, level: usize) -> Result<HipCtx> {
        let mut out: *const ffi::FheCtx = ptr::null();
        check(unsafe { ffi::fhe_ctx_at_level(self.ptr, level, &mut out) })?;
        Ok(HipCtx { ptr: out, owned: false })
    }
    pub fn as_ptr(&self) -> *const ffi::FheCtx { self.ptr }
    pub fn degree(&self) -> usize { unsafe { ffi::fhe_ctx_degree(self.ptr) } }
    pub fn nmoduli(&self) -> usize { unsafe { ffi::fhe_ctx_nmoduli(self.ptr) } }
    fn batch(&self, len: usize, rows: usize) -> usize {
        let per = rows * self.degree();
        assert!(per > 0 && len % per == 0, "buffer is not a whole number of polynomials");
        len / per
    }
    /// `Poly::ntt_forward` (rq/mod.rs:335-343) on `polys.len() / (L*N)` polynomials, in place.
    pub fn ntt_forward(&self, polys: &mut [u64]) -> Result<()> {
        check(unsafe { ffi::fhe_ntt_forward(self.ptr, polys.as_mut_ptr(), self.batch(polys.len(), self.nmoduli())) })
    }
    /// `Poly::ntt_backward` (rq/mod.rs:346-354).
    pub fn ntt_backward(&self, polys: &mut [u64]) -> Result<()> {
        check(unsafe { ffi::fhe_ntt_backward(self.ptr, polys.as_mut_ptr(), self.batch(polys.len(), self.nmoduli())) })
    }
    /// `AddAssign` / `SubAssign` / `MulAssign<&Poly<Ntt>>` (rq/ops.rs:10-206): `a op= b`.
    pub fn add_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        assert_eq!(a.len(), b.len());
        check(unsafe { ffi::fhe_poly_add(self.ptr, a.as_mut_ptr(), b.as_ptr(), self.batch(a.len(), self.nmoduli())) })
    }
    pub fn sub_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        assert_eq!(a.len(), b.len());
        check(unsafe { ffi::fhe_poly_sub(self.ptr, a.as_mut_ptr(), b.as_ptr(), self.batch(a.len(), self.nmoduli())) })
    }
    pub fn mul_assign(&self, a: &mut [u64], b: &[u64]) -> Result<()> {
        assert_eq!(a.len(), b.len());
        check(unsafe { ffi::fhe_poly_mul(self.ptr, a.as_mut_ptr(), b.as_ptr(), self.batch(a.len(), self.nmoduli())) })
    }
    /// `MulAssign<&Poly<NttShoup>>` (rq/ops.rs:208-245).
    pub fn mul_shoup_assign(&self, a: &mut [u64], b: &[u64], b_shoup: &[u64]) -> Result<()> {
        assert!(a.len() == b.len() && b.len() == b_shoup.len());
        check(unsafe {
            ffi::fhe_poly_mul_shoup(self.ptr, a.as_mut_ptr(), b.as_ptr(), b_shoup.as_ptr(), self.batch(a.len(), self.nmoduli()))
        })
    }
    pub fn neg_assign(&self, a: &mut [u64]) -> Result<()> {
        check(unsafe { ffi::fhe_poly_neg(self.ptr, a.as_mut_ptr(), self.batch(a.len(), self.nmoduli())) })
    }
    /// `Poly::substitute` (rq/mod.rs:360-412); `exponent` as in `SubstitutionExponent::new`.
    pub fn substitute(&self, exponent: usize, input: &[u64], out: &mut [u64], repr_is_ntt: bool) -> Result<()> {
        assert_eq!(input.len(), out.len());
        check(unsafe {
            ffi::fhe_poly_substitute(self.ptr, exponent, input.as_ptr(), out.as_mut_ptr(),
                self.batch(input.len(), self.nmoduli()), repr_is_ntt as c_int)
        })
    }
    /// `Poly::<PowerBasis>::switch_down` (rq/mod.rs:433-492): `[batch][L][N]` -> `[batch][L-1][N]`.
    pub fn switch_down(&self, input: &[u64], out: &mut [u64]) -> Result<()> {
        let b = self.batch(input.len(), self.nmoduli());
        assert_eq!(out.len(), b * (self.nmoduli() - 1) * self.degree());
        check(unsafe { ffi::fhe_poly_switch_down(self.ptr, input.as_ptr(), out.as_mut_ptr(), b) })
    }
    /// `Ciphertext::switch_down` (bfv/ciphertext.rs:148-161) on `nparts`-part ciphertexts in Ntt form.
    pub fn ciphertext_switch_down(&self, nparts: usize, ct: &[u64], out: &mut [u64]) -> Result<()> {
        let b = self.batch(ct.len(), nparts * self.nmoduli());
        check(unsafe { ffi::fhe_bfv_switch_down(self.ptr, nparts, ct.as_ptr(), out.as_mut_ptr(), b) })
    }
}
impl Drop for HipCtx {
    fn drop(&mut self) {
        if self.owned {
            unsafe { ffi::fhe_ctx_destroy(self.ptr as *mut ffi::FheCtx) }
        }
    }
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
pub struct HipScaler {
    ptr: *mut ffi::FheScaler,
    to_rows: usize,
    from_rows: usize,
    degree: usize,
}
unsafe impl Send for HipScaler {}
unsafe impl Sync for HipScaler {}
impl HipScaler {
    /// `Scaler::new` (rq/scaler.rs:27-52) with the constants the host's `RnsScaler::new` already computed.
    pub fn from_constants(from: &HipCtx, to: &HipCtx, number_common_moduli: usize, is_one: bool,
                          k: &RnsScalerConstants<'_>) -> Result<Self> {
        let mut out: *mut ffi::FheScaler = ptr::null_mut();
        check(unsafe {
            ffi::fhe_scaler_create_from_constants(from.as_ptr(), to.as_ptr(), number_common_moduli, is_one as c_int,
                k.gamma.as_ptr(), k.gamma_shoup.as_ptr(), k.omega.as_ptr(), k.omega_shoup.as_ptr(), k.theta_gamma_lo,
                k.theta_gamma_hi, k.theta_gamma_sign as c_int, k.theta_omega_lo.as_ptr(), k.theta_omega_hi.as_ptr(),
                k.theta_omega_sign.as_ptr(), k.theta_garner_lo.as_ptr(), k.theta_garner_hi.as_ptr(),
                k.theta_garner_shift, &mut out)
        })?;
        Ok(Self { ptr: out, to_rows: to.nmoduli(), from_rows: from.nmoduli(), degree: from.degree() })
    }
    pub fn as_ptr(&self) -> *const ffi::FheScaler { self.ptr }
    /// `Scaler::scale` (rq/scaler.rs:55-127): `[batch][from.L][N]` -> `[batch][to.L][N]`.
    pub fn scale(&self, input: &[u64], out: &mut [u64], repr_is_ntt: bool) -> Result<()> {
        let b = input.len() / (self.from_rows * self.degree);
        assert!(input.len() == b * self.from_rows * self.degree && out.len() == b * self.to_rows * self.degree);
        check(unsafe { ffi::fhe_poly_scale(self.ptr, input.as_ptr(), out.as_mut_ptr(), b, repr_is_ntt as c_int) })
    }
}
impl Drop for HipScaler {
    fn drop(&mut self) { unsafe { ffi::fhe_scaler_destroy(self.ptr) } }
}

// ------------------------------------------------------------------------------------- KeySwitchingKey
pub struct HipKsk {
    ptr: *mut ffi::FheKsk,
}
unsafe impl Send for HipKsk {}
unsafe impl Sync for HipKsk {}
impl HipKsk {
    /// Uploads `c0`, `c1` (`[ndigits][Lk][N]`, NttShoup polys' coefficients and Shoup twins).
    pub fn new(ct_ctx: &HipCtx, ksk_ctx: &HipCtx, ndigits: usize, c0: &[u64], c0_shoup: &[u64], c1: &[u64],
               c1_shoup: &[u64], log_base: usize) -> Result<Self> {
        let n = ndigits * ksk_ctx.nmoduli() * ksk_ctx.degree();
        assert!(c0.len() == n && c1.len() == n && c0_shoup.len() == n && c1_shoup.len() == n);
        let mut out: *mut ffi::FheKsk = ptr::null_mut();
        check(unsafe {
            ffi::fhe_ksk_create(ct_ctx.as_ptr(), ksk_ctx.as_ptr(), ndigits, c0.as_ptr(), c0_shoup.as_ptr(), c1.as_ptr(),
                c1_shoup.as_ptr(), log_base, &mut out)
        })?;
        Ok(Self { ptr: out })
    }
    pub fn as_ptr(&self) -> *const ffi::FheKsk { self.ptr }
    /// `KeySwitchingKey::key_switch` (keys/key_switching_key.rs:241-270); `p` PowerBasis.
    pub fn key_switch(&self, p: &[u64], c0: &mut [u64], c1: &mut [u64], batch: usize) -> Result<()> {
        check(unsafe { ffi::fhe_key_switch(self.ptr, p.as_ptr(), c0.as_mut_ptr(), c1.as_mut_ptr(), batch) })
    }
    /// `RelinearizationKey::relinearizes` (keys/relinearization_key.rs:69-102): 3 parts in, 2 out.
    pub fn relinearize(&self, ct3: &[u64], out: &mut [u64], batch: usize) -> Result<()> {
        check(unsafe { ffi::fhe_bfv_relinearize(self.ptr, ct3.as_ptr(), out.as_mut_ptr(), batch) })
    }
    /// `GaloisKey::relinearize` (keys/galois_key.rs:63-86).
    pub fn galois(&self, exponent: usize, ct: &[u64], out: &mut [u64], batch: usize) -> Result<()> {
        check(unsafe { ffi::fhe_bfv_galois(self.ptr, exponent, ct.as_ptr(), out.as_mut_ptr(), batch) })
    }
}
impl Drop for HipKsk {
    fn drop(&mut self) { unsafe { ffi::fhe_ksk_destroy(self.ptr) } }
}

// ----------------------------------------------------------------------------------------- Multiplicator
pub struct HipMul {
    ptr: *mut ffi::FheMul,
}
unsafe impl Send for HipMul {}
unsafe impl Sync for HipMul {}
impl HipMul {
    /// `Multiplicator::new_leveled_internal` (+ `enable_relinearization`, `enable_mod_switching`), ops/mul.rs:74-163.
    pub fn new(extender_lhs: &HipScaler, extender_rhs: &HipScaler, down_scaler: &HipScaler, rk: Option<&HipKsk>,
               mod_switch: bool) -> Result<Self> {
        let mut out: *mut ffi::FheMul = ptr::null_mut();
        check(unsafe {
            ffi::fhe_mul_create(extender_lhs.as_ptr(), extender_rhs.as_ptr(), down_scaler.as_ptr(),
                rk.map_or(ptr::null(), |k| k.as_ptr()), mod_switch as c_int, &mut out)
        })?;
        Ok(Self { ptr: out })
    }
    /// (parts, rows) of one output ciphertext.
    pub fn out_shape(&self) -> Result<(usize, usize)> {
        let (mut p, mut r) = (0usize, 0usize);
        check(unsafe { ffi::fhe_mul_out_shape(self.ptr, &mut p, &mut r) })?;
        Ok((p, r))
    }
    /// `Multiplicator::multiply` (ops/mul.rs:165-243) on `batch` ciphertext pairs `[batch][2][L][N]`, Ntt form.
    pub fn multiply(&self, lhs: &[u64], rhs: &[u64], out: &mut [u64], batch: usize) -> Result<()> {
        assert_eq!(lhs.len(), rhs.len());
        check(unsafe { ffi::fhe_bfv_mul(self.ptr, lhs.as_ptr(), rhs.as_ptr(), out.as_mut_ptr(), batch) })
    }
    /// Device-resident variant on a HIP stream (`fhe_bfv_mul_dev`): raw device pointers.
    ///
    /// # Safety
    /// The pointers must be device allocations of the shapes `multiply` documents, valid until the stream drains.
    pub unsafe fn multiply_dev(&self, lhs: *const u64, rhs: *const u64, out: *mut u64, batch: usize,
                               stream: *mut c_void) -> Result<()> {
        check(unsafe { ffi::fhe_bfv_mul_dev(self.ptr, lhs, rhs, out, batch, stream) })
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
                       tables: &mut TablesFn<'_>) -> Result<Self> {
        let mut out: *mut ffi::FheParams = ptr::null_mut();
        let mut fat: &mut TablesFn<'_> = tables;
        check(unsafe {
            ffi::fhe_params_create_with_tables(device as c_int, degree, moduli.len(), moduli.as_ptr(), plaintext_modulus,
                Some(tables_trampoline), &mut fat as *mut &mut TablesFn<'_> as *mut c_void, &mut out)
        })?;
        Ok(Self { ptr: out })
    }
    /// `Multiplicator::default(rk)` (+ `enable_mod_switching`) at `level`; `rk = None`: `&ct * &ct`.
    /// (The engine cached every table `with_tables` supplied; the closure is not called again.)
    pub fn multiplicator(&self, level: usize, rk: Option<&HipKsk>, mod_switch: bool) -> Result<HipMul> {
        let mut out: *mut ffi::FheMul = ptr::null_mut();
        check(unsafe {
            ffi::fhe_mul_create_default(self.ptr, level, rk.map_or(ptr::null(), |k| k.as_ptr()), mod_switch as c_int, &mut out)
        })?;
        Ok(HipMul { ptr: out })
    }
    /// `BfvParameters::context_at_level`: borrowed from `self`.
    pub fn context_at_level(&self, level: usize) -> Result<HipCtx> {
        let mut out: *const ffi::FheCtx = ptr::null();
        check(unsafe { ffi::fhe_params_ctx(self.ptr, level, &mut out) })?;
        Ok(HipCtx { ptr: out, owned: false })
    }
}
impl Drop for HipParams {
    fn drop(&mut self) { unsafe { ffi::fhe_params_destroy(self.ptr) } }
}
