//! User code against the patched `fhe` crate (feature `hip`): one upload, a chain of homomorphic operations that never
//! leaves the GPU, one download.  Every item used here is PUBLIC -- in the reference, in `fhe-math-hip` or added by
//! `rust/patches/` -- which `tests/test_rust_shim.py::test_examples_use_only_public_items` checks against the
//! reference's own `pub` / `pub(crate)` declarations (round 3's example reached `GaloisKey.ksk`, which is pub(crate)).
//! (No Rust toolchain exists in the build image: reviewed source, not compiled there.)
use std::sync::Arc;

use fhe::bfv::{BfvParameters, Ciphertext, EvaluationKey, Multiplicator, RelinearizationKey};
use fhe_math_hip::{DeviceCiphertexts, Stream};

/// sum over the slots of (a_i * b_i rotated by one column), for a batch of ciphertext pairs, at level `out_level`
pub fn rotated_products(
    par: &Arc<BfvParameters>,
    multiplicator_without_key: &Multiplicator, // Multiplicator::default-shaped, no relinearisation key: three-part output
    rk: &RelinearizationKey,
    ek: &EvaluationKey,
    lhs: &[Ciphertext],
    rhs: &[Ciphertext],
    out_level: usize,
) -> fhe::Result<Vec<Ciphertext>> {
    let s = Stream::new(fhe_math_hip::default_device()).map_err(hip)?;
    // one upload per operand batch (ciphertexts of one level, two parts each)
    let da: DeviceCiphertexts = Ciphertext::to_device(lhs, &s)?;
    let db: DeviceCiphertexts = Ciphertext::to_device(rhs, &s)?;
    // everything below is stream-ordered; nothing crosses PCIe and nothing waits
    let prod3 = multiplicator_without_key.multiply_dev(&da, &db, &s)?;
    let prod = rk.relinearizes_dev(&prod3, &s)?;
    let rot = ek.rotates_columns_by_dev(&prod, 1, &s)?;
    let sum = ek.computes_inner_sum_dev(&rot, &s)?;
    let low = Ciphertext::switch_to_level_dev(par, &sum, out_level, &s)?;
    // operands and intermediates go back to the pool behind the work that still reads them (no device synchronisation)
    for d in [da, db, prod3, prod, rot, sum] {
        d.release_on(&s).map_err(hip)?;
    }
    // the only wait of the whole chain
    Ciphertext::from_device(par, &low, &s)
}

fn hip(e: fhe_math_hip::HipError) -> fhe::Error {
    panic!("{e}") // (user code has no access to the crate-private status mapping; a HIP failure here is fatal anyway)
}
