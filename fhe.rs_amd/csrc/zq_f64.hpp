// zq_f64.hpp -- exact modular arithmetic on doubles holding integers, for moduli p < 2^50.
//
// Why: every ciphertext / key modulus of the reference's stock parameter sets is 36-49 bits
// (F/bfv/parameters.rs:222-251).  gfx950 has no 64 x 64 -> 128 multiplier -- a lazy Shoup product is ten
// v_mad_u64_u32 / v_mul_lo_u32 (zq_dev.hpp) -- but it has a full-rate FP64 FMA pipe, and for operands that fit the
// 53-bit significand a fused multiply-add returns the exact low half of a product:
//
//     h = RN(x w)            l = fma(x, w, -h)  ==  x w - h   exactly (error-free product, Dekker / Veltkamp)
//     q = rint(x * (w/p))    an estimate of x w / p, off by at most `c` (below)
//     r = fma(-q, p, h) + l  ==  x w - q p      exactly: h - q p is an integer below 2^53, so the fma does not round
//
// six full-rate operations for a lazy modular product with a precomputed w/p (the role Shoup's floor(w 2^64 / p)
// plays in M/zq/mod.rs:224-234), against ~13.7 issue slots for the integer form (profiles/r06_f64_gate.json).
// VALUES ARE SIGNED: a residue class is represented by any integer v with |v| < 2^53 congruent to it; canonical
// [0, p) u64 words are formed only at kernel boundaries (to_u64_canonical), so nothing crosses the C ABI in this form
// and the outputs -- canonical residues, a function of the inputs -- stay bit-identical to the reference.
//
// Bounds (p < 2^(53-H), H >= 3; |x| = a p, 0 <= w < p, wp = RN(w / p)):
//   * x * wp carries two roundings: |x wp - x w / p| <= |x| 2^-52 (1 + 2^-53) =: eps <= a 2^(1-H);
//     q = rint(.) is within 0.5 + eps of the true quotient, so |r| <= (0.5 + eps) p.
//   * exactness of fma(-q, p, h): h - q p = r - l with |l| <= ulp(h) / 2 <= |x| p 2^-53 <= a p 2^-H... in integers:
//     |h - q p| <= (0.5 + eps) p + a p 2^-H p / p  -- with a <= 2^(H-1) both terms are below p, so |h - q p| < 2 p < 2^53:
//     an integer of that size is representable, the fma is exact, and so is the final + l.
//   * callers keep a <= 2^(H-1) (|x| < 2^52): eps <= 1, |r| <= 1.5 p; with a <= 2^(H-2): |r| <= p; a <= 1: |r| < 0.51 p.
// reduce_f64 (x - p rint(x / p), three operations) brings any |x| < 2^52 to |r| <= 0.5 p + |x| 2^-52 p.
// f64_bound() tracks `a` through the stages of a transform at compile time, as fwd_narrow_bound() does for the
// integer narrow butterflies, and says where a reduction has to sit.
#pragma once
#include <cmath>
#include <cstdint>

#include "zq_dev.hpp"

namespace fhe {

// p and 1/p as doubles (wave-uniform; scalar registers)
struct PF {
    double p, ip;
};

FHE_HD double f64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
FHE_HD double f64_rint(double a) { return __builtin_rint(a); }   // v_rndne_f64 (round to nearest even: the default mode)

// integer u64 below 2^52 -> double, exactly: the 2^52 bias trick (one OR on the high word, one subtract)
FHE_HD double f64_from_u64(u64 x) {
#if defined(FHE_HOST_EMULATION)
    if (x >> 52) __builtin_trap();
#endif
    union {
        u64 u;
        double d;
    } v;
    v.u = x | 0x4330000000000000ull;
    return v.d - 4503599627370496.0;
}
// double holding an integer in [0, 2^52) -> u64
FHE_HD u64 f64_to_u64(double x) {
#if defined(FHE_HOST_EMULATION)
    if (!(x >= 0.0 && x < 4503599627370496.0) || x != std::floor(x)) __builtin_trap();
#endif
    union {
        u64 u;
        double d;
    } v;
    v.d = x + 4503599627370496.0;
    return v.u & 0x000FFFFFFFFFFFFFull;
}

// x w mod p, lazily: |result| <= (0.5 + |x| 2^-52) p.  wp = RN(w / p) (or any value within 2^-53 relative of w / p).
FHE_HD double mulmod_f64(double x, double w, double wp, double p) {
    const double h = x * w;
    const double l = f64_fma(x, w, -h);
    const double q = f64_rint(x * wp);
#if defined(FHE_HOST_EMULATION)
    if (!(std::fabs(x) < 9007199254740992.0) || x != std::floor(x)) __builtin_trap();   // range tracking broken
#endif
    return f64_fma(-q, p, h) + l;
}
// acc + x w mod p, lazily (the addend joins the exact low half: one operation, like the integer form's free addend)
FHE_HD double mulmod_add_f64(double acc, double x, double w, double wp, double p) {
    const double h = x * w;
    const double l = f64_fma(x, w, -h);
    const double q = f64_rint(x * wp);
    return (f64_fma(-q, p, h) + l) + acc;
}
// product of two residues with no precomputed quotient (tensor products): q from h / p
FHE_HD double mulmod2_f64(double a, double b, const PF &m) {
    const double h = a * b;
    const double l = f64_fma(a, b, -h);
    const double q = f64_rint(h * m.ip);
    return f64_fma(-q, m.p, h) + l;
}
// |x| < 2^52 -> |r| <= 0.5 p (1 + 2^-40)
FHE_HD double reduce_f64(double x, const PF &m) { return f64_fma(-f64_rint(x * m.ip), m.p, x); }
// any representative with |x| < 2^52 -> the canonical residue in [0, p) as a u64 word
FHE_HD u64 to_u64_canonical(double x, const PF &m) {
    double r = reduce_f64(x, m);          // |r| <= ~0.5 p
    r = r < 0.0 ? r + m.p : r;            // [0, p]: r == p cannot happen (r + p < p when r < 0), r == -0.0 stays 0
    r = r >= m.p ? r - m.p : r;           // (the reduction's own rounding slack: r slightly above 0.5 p is still below p)
    return f64_to_u64(r);
}

// Harvey-shaped lazy butterflies on signed representatives (M/ntt/native.rs:256-269 / 288-300 compute the same
// residues): forward (x, y) <- (x + w y, x - w y); the outputs grow by |w y| <= (0.5 + eps) p per stage.
FHE_HD void fwd_butterfly_f64(double &x, double &y, double w, double wp, double p) {
    const double t = mulmod_f64(y, w, wp, p);
    y = x - t;
    x = x + t;
}
// inverse (Gentleman-Sande): (x, y) <- (x + y, (x - y) z); the sum doubles per stage, the product is below ~p
FHE_HD void inv_butterfly_f64(double &x, double &y, double z, double zp, double p) {
    const double d = x - y;
    x = x + y;
    y = mulmod_f64(d, z, zp, p);
}

// Bound tracking, in units of p, for p < 2^(53 - H): representatives must stay below 2^(H-1) p = 2^52 where they enter a
// product.  Forward: a' = a + 0.5 + a 2^(1-H) per stage; f64_fwd_reduce(stage) says whether the values are reduced
// (to 0.5 p) before that stage.  Computed in 1/1024ths to stay in integers.
constexpr int f64_fwd_step(int a1024, int H) { return a1024 + 512 + ((a1024 * 2) >> H) + 1; }
constexpr int f64_fwd_bound(int stage, int H, int a0_1024 = 1024) {
    int a = a0_1024;
    for (int s = 0; s < stage; s++) {
        if (f64_fwd_step(a, H) > (1024 << (H - 1))) a = 513;   // reduce first
        a = f64_fwd_step(a, H);
    }
    return a;
}
constexpr bool f64_fwd_reduces(int stage, int H, int a0_1024 = 1024) {
    return f64_fwd_step(f64_fwd_bound(stage, H, a0_1024), H) > (1024 << (H - 1));
}

}  // namespace fhe
