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
// Second form, same six operations, NO precomputed quotient: q = rint(h (1/p)) (mulmod2_add_f64) -- the operand is then ONE
// word.  The fused key switch uses it for its key words and its per-lane twiddles (half the key bytes through L2, half the
// twiddle registers: the N = 16384 tile runs radix-8 passes throughout); the transforms proper keep {w, w/p} pairs, whose two
// multiplies are independent (8 % faster register resident).
// VALUES ARE SIGNED: a residue class is represented by any integer v with |v| < 2^53 congruent to it; canonical
// [0, p) u64 words are formed only at kernel boundaries (to_u64_canonical), so nothing crosses the C ABI in this form
// and the outputs -- canonical residues, a function of the inputs -- stay bit-identical to the reference.
//
// Bounds (p < 2^(53-H), H >= 3; |x| = a p, 0 <= w < p, wp = RN(w / p)):
//   * x * wp carries two roundings: |x wp - x w / p| <= |x| 2^-52 (1 + 2^-53) =: eps <= a 2^(1-H);
//     q = rint(.) is within 0.5 + eps of the true quotient, so |r| <= (0.5 + eps) p.
//   * exactness of fma(-q, p, h): h - q p = r - l with |l| <= ulp(h) / 2 <= |x| w 2^-53 < |x| p 2^-53.  For |x| < 2^53:
//     eps < 2, |r| < 2.5 p, |l| < p, so |h - q p| < 3.5 p < 2^52: an integer of that size is representable, the fma does
//     not round, and neither does the final + l (two integers whose sum is below 2^52).
//   * so the one invariant is REPRESENTABILITY: every value stays an integer below 2^53 in magnitude; the lazy product
//     of y is then below (0.5 + |y| 2^-52) p.  f64_fwd_bound() below tracks it.
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
// acc + x k mod p for a per-lane k with NO precomputed quotient (the key switch's accumulate: dropping k / p halves the key
// bytes a workgroup pulls through L2 -- 16 instead of 32 per coefficient and digit -- for the same six operations): the quotient
// estimate h (1/p) carries three roundings, so |x k - q p| <= (0.5 + 3 |x| 2^-53) p; exact for |x| < 2^53 like mulmod_f64.
FHE_HD double mulmod2_add_f64(double acc, double x, double k, const PF &m) {
    const double h = x * k;
    const double l = f64_fma(x, k, -h);
    const double q = f64_rint(h * m.ip);
    return (f64_fma(-q, m.p, h) + l) + acc;
}
// |x| < 2^52 -> |r| <= 0.5 p (1 + 2^-40)
FHE_HD double reduce_f64(double x, const PF &m) { return f64_fma(-f64_rint(x * m.ip), m.p, x); }
// any representative (|x| < 2^53) -> the canonical residue in [0, p) as a u64 word.  reduce_f64 leaves |r| <= 0.5 p + 8
// (the quotient estimate x * (1/p) is within |x / p| 2^-52 of the true quotient -- a few ulps more if 1/p itself is not
// correctly rounded -- and r = x - q p is exact), so ONE conditional add of p makes it canonical: r < 0 gives r + p in
// [0.5 p - 8, p), r >= 0 is at most 0.5 p + 8 < p.
FHE_HD u64 to_u64_canonical(double x, const PF &m) {
    double r = reduce_f64(x, m);
    r = r < 0.0 ? r + m.p : r;            // (-0.0 stays 0)
#if defined(FHE_HOST_EMULATION)
    if (!(r >= 0.0 && r < m.p)) __builtin_trap();
#endif
    return f64_to_u64(r);
}

// Harvey-shaped lazy butterflies on signed representatives (M/ntt/native.rs:256-269 / 288-300 compute the same
// residues): forward (x, y) <- (x + w y, x - w y); the outputs grow by |w y| <= (0.5 + eps) p per stage.
// (w: the twiddle as a double; the quotient comes from h (1/p) -- mulmod2_add_f64's form -- so a twiddle is ONE word: the
// per-lane twiddles of the late passes cost half the registers and half the L2 bytes of a {w, w/p} pair, which is what lets the
// N = 16384 key switch run radix-8 passes throughout without scratch and takes the forward transform from 110 to 68 VGPRs.)
FHE_HD void fwd_butterfly_f64(double &x, double &y, double w, const PF &m) {
    const double t = mulmod2_add_f64(0.0, y, w, m);
    y = x - t;
    x = x + t;
}
// The same with the quotient from a precomputed w / p (mulmod_f64): for passes whose twiddles are WAVE-UNIFORM -- the pair
// sits in scalar registers, so the second word costs nothing, and the two multiplies of a product are independent (measured,
// register resident: 3.59 against 3.31 T butterflies/s; profiles/r06_f64_gate.json vs r06_h_f64_rates.json).
FHE_HD void fwd_butterfly_wp_f64(double &x, double &y, double w, double wp, const PF &m) {
    const double t = mulmod_f64(y, w, wp, m.p);
    y = x - t;
    x = x + t;
}
// inverse (Gentleman-Sande): (x, y) <- (x + y, (x - y) z); the sum doubles per stage, the product is below ~p.  (The inverse
// passes run in the transform kernels only, which keep {z, z/p} pairs in every pass.)
FHE_HD void inv_butterfly_f64(double &x, double &y, double z, double zp, const PF &m) {
    const double d = x - y;
    x = x + y;
    y = mulmod_f64(d, z, zp, m.p);
}

// A modulus as the F64 passes carry it: the kernels' PM record (four 64-bit words in scalar registers) holds the bit
// patterns of p and 1/p as doubles in its first two words, so that the pass templates keep one signature.
FHE_HD double f64_of_bits(u64 v) {
    union {
        u64 u;
        double d;
    } c;
    c.u = v;
    return c.d;
}
FHE_HD u64 bits_of_f64(double d) {
    union {
        u64 u;
        double d;
    } c;
    c.d = d;
    return c.u;
}
FHE_HD PM make_pm_f64(const DevMod &m) {
    const double p = (double)m.p;          // exact: p < 2^50
    return PM{bits_of_f64(p), bits_of_f64(1.0 / p), 0, 0};
}
FHE_HD PF pf_of(const PM &pm) { return PF{f64_of_bits(pm.p), f64_of_bits(pm.p2)}; }

// Bound tracking for a launch whose moduli are all below 2^(53 - HR) (HR = 3: 50-bit primes, 4: 49-bit, 5: 48-bit and
// less).  Bounds are ABSOLUTE, in units of U = 2^(53 - HR) / 1024: a digit row lifted to another modulus of the launch is
// simply a representative below 1024 U, whatever the ratio of the two moduli.  Representatives must stay below
// 2^53 = 2^HR * 1024 U.  A lazy product of y is below (0.5 + |y| 2^-52) p < (512 + (|y| / U) 2^(1 - HR)) U.
constexpr int F64_ONE = 1024;
constexpr int f64_limit(int HR) { return F64_ONE << HR; }
// (the kernels take every quotient from h (1/p): |x k - q p| <= (0.5 + 3 |x| 2^-53) p; +2: rounding of this model)
constexpr int f64_product_bound(int y, int HR) { return 512 + ((3 * y) >> HR) + 2; }
constexpr int f64_product_bound2(int y, int HR) { return f64_product_bound(y, HR); }
constexpr int F64_REDUCED = 520;   // after reduce_f64: 0.5 p + |x| 2^-52 p, far below 520 / 1024
// forward (Cooley-Tukey) stages: every value of a stage shares one bound; a' = a + product_bound(a)
constexpr int f64_fwd_step(int a, int HR) { return a + f64_product_bound(a, HR); }
constexpr int f64_fwd_bound(int stage, int HR, int a0 = F64_ONE) {   // bound BEFORE `stage` (after its reduction, if any)
    int a = a0;
    for (int s = 0; s < stage; s++) {
        if (f64_fwd_step(a, HR) >= f64_limit(HR)) a = F64_REDUCED;
        a = f64_fwd_step(a, HR);
    }
    return a;
}
constexpr bool f64_fwd_reduces(int stage, int HR, int a0 = F64_ONE) {   // are the values reduced before `stage`?
    return f64_fwd_step(f64_fwd_bound(stage, HR, a0), HR) >= f64_limit(HR);
}
constexpr int f64_fwd_out_bound(int stages, int HR, int a0 = F64_ONE) {   // bound after the last of `stages` stages
    const int a = f64_fwd_bound(stages - 1, HR, a0);
    return f64_fwd_step(f64_fwd_reduces(stages - 1, HR, a0) ? F64_REDUCED : a, HR);
}
static_assert(f64_fwd_out_bound(16, 5) < f64_limit(5), "48-bit moduli: no reduction inside a forward transform up to N = 65536");
static_assert(!f64_fwd_reduces(15, 5) && f64_fwd_reduces(4, 3) && !f64_fwd_reduces(3, 3) && f64_fwd_reduces(9, 4) && !f64_fwd_reduces(8, 4), "forward reduction schedule");

}  // namespace fhe
