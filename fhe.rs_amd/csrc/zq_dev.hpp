// zq_dev.hpp -- 64-bit modular arithmetic for the kernels (moduli < 2^62).
//
// Value semantics follow fhe_math::zq::Modulus (M/zq/mod.rs): Shoup multiplication for every
// constant operand (lazy_mul_shoup :224-234), lazy [0,2p)/[0,4p) ranges inside the NTT
// butterflies (M/ntt/native.rs:256-300) and canonical [0,p) at every kernel boundary.  Only
// canonical values cross the C ABI, and a canonical residue is a mathematical function of the
// inputs, so any exact reduction gives bit-identical outputs; the single-word Barrett below
// replaces the reference's 128-bit-ratio Barrett (:693-707) / NFLlib "opt" path (:730-740)
// at 11 instead of ~18 32-bit multiplies and works for every modulus (no supports_opt split).
//
// There is no 64x64->128 multiplier on CDNA4: `__umul64hi` lowers to four v_mad_u64_u32 and
// a low 64-bit product to one v_mad_u64_u32 + two v_mul_lo_u32, so a Shoup modmul costs ten
// 32-bit multiplies.  Integer multiply issue, not HBM, bounds these kernels.
#pragma once
#include <cstdint>

#include "knobs.hpp"

#if defined(__HIPCC__) || defined(__HIP__)
#define FHE_HD __host__ __device__ __forceinline__
#else
#define FHE_HD inline
#endif

namespace fhe {

typedef uint64_t u64;
typedef unsigned __int128 u128_t;

struct DevMod {  // layout == hostmath.hpp ModConsts
    u64 p, p2, mu, brt_hi, brt_lo;
    uint32_t k, pad;
    u64 np, np2;  // 2^64 - p, 2^64 - 2p (loaded, so the compiler cannot fold x + np back into x - p)
};

// (Round 2 priced the instruction stream with timing-only builds that computed WRONG residues on purpose -- fewer
// partial products, no barriers, a pseudo-Mersenne fold; results in DESIGN.md section 6, sources in the history up to
// commit db30352.  None of that is in this file any more: every path here is exact.)
// The high product through v_mad_u64_u32's carry-out (FHE_MAD_CARRY, default on; the device compiler has no way to
// ask for it): the two cross products are summed by the multiply-add itself, their carry leaves in an SGPR pair and
// joins the upper half of the sum as the 64-bit addend of the last multiply -- one v_mul_hi, three multiply-adds, a
// move and a select instead of the four multiplies plus ~six moves / 64-bit adds of the generic expansion (gfx950
// needs even-aligned register pairs, so every 32-bit piece that enters a 64-bit addend costs the compiler a move).
// SU: `b` is wave-uniform and stays in scalar registers (one constant-bus operand per instruction).
// Every asm statement is a single instruction, so the scheduler still interleaves neighbouring butterflies; the
// s_nop covers the two wait states gfx950 wants between a VALU write of an SGPR and a VALU read of it.
#if defined(__HIP_DEVICE_COMPILE__) && FHE_MAD_CARRY
#define FHE_HAVE_MAD_CARRY 1
template <bool SU>
__device__ __forceinline__ u64 mad64_carry(uint32_t a, uint32_t b, u64 add, u64 &carry) {
    u64 r;
    if constexpr (SU)
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "s"(b), "v"(add));
    else
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(add));
    return r;
}
__device__ __forceinline__ uint32_t carry_bit(u64 carry) {
    uint32_t r;
    asm("s_nop 1\n\tv_cndmask_b32 %0, 0, 1, %1" : "=v"(r) : "s"(carry));
    return r;
}
// floor(a * b / 2^64)
template <bool SU = false>
__device__ __forceinline__ u64 mulhi64_c(u64 a, u64 b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const u64 m = (u64)a0 * b1 + (u64)__umulhi(a0, b0);   // < 2^64
    u64 c;
    const u64 n = mad64_carry<SU>(a1, b0, m, c);
    return (u64)a1 * b1 + ((u64)(uint32_t)(n >> 32) | ((u64)carry_bit(c) << 32));
}
// the same without the a0 * b0 partial product: the true value or one less
template <bool SU = false>
__device__ __forceinline__ u64 mulhi64_approx_c(u64 a, u64 b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const u64 m = (u64)a0 * b1;
    u64 c;
    const u64 n = mad64_carry<SU>(a1, b0, m, c);
    return (u64)a1 * b1 + ((u64)(uint32_t)(n >> 32) | ((u64)carry_bit(c) << 32));
}
// (Issuing the two 32-bit cross products of the Shoup low word behind the carry-producing multiply in the same asm
// statement, as its wait states, measured 1.2 % slower than the s_nop: the statement pins three instructions.)
#else
#define FHE_HAVE_MAD_CARRY 0
#endif

// add + a*b + q*np (mod 2^64): the low word of a lazy Shoup product, with an optional addend.
// FHE_MAD_CROSS: the four 32-bit cross products (a0*b1, a1*b0, q0*np1, q1*np0: only their low words count) are
// summed by a chain of v_mad_u64_u32 used as a 32-bit multiply-add (the upper half of its result is junk), and
// their sum joins the upper word with one add -- 6 multiply-adds + 1 add against the compiler's 2 multiply-adds,
// 4 v_mul_lo_u32 and 2 v_add3_u32.  SU: b is wave-uniform (np always is).
template <bool SU = false, bool ADD = true>
FHE_HD u64 shoup_lo(u64 add, u64 a, u64 b, u64 q, u64 np) {
#if defined(__HIP_DEVICE_COMPILE__) && FHE_MAD_CROSS
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32), n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    u64 t, r, sd;
    constexpr bool no_add = !ADD;   // (no addend: the inline constant 0, not a register pair holding it)
    if constexpr (SU) {
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t), "=s"(sd) : "v"(a0), "s"(b1));
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(sd) : "v"(a1), "s"(b0), "v"(t));
        if constexpr (no_add)
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(sd) : "v"(a0), "s"(b0));
        else
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sd) : "v"(a0), "s"(b0), "v"(add));
    } else {
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(t), "=s"(sd) : "v"(a0), "v"(b1));
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(sd) : "v"(a1), "v"(b0), "v"(t));
        if constexpr (no_add)
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(sd) : "v"(a0), "v"(b0));
        else
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sd) : "v"(a0), "v"(b0), "v"(add));
    }
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(sd) : "v"(q0), "s"(n1), "v"(t));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(sd) : "v"(q1), "s"(n0), "v"(t));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sd) : "v"(q0), "s"(n0), "v"(r));
    // (written as a 64-bit add the compiler builds the pair {0, t.lo} with two moves and adds it with v_lshl_add_u64)
    uint32_t rh;
    asm("v_add_u32 %0, %1, %2" : "=v"(rh) : "v"((uint32_t)(r >> 32)), "v"((uint32_t)t));
    return ((u64)rh << 32) | (uint32_t)r;
#else
    return (ADD ? add : 0) + a * b + q * np;
#endif
}

FHE_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && FHE_MAD_CARRY
    return mulhi64_c<false>(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128_t)a * b) >> 64);
#endif
}

// x in [0, 2m) -> [0, m)
FHE_HD u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }
// same with nm = 2^64 - m supplied: gfx950 has a one-instruction 64-bit add (v_lshl_add_u64) but
// subtracts through a v_sub_co/v_subb pair plus a VCC wait state, so hot loops add -m instead.
// (The carry of x + nm IS the comparison; add / add-with-carry / two selects through __builtin_addc measured the
// same as this compare-select-add form at N = 8192 and costs registers: 272 B of scratch in the N = 16384 key switch.)
FHE_HD u64 csub_n(u64 x, u64 m, u64 nm) {
    return x + (x >= m ? nm : 0);
}

// M/zq/mod.rs:224-234: any a < 2^64, b < p, bs = floor(b * 2^64 / p); result in [0, 2p).
FHE_HD u64 mul_shoup_lazy(u64 a, u64 b, u64 bs, u64 p) {
    u64 q = mulhi64(a, bs);
    return a * b - q * p;
}
FHE_HD u64 mul_shoup(u64 a, u64 b, u64 bs, u64 p) { return csub(mul_shoup_lazy(a, b, bs, p), p); }

// Barrett reduction of x = hi:lo < 2^(2k+1) (a product of two residues, or the sum of two) to [0, p).
// q = floor(floor(x / 2^(k-1)) * floor(2^(2k)/p) / 2^(k+1)) underestimates floor(x/p) by <= 2
// for x < 2^(2k) and by <= 3 for x < 2^(2k+1), so x - q*p < 4p < 2^64.
FHE_HD u64 barrett_reduce_wide(u64 hi, u64 lo, const DevMod &m) {
    const uint32_t s = m.k - 1;
    u64 xs = (lo >> s) | (hi << (64 - s));  // x >> (k-1), < 2^(k+1)   (p >= 2, hostmath.hpp: k >= 2, s in [1, 61])
    u64 q = mulhi64(xs, m.mu);
    u64 r = lo + q * m.np;  // lo - q*p < 3p < 2^64
    r = csub_n(r, m.p2, m.np2);
    return csub_n(r, m.p, m.np);
}
// Full product of two residues, both below 2^62 (the largest modulus size): four 32 x 32 multiply-adds with the
// cross products summed by the multiply-add itself -- a0*b1 + a1*b0 + 2^32 cannot leave 64 bits when a1, b1 < 2^30.
FHE_HD void mul_wide62(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(FHE_HOST_EMULATION)
    if ((a | b) >> 62) __builtin_trap();
#endif
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const u64 l = (u64)a0 * b0;
    u64 m = (u64)a0 * b1 + (l >> 32);
    m += (u64)a1 * b0;
    hi = (u64)a1 * b1 + (m >> 32);
    lo = (u64)(uint32_t)l | (m << 32);
}
// a*b + c*d for four such residues (below 2^125): eight multiply-adds, one carry (the two low products)
FHE_HD void mac2_wide62(u64 a, u64 b, u64 c, u64 d, u64 &hi, u64 &lo) {
#if FHE_HAVE_MAD_CARRY
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32), d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32);
    u64 cl;
    const u64 l = mad64_carry<false>(c0, d0, (u64)a0 * b0, cl);
    u64 m = (u64)a0 * b1 + ((u64)(uint32_t)(l >> 32) | ((u64)carry_bit(cl) << 32));
    m += (u64)a1 * b0;   // four cross products below 2^62 each and 2^33: no carry out
    m += (u64)c0 * d1;
    m += (u64)c1 * d0;
    hi = (u64)a1 * b1 + (m >> 32);
    hi += (u64)c1 * d1;
    lo = (u64)(uint32_t)l | (m << 32);
#else
#if defined(FHE_HOST_EMULATION)
    if ((a | b | c | d) >> 62) __builtin_trap();
#endif
    const u128_t sum = (u128_t)a * b + (u128_t)c * d;
    hi = (u64)(sum >> 64), lo = (u64)sum;
#endif
}
// lazy: below 2p (what the inverse transform's first pass takes)
FHE_HD u64 barrett_reduce_wide_lazy(u64 hi, u64 lo, const DevMod &m) {
    const uint32_t s = m.k - 1;
    u64 xs = (lo >> s) | (hi << (64 - s));
    u64 q = mulhi64(xs, m.mu);
    return csub_n(lo + q * m.np, m.p2, m.np2);
}
FHE_HD u64 mul_mod(u64 a, u64 b, const DevMod &m) {  // a, b < p
    u64 lo, hi;
    mul_wide62(a, b, hi, lo);
    return barrett_reduce_wide(hi, lo, m);
}
FHE_HD u64 mul_mod_lazy(u64 a, u64 b, const DevMod &m) {  // a, b < p -> below 2p
    u64 lo, hi;
    mul_wide62(a, b, hi, lo);
    return barrett_reduce_wide_lazy(hi, lo, m);
}

// Reduction of an arbitrary 64-bit value to [0, p): q = floor(a * floor(2^64/p) / 2^64).
// floor(2^64/p) is brt_hi when p > 1 (M/zq/mod.rs:712-723 keeps the low word too; dropping it
// costs at most one extra conditional subtraction).
FHE_HD u64 reduce_u64(u64 a, const DevMod &m) {
    u64 q = mulhi64(a, m.brt_hi);
    u64 r = a + q * m.np;  // a - q*p < 3p
    r = csub_n(r, m.p2, m.np2);
    return csub_n(r, m.p, m.np);
}

// Full 128-bit reduction, M/zq/mod.rs:693-707 (needed only for the scaler's v and w words).
FHE_HD u64 reduce_u128(u64 hi, u64 lo, const DevMod &m) {
    u64 p_lo_lo = mulhi64(lo, m.brt_lo);
    // (lo*brt_hi + hi*brt_lo + p_lo_lo) >> 64, with 128-bit carries
    u64 a0 = lo * m.brt_hi, a1 = mulhi64(lo, m.brt_hi);
    u64 b0 = hi * m.brt_lo, b1 = mulhi64(hi, m.brt_lo);
    u64 s0 = a0 + b0;
    u64 c0 = s0 < a0;
    u64 s0b = s0 + p_lo_lo;
    u64 c1 = s0b < s0;
    u64 q = a1 + b1 + c0 + c1 + hi * m.brt_hi;
    u64 r = lo + q * m.np;  // lo - q*p < 2p
    return csub_n(r, m.p, m.np);
}

FHE_HD u64 add_mod(u64 a, u64 b, u64 p) { return csub(a + b, p); }
FHE_HD u64 sub_mod(u64 a, u64 b, u64 p) { return csub(a + p - b, p); }
FHE_HD u64 neg_mod(u64 a, u64 p) { return csub(p - a, p); }

// A (wave-uniform) modulus with its two's-complement negations, as the NTT passes carry it.
struct PM {
    u64 p, p2, np, np2;
};
FHE_HD PM make_pm(const DevMod &m) { return PM{m.p, m.p2, m.np, m.np2}; }
// mul_shoup_lazy with a*b - q*p written as a*b + q*(2^64 - p)  (mod 2^64)
// SU (here and in the butterflies): the twiddle pair is wave-uniform (scalar registers); see mad64_carry
template <bool SU = false>
FHE_HD u64 mulhi64_t(u64 a, u64 b) {
#if FHE_HAVE_MAD_CARRY
    return mulhi64_c<SU>(a, b);
#else
    return mulhi64(a, b);
#endif
}
template <bool SU = false>
FHE_HD u64 mul_shoup_lazy_n(u64 a, u64 b, u64 bs, u64 np) {
    return shoup_lo<SU, false>(0, a, b, mulhi64_t<SU>(a, bs), np);
}
// add + a*b mod p, lazily: the addend enters the product chain (below add + 2p)
template <bool SU = false>
FHE_HD u64 mul_shoup_lazy_add_n(u64 add, u64 a, u64 b, u64 bs, u64 np) {
    return shoup_lo<SU, true>(add, a, b, mulhi64_t<SU>(a, bs), np);
}
FHE_HD u64 add_mod_n(u64 a, u64 b, const PM &m) { return csub_n(a + b, m.p, m.np); }

// Harvey lazy butterflies, M/ntt/native.rs:256-269 / 288-300.
template <bool SU = false>
FHE_HD void fwd_butterfly(u64 &x, u64 &y, u64 w, u64 ws, const PM &m) {
    x = csub_n(x, m.p2, m.np2);
    // x + t with x as the addend of the product chain (v_mad_u64_u32 adds a 64-bit value for free), and
    // x + 2p - t = (2x + 2p) - (x + t): one 64-bit operation less than forming t, x + t and x + 2p - t separately
    const u64 xt = shoup_lo<SU>(x, y, w, mulhi64_t<SU>(y, ws), m.np);
    y = ((x << 1) + m.p2) - xt;
    x = xt;
}
// Forward butterflies for moduli below 2^60 (16p < 2^64): the conditional subtraction on x is not
// needed every stage.  With every value below b*p before a stage, both outputs are below (b+2)*p
// (t < 2p whatever y is); fwd_narrow_bound() tracks b over the stages of a transform whose input is
// below b0*p (1: canonical) and says where the one strong correction (x < 16p -> x < 4p) has to sit.
//
// FHE_APPROX_SHOUP (default on): the headroom also pays for a cheaper quotient.  floor(y * ws / 2^64) is formed
// from three of its four 32 x 32 partial products (y0 * ws0 only feeds a carry): the estimate is the true quotient
// or one less, so t = y*w - q'*p is below 3p instead of 2p -- one multiply and its register shuffling less per
// butterfly -- and every stage grows the bound by 3p: outputs x + t and x + 3p - t are below (b + 3) p.
constexpr int FWD_NARROW_STEP = FHE_APPROX_SHOUP ? 3 : 2;
constexpr int fwd_narrow_bound(int stage, int b0 = 1) {  // b before `stage`, b0 before stage 0
    int b = b0;
    for (int s = 0; s < stage; s++) b = (b > 16 - FWD_NARROW_STEP ? 4 : b) + FWD_NARROW_STEP;
    return b;
}
constexpr bool fwd_narrow_corrects(int stage, int b0 = 1) { return fwd_narrow_bound(stage, b0) > 16 - FWD_NARROW_STEP; }
// floor(a * s / 2^64) or one less: the partial product a0 * s0 is left out (it contributes at most a carry of 1)
template <bool SU = false>
FHE_HD u64 mulhi64_approx(u64 a, u64 s) {
#if FHE_HAVE_MAD_CARRY
    return mulhi64_approx_c<SU>(a, s);
#else
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
    const u64 m = (u64)a0 * s1;
    const u64 m2 = (u64)a1 * s0 + (uint32_t)m;
    return (u64)a1 * s1 + (m >> 32) + (m2 >> 32);
#endif
}
template <bool SU = false>
FHE_HD void fwd_butterfly_narrow(u64 &x, u64 &y, u64 w, u64 ws, const PM &m, bool correct) {
    if (correct) {  // x < 16p -> < 4p
        const u64 p4 = m.p2 << 1, p8 = m.p2 << 2, np4 = m.np2 << 1, np8 = m.np2 << 2;
        x = csub_n(x, p8, np8);
        x = csub_n(x, p4, np4);
    }
    // (x + t formed inside the product chain, x + pk - t as (2x + pk) - (x + t): see fwd_butterfly; 2x may wrap
    // around 2^64, the difference is exact because x + pk - t < 16p < 2^64)
#if FHE_APPROX_SHOUP
    const u64 xt = shoup_lo<SU>(x, y, w, mulhi64_approx<SU>(y, ws), m.np);   // t below 3p
    const u64 pk = m.p2 + m.p;
#else
    const u64 xt = shoup_lo<SU>(x, y, w, mulhi64_t<SU>(y, ws), m.np);
    const u64 pk = m.p2;
#endif
#if defined(FHE_HOST_EMULATION)
    if (xt - x >= pk || x > ~0ull - pk) __builtin_trap();  // range tracking broken
#endif
    y = ((x << 1) + pk) - xt;
    x = xt;
}
template <bool SU = false>
FHE_HD void inv_butterfly(u64 &x, u64 &y, u64 z, u64 zs, const PM &m) {
    u64 t = x;
    x = csub_n(y + t, m.p2, m.np2);
    y = mul_shoup_lazy_n<SU>(m.p2 + t - y, z, zs, m.np);
}

FHE_HD u64 splitmix64(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// RnsScaler::scale's 256-bit fixed-point sums (ethnum::U256 in M/rns/scaler.rs:260-313): sums of 64 x 128-bit products
// held as five 64-bit columns while terms are added and resolved into 32-bit limbs once (cols5_limbs below).
// (Rounds 1-2 kept them as U256 / four u128 columns in C; the history up to commit b60ee88 has that form.)
// The carry handling is the multiplier's own: the eight 32 x 32 partial products
// of r * (lo | hi << 64) go straight into five 64-bit column accumulators (weights 2^0, 2^32, ..., 2^128) THROUGH
// v_mad_u64_u32's addend, and each accumulator's carry-out is banked in a 32-bit counter by one v_addc -- 17
// instructions per term, against ~35 (plus wait states) for a u128 formulation, whose zero-extending adds the
// compiler expands into add/addc chains and register moves.  The hazard recognizer does not see inside asm: a
// VALU-written SGPR needs two wait states before a VALU reads it as carry-in; the instruction order provides them
// (one s_nop before the last addc).  Exact: value = sum_k (c_k + o_k 2^64) 2^(32 k).
struct Cols5 {
    u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
};
// r: per-lane; (lo, hi): a WAVE-UNIFORM constant (SGPR operands, one constant-bus read per multiply).
// FIRST: the accumulator is empty (first term of a sum): the columns are written, not added to, and only the three
// columns that take two partial products can carry -- 11 instructions and no zeroed registers.
template <bool FIRST = false>
FHE_HD void cols5_mac_64x128(Cols5 &a, u64 r, u64 lo, u64 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t r0 = (uint32_t)r, r1 = (uint32_t)(r >> 32);
    const uint32_t t0 = (uint32_t)lo, t1 = (uint32_t)(lo >> 32), t2 = (uint32_t)hi, t3 = (uint32_t)(hi >> 32);
    u64 sa, sb, sc;  // carry-outs (SGPR pairs)
    if constexpr (FIRST) {
        u64 sx;
        asm("v_mad_u64_u32 %[c0], %[sx], %[r0], %[t0], 0\n\t"
            "v_mad_u64_u32 %[c1], %[sx], %[r0], %[t1], 0\n\t"
            "v_mad_u64_u32 %[c2], %[sx], %[r0], %[t2], 0\n\t"
            "v_mad_u64_u32 %[c3], %[sx], %[r0], %[t3], 0\n\t"
            "v_mad_u64_u32 %[c1], %[sa], %[r1], %[t0], %[c1]\n\t"
            "v_mad_u64_u32 %[c2], %[sb], %[r1], %[t1], %[c2]\n\t"
            "v_mad_u64_u32 %[c3], %[sc], %[r1], %[t2], %[c3]\n\t"
            "v_mad_u64_u32 %[c4], %[sx], %[r1], %[t3], 0\n\t"
            "v_addc_co_u32 %[o1], vcc, 0, 0, %[sa]\n\t"
            "v_addc_co_u32 %[o2], vcc, 0, 0, %[sb]\n\t"
            "v_addc_co_u32 %[o3], vcc, 0, 0, %[sc]"
            : [c0] "=&v"(a.c0), [c1] "=&v"(a.c1), [c2] "=&v"(a.c2), [c3] "=&v"(a.c3), [c4] "=&v"(a.c4), [o1] "=&v"(a.o1),
              [o2] "=&v"(a.o2), [o3] "=&v"(a.o3), [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc), [sx] "=&s"(sx)
            : [r0] "v"(r0), [r1] "v"(r1), [t0] "s"(t0), [t1] "s"(t1), [t2] "s"(t2), [t3] "s"(t3)
            : "vcc");
        a.o0 = 0, a.o4 = 0;
        return;
    }
    asm("v_mad_u64_u32 %[c0], %[sa], %[r0], %[t0], %[c0]\n\t"
        "v_mad_u64_u32 %[c1], %[sb], %[r0], %[t1], %[c1]\n\t"
        "v_mad_u64_u32 %[c2], %[sc], %[r0], %[t2], %[c2]\n\t"
        "v_addc_co_u32 %[o0], vcc, 0, %[o0], %[sa]\n\t"
        "v_mad_u64_u32 %[c1], %[sa], %[r1], %[t0], %[c1]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[sb]\n\t"
        "v_addc_co_u32 %[o2], vcc, 0, %[o2], %[sc]\n\t"
        "v_mad_u64_u32 %[c3], %[sb], %[r0], %[t3], %[c3]\n\t"
        "v_mad_u64_u32 %[c2], %[sc], %[r1], %[t1], %[c2]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[sa]\n\t"
        "v_mad_u64_u32 %[c3], %[sa], %[r1], %[t2], %[c3]\n\t"
        "v_addc_co_u32 %[o3], vcc, 0, %[o3], %[sb]\n\t"
        "v_addc_co_u32 %[o2], vcc, 0, %[o2], %[sc]\n\t"
        "v_mad_u64_u32 %[c4], %[sb], %[r1], %[t3], %[c4]\n\t"
        "v_addc_co_u32 %[o3], vcc, 0, %[o3], %[sa]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %[o4], vcc, 0, %[o4], %[sb]"
        : [c0] "+v"(a.c0), [c1] "+v"(a.c1), [c2] "+v"(a.c2), [c3] "+v"(a.c3), [c4] "+v"(a.c4), [o0] "+v"(a.o0),
          [o1] "+v"(a.o1), [o2] "+v"(a.o2), [o3] "+v"(a.o3), [o4] "+v"(a.o4), [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc)
        : [r0] "v"(r0), [r1] "v"(r1), [t0] "s"(t0), [t1] "s"(t1), [t2] "s"(t2), [t3] "s"(t3)   // (lo, hi): wave-uniform
        : "vcc");
#else  // host pass / host emulation: the same columns in plain C
    if (FIRST) a = Cols5{};
    const u64 rr[2] = {(uint32_t)r, r >> 32}, tt[4] = {(uint32_t)lo, lo >> 32, (uint32_t)hi, hi >> 32};
    u64 *const cs[5] = {&a.c0, &a.c1, &a.c2, &a.c3, &a.c4};
    uint32_t *const os[5] = {&a.o0, &a.o1, &a.o2, &a.o3, &a.o4};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 4; j++) {
            const u64 t = *cs[i + j] + rr[i] * tt[j];
            *os[i + j] += t < *cs[i + j];
            *cs[i + j] = t;
        }
#endif
}
// ---- 32-bit limb chains: what the scaler's glue (column resolves, 256-bit shifts, rounding) is written in since
// round 3.  The compiler expands u128 / 256-bit C arithmetic into zero-extensions, 64-bit shifts and compare-select
// carries (about half of scale_kernel's VALU instructions); a chain of add-with-carry on 32-bit limbs is one
// instruction per limb.
// a + b + k; k (0 or 1) is the carry in and out
FHE_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t &k) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned ko;
    const uint32_t r = __builtin_addc(a, b, k, &ko);
    k = ko;
    return r;
#else
    const u64 t = (u64)a + b + k;
    k = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
// a - b - k; k (0 or 1) is the borrow in and out
FHE_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t &k) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned ko;
    const uint32_t r = __builtin_subc(a, b, k, &ko);
    k = ko;
    return r;
#else
    const u64 t = (u64)a - b - k;
    k = (uint32_t)(t >> 32) & 1;
    return (uint32_t)t;
#endif
}
// low word of {hi, lo} >> s, s in [0, 31]
FHE_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return s ? (lo >> s) | (hi << (32 - s)) : lo;
#endif
}
FHE_HD uint32_t lo32(u64 v) { return (uint32_t)v; }
FHE_HD uint32_t hi32(u64 v) { return (uint32_t)(v >> 32); }
FHE_HD u64 pack64(uint32_t lo, uint32_t hi) { return (u64)lo | ((u64)hi << 32); }
// ceil(X / 2) of a three-limb value
FHE_HD void ceil_half3(uint32_t &x0, uint32_t &x1, uint32_t &x2) {
    const uint32_t odd = x0 & 1;
    uint32_t k = 0;
    x0 = addc32(funnel32(x1, x0, 1), odd, k);
    x1 = addc32(funnel32(x2, x1, 1), 0, k);
    x2 = (x2 >> 1) + k;
}

// The value of a Cols5 accumulator, sum_k (c_k + o_k 2^64) 2^(32 k), mod 2^(32 NL) as limbs L[0 .. NL): two carry
// chains, (c0 | c2 << 64 | c4 << 128) + (c1 << 32 | c3 << 96) and + (o0 | o1 << 32 | ...) << 64.
template <int NL>
FHE_HD void cols5_limbs(const Cols5 &a, uint32_t (&L)[8]) {
    static_assert(NL >= 6 && NL <= 8, "limb count");
    const uint32_t A[8] = {lo32(a.c0), hi32(a.c0), lo32(a.c2), hi32(a.c2), lo32(a.c4), hi32(a.c4), 0, 0};
    const uint32_t B[8] = {0, lo32(a.c1), hi32(a.c1), lo32(a.c3), hi32(a.c3), 0, 0, 0};
    const uint32_t C[8] = {0, 0, a.o0, a.o1, a.o2, a.o3, a.o4, 0};
    uint32_t k = 0;
    L[0] = A[0];
#pragma unroll
    for (int i = 1; i < NL; i++) L[i] = addc32(A[i], B[i], k);
    k = 0;
#pragma unroll
    for (int i = 2; i < NL; i++) L[i] = addc32(L[i], C[i], k);
#pragma unroll
    for (int i = NL; i < 8; i++) L[i] = 0;
}


}  // namespace fhe
