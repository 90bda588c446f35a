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
    u64 xs = (s == 0) ? lo : ((lo >> s) | (hi << (64 - s)));  // x >> (k-1), < 2^(k+1)
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
    u64 xs = (s == 0) ? lo : ((lo >> s) | (hi << (64 - s)));
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

// 256-bit wrap-around accumulator == ethnum::U256 as used by RnsScaler::scale
// (M/rns/scaler.rs:260-313), held as two 128-bit halves so that additions compile to
// hardware carry chains (v_add_co / v_addc) instead of compare-and-select sequences.
struct U256 {
    u128_t lo, hi;
};
// acc +/-= r * (lo | hi << 64)   (mod 2^256)
FHE_HD void u256_mac_64x128(U256 &acc, u64 r, u64 lo, u64 hi, bool negate) {
    const u128_t p0 = (u128_t)r * lo, p1 = (u128_t)r * hi;  // product = p0 + (p1 << 64), < 2^192
    u128_t t_lo;
    const bool c = __builtin_add_overflow(p0, p1 << 64, &t_lo);
    const u128_t t_hi = (p1 >> 64) + (c ? 1 : 0);
    if (!negate) {
        const bool c2 = __builtin_add_overflow(acc.lo, t_lo, &acc.lo);
        acc.hi += t_hi + (c2 ? 1 : 0);
    } else {
        const bool b2 = __builtin_sub_overflow(acc.lo, t_lo, &acc.lo);
        acc.hi -= t_hi + (b2 ? 1 : 0);
    }
}
// The same sums without carry detection: 64-bit columns of the 64 x 128-bit products are added
// into separate 128-bit accumulators (a column sum of < 2^32 terms stays below 2^96, so the
// zero-extending adds cannot overflow and compile to plain add/addc chains); the 256-bit value
// c0 + c1*2^64 + c2*2^128 + c3*2^192 (mod 2^256) is formed once.
struct Cols256 {
    u128_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
};
// c += r * (lo | hi << 64)
FHE_HD void cols_mac_64x128(Cols256 &c, u64 r, u64 lo, u64 hi) {
    const u128_t p0 = (u128_t)r * lo, p1 = (u128_t)r * hi;
    c.c0 += (u64)p0;
    c.c1 += (u64)(p0 >> 64);
    c.c1 += (u64)p1;
    c.c2 += (u64)(p1 >> 64);
}
// c += (r * (lo | hi << 64)) << 64
FHE_HD void cols_mac_64x128_shl64(Cols256 &c, u64 r, u64 lo, u64 hi) {
    const u128_t p0 = (u128_t)r * lo, p1 = (u128_t)r * hi;
    c.c1 += (u64)p0;
    c.c2 += (u64)(p0 >> 64);
    c.c2 += (u64)p1;
    c.c3 += (u64)(p1 >> 64);
}
FHE_HD U256 cols_resolve(const Cols256 &c) {
    const u128_t m1 = c.c1 + (c.c0 >> 64);
    const u128_t m2 = c.c2 + (m1 >> 64);
    const u128_t m3 = c.c3 + (m2 >> 64);  // bits >= 2^256 fall off: U256 arithmetic wraps
    return U256{(u128_t)(u64)c.c0 | (m1 << 64), (u128_t)(u64)m2 | (m3 << 64)};
}
// The same sums on the device with the carry handling of the multiplier itself: the eight 32 x 32 partial products
// of r * (lo | hi << 64) go straight into five 64-bit column accumulators (weights 2^0, 2^32, ..., 2^128) THROUGH
// v_mad_u64_u32's addend, and each accumulator's carry-out is banked in a 32-bit counter by one v_addc -- 17
// instructions per term, against ~35 (plus wait states) for the u128 formulation above, whose zero-extending adds the
// compiler expands into add/addc chains and register moves.  The hazard recognizer does not see inside asm: a
// VALU-written SGPR needs two wait states before a VALU reads it as carry-in; the instruction order provides them
// (one s_nop before the last addc).  Exact: value = sum_k (c_k + o_k 2^64) 2^(32 k).
struct Cols5 {
    u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
};
// r: per-lane; (lo, hi): a WAVE-UNIFORM constant (SGPR operands, one constant-bus read per multiply).
FHE_HD void cols5_mac_64x128(Cols5 &a, u64 r, u64 lo, u64 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t r0 = (uint32_t)r, r1 = (uint32_t)(r >> 32);
    const uint32_t t0 = (uint32_t)lo, t1 = (uint32_t)(lo >> 32), t2 = (uint32_t)hi, t3 = (uint32_t)(hi >> 32);
    u64 sa, sb, sc;  // carry-outs (SGPR pairs)
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
// -> the 64-bit columns of Cols256 (weights 2^0, 2^64, 2^128, 2^192; each sum stays far below 2^128)
FHE_HD Cols256 cols5_to_cols256(const Cols5 &a) {
    Cols256 c;
    const u64 M32 = 0xffffffffull;
    c.c0 = (u128_t)a.c0 + ((a.c1 & M32) << 32);
    c.c1 = (u128_t)(a.c1 >> 32) + a.c2 + ((a.c3 & M32) << 32) + a.o0 + ((u64)a.o1 << 32);
    c.c2 = (u128_t)(a.c3 >> 32) + a.c4 + a.o2 + ((u64)a.o3 << 32);
    c.c3 = a.o4;
    return c;
}
FHE_HD U256 u256_sub(const U256 &a, const U256 &b) {  // wrapping
    U256 r;
    const bool borrow = __builtin_sub_overflow(a.lo, b.lo, &r.lo);
    r.hi = a.hi - b.hi - (borrow ? 1 : 0);
    return r;
}
// bits [s, s+128) of a, for 1 <= s <= 127
FHE_HD void u256_shr_lo128(const U256 &a, uint32_t s, u64 &lo, u64 &hi) {
    const u128_t v = (a.lo >> s) | (a.hi << (128 - s));
    lo = (u64)v;
    hi = (u64)(v >> 64);
}
FHE_HD U256 u256_not(const U256 &a) { return U256{~a.lo, ~a.hi}; }
// any bit at position >= 191 set (the reference's sign test, scaler.rs:303)
FHE_HD bool u256_ge_2_191(const U256 &a) { return (a.hi >> 63) != 0; }

}  // namespace fhe
