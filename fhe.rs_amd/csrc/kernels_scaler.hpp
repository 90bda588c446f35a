// kernels_scaler.hpp -- RnsScaler::scale per coefficient column (scale_kernel) and its multi-word accumulators.
#pragma once
#include "kernels_common.hpp"

namespace fhe {
namespace k {

// ------------------------------------------------------------------ RNS scaler ----
struct ScalerDev {
    const u64 *gamma_neg;                                // [nto]      (q - gamma) mod q
    const u64 *omega;                                    // [nto][nfrom]
    const u64 *vhi_tab;                                  // [nto][16]  k * 2^64 * gamma_neg mod q
    const u64 *c64_tab;                                  // [nto][16]  k * 2^64  mod q
    const u64 *c128_tab;                                 // [nto][16]  k * 2^128 mod q
    const u64 *theta_omega_lo, *theta_omega_hi;          // [nfrom]
    const u64 *theta_omega_sign;                         // [nfrom] (0/1)
    const u64 *theta_omega_mask;                         // [nfrom] 0 (term added) or ~0 (term subtracted)
    u64 w_const[4];                                      // the constant the one-accumulator form of w subtracts (below)
    const u64 *theta_garner_lo, *theta_garner_hi;        // [nfrom]
    u64 theta_gamma_lo, theta_gamma_hi;
    u64 narrow_mask;  // bit j: the output sum for target modulus j provably stays below 2^(2k_j+1) (see scaler_upload)
    u64 fold_mask;    // bit j: it stays below 2^(2k_j+6): bits >= 2^(2k_j) are folded through fold_tab first
    const u64 *fold_tab;                                 // [nto][64]  i * 2^(2k_j) mod q_j
    uint32_t theta_gamma_sign, is_one, shift, nfrom, nto, ncommon;
    uint32_t v_fits_64;  // v < 2^64 for every input (factor-one scalers over few moduli): no v_hi term
};

// Sum of 64x64-bit products on the device: the four 32x32 partial products of a term go straight
// into three 64-bit column accumulators (weights 2^0, 2^32, 2^64) THROUGH v_mad_u64_u32's addend,
// and each accumulator's carry-out -- which the compiler never uses -- is banked in a 32-bit
// overflow counter by a v_addc.  8 VALU instructions per term and no register shuffling, against
// 14 for the 128-bit formulation below (the multiply needs zero-extended register pairs there).
// The hazard recognizer does not see inside asm: a VALU-written SGPR needs two wait states
// before a VALU reads it as carry-in; the instruction order below provides them.
struct Acc3x64 {
    u64 c0 = 0, c1 = 0, c2 = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0;
};
// x: per-lane value; y: WAVE-UNIFORM constant (scaler tables): its halves are SGPR operands of the multiplies
// (one constant-bus read per instruction), which saves the two copies into VGPRs a "v" constraint costs per term.
FHE_HD void mac3x64(Acc3x64 &a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    u64 s0, s1, s2;  // carry-outs (SGPR pairs)
    asm("v_mad_u64_u32 %[c0], %[s0], %[xl], %[yl], %[c0]\n\t"
        "v_mad_u64_u32 %[c1], %[s1], %[xl], %[yh], %[c1]\n\t"
        "v_mad_u64_u32 %[c2], %[s2], %[xh], %[yh], %[c2]\n\t"
        "v_addc_co_u32 %[o0], vcc, 0, %[o0], %[s0]\n\t"
        "v_mad_u64_u32 %[c1], %[s0], %[xh], %[yl], %[c1]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s1]\n\t"
        "v_addc_co_u32 %[o2], vcc, 0, %[o2], %[s2]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s0]"
        : [c0] "+v"(a.c0), [c1] "+v"(a.c1), [c2] "+v"(a.c2), [o0] "+v"(a.o0), [o1] "+v"(a.o1), [o2] "+v"(a.o2),
          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
        : [xl] "v"(xl), [xh] "v"(xh), [yl] "s"(yl), [yh] "s"(yh)   // y: the wave-uniform constant, straight from SGPRs
        : "vcc");
#else  // host pass / host emulation: the same columns in plain C
    const u64 xl = (uint32_t)x, xh = x >> 32, yl = (uint32_t)y, yh = y >> 32;
    const u64 pr[4] = {xl * yl, xl * yh, xh * yh, xh * yl};
    u64 *const cs[4] = {&a.c0, &a.c1, &a.c2, &a.c1};
    uint32_t *const os[4] = {&a.o0, &a.o1, &a.o2, &a.o1};
    for (int k = 0; k < 4; k++) {
        const u64 t = *cs[k] + pr[k];
        *os[k] += t < *cs[k];
        *cs[k] = t;
    }
#endif
}
// value = (c0 + o0 2^64) + (c1 + o1 2^64) 2^32 + (c2 + o2 2^64) 2^64  ->  low 128 bits and the rest
FHE_HD void acc3x64_resolve(const Acc3x64 &a, u64 extra, u128_t &low, u64 &top) {
    const u128_t l = (u128_t)a.c0 + ((u128_t)a.c1 << 32) + extra;                     // < 2^98
    const u128_t m = (u128_t)a.c2 + a.o0 + ((u128_t)a.o1 << 32) + (l >> 64);         // weight 2^64, < 2^67
    low = (u128_t)(u64)l | (m << 64);
    top = (u64)(m >> 64) + a.o2;
}

// Sum of 64x64-bit products without carry detection: the low and the high 64-bit halves of the
// products are summed separately (each sum of up to 2^32 terms fits 96 bits, so a plain
// zero-extending 128-bit add never overflows and the compiler emits one add/addc chain, no
// compares); value = lo + (hi << 64), resolved once at the end.
struct Acc192 {
    u128_t lo = 0, hi = 0;
};
FHE_HD void mac192(Acc192 &acc, u64 a, u64 b) {
    const u128_t p = (u128_t)a * b;
    acc.lo += (u64)p;
    acc.hi += (u64)(p >> 64);
}
// -> low 128 bits and the bits above them (`top`)
FHE_HD void acc192_resolve(const Acc192 &acc, u128_t &low, u64 &top) {
    const u128_t mid = acc.hi + (acc.lo >> 64);
    low = (u128_t)(u64)acc.lo | (mid << 64);
    top = (u64)(mid >> 64);
}

// One lane per coefficient column (RnsScaler::scale, M/rns/scaler.rs:249-352).  The 256-bit
// fixed-point sums v and w are reproduced limb for limb (they define the rounding).  The
// per-target value y = -v*gamma (+/- w) + sum_j r_j*omega_j only matters mod q (the reference
// ends with reduce_u128), so instead of one Shoup product per term it is accumulated as
// exact 128-bit products in a 192-bit register and reduced ONCE (4 instead of 10 32-bit
// multiplies per term); the few bits of v, w and of the accumulator above 2^64 / 2^128 are
// folded through 16-entry tables (v, |w| < 2^68 and top < 16 for up to 64 source moduli).
// in: [npolys][nfrom][N] PowerBasis; out: rows [ncommon, nto) of [npolys][nto][N].
// NF >= nfrom: the column's residues are loaded once, together, into registers (coalesced
// along N; one batch of loads in flight); all scaler constants are wave-uniform scalar loads.
template <int NF>
__global__ void __launch_bounds__(256, NF <= 4 ? 8 : 1)   // (NF <= 4: 64 VGPRs / 8 waves per SIMD measured 3 % faster)
    scale_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                             u64 out_poly_stride, ScalerDev s, const DevMod *__restrict__ to_mods, uint32_t logn,
                             u64 total) {
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    // Columns are handed out from the LAST polynomial backwards: the kernel that wrote `in` (an inverse NTT, the
    // fused tensor kernel) went through the polynomials in ascending order, so its most recent output is what
    // still sits in the 256 MiB Infinity Cache; and the forward NTT that follows this kernel (ascending again)
    // starts on what was written here last.  Same-box A/B: -1.2 % per ct x ct step, -3 % on that forward NTT.
    gid = total - 1 - gid;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *src = in + poly * in_poly_stride + col;
    u64 rests[NF];
#pragma unroll
    for (int i = 0; i < NF; i++) rests[i] = (uint32_t)i < s.nfrom ? src[(u64)i * n] : 0;

    // (all per-source tables are zero-padded to NF entries by the host, scaler_upload: the term loops run without
    // per-term bounds checks -- a padded term multiplies a zero residue by a zero constant -- so the constants of
    // a sum are fetched together, one scalar wait per sum instead of one per term)
    Cols5 vc;
#pragma unroll
    for (int i = 0; i < NF; i++) cols5_mac_64x128(vc, rests[i], s.theta_garner_lo[i], s.theta_garner_hi[i]);
    const U256 sum = cols_resolve(cols5_to_cols256(vc));
    u64 vlo, vhi;
    u256_shr_lo128(sum, s.shift - 1, vlo, vhi);
    {  // v = div_ceil(v, 2)
        const u64 odd = vlo & 1;
        vlo = (vlo >> 1) | (vhi << 63);
        vhi >>= 1;
        vlo += odd;
        vhi += (vlo < odd);
    }
    u64 wlo = 0, whi = 0;
    bool w_sign = false;
    if (!s.is_one) {
        // t = sum_i +/- r_i * theta_omega_i  -/+  v * theta_gamma  (mod 2^256, scaler.rs:278-301).  ONE accumulator:
        // a subtracted term  -x * theta  is written  (~x) * theta - (2^64 - 1) * theta  (mod 2^256), so every term is
        // an addition of (x ^ mask) * theta with a wave-uniform mask of 0 or ~0, and the constants
        // (2^64 - 1) * theta of the subtracted terms are one 256-bit constant the host summed (ScalerDev::w_const).
        // Round 3: the two-accumulator form (added and subtracted terms summed separately) chose its accumulator by a
        // uniform branch per term, and every merge of the two paths cost a copy of the ten accumulator registers.
        Cols5 acc5;
#pragma unroll
        for (int i = 0; i < NF; i++) {
                // theta_omega_i = 0 whenever the scaled Garner coefficient is an integer -- e.g. every
                // source modulus outside the denominator when scaling Q*P -> Q by t/Q (5 of C2's 9)
                const u64 tlo = s.theta_omega_lo[i], thi = s.theta_omega_hi[i];
                if ((tlo | thi) == 0) continue;
                cols5_mac_64x128(acc5, rests[i] ^ s.theta_omega_mask[i], tlo, thi);
            }
        // v * theta_gamma (128 x 128 -> 256 wrapping): low word of v, then (high word) << 64; subtracted unless
        // theta_gamma_sign
        const u64 gmask = s.theta_gamma_sign ? 0ull : ~0ull;
        cols5_mac_64x128(acc5, vlo ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
        Cols256 acc = cols5_to_cols256(acc5);
        cols_mac_64x128_shl64(acc, vhi ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
        const U256 wk{(u128_t)s.w_const[0] | ((u128_t)s.w_const[1] << 64), (u128_t)s.w_const[2] | ((u128_t)s.w_const[3] << 64)};
        const U256 t = u256_sub(cols_resolve(acc), wk);
        w_sign = u256_ge_2_191(t);
        if (w_sign) {
            u256_shr_lo128(u256_not(t), 126, wlo, whi);
            wlo += 1;
            whi += (wlo == 0);
            wlo = (wlo >> 1) | (whi << 63);
            whi >>= 1;
        } else {
            u256_shr_lo128(t, 126, wlo, whi);
            const u64 odd = wlo & 1;
            wlo = (wlo >> 1) | (whi << 63);
            whi >>= 1;
            wlo += odd;
            whi += (wlo < odd);
        }
    }
    const uint32_t vh = (uint32_t)vhi & 15, wh = (uint32_t)whi & 15;
    u64 *o = out + poly * out_poly_stride + col;
    for (uint32_t jt = s.ncommon; jt < s.nto; jt++) {
        const DevMod q = to_mods[jt];
        const u64 *om = s.omega + (u64)jt * NF;   // rows zero-padded to NF
        Acc3x64 a192;
        u128_t extra = 0;                                      // small addends of the sum (< 2^66)
        mac3x64(a192, vlo, s.gamma_neg[jt]);                   // -v_lo * gamma
        // -v_hi * 2^64 * gamma (< q) through a 16-entry table -- a per-lane load, skipped when the host-side bound
        // on v (scaler_upload: v <= sum_i (q_i - 1) + 1) says v_hi is always zero
        u64 small = s.v_fits_64 ? 0 : s.vhi_tab[jt * 16 + vh];
        if (!s.is_one) {
            // +/- w = +/- (w_hi * 2^64 + w_lo): the high part through the table, the low word straight
            // into the 192-bit sum -- as w_lo, or as K - w_lo with K = q * ceil(2^64 / q) = 2^64 + K_lo = 0 (mod q)
            const u64 c = s.c64_tab[jt * 16 + wh];             // w_hi * 2^64 mod q
            small += w_sign ? (c ? q.p - c : 0) : c;           // < 2q
            const u64 k_lo = q.p * (q.brt_hi + 1);             // K mod 2^64 (K >= 2^64 > w_lo)
            extra = w_sign ? ((((u128_t)1 << 64) | k_lo) - wlo) : (u128_t)wlo;
        }
#pragma unroll
        for (int i = 0; i < NF; i++) mac3x64(a192, rests[i], om[i]);
        extra += small;
        // (extra < 2^66 does not fit the u64 parameter: split it)
        u128_t acc;
        u64 top;
        acc3x64_resolve(a192, (u64)extra, acc, top);
        {
            const u128_t hi_extra = (extra >> 64) << 64;       // at most 3 * 2^64
            const bool c = __builtin_add_overflow(acc, hi_extra, &acc);
            top += c ? 1 : 0;
        }
        u64 r;
        if ((s.narrow_mask >> (jt & 63)) & 1) {
            // the whole sum is < 2^(2k+1) (hence top == 0): the single-word Barrett of zq_dev.hpp does it
            r = barrett_reduce_wide((u64)(acc >> 64), (u64)acc, q);
        } else if ((s.fold_mask >> (jt & 63)) & 1) {
            // < 2^(2k+6): replace the bits above 2^(2k) by their residue (64-entry table), which leaves
            // < 2^(2k) + q < 2^(2k+1) for the same single-word Barrett
            const uint32_t f = 2 * q.k;
            const uint32_t idx = (uint32_t)(acc >> f);
            acc = (acc & ((((u128_t)1) << f) - 1)) + s.fold_tab[jt * 64 + idx];
            r = barrett_reduce_wide((u64)(acc >> 64), (u64)acc, q);
        } else {
            r = reduce_u128((u64)(acc >> 64), (u64)acc, q);    // [0, q)
            r = csub_n(r + s.c128_tab[jt * 16 + ((uint32_t)top & 15)], q.p, q.np);
        }
        o[(u64)jt * n] = r;
    }
}

}  // namespace k
}  // namespace fhe
