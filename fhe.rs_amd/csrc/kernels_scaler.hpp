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
    const u64 *w_tab;                                    // [nto][32]  +w: k * 2^64 mod q;  -w: (1 - (k + 1) 2^64) mod q at 16 + k
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
    uint32_t wide_w;     // |t| can reach 2^191 (host-side bound, scaler_upload): the launch takes the WIDE_W instance
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
// C2_CARRY = false: the caller guarantees that the sum of the upper partial products xh * yh cannot leave 64 bits
// (scale_kernel: one term below 2^62 and at most 11 below 2^60), so the third column keeps no overflow counter.
// FIRST: the accumulator is zero (first term of a sum): only the second cross product can carry.
template <bool C2_CARRY = true, bool FIRST = false>
FHE_HD void mac3x64(Acc3x64 &a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    u64 s0, s1, s2;  // carry-outs (SGPR pairs)
    if constexpr (FIRST) {
        asm("v_mad_u64_u32 %[c1], %[s1], %[xl], %[yh], 0\n\t"
            "v_mad_u64_u32 %[c0], %[s0], %[xl], %[yl], 0\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[xh], %[yh], 0\n\t"
            "v_mad_u64_u32 %[c1], %[s0], %[xh], %[yl], %[c1]\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %[o1], vcc, 0, 0, %[s0]"
            : [c0] "=&v"(a.c0), [c1] "=&v"(a.c1), [c2] "=&v"(a.c2), [o1] "=&v"(a.o1),
              [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [xl] "v"(xl), [xh] "v"(xh), [yl] "s"(yl), [yh] "s"(yh)
            : "vcc");
        a.o0 = 0, a.o2 = 0;
    } else if constexpr (C2_CARRY) {
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
    } else {
        asm("v_mad_u64_u32 %[c0], %[s0], %[xl], %[yl], %[c0]\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[xl], %[yh], %[c1]\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[xh], %[yh], %[c2]\n\t"
            "v_addc_co_u32 %[o0], vcc, 0, %[o0], %[s0]\n\t"
            "v_mad_u64_u32 %[c1], %[s0], %[xh], %[yl], %[c1]\n\t"
            "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s1]\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s0]"
            : [c0] "+v"(a.c0), [c1] "+v"(a.c1), [c2] "+v"(a.c2), [o0] "+v"(a.o0), [o1] "+v"(a.o1),
              [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [xl] "v"(xl), [xh] "v"(xh), [yl] "s"(yl), [yh] "s"(yh)
            : "vcc");
    }
#else  // host pass / host emulation: the same columns in plain C
    const u64 xl = (uint32_t)x, xh = x >> 32, yl = (uint32_t)y, yh = y >> 32;
    const u64 pr[4] = {xl * yl, xl * yh, xh * yh, xh * yl};
    u64 *const cs[4] = {&a.c0, &a.c1, &a.c2, &a.c1};
    uint32_t *const os[4] = {&a.o0, &a.o1, &a.o2, &a.o1};
    for (int k = 0; k < 4; k++) {
        const u64 t = *cs[k] + pr[k];
        if (!C2_CARRY && k == 2 && t < *cs[k]) __builtin_trap();   // the caller's bound on the third column is wrong
        *os[k] += t < *cs[k];
        *cs[k] = t;
    }
#endif
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
// PLAIN: a factor-one scaler whose v fits one word (is_one && v_fits_64, every basis extension of the BFV
// parameter sets): no w, no v_hi -- the instance carries none of that code.
// Round 3: everything between the multiply-add blocks (column resolves, the 256-bit shifts, rounding, the small
// addends of the output sums) is written on 32-bit limbs with add-with-carry chains (zq_dev.hpp): the u128 / U256 C
// of rounds 1-2 compiled to roughly as many instructions as the multiplies themselves.
// WIDE_W: the fast path takes w's sign from bit 255 of t and keeps 68 bits of w -- equal to the
// reference's `t >> 191 > 0` test and 128-bit w (scaler.rs:303-313) exactly as long as |t| < 2^191, which the host
// proves from the moduli and thetas for every BFV scaler (the theta_omega of the extension moduli are zero).  A
// non-unit factor over many wide moduli can leave that range; there the reference's result is defined by its bit
// tests, not by the mathematics, and this instance reproduces them: sign = any of bits 191 ... 255, w = the low 128
// bits of (sign ? ~t : t) >> 126, +1 (wrapping) and halved / halved upwards, reduced mod q per target.
template <int NF, bool PLAIN, bool WIDE_W = false>
__global__ void __launch_bounds__(256, NF <= 4 ? 8 : 1)   // (NF <= 4: 64 VGPRs / 8 waves per SIMD measured 3 % faster)
    scale_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                             u64 out_poly_stride, ScalerDev s, const DevMod *__restrict__ to_mods, uint32_t logn,
                             u64 total, uint32_t ascending) {
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    // Columns are handed out from the LAST polynomial backwards: the kernel that wrote `in` (an inverse NTT, the
    // fused tensor kernel) went through the polynomials in ascending order, so its most recent output is what
    // still sits in the 256 MiB Infinity Cache; and the forward NTT that follows this kernel (ascending again)
    // starts on what was written here last.  Same-box A/B: -1.2 % per ct x ct step, -3 % on that forward NTT.
    // (`ascending`: the caller knows that the producer ran backwards -- the multiply's tensor kernel since round 4)
    if (!ascending) gid = total - 1 - gid;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *src = in + poly * in_poly_stride + col;
    u64 rests[NF];
#pragma unroll
    for (int i = 0; i < NF; i++) rests[i] = (uint32_t)i < s.nfrom ? load_last<(FHE_PIPE_NT & 1) != 0>(src + (u64)i * n) : 0;

    // (all per-source tables are zero-padded to NF entries by the host, scaler_upload: the term loops run without
    // per-term bounds checks -- a padded term multiplies a zero residue by a zero constant -- so the constants of
    // a sum are fetched together, one scalar wait per sum instead of one per term)
    // ---- v = ceil((sum_i r_i theta_i >> (shift - 1)) / 2), scaler.rs:260-276.  The sum is below 2^198 (seven limbs);
    // shift - 1 is in [96, 126] (scaler_upload checks), so the bits wanted start in limb 3; v is below 2^70.
    u64 vlo;
    uint32_t vhi;
    {
        Cols5 vc;
        cols5_mac_64x128<true>(vc, rests[0], s.theta_garner_lo[0], s.theta_garner_hi[0]);
#pragma unroll
        for (int i = 1; i < NF; i++) cols5_mac_64x128(vc, rests[i], s.theta_garner_lo[i], s.theta_garner_hi[i]);
        uint32_t V[8];
        cols5_limbs<7>(vc, V);
        const uint32_t b = (s.shift - 1) & 31;
        uint32_t x0 = funnel32(V[4], V[3], b), x1 = funnel32(V[5], V[4], b), x2 = funnel32(V[6], V[5], b);
        ceil_half3(x0, x1, x2);
        vlo = pack64(x0, x1);
        vhi = x2;
    }
    u64 wx = 0;            // w's low word, complemented when w is subtracted (see the output sums)
    uint32_t widx = 0;     // index of w's contribution in w_tab: 16 * (w subtracted) + (w >> 64)
    u64 ww_lo = 0, ww_hi = 0;   // WIDE_W: the reference's 128-bit w
    bool ww_neg = false;
    if constexpr (!PLAIN) {
        if (!s.is_one) {
            // t = sum_i +/- r_i * theta_omega_i  -/+  v * theta_gamma  (mod 2^256, scaler.rs:278-301).  ONE accumulator:
            // a subtracted term  -x * theta  is written  (~x) * theta - (2^64 - 1) * theta  (mod 2^256), so every term
            // is an addition of (x ^ mask) * theta with a wave-uniform mask of 0 or ~0, and the constants
            // (2^64 - 1) * theta of the subtracted terms are one 256-bit constant the host summed (ScalerDev::w_const).
            // (the v * theta_gamma term first -- low word of v here, (high word) << 64 below; subtracted unless
            // theta_gamma_sign -- so that the accumulator starts from a term that is always there)
            const u64 gmask = s.theta_gamma_sign ? 0ull : ~0ull;
            Cols5 acc5;
            cols5_mac_64x128<true>(acc5, vlo ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
#pragma unroll
            for (int i = 0; i < NF; i++) {
                // theta_omega_i = 0 whenever the scaled Garner coefficient is an integer -- e.g. every
                // source modulus outside the denominator when scaling Q*P -> Q by t/Q (5 of C2's 9)
                const u64 tlo = s.theta_omega_lo[i], thi = s.theta_omega_hi[i];
                if ((tlo | thi) == 0) continue;
                cols5_mac_64x128(acc5, rests[i] ^ s.theta_omega_mask[i], tlo, thi);
            }
            uint32_t W[8];
            cols5_limbs<8>(acc5, W);
            {
                Cols5 h5;
                cols5_mac_64x128<true>(h5, (u64)vhi ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
                uint32_t H[8];
                cols5_limbs<6>(h5, H);     // (<< 64 mod 2^256 keeps 192 bits of it)
                uint32_t k = 0;
#pragma unroll
                for (int i = 2; i < 8; i++) W[i] = addc32(W[i], H[i - 2], k);
            }
            {
                uint32_t k = 0;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    W[i] = subb32(W[i], (i & 1) ? hi32(s.w_const[i >> 1]) : lo32(s.w_const[i >> 1]), k);
            }
            // w = ceil(X / 2) with X = (t negative ? ~t : t) >> 126  -- scaler.rs:303-313 writes the negative case as
            // ((!t >> 126) + 1) >> 1, which is the same rounding; |w| < 2^68, so three limbs of X are enough
            if constexpr (WIDE_W) {
                // scaler.rs:303-313 to the letter (see the template's comment)
                ww_neg = ((W[5] >> 31) | W[6] | W[7]) != 0;
                const uint32_t mm = ww_neg ? ~0u : 0u;
                // X = the low 128 bits of (t ^ mm) >> 126 (`as_u128()`): bit 126 is bit 30 of limb 3; the two top bits
                // of the shifted value (bits 254, 255 of t ^ mm) would land in bits 128, 129 and are dropped
                const u128_t X =
                    ((u128_t)pack64(funnel32(W[6] ^ mm, W[5] ^ mm, 30), funnel32(W[7] ^ mm, W[6] ^ mm, 30)) << 64) |
                    pack64(funnel32(W[4] ^ mm, W[3] ^ mm, 30), funnel32(W[5] ^ mm, W[4] ^ mm, 30));
                const u128_t w128 = ww_neg ? (X + 1) >> 1 : (X >> 1) + (X & 1);
                ww_lo = (u64)w128, ww_hi = (u64)(w128 >> 64);
            }
            const uint32_t m = WIDE_W ? 0u : (uint32_t)((int32_t)W[7] >> 31);   // all ones: t is negative, w is subtracted
            uint32_t x0 = funnel32(W[4] ^ m, W[3] ^ m, 30), x1 = funnel32(W[5] ^ m, W[4] ^ m, 30),
                     x2 = funnel32(W[6] ^ m, W[5] ^ m, 30);
            ceil_half3(x0, x1, x2);
            // -w = -(w_hi 2^64 + w_lo) = ~w_lo + (1 - (w_hi + 1) 2^64): the complemented low word goes into the output
            // sums as it is, the rest is a per-target table entry (w_tab, 16 + w_hi)
            wx = WIDE_W ? 0 : pack64(x0 ^ m, x1 ^ m);
            widx = (m & 16) + (x2 & 15);
        }
    }
    const uint32_t vh = vhi & 15;
    u64 *o = out + poly * out_poly_stride + col;
    // third column of the output sums: v_hi32 * gamma_hi < 2^62 and NF products of two upper words below 2^30 each
    constexpr bool C2C = NF > 11;
    for (uint32_t jt = s.ncommon; jt < s.nto; jt++) {
        const DevMod q = to_mods[jt];
        const u64 *om = s.omega + (u64)jt * NF;   // rows zero-padded to NF
        Acc3x64 a;
        mac3x64<C2C, true>(a, vlo, s.gamma_neg[jt]);           // -v_lo * gamma
#pragma unroll
        for (int i = 0; i < NF; i++) mac3x64<C2C>(a, rests[i], om[i]);
        // S = c0 + c1 2^32 + (c2 + o0) 2^64 + o1 2^96 (+ o2 2^128) + E, E = the small addends (below 2^65):
        // limbs l0..l3 and `top`
        uint32_t l0, l1, l2, l3, top;
        if constexpr (PLAIN) {
            uint32_t k = 0;
            l0 = lo32(a.c0);
            l1 = addc32(hi32(a.c0), lo32(a.c1), k);
            l2 = addc32(lo32(a.c2), hi32(a.c1), k);
            l3 = addc32(hi32(a.c2), a.o1, k);
            top = k + (C2C ? a.o2 : 0);
            k = 0;
            l2 = addc32(l2, a.o0, k);
            l3 = addc32(l3, 0, k);
            top += k;
        } else {
            // -v_hi * 2^64 * gamma (< q) through a 16-entry table -- a per-lane load, skipped when the host-side
            // bound on v (scaler_upload: v <= sum_i (q_i - 1) + 1) says v_hi is always zero; +/- w: its high part and
            // the constant of the complement through w_tab, its (complemented) low word straight into the sum
            u64 small = s.v_fits_64 ? 0 : s.vhi_tab[jt * 16 + vh];
            if constexpr (WIDE_W) {
                // +/- (w mod q): any representative of the class gives the same canonical output
                if (!s.is_one) {
                    const u64 wr = reduce_u128(ww_hi, ww_lo, q);
                    small += ww_neg ? (wr ? q.p - wr : 0) : wr;
                }
            } else {
                if (!s.is_one) small += s.w_tab[jt * 32 + widx];   // below 2q
            }
            uint32_t k = 0;
            const uint32_t e0 = addc32(lo32(wx), lo32(small), k), e1 = addc32(hi32(wx), hi32(small), k), e2 = k;
            k = 0;
            l0 = addc32(lo32(a.c0), e0, k);
            l1 = addc32(hi32(a.c0), lo32(a.c1), k);
            l2 = addc32(lo32(a.c2), hi32(a.c1), k);
            l3 = addc32(hi32(a.c2), a.o1, k);
            top = k + (C2C ? a.o2 : 0);
            k = 0;
            l1 = addc32(l1, e1, k);
            l2 = addc32(l2, a.o0 + e2, k);
            l3 = addc32(l3, 0, k);
            top += k;
        }
        u64 lo = pack64(l0, l1), hi = pack64(l2, l3);
        u64 r;
        if ((s.narrow_mask >> (jt & 63)) & 1) {
            // the whole sum is < 2^(2k+1) (hence top == 0): the single-word Barrett of zq_dev.hpp does it
            r = barrett_reduce_wide(hi, lo, q);
        } else if ((s.fold_mask >> (jt & 63)) & 1) {
            // < 2^(2k+6): replace the bits above 2^(2k) by their residue (64-entry table), which leaves
            // < 2^(2k) + q < 2^(2k+1) for the same single-word Barrett
            const uint32_t f = 2 * q.k;
            uint32_t idx;
            if (f >= 64) {
                idx = (uint32_t)(hi >> (f - 64));
                hi &= (1ull << (f - 64)) - 1;
            } else {
                idx = (uint32_t)((lo >> f) | (hi << (64 - f)));   // (f >= 2; the sum is below 2^(f+6))
                lo &= (1ull << f) - 1;
                hi = 0;
            }
            const u64 t = s.fold_tab[jt * 64 + idx];
            uint32_t k = 0;
            const uint32_t m0 = addc32(lo32(lo), lo32(t), k), m1 = addc32(hi32(lo), hi32(t), k);
            hi += k;
            r = barrett_reduce_wide(hi, pack64(m0, m1), q);
        } else {
            r = reduce_u128(hi, lo, q);    // [0, q)
            r = csub_n(r + s.c128_tab[jt * 16 + (top & 15)], q.p, q.np);
        }
        o[(u64)jt * n] = r;
    }
}

}  // namespace k
}  // namespace fhe
