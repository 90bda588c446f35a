// kernels_ks.hpp -- the fused key switch: ks_fused_kernel (rows that fit LDS) and ks_fused_split_kernel (sub-blocks).
#pragma once
#include "kernels_passes.hpp"

namespace fhe {
namespace k {

// ------------------------------------------------------------ fused key switch ----
// (c0, c1)[b][j] (+)= sum_i NTT_j( [p_i]_{q_j} ) (.) (k0, k1)[i][j]   for one (b, j) per workgroup.
// The digit rows p[b][i][:] are lifted (reduced mod q_j), transformed in LDS and multiplied
// into per-thread register accumulators; the key streams from L2/MALL (shared by the batch).
// A non-null addend0/addend1 is added to the respective output (fused relinearisation add,
// F/bfv/ops/mul.rs:224-225; rotation adds substitute(c0) to c0 only); canonical outputs.
// ks_threads_c(LOGN) threads; thread t owns the 16-byte chunks {c*T + t}, c < CH (or the single
// coefficient t when the row is smaller than one chunk per thread).
// (FHE_KS_TWPF=true: the transforms' per-lane twiddles requested one pass ahead at N = 8192 -- 112 VGPRs, no scratch,
// and no change in same-box A/B, profiles/r02_ks_twpf_ab.txt)
// (FHE_KS_PERSIST14=1: resident workgroups at N = 16384 as well -- no row-prefetch registers there, so nothing to
// overlap: C3 relinearise 108.1-109.7 k against 111.0-111.3 k ops/s, profiles/r02_ks_persist_ab.txt)
// TT (threads per workgroup, 0 = ks_threads_c(LOGN)): TT = N / 16 at N = 8192 is the two-workgroups-per-CU cut -- 512
// threads x 16 coefficients, BOTH accumulator sets in registers (64 VGPRs, as at N = 16384), only the 68 KiB row tile
// in LDS, so that a second workgroup is resident and runs its butterflies while this one sits in a barrier.
constexpr int ks_threads_tt(int logn, int tt) { return tt ? tt : ks_threads_c(logn); }
constexpr bool ks_acc1_in_lds_tt(int logn, int tt) { return tt == 0 && ks_acc1_in_lds_c(logn); }
// RNS (the host sets it for LOGN >= 12 when digit_arg says so): the digits are residue rows of same-width moduli
// (digit_shift_bits == 0, lift_mode == 1 -- relinearisation and Galois keys), lifted by one conditional subtraction.
// (TT = 512 at N = 16384, lab: 32 coefficients per thread and a 256-register budget -- two waves per SIMD -- so that
// both accumulator sets AND a radix-16 pass fit: 4 passes instead of 6, see engine.hpp FHE_LAB_KS14_T512)
constexpr int ks_min_waves(int logn, int tt) { return (logn == 14 && tt == 512) ? 2 : 4; }
// G0 > 0 (round 4; LOGN = 14: N = 32768 as two 16384-point halves, N = 65536 as four quarters): the tile is one of 2^G0
// parts of a row of 2^(LOGN+G0) points and the first G0 Cooley-Tukey stages are folded into the loader.  G0 = 1: half
// `sub` needs x[e] +/- w x[e + 2^LOGN] -- ONE Shoup product per coefficient (what a regular stage costs per butterfly
// pair) and two source reads, where ks_fused_split_kernel's 8192-point quarter rows pay three products and four reads
// for their two folded stages (C5 relinearise 1.46 -> 1.25 ms, profiles/r04_ks_half15_ab.txt); G0 = 2: three products
// instead of the seven of eight 8192-point parts.  The remaining LOGN stages run in LDS with twiddle base 2^G0 + sub.
// Workgroups are (ciphertext, key modulus, part).
// GAL (round 5): the instance GaloisKey::relinearize launches -- see `gal` below.  A template flag, not a runtime branch:
// with the gathers behind `if (gal)` EVERY instance spilled (72-96 B of scratch at N = 8192, 532-548 B at N = 16384, whose
// digit loop sits at 124 of 128 VGPRs); the relinearisation / multiply instances are the round-4 kernels, bit for bit.
// F64 = HR > 0 (round 6): every key modulus below 2^(53 - HR) (the reference's stock parameter sets: 36-49 bits): the digit
// transforms and both multiply-accumulates run on doubles holding integers (zq_f64.hpp).  Then `tw` is the key context's
// F64 twiddle table and k0 / k1 are the key words as doubles (Ksk::c0f, c1f; k0s / k1s are not read: the accumulate takes its
// quotient from h / p, which halves the key bytes pulled through L2); a digit row of
// another modulus of the basis IS a representative under this one -- there is no lift at all.  RNS digits, whole-row tiles
// (G0 = 0); TT = 512 at N = 8192 (radix-8 passes fit beside both accumulator sets once the per-lane twiddles are one word:
// 126 VGPRs, no scratch -- the two-workgroups-per-CU cut that lost with two-word twiddles is 6 % ahead here, engine.hpp
// launch_ks_fused), TT = 0 otherwise.  The integer instances (F64 = 0) are the round-5 kernels bit for bit (tests/test_isa_guards.py).
template <int LOGN, bool NARROW = false, int GM = KS_GMAX, int TT = 0, bool RNS = false, int G0 = 0, bool GAL = false, int F64 = 0>
__global__ void __launch_bounds__(ks_threads_tt(LOGN, TT), ks_min_waves(LOGN, TT))
    ks_fused_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0, u64 *__restrict__ out1,
                    u64 out_poly_stride, const u64 *__restrict__ addend0, const u64 *__restrict__ addend1,
                    u64 addend_poly_stride, const u64 *__restrict__ k0, const u64 *__restrict__ k0s,
                    const u64 *__restrict__ k1, const u64 *__restrict__ k1s, const DevMod *__restrict__ mods,
                    const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t lk, uint32_t digit_arg,
                    const u64 *__restrict__ xhat, u64 xhat_poly_stride, uint32_t total, uint32_t gal) {
    // gal (round 5; GAL instances only, otherwise ignored): GaloisKey::relinearize (F/bfv/keys/galois_key.rs:63-86) -- `xhat` and `addend0` are the
    // caller's UNPERMUTED Ntt rows (c1 and c0) and are read through the substitution x -> x^gal (galois_src_index):
    // the rotation's separate permutation pass is gone.  Block-uniform; only the item prologue and epilogue look at it.
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ks_threads_tt(LOGN, TT);
    constexpr int N = 1 << LOGN;             // the tile: the whole row, or (G0 = 1) one half of it
    constexpr u64 NROW = (u64)N << G0;       // coefficients of a row
    constexpr int NS = 1 << G0;
    static_assert(G0 == 0 || (G0 <= 2 && !ks_acc1_in_lds_tt(LOGN, TT) && tile_chunks_c(LOGN, ks_threads_tt(LOGN, TT)) >= 2),
                  "the folded first stages are written for the register-accumulator form (N = 16384 tiles)");
    static_assert(F64 == 0 || (RNS && G0 == 0 && !NARROW && tile_chunks_c(LOGN, ks_threads_tt(LOGN, TT)) >= 1),
                  "the F64 instances: RNS digits, whole-row tiles");
    // F64 bounds (units: zq_f64.hpp).  The transform leaves values below VB; they are reduced before the products when that
    // buys accumulator room (HR = 3, 4); a term is then below PB, the caller's own Ntt row (canonical) gives one below
    // POWN, and F64_CAP digits fit an accumulator together with a canonical addend before it must be reduced.
    constexpr int F64_VB = F64 ? f64_fwd_out_bound(LOGN, F64 ? F64 : 3) : 0;
    // (reduce the transformed values first only when the accumulators would otherwise hold fewer than sixteen terms)
    // (products by key words take their quotient from h / p -- mulmod2_add_f64: no k / p twin is read)
    constexpr bool F64_REDV = F64 > 0 && (f64_limit(F64 ? F64 : 3) - 2 * F64_ONE - F64_REDUCED) / f64_product_bound2(F64_VB, F64 ? F64 : 3) < 16;
    constexpr int F64_PB = F64 ? f64_product_bound2(F64_REDV ? F64_REDUCED : F64_VB, F64 ? F64 : 3) : 1;
    constexpr int F64_POWN = F64 ? f64_product_bound2(F64_ONE, F64 ? F64 : 3) : 1;
    constexpr int F64_CAP = F64 ? (f64_limit(F64 ? F64 : 3) - F64_ONE - F64_POWN - F64_REDUCED) / F64_PB : 1;
    static_assert(F64 == 0 || F64_CAP >= 8, "accumulator room");
    constexpr int CH = tile_chunks_c(LOGN, T);
    constexpr int NE = CH > 0 ? 2 * CH : 1;  // coefficients owned by a thread
    // GM: radix (log2) of the LDS passes.  8 everywhere but N = 16384, whose 1024 threads hold both accumulator sets in
    // registers (64 VGPRs): beside a per-lane-twiddle radix-8 pass that spills 24 VGPRs (100 B of scratch per lane,
    // 5 GB of extra HBM traffic per 512-polynomial launch, PMC).  GM_MIXED keeps radix 8 for the passes whose
    // twiddles are scalar and takes radix 4 for the rest (3+3+2+2+2+2 stages, 124 VGPRs, no scratch): C3 relinearise
    // 99.2 k (radix 8) -> 110.5 k (radix 4 throughout) -> 111.7-112.9 k ops/s.
    const uint32_t tid0 = threadIdx.x;
    // (an XCD-aware order that puts the lk workgroups of one polynomial on one L2, as tensor_intt_kernel
    // does, was measured: no change -- this kernel is nowhere near the HBM limit)
    // The (ciphertext, key modulus) items of a launch are dealt round-robin to the gridDim.x workgroups (`total` of
    // them; the host launches one workgroup per item except at N = 8192, where a workgroup owns its CU: there
    // gridDim.x is the number of CUs and, while an item's result is on its way out, the next item's first digit row
    // is already coming in -- neither that load nor a workgroup launch sits between two items).
    u64x2 pre[ks_acc1_in_lds_tt(LOGN, TT) ? CH : 1];
    bool have_pre = false;   // (block-uniform) `pre` already holds this item's first digit row
#if defined(FHE_HOST_EMULATION)
    constexpr bool ITEM_LOOP = true;    // (every size, so that the emulated suite walks the loop)
#else
    constexpr bool ITEM_LOOP = (LOGN == 13 && TT == 0) || (FHE_KS_PERSIST14 && LOGN == 14);
#endif
    uint32_t item = blockIdx.x;
    if (item >= total) return;
    FHE_TS_BEGIN();
    do {
    const uint32_t bj = item >> G0, sub = item & (NS - 1);
    const uint32_t b = to_sgpr(bj / lk), j = bj - b * lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = F64 ? make_pm_f64(md) : make_pm(md);
    const PF pf = pf_of(pm);                  // (F64 instances only)
    const u64x2 *twr = tw + (u64)j * NROW;
    const u64 suboff = (u64)sub * N;          // this half inside a row (0 when the tile is the row)
    // N = 8192: 1024 threads cap a thread at 128 VGPRs, which 2 x 16 accumulators plus a radix-8
    // pass do not fit; the c1 accumulators live in LDS behind the row tile instead (each thread
    // only ever touches its own 16-byte chunks, so no extra barrier).
    constexpr bool ACC1_LDS = ks_acc1_in_lds_tt(LOGN, TT);
    u64 acc0[NE], acc1[ACC1_LDS ? 1 : NE];
    u64x2 *const acc1_lds = reinterpret_cast<u64x2 *>(lds + lds_words(N));
#pragma unroll
    for (int e = 0; e < NE; e++) acc0[e] = 0;
    if constexpr (ACC1_LDS) {
#pragma unroll
        for (int c = 0; c < CH; c++) acc1_lds[c * T + tid0] = u64x2{0, 0};
    } else {
#pragma unroll
        for (int e = 0; e < NE; e++) acc1[e] = 0;
    }
    // `digit_arg` = digit_shift_bits | lift_mode << 8.  lift_mode says how far a source residue can exceed
    // the key moduli (host-side, from the moduli): 1 -> below 2 q_j (one conditional subtraction lifts
    // it), 2 -> below 4 q_j (two), 0 -> anything (Barrett).  RNS digits of same-width moduli are mode 1.
    const uint32_t digit_shift_bits = digit_arg & 0xff, lift_mode = digit_arg >> 8;
    constexpr bool rns_fast = RNS;
    auto lift = [&](u64 v) -> u64 {
        if (lift_mode == 1) return csub_n(v, p, pm.np);
        if (lift_mode == 2) return csub_n(csub_n(v, p2, pm.np2), p, pm.np);
        return reduce_u64(v, md);
    };
    // digit_shift_bits == 0: digit i is residue row i of p (RNS decomposition, :256-268).
    // otherwise: base-2^bits digits of the single row 0 (key_switch_decomposition, :323-362).
    const u64 *const src0 = pin + (u64)b * src_poly_stride;
    const u64 dstride = digit_shift_bits ? 0 : NROW;
    const u64 mask = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
    // `xhat` (callers that hold the digit polynomial in Ntt form -- relinearise, Galois, RGSW: [digits][N] per
    // polynomial over the ciphertext moduli, canonical): the RNS digit j reduced mod q_j is row j itself and its
    // transform under key modulus j IS xhat's row j (the ciphertext moduli are a prefix of the key moduli), so that
    // one of the L transforms of this workgroup is not computed: its product initialises the accumulators.
    const bool own = xhat != nullptr && digit_shift_bits == 0 && j < ndigits;   // (block-uniform)
    if (own) {
        const u64 koff = ((u64)j * lk + j) * NROW + suboff;
        const u64 *xr = xhat + (u64)b * xhat_poly_stride + (u64)j * NROW + suboff;
        if constexpr (CH > 0) {
            const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
            const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t ci = c * T + tid0;
                u64x2 v;
                if constexpr (GAL) {
                    const uint32_t d = (uint32_t)suboff + 2 * ci;
                    v.x = (xr - suboff)[galois_src_index(d, gal, LOGN + G0)];
                    v.y = (xr - suboff)[galois_src_index(d + 1, gal, LOGN + G0)];
                } else {
                    v = reinterpret_cast<const u64x2 *>(xr)[ci];
                }
                const u64x2 q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
                if constexpr (F64 > 0) {
                    const double vx = f64_from_u64(v.x), vy = f64_from_u64(v.y);
                    acc0[2 * c] = bits_of_f64(mulmod2_add_f64(0.0, vx, f64_of_bits(q0.x), pf));
                    acc0[2 * c + 1] = bits_of_f64(mulmod2_add_f64(0.0, vy, f64_of_bits(q0.y), pf));
                    const u64x2 af{bits_of_f64(mulmod2_add_f64(0.0, vx, f64_of_bits(q1.x), pf)),
                                   bits_of_f64(mulmod2_add_f64(0.0, vy, f64_of_bits(q1.y), pf))};
                    if constexpr (ACC1_LDS) {
                        acc1_lds[ci] = af;
                    } else {
                        acc1[ACC1_LDS ? 0 : 2 * c] = af.x;
                        acc1[ACC1_LDS ? 0 : 2 * c + 1] = af.y;
                    }
                    continue;
                }
                acc0[2 * c] = mul_shoup_lazy_n(v.x, q0.x, q0s.x, pm.np);        // below 2p, like every accumulator value
                acc0[2 * c + 1] = mul_shoup_lazy_n(v.y, q0.y, q0s.y, pm.np);
                const u64x2 a{mul_shoup_lazy_n(v.x, q1.x, q1s.x, pm.np), mul_shoup_lazy_n(v.y, q1.y, q1s.y, pm.np)};
                if constexpr (ACC1_LDS) {
                    acc1_lds[ci] = a;
                } else {
                    acc1[ACC1_LDS ? 0 : 2 * c] = a.x;
                    acc1[ACC1_LDS ? 0 : 2 * c + 1] = a.y;
                }
            }
        } else if (tid0 < N) {
            const u64 v = GAL ? xr[galois_src_index(tid0, gal, LOGN)] : xr[tid0];   // (CH == 0: whole small rows, G0 == 0)
            acc0[0] = mul_shoup_lazy_n(v, k0[koff + tid0], k0s[koff + tid0], pm.np);
            acc1[0] = mul_shoup_lazy_n(v, k1[koff + tid0], k1s[koff + tid0], pm.np);
        }
    }
    const uint32_t nloop = ndigits - (own ? 1u : 0u);          // digits that go through the transform
    auto digit_of = [&](uint32_t ii) -> uint32_t { return ii + ((own && ii >= j) ? 1u : 0u); };
    FHE_TSK(20);   // item prologue: index arithmetic, accumulator clears, the caller's own Ntt row (xhat)
    // The workgroup is alone on its CU (LDS), so nothing else hides the row load: digit i+1's
    // row is fetched into registers while digit i goes through its passes.
    constexpr bool PREFETCH = ks_acc1_in_lds_tt(LOGN, TT);   // (needs the VGPRs the LDS accumulators free)
    // (Feeding the first pass from these registers instead of staging the lifted row in LDS was
    // measured: 2.5 % slower -- the extra register shuffling outweighs the saved barrier.)
    if constexpr (PREFETCH) {
        if (nloop > 0 && !have_pre) {
            const u64x2 *first = reinterpret_cast<const u64x2 *>(src0 + (u64)digit_of(0) * dstride);
#pragma unroll
            for (int c = 0; c < CH; c++) pre[c] = first[c * T + tid0];
        }
    }
    uint32_t f64_terms = 1;   // (F64 instances: terms an accumulator holds since it was last reduced; the own row counts)
    for (uint32_t ii = 0; ii < nloop; ii++) {
        const uint32_t i = digit_of(ii);
        const uint32_t tid = opaque(tid0);
        const uint32_t sh = i * digit_shift_bits;
        FHE_TSK(0);
        // RNS instances: one conditional subtraction per coefficient.  The generic lambda keeps the shift, the mask
        // and three uniform branches PER ELEMENT (round 3, from the ISA).
        if constexpr (PREFETCH) {
            if constexpr (F64 > 0) {   // (a residue of another modulus of the basis is a representative as it is)
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const uint32_t e = 2 * (c * T + tid);
                    lds[padi(e)] = bits_of_f64(f64_from_u64(pre[c].x));
                    lds[padi(e + 1)] = bits_of_f64(f64_from_u64(pre[c].y));
                }
            } else if constexpr (rns_fast) {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const uint32_t e = 2 * (c * T + tid);
                    lds[padi(e)] = csub_n(pre[c].x, p, pm.np);
                    lds[padi(e + 1)] = csub_n(pre[c].y, p, pm.np);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const uint32_t e = 2 * (c * T + tid);
                    lds[padi(e)] = lift((pre[c].x >> sh) & mask);
                    lds[padi(e + 1)] = lift((pre[c].y >> sh) & mask);
                }
            }
        } else {
            // (address and mask are recomputed per digit on purpose: hoisted, they cost VGPRs that the
            // N = 16384 variant does not have)
            const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * NROW);
            if constexpr (G0 == 1) {
                // stage 0 of the 2N-point transform, only the branch that leads to this half: lo +/- w * hi on the lifted
                // values (canonical, so lo needs no correction): below 3p.  Four batches of CH / 4 chunks: 2 x CH / 4
                // loads of 16 bytes in flight next to the two accumulator sets.
                const u64 mask_i = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
                auto lf = [&](u64 v) -> u64 {
                    if constexpr (rns_fast) return csub_n(v, p, pm.np);
                    return lift((v >> sh) & mask_i);
                };
                const u64x2 w0 = twr[1];
                const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src);
                constexpr int HB = CH >= 4 ? CH / 4 : 1;
#pragma unroll
                for (int h = 0; h < CH; h += HB) {
                    u64x2 lo[HB], hi[HB];
#pragma unroll
                    for (int c = 0; c < HB; c++) {
                        lo[c] = s2[(h + c) * T + tid];
                        hi[c] = s2[(h + c) * T + tid + N / 2];
                    }
#pragma unroll
                    for (int c = 0; c < HB; c++) {
                        const uint32_t e = 2 * ((h + c) * T + tid);
                        const u64 ax = lf(lo[c].x), ay = lf(lo[c].y), bx = lf(hi[c].x), by = lf(hi[c].y);
                        if (sub) {   // (uniform over the workgroup)
                            lds[padi(e)] = ax + p2 - mul_shoup_lazy_n<true>(bx, w0.x, w0.y, pm.np);
                            lds[padi(e + 1)] = ay + p2 - mul_shoup_lazy_n<true>(by, w0.x, w0.y, pm.np);
                        } else {
                            lds[padi(e)] = mul_shoup_lazy_add_n<true>(ax, bx, w0.x, w0.y, pm.np);
                            lds[padi(e + 1)] = mul_shoup_lazy_add_n<true>(ay, by, w0.x, w0.y, pm.np);
                        }
                    }
                    sched_fence();
                }
            } else if constexpr (G0 > 1) {
                // the first G0 stages of the 2^G0 N-point transform, only the branches that lead to this part of the
                // row (ks_fused_split_kernel's loader on 16384-point parts; G0 = 2: three Shoup products per
                // coefficient, values below 4p), one chunk = NS loads of 16 bytes at a time.  (G0 = 1 written out above:
                // the general form costs that instance 12 bytes of scratch.)
                const u64 mask_i = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
                auto lf = [&](u64 v) -> u64 {
                    if constexpr (rns_fast) return csub_n(v, p, pm.np);
                    return lift((v >> sh) & mask_i);
                };
                const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src);
                constexpr int HB = 1;
                u64x2 wst[G0];
#pragma unroll
                for (int st = 0; st < G0; st++) wst[st] = twr[(1u << st) + (sub >> (G0 - st))];
#pragma unroll
                for (int h = 0; h < CH; h += HB) {
                    u64x2 v[HB][NS];
#pragma unroll
                    for (int c = 0; c < HB; c++)
#pragma unroll
                        for (int k = 0; k < NS; k++) v[c][k] = s2[(h + c) * T + tid + (u64)k * (N / 2)];
#pragma unroll
                    for (int c = 0; c < HB; c++) {
#pragma unroll
                        for (int k = 0; k < NS; k++) v[c][k].x = lf(v[c][k].x), v[c][k].y = lf(v[c][k].y);
#pragma unroll
                        for (int st = 0; st < G0; st++) {   // stage st keeps the half of the pairs whose output leads to `sub`
                            const int half = NS >> (st + 1);
                            const u64x2 wv = wst[st];
                            const bool minus = (sub >> (G0 - st - 1)) & 1;   // (uniform over the workgroup)
                            // (stage 0 works on canonical values: no correction of the first operand there)
                            if (minus) {
#pragma unroll
                                for (int m = 0; m < half; m++) {
                                    const u64 lx = st ? csub_n(v[c][m].x, p2, pm.np2) : v[c][m].x;
                                    const u64 ly = st ? csub_n(v[c][m].y, p2, pm.np2) : v[c][m].y;
                                    v[c][m].x = lx + p2 - mul_shoup_lazy_n<true>(v[c][m + half].x, wv.x, wv.y, pm.np);
                                    v[c][m].y = ly + p2 - mul_shoup_lazy_n<true>(v[c][m + half].y, wv.x, wv.y, pm.np);
                                }
                            } else {
#pragma unroll
                                for (int m = 0; m < half; m++) {
                                    const u64 lx = st ? csub_n(v[c][m].x, p2, pm.np2) : v[c][m].x;
                                    const u64 ly = st ? csub_n(v[c][m].y, p2, pm.np2) : v[c][m].y;
                                    v[c][m].x = mul_shoup_lazy_add_n<true>(lx, v[c][m + half].x, wv.x, wv.y, pm.np);
                                    v[c][m].y = mul_shoup_lazy_add_n<true>(ly, v[c][m + half].y, wv.x, wv.y, pm.np);
                                }
                            }
                        }
                        const uint32_t e = 2 * ((h + c) * T + tid);
                        lds[padi(e)] = v[c][0].x;
                        lds[padi(e + 1)] = v[c][0].y;
                    }
                    sched_fence();
                }
            } else if constexpr (F64 > 0) {
                tile_to_lds<CH, N, T>(lds, src, tid, [](u64 v) { return bits_of_f64(f64_from_u64(v)); });
            } else if constexpr (rns_fast) {
                tile_to_lds<CH, N, T>(lds, src, tid, [&](u64 v) { return csub_n(v, p, pm.np); });
            } else {
                const u64 mask_i = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
                tile_to_lds<CH, N, T>(lds, src, tid, [&](u64 v) { return lift((v >> sh) & mask_i); });
            }
        }
        FHE_TSK(1);
        FHE_BARRIER();
        FHE_TSK(2);
        if constexpr (PREFETCH) {
            if (ii + 1 < nloop) {
                const u64x2 *nx = reinterpret_cast<const u64x2 *>(src0 + (u64)digit_of(ii + 1) * dstride);
#pragma unroll
                for (int c = 0; c < CH; c++) pre[c] = nx[c * T + tid];
            }
        }
        const u64 koff = ((u64)i * lk + j) * NROW + suboff;
        if constexpr (CH > 0) {
            const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
            const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
            // KPF: the key words of the first two chunks are requested before the barrier that ends the
            // transform, so their L2 latency is spent waiting for the other waves, not after them
            constexpr bool KPF = PREFETCH && CH >= 2;
            // (twiddle prefetch measured: no gain here; NARROW: values < 16p on exit, fine for the Shoup MAC)
            ntt_fwd_lds<LOGN, T, GM, FHE_KS_TWPF && LOGN == 13, !KPF, (F64 ? -(F64 | 8) : NARROW ? (G0 ? 4 : 1) : 0), NoSrc, KS_LATE>(lds, twr, NS + sub, pm, tid);
            FHE_TSK(7);
            // (all four chunks prefetched -- 118 VGPRs, no scratch: no change; three: 2 % slower.  ABBA runs in
            // profiles/r02_ks_kpf_ab.txt: the key words' latency is not what the MAC waits for)
            constexpr int KPFN = KPF ? (FHE_KS_KPF_CHUNKS < CH ? FHE_KS_KPF_CHUNKS : CH) : 0;   // chunks whose key words are prefetched
            u64x2 kq[KPF ? 4 * KPFN : 1];
            if constexpr (KPF) {
#pragma unroll
                for (int c = 0; c < KPFN; c++) {
                    const uint32_t ci = c * T + tid;
                    kq[4 * c] = a0[ci], kq[4 * c + 1] = a0s[ci], kq[4 * c + 2] = a1[ci], kq[4 * c + 3] = a1s[ci];
                }
                FHE_TSK(3);
                FHE_BARRIER();
                FHE_TSK(4);
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t ci = c * T + tid;
                u64x2 q0, q0s, q1, q1s;
                if (KPF && c < KPFN) {
                    q0 = kq[KPF ? 4 * c : 0], q0s = kq[KPF ? 4 * c + 1 : 0], q1 = kq[KPF ? 4 * c + 2 : 0], q1s = kq[KPF ? 4 * c + 3 : 0];
                } else {
                    q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
                }
                if constexpr (F64 > 0) {
                    double fx = f64_of_bits(lds[padi(2 * ci)]), fy = f64_of_bits(lds[padi(2 * ci + 1)]);
                    if constexpr (F64_REDV) fx = reduce_f64(fx, pf), fy = reduce_f64(fy, pf);
                    // (block-uniform) every F64_CAP terms the accumulators are brought back below p / 2
                    const bool fold_acc = f64_terms >= (uint32_t)F64_CAP;
                    auto mac = [&](u64 acc, double v, u64 kk) {
                        double a = f64_of_bits(acc);
                        if (fold_acc) a = reduce_f64(a, pf);
                        return bits_of_f64(mulmod2_add_f64(a, v, f64_of_bits(kk), pf));
                    };
                    acc0[2 * c] = mac(acc0[2 * c], fx, q0.x);
                    acc0[2 * c + 1] = mac(acc0[2 * c + 1], fy, q0.y);
                    if constexpr (ACC1_LDS) {
                        u64x2 a = acc1_lds[ci];
                        a.x = mac(a.x, fx, q1.x);
                        a.y = mac(a.y, fy, q1.y);
                        acc1_lds[ci] = a;
                    } else {
                        acc1[ACC1_LDS ? 0 : 2 * c] = mac(acc1[ACC1_LDS ? 0 : 2 * c], fx, q1.x);
                        acc1[ACC1_LDS ? 0 : 2 * c + 1] = mac(acc1[ACC1_LDS ? 0 : 2 * c + 1], fy, q1.y);
                    }
                } else {
                const u64 vx = lds[padi(2 * ci)], vy = lds[padi(2 * ci + 1)];  // < 4p: Shoup accepts any u64
                acc0[2 * c] = csub_n(mul_shoup_lazy_add_n(acc0[2 * c], vx, q0.x, q0s.x, pm.np), p2, pm.np2);
                acc0[2 * c + 1] = csub_n(mul_shoup_lazy_add_n(acc0[2 * c + 1], vy, q0.y, q0s.y, pm.np), p2, pm.np2);
                if constexpr (ACC1_LDS) {
                    u64x2 a = acc1_lds[ci];
                    a.x = csub_n(mul_shoup_lazy_add_n(a.x, vx, q1.x, q1s.x, pm.np), p2, pm.np2);
                    a.y = csub_n(mul_shoup_lazy_add_n(a.y, vy, q1.y, q1s.y, pm.np), p2, pm.np2);
                    acc1_lds[ci] = a;
                } else {
                    acc1[2 * c] = csub_n(mul_shoup_lazy_add_n(acc1[2 * c], vx, q1.x, q1s.x, pm.np), p2, pm.np2);
                    acc1[2 * c + 1] = csub_n(mul_shoup_lazy_add_n(acc1[2 * c + 1], vy, q1.y, q1s.y, pm.np), p2, pm.np2);
                }
                }
                if (c & 1) sched_fence();  // at most two chunks of key loads (32 VGPRs) in flight
            }
        } else {
            ntt_fwd_lds<LOGN, T, GM, false, true, (NARROW ? 1 : 0), NoSrc, KS_LATE>(lds, twr, 1, pm, tid);
            if (tid < N) {
            const u64 v = lds[padi(tid)];
            acc0[0] = csub_n(mul_shoup_lazy_add_n(acc0[0], v, k0[koff + tid], k0s[koff + tid], pm.np), p2, pm.np2);
            acc1[0] = csub_n(mul_shoup_lazy_add_n(acc1[0], v, k1[koff + tid], k1s[koff + tid], pm.np), p2, pm.np2);
            }
        }
        if constexpr (F64 > 0) f64_terms = f64_terms >= (uint32_t)F64_CAP ? 2u : f64_terms + 1;   // (a reduced accumulator counts as one term)
        // (wave-contiguous chunk ownership, which makes this barrier and the one before the MAC wave-local as
        // well, was measured: nothing beyond what the late pass plan already gives)
        FHE_TSK(5);
        FHE_BARRIER();
        FHE_TSK(6);
    }
    // (FHE_DEBUG_KS_NOMEM -- every polynomial aliased to the first: rows, addends and outputs out of L2 -- makes this
    // kernel 11 % faster at C2: what its one workgroup per CU cannot hide.  Requesting the addends during the last
    // digit, into the row-prefetch registers that are idle then, was built: those 16 registers stay live through the
    // last MAC and spill (84-164 B of scratch); not kept.)
    const uint32_t tid = opaque(tid0);  // keeps the epilogue's address arithmetic below the digit loop
    have_pre = false;
    if constexpr (PREFETCH && ITEM_LOOP) {
        const uint32_t nitem = item + gridDim.x;
        if (nitem < total) {   // the next item's first digit row (the same selection as at the top of the loop)
            const uint32_t nb = (nitem >> G0) / lk, nj = (nitem >> G0) - nb * lk;
            const bool nown = xhat != nullptr && digit_shift_bits == 0 && nj < ndigits;
            if (ndigits - (nown ? 1u : 0u) > 0) {
                const uint32_t nd = (nown && nj == 0) ? 1u : 0u;
                const u64x2 *first = reinterpret_cast<const u64x2 *>(pin + (u64)nb * src_poly_stride + (u64)nd * dstride);
#pragma unroll
                for (int c = 0; c < CH; c++) pre[c] = first[c * T + tid];
                have_pre = true;
            }
        }
    }
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * NROW + suboff;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * NROW + suboff;
    if constexpr (CH > 0) {
        u64x2 *o0 = reinterpret_cast<u64x2 *>(out0 + ooff), *o1 = reinterpret_cast<u64x2 *>(out1 + ooff);
        const u64x2 *d0 = reinterpret_cast<const u64x2 *>(addend0 ? addend0 + aoff : nullptr);
        const u64x2 *d1 = reinterpret_cast<const u64x2 *>(addend1 ? addend1 + aoff : nullptr);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            u64x2 r0, r1;
            if constexpr (F64 > 0) {
                // accumulator (+ canonical addend, added as a double: the sum stays below 2^53) -> canonical word
                const u64x2 af = ACC1_LDS ? acc1_lds[ci] : u64x2{acc1[ACC1_LDS ? 0 : 2 * c], acc1[ACC1_LDS ? 0 : 2 * c + 1]};
                double f0x = f64_of_bits(acc0[2 * c]), f0y = f64_of_bits(acc0[2 * c + 1]);
                double f1x = f64_of_bits(af.x), f1y = f64_of_bits(af.y);
                if (d0) {
                    u64x2 a;
                    if constexpr (GAL) {
                        const u64 *arow = addend0 + aoff - suboff;
                        const uint32_t d = (uint32_t)suboff + 2 * ci;
                        a.x = arow[galois_src_index(d, gal, LOGN + G0)];
                        a.y = arow[galois_src_index(d + 1, gal, LOGN + G0)];
                    } else {
                        a = d0[ci];
                    }
                    f0x += f64_from_u64(a.x), f0y += f64_from_u64(a.y);
                }
                if (d1) {
                    const u64x2 a = d1[ci];
                    f1x += f64_from_u64(a.x), f1y += f64_from_u64(a.y);
                }
                o0[ci] = u64x2{to_u64_canonical(f0x, pf), to_u64_canonical(f0y, pf)};
                o1[ci] = u64x2{to_u64_canonical(f1x, pf), to_u64_canonical(f1y, pf)};
                continue;
            }
            r0.x = csub_n(acc0[2 * c], p, pm.np);
            r0.y = csub_n(acc0[2 * c + 1], p, pm.np);
            const u64x2 a1 = ACC1_LDS ? acc1_lds[ci] : u64x2{acc1[ACC1_LDS ? 0 : 2 * c], acc1[ACC1_LDS ? 0 : 2 * c + 1]};
            r1.x = csub_n(a1.x, p, pm.np);
            r1.y = csub_n(a1.y, p, pm.np);
            if (d0) {
                u64x2 a;
                if constexpr (GAL) {
                    const u64 *arow = addend0 + aoff - suboff;
                    const uint32_t d = (uint32_t)suboff + 2 * ci;
                    a.x = arow[galois_src_index(d, gal, LOGN + G0)];
                    a.y = arow[galois_src_index(d + 1, gal, LOGN + G0)];
                } else {
                    a = d0[ci];
                }
                r0.x = add_mod_n(r0.x, a.x, pm);
                r0.y = add_mod_n(r0.y, a.y, pm);
            }
            if (d1) {
                const u64x2 a = d1[ci];
                r1.x = add_mod_n(r1.x, a.x, pm);
                r1.y = add_mod_n(r1.y, a.y, pm);
            }
            o0[ci] = r0;
            o1[ci] = r1;
        }
    } else if (tid < N) {
        u64 r0 = csub(acc0[0], p), r1 = csub(acc1[0], p);
        if (addend0) r0 = add_mod(r0, addend0[aoff + (GAL ? galois_src_index(tid, gal, LOGN) : tid)], p);
        if (addend1) r1 = add_mod(r1, addend1[aoff + tid], p);
        out0[ooff + tid] = r0;
        out1[ooff + tid] = r1;
    }
    FHE_TSK(21);   // epilogue: next item's row prefetch, final reductions, addends, stores
    } while (ITEM_LOOP && (item += gridDim.x) < total);
    FHE_TS_END();
}

// The same for rows that do not fit LDS (N = 2^(13+G0) >= 32768): one workgroup per (ciphertext,
// key modulus j, 8192-point sub-block).  The first G0 Cooley-Tukey stages (native.rs:142-175,
// blocks larger than the tile) are folded into the loader: coefficient e of sub-block `sub`
// depends on the 2^G0 source coefficients e + k*8192 through G0 butterflies of which only the
// branch leading to `sub` is evaluated (2^G0 - 1 Shoup multiplications per coefficient instead
// of G0/2 amortised, but no round trip of the lifted row through HBM); the remaining 13 stages
// run in LDS with twiddle base 2^G0 + sub, exactly like ntt_kernel's sub-block mode.
template <int G0, int LOGM = 13, bool NARROW = false, bool RNS = false>
__global__ void __launch_bounds__((1 << LOGM) / 8, 4)
    ks_fused_split_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0,
                          u64 *__restrict__ out1, u64 out_poly_stride, const u64 *__restrict__ addend0,
                          const u64 *__restrict__ addend1, u64 addend_poly_stride, const u64 *__restrict__ k0,
                          const u64 *__restrict__ k0s, const u64 *__restrict__ k1, const u64 *__restrict__ k1s,
                          const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t lk,
                          uint32_t digit_arg, const u64 *__restrict__ xhat, u64 xhat_poly_stride, uint32_t gal) {
    FHE_DYN_SMEM(u64, lds);   // (gal: see ks_fused_kernel)
    constexpr int M = 1 << LOGM, T = M / 8, CH = M / (2 * T), NS = 1 << G0;
    constexpr u64 N = (u64)M << G0;
    const uint32_t tid0 = threadIdx.x;
    const uint32_t sub = blockIdx.x & (NS - 1);
    const uint32_t bj = blockIdx.x >> G0;
    const uint32_t b = to_sgpr(bj / lk), j = bj - b * lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64x2 *twr = tw + (u64)j * N;
    const uint32_t digit_shift_bits = digit_arg & 0xff, lift_mode = digit_arg >> 8;  // see ks_fused_kernel
    auto lift = [&](u64 v) -> u64 {
        if (lift_mode == 1) return csub_n(v, p, pm.np);
        if (lift_mode == 2) return csub_n(csub_n(v, p2, pm.np2), p, pm.np);
        return reduce_u64(v, md);
    };
    u64 acc0[2 * CH];
    u64x2 *const acc1_lds = reinterpret_cast<u64x2 *>(lds + lds_words(M));
#pragma unroll
    for (int e = 0; e < 2 * CH; e++) acc0[e] = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc1_lds[c * T + tid0] = u64x2{0, 0};
    // (see ks_fused_kernel: digit j under key modulus j is the caller's Ntt-form row j -- no transform)
    const bool own = xhat != nullptr && digit_shift_bits == 0 && j < ndigits;
    if (own) {
        const u64 koff = ((u64)j * lk + j) * N + (u64)sub * M;
        const u64x2 *xr = reinterpret_cast<const u64x2 *>(xhat + (u64)b * xhat_poly_stride + (u64)j * N + (u64)sub * M);
        const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
        const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid0;
            u64x2 v;
            if (gal) {
                const u64 *xrow = xhat + (u64)b * xhat_poly_stride + (u64)j * N;
                const uint32_t d = sub * M + 2 * ci;
                v.x = xrow[galois_src_index(d, gal, LOGM + G0)];
                v.y = xrow[galois_src_index(d + 1, gal, LOGM + G0)];
            } else {
                v = xr[ci];
            }
            const u64x2 q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
            acc0[2 * c] = mul_shoup_lazy_n(v.x, q0.x, q0s.x, pm.np);
            acc0[2 * c + 1] = mul_shoup_lazy_n(v.y, q0.y, q0s.y, pm.np);
            acc1_lds[ci] = u64x2{mul_shoup_lazy_n(v.x, q1.x, q1s.x, pm.np), mul_shoup_lazy_n(v.y, q1.y, q1s.y, pm.np)};
        }
    }
    const uint32_t nloop = ndigits - (own ? 1u : 0u);
    for (uint32_t ii = 0; ii < nloop; ii++) {
        const uint32_t i = ii + ((own && ii >= j) ? 1u : 0u);
        const uint32_t tid = opaque(tid0);
        const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * N);
        const uint32_t sh = i * digit_shift_bits;
        const u64 mask = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            u64x2 v[NS];
#pragma unroll
            for (int k = 0; k < NS; k++) v[k] = reinterpret_cast<const u64x2 *>(src + (u64)k * M)[ci];
            if constexpr (RNS) {   // (see ks_fused_kernel's staging loop)
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    v[k].x = csub_n(v[k].x, p, pm.np);
                    v[k].y = csub_n(v[k].y, p, pm.np);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    v[k].x = lift((v[k].x >> sh) & mask);
                    v[k].y = lift((v[k].y >> sh) & mask);
                }
            }
            // stage s keeps the half of the pairs whose output leads to `sub`
#pragma unroll
            for (int st = 0; st < G0; st++) {
                const int half = NS >> (st + 1);
                const u64x2 w = twr[(1u << st) + (sub >> (G0 - st))];
                const bool minus = (sub >> (G0 - st - 1)) & 1;   // (uniform over the workgroup: one branch per stage)
                if (minus) {
#pragma unroll
                    for (int m = 0; m < half; m++) {
                        const u64 lx = csub_n(v[m].x, p2, pm.np2), ly = csub_n(v[m].y, p2, pm.np2);
                        v[m].x = lx + p2 - mul_shoup_lazy_n(v[m + half].x, w.x, w.y, pm.np);
                        v[m].y = ly + p2 - mul_shoup_lazy_n(v[m + half].y, w.x, w.y, pm.np);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < half; m++) {
                        const u64 lx = csub_n(v[m].x, p2, pm.np2), ly = csub_n(v[m].y, p2, pm.np2);
                        v[m].x = mul_shoup_lazy_add_n(lx, v[m + half].x, w.x, w.y, pm.np);
                        v[m].y = mul_shoup_lazy_add_n(ly, v[m + half].y, w.x, w.y, pm.np);
                    }
                }
            }
            lds[padi(2 * ci)] = v[0].x;
            lds[padi(2 * ci + 1)] = v[0].y;
        }
        FHE_BARRIER();
        // (NARROW: the folded loader stages leave values below 4p)
        ntt_fwd_lds<LOGM, T, KS_GMAX, false, true, (NARROW ? 4 : 0), NoSrc, KS_LATE>(lds, twr, NS + sub, pm, tid);
        const u64 koff = ((u64)i * lk + j) * N + (u64)sub * M;
        const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
        const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            const u64x2 q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
            const u64 vx = lds[padi(2 * ci)], vy = lds[padi(2 * ci + 1)];
            acc0[2 * c] = csub_n(mul_shoup_lazy_add_n(acc0[2 * c], vx, q0.x, q0s.x, pm.np), p2, pm.np2);
            acc0[2 * c + 1] = csub_n(mul_shoup_lazy_add_n(acc0[2 * c + 1], vy, q0.y, q0s.y, pm.np), p2, pm.np2);
            u64x2 a = acc1_lds[ci];
            a.x = csub_n(mul_shoup_lazy_add_n(a.x, vx, q1.x, q1s.x, pm.np), p2, pm.np2);
            a.y = csub_n(mul_shoup_lazy_add_n(a.y, vy, q1.y, q1s.y, pm.np), p2, pm.np2);
            acc1_lds[ci] = a;
            if (c & 1) sched_fence();
        }
        FHE_BARRIER();
    }
    const uint32_t tid = opaque(tid0);
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * N + (u64)sub * M;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * N + (u64)sub * M;
    u64x2 *o0 = reinterpret_cast<u64x2 *>(out0 + ooff), *o1 = reinterpret_cast<u64x2 *>(out1 + ooff);
    const u64x2 *d0 = reinterpret_cast<const u64x2 *>(addend0 ? addend0 + aoff : nullptr);
    const u64x2 *d1 = reinterpret_cast<const u64x2 *>(addend1 ? addend1 + aoff : nullptr);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t ci = c * T + tid;
        u64x2 r0, r1;
        r0.x = csub_n(acc0[2 * c], p, pm.np);
        r0.y = csub_n(acc0[2 * c + 1], p, pm.np);
        const u64x2 a1v = acc1_lds[ci];
        r1.x = csub_n(a1v.x, p, pm.np);
        r1.y = csub_n(a1v.y, p, pm.np);
        if (d0) {
            u64x2 a;
            if (gal) {
                const u64 *arow = addend0 + (u64)b * addend_poly_stride + (u64)j * N;
                const uint32_t d = sub * M + 2 * ci;
                a.x = arow[galois_src_index(d, gal, LOGM + G0)];
                a.y = arow[galois_src_index(d + 1, gal, LOGM + G0)];
            } else {
                a = d0[ci];
            }
            r0.x = add_mod_n(r0.x, a.x, pm);
            r0.y = add_mod_n(r0.y, a.y, pm);
        }
        if (d1) {
            const u64x2 a = d1[ci];
            r1.x = add_mod_n(r1.x, a.x, pm);
            r1.y = add_mod_n(r1.y, a.y, pm);
        }
        o0[ci] = r0;
        o1[ci] = r1;
    }
}

// ------------------------------------------------------------ unfused key switch ----
// The same sum as ks_fused_kernel, cut into two launches (round 4; F/bfv/keys/key_switching_key.rs:241-320 with the
// lazy accumulation pattern of F/bfv/ops/dot_product.rs:54-180):
//   stage A  ks_ntt_kernel:  W[b][i][jj][:] = NTT_{q_j}( [p_i]_{q_j} ),  j = j0 + jj   (one workgroup per row tile)
//   stage B  ks_mac_kernel:  (c0, c1)[b][j] (+)= sum_i W[b][i][jj] (.) (k0, k1)[i][j]  (one lane per coefficient pair)
// Why: the fused kernel holds two accumulator sets next to its transform, which forces 8 coefficients per thread,
// radix-8/4 passes and ONE workgroup per CU (N >= 8192); stage A is the plain NTT geometry instead (16 coefficients
// per thread, radix-16 passes, two workgroups per CU at 8192-point tiles), and stage B is a streaming kernel that
// multiplies 64 x 64 -> 128 bits into lazy 128-bit accumulators with ONE reduction per output -- no Shoup twins, so
// only half of the key's bytes are read.  The price is W: L * Lk rows per polynomial written and read once, which the
// host keeps small enough to stay on-die (groups of key moduli / of polynomials, engine.hpp).
//
// ks_ntt_kernel<LOGM, G0>: tile of 2^LOGM points; G0 > 0: the row has 2^(LOGM+G0) points and the first G0
// Cooley-Tukey stages are folded into the loader exactly as in ks_fused_split_kernel.
// Output range: NARROW (key moduli below 2^60) values below 4p < 2^62, otherwise canonical: both below 2^62, which
// is what stage B's four-multiply product (mul_wide62) takes.
template <int V>
struct KsLiftMode {
    static constexpr int value = V;
};
// F64 = HR > 0 (round 6; whole-row tiles, RNS digits, every key modulus below 2^(53 - HR)): the transforms run on doubles
// (`tw` is then the key context's F64 table); no lift; W receives canonical words, which is what stage B takes.
template <int LOGM, int G0, bool NARROW, bool RNS, int F64 = 0>
__global__ void __launch_bounds__(ntt_threads_c(LOGM), 4)
    ks_ntt_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ w, const DevMod *__restrict__ mods,
                  const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t j0, uint32_t jg, uint32_t digit_arg,
                  uint32_t skip_own, u64 *extra, u64 extra_poly_stride, uint32_t extra_rows, uint32_t main_rows) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ntt_threads_c(LOGM);
    constexpr int M = 1 << LOGM, NS = 1 << G0;
    constexpr int CH = tile_chunks_c(LOGM, T);
    constexpr u64 N = (u64)M << G0;
    const uint32_t tid = threadIdx.x;
    const uint32_t sub = blockIdx.x & (NS - 1);
    // Round 5 (G0 == 0 only; extra == nullptr otherwise): workgroups behind the `main_rows` digit tiles transform
    // EXTRA residue rows in place -- `extra` [polys][extra_rows][N], row r under modulus r, canonical in and out: the
    // forward NTT of (c0, c1) that Multiplicator::multiply runs next to its key switch (F/bfv/ops/mul.rs:207-227).  One
    // launch instead of two: a launch that does not fill the device takes one workgroup's time whatever it contains
    // (C2, one pair: ntt_fwd 12 us + ks_digit_ntt 12 us -> 12 us).
    if constexpr (G0 == 0) {
        if (blockIdx.x >= main_rows) {   // (block-uniform)
            const uint32_t er = blockIdx.x - main_rows;
            const uint32_t ep = to_sgpr(er / extra_rows), r = er - ep * extra_rows;
            const DevMod md = mods[r];
            const PM pm = make_pm(md);
            const u64 p2 = md.p2;
            u64 *row = extra + (u64)ep * extra_poly_stride + (u64)r * N;
            const u64x2 *twr = tw + (u64)r * N;
            if constexpr (F64 > 0) {
                const PM pmf = make_pm_f64(md);
                const PF pf = pf_of(pmf);
                ntt_fwd_lds<LOGM, T, GMAX, true, true, -F64>(lds, twr, 1, pmf, tid, [&](uint32_t idx, uint32_t) { return bits_of_f64(f64_from_u64(row[idx])); });
                lds_to_tile<CH, M, T>(lds, row, tid, [&](u64 v) { return to_u64_canonical(f64_of_bits(v), pf); });
                return;
            }
            ntt_fwd_lds<LOGM, T, GMAX, true, true, (NARROW ? 1 : 0)>(lds, twr, 1, pm, tid, [&](uint32_t idx, uint32_t) { return row[idx]; });
            if constexpr (NARROW) {   // below 16p -> canonical
                const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
                lds_to_tile<CH, M, T>(lds, row, tid, [&](u64 v) {
                    return csub_n(csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2), md.p, pm.np);
                });
            } else {
                lds_to_tile<CH, M, T>(lds, row, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), md.p, pm.np); });
            }
            return;
        }
    }
    const uint32_t rowb = blockIdx.x >> G0;                       // (b * ndigits + i) * jg + jj
    const uint32_t bi = to_sgpr(rowb / jg), jj = rowb - bi * jg, j = j0 + jj;
    const uint32_t b = to_sgpr(bi / ndigits), i = bi - b * ndigits;
    // (block-uniform) digit j under key modulus j is the caller's Ntt-form row (`xhat`): stage B reads it there
    if (skip_own && i == j) return;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64x2 *twr = tw + (u64)j * N;
    const uint32_t digit_shift_bits = digit_arg & 0xff, lift_mode = digit_arg >> 8;  // see ks_fused_kernel
    const uint32_t sh = i * digit_shift_bits;
    const u64 mask = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
    const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * N) + (G0 ? 0 : (u64)sub * M);
    if constexpr (F64 > 0) {
        static_assert(F64 == 0 || (G0 == 0 && RNS && !NARROW), "the F64 instances: whole-row tiles, RNS digits");
        const PM pmf = make_pm_f64(md);
        const PF pf = pf_of(pmf);
        ntt_fwd_lds<LOGM, T, GMAX, true, true, -F64>(lds, twr, NS + sub, pmf, tid, [&](uint32_t idx, uint32_t) { return bits_of_f64(f64_from_u64(src[idx])); });
        lds_to_tile<CH, M, T>(lds, w + (u64)rowb * N + (u64)sub * M, tid, [&](u64 v) { return to_u64_canonical(f64_of_bits(v), pf); });
        return;
    }
    // The lift mode is a compile-time constant of the transform's loader (three uniform branches around the whole
    // transform, below).  Round 5: as a run-time test inside the loader it put a branch diamond behind every one of the
    // first pass's sixteen global loads, and the compiler then waited for each load before issuing the next (global
    // loads are not speculated across a branch): `global_load -> s_waitcnt vmcnt(0) -> s_cbranch` sixteen times per
    // thread, 5-6 us of a 12 us single-workgroup transform on every basis whose moduli differ in width -- the
    // reference's stock sets (43/44-bit and 48/49-bit rows: lift mode 2), which have no RNS instance.
    auto run = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;   // 1: digit < 2p, 2: digit < 4p, 0: anything (Barrett), 3: RNS instance
        auto lift = [&](u64 v) -> u64 {
            u64 r;   // (one return: g++ -- the emulation build -- misreads returns inside `if constexpr` of a nested lambda)
            if constexpr (MODE == 3) {
                r = csub_n(v, p, pm.np);
            } else {
                v = (v >> sh) & mask;
                if constexpr (MODE == 1) r = csub_n(v, p, pm.np);
                else if constexpr (MODE == 2) r = csub_n(csub_n(v, p2, pm.np2), p, pm.np);
                else r = reduce_u64(v, md);
            }
            return r;
        };
        auto load = [&](uint32_t idx, uint32_t) -> u64 {
            if constexpr (G0 == 0) {
                return lift(src[idx]);
            } else {
                u64 v[NS];
#pragma unroll
                for (int k = 0; k < NS; k++) v[k] = lift(src[idx + (u64)k * M]);
#pragma unroll
                for (int st = 0; st < G0; st++) {   // stage st keeps the half of the pairs whose output leads to `sub`
                    const int half = NS >> (st + 1);
                    const u64x2 wv = twr[(1u << st) + (sub >> (G0 - st))];
                    const bool minus = (sub >> (G0 - st - 1)) & 1;   // (uniform over the workgroup)
                    // x + t or x + 2p - t as a SELECT, not a branch: a branch diamond here sits between this group's
                    // global loads and the next group's, and the compiler then finishes each group before it issues the
                    // next loads (ISA, round 5: `2 x global_load -> s_waitcnt vmcnt(0) -> s_cbranch` sixteen times per
                    // thread) -- the same serialisation the lift mode caused above
#pragma unroll
                    for (int m = 0; m < half; m++) {
                        const u64 t = mul_shoup_lazy_n<true>(v[m + half], wv.x, wv.y, pm.np);   // below 2p
                        v[m] = csub_n(v[m], p2, pm.np2) + (minus ? p2 - t : t);
                    }
                }
                return v[0];   // below 4p
            }
        };
        ntt_fwd_lds<LOGM, T, GMAX, true, true, (NARROW ? (G0 ? 4 : 1) : 0)>(lds, twr, NS + sub, pm, tid, load);
    };
    if constexpr (RNS) {
        run(KsLiftMode<3>{});
    } else {
        if (lift_mode == 1) run(KsLiftMode<1>{});          // (block-uniform: digit_arg is a kernel argument)
        else if (lift_mode == 2) run(KsLiftMode<2>{});
        else run(KsLiftMode<0>{});
    }
    u64 *dst = w + (u64)rowb * N + (u64)sub * M;
    if constexpr (NARROW) {   // below 16p -> below 4p
        const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p8, np8), p4, np4); });
    } else {                  // below 4p -> canonical
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); });
    }
}

// Stage B.  One lane per 16-byte chunk of an output row (b, j); grid = (key range, polynomial) in an XCD-aware
// order: the lanes of a block read nd * 2 * 4 KiB of key words that every polynomial of the launch needs again, so
// the blocks of one key range get ids 8 apart (same XCD, dispatched back to back: the re-reads hit that L2).
// Lazy accumulation: x < 2^62, k < 2^62 -> x * k < 2^124, sixteen terms fit 128 bits; longer digit loops fold the
// accumulator through the 128-bit reduction every sixteen terms.
__global__ void __launch_bounds__(256)
    ks_mac_kernel(const u64 *__restrict__ w, u64 *__restrict__ out0, u64 *__restrict__ out1, u64 out_poly_stride,
                  const u64 *__restrict__ addend0, const u64 *__restrict__ addend1, u64 addend_poly_stride,
                  const u64 *__restrict__ k0, const u64 *__restrict__ k1, const DevMod *__restrict__ mods,
                  uint32_t ndigits, uint32_t lk, uint32_t j0, uint32_t jg, uint32_t logn, const u64 *__restrict__ xhat,
                  u64 xhat_poly_stride, uint32_t npolys, uint32_t gal) {   // (gal: see ks_fused_kernel)
    const uint32_t n = 1u << logn;
    const uint32_t cpr = n >= 512 ? n / 512 : 1;          // 256-lane chunks per row
    const uint32_t nkr = jg * cpr;                        // key ranges of this launch
    const uint32_t grp = blockIdx.x >> 3, x8 = blockIdx.x & 7;
    const uint32_t krhi = grp / npolys, b = grp - krhi * npolys;
    const uint32_t kr = krhi * 8 + x8;
    if (kr >= nkr) return;                                // (block-uniform) tail of the rounded-up grid
    const uint32_t jj = kr / cpr, cb = kr - jj * cpr, j = j0 + jj;
    const uint32_t ci = cb * 256 + threadIdx.x;           // 16-byte chunk inside the row
    if (2 * ci >= n) return;
    const DevMod md = mods[j];
    const u64 roff = 2 * (u64)ci;
    const u64 *wp = w + (((u64)b * ndigits) * jg + jj) * n + roff;      // + i * jg * n per digit
    const u64 *kp0 = k0 + (u64)j * n + roff, *kp1 = k1 + (u64)j * n + roff;   // + i * lk * n per digit
    const bool own = xhat != nullptr && j < ndigits;      // (block-uniform)
    u128_t a0x = 0, a0y = 0, a1x = 0, a1y = 0;
    auto mac = [&](u128_t &acc, u64 x, u64 k) {
        u64 hi, lo;
        mul_wide62(x, k, hi, lo);
        acc += ((u128_t)hi << 64) | lo;
    };
    auto fold = [&](u128_t &acc) { acc = reduce_u128((u64)(acc >> 64), (u64)acc, md); };
    for (uint32_t i0 = 0; i0 < ndigits; i0 += 16) {
        const uint32_t i1 = i0 + 16 < ndigits ? i0 + 16 : ndigits;
        if (i0) fold(a0x), fold(a0y), fold(a1x), fold(a1y);
#pragma unroll 4
        for (uint32_t i = i0; i < i1; i++) {
            // (W is read exactly once, by this lane: a streaming load; the key words are every polynomial's: cached)
            u64x2 xv;
            if (own && i == j) {
                const u64 *xrow = xhat + (u64)b * xhat_poly_stride + (u64)j * n;
                if (gal) {
                    xv.x = xrow[galois_src_index((uint32_t)roff, gal, logn)];
                    xv.y = xrow[galois_src_index((uint32_t)roff + 1, gal, logn)];
                } else {
                    xv = *reinterpret_cast<const u64x2 *>(xrow + roff);
                }
            } else {
                xv = load_stream(reinterpret_cast<const u64x2 *>(wp + (u64)i * jg * n));
            }
            const u64x2 q0 = *reinterpret_cast<const u64x2 *>(kp0 + (u64)i * lk * n);
            const u64x2 q1 = *reinterpret_cast<const u64x2 *>(kp1 + (u64)i * lk * n);
            mac(a0x, xv.x, q0.x);
            mac(a0y, xv.y, q0.y);
            mac(a1x, xv.x, q1.x);
            mac(a1y, xv.y, q1.y);
        }
    }
    u64x2 r0, r1;
    r0.x = reduce_u128((u64)(a0x >> 64), (u64)a0x, md);
    r0.y = reduce_u128((u64)(a0y >> 64), (u64)a0y, md);
    r1.x = reduce_u128((u64)(a1x >> 64), (u64)a1x, md);
    r1.y = reduce_u128((u64)(a1y >> 64), (u64)a1y, md);
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * n + roff;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * n + roff;
    if (addend0) {
        u64x2 a;
        if (gal) {
            const u64 *arow = addend0 + aoff - roff;
            a.x = arow[galois_src_index((uint32_t)roff, gal, logn)];
            a.y = arow[galois_src_index((uint32_t)roff + 1, gal, logn)];
        } else {
            a = *reinterpret_cast<const u64x2 *>(addend0 + aoff);
        }
        r0.x = add_mod(r0.x, a.x, md.p);
        r0.y = add_mod(r0.y, a.y, md.p);
    }
    if (addend1) {
        const u64x2 a = *reinterpret_cast<const u64x2 *>(addend1 + aoff);
        r1.x = add_mod(r1.x, a.x, md.p);
        r1.y = add_mod(r1.y, a.y, md.p);
    }
    *reinterpret_cast<u64x2 *>(out0 + ooff) = r0;
    *reinterpret_cast<u64x2 *>(out1 + ooff) = r1;
}

}  // namespace k
}  // namespace fhe
