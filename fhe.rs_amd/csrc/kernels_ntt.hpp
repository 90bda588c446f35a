// kernels_ntt.hpp -- ntt_kernel (NttOperator::forward / backward), tensor_intt_kernel (tensor step fused with the
// inverse transform that follows it) and ntt_global_kernel (stages that span sub-blocks, N >= 32768).
#pragma once
#include "kernels_passes.hpp"

namespace fhe {
namespace k {

// ------------------------------------------------------------------- NTT kernel ----
// One workgroup (ntt_threads_c(LOGM) threads) per (row, sub-block).
// grid.x = npolys * map.rows * nsub, nsub = 2^(logn - LOGM).
// LOGM == logn: whole row in LDS (N <= 16384).  LOGM < logn: this is the LDS half of the
// two-kernel transform for N >= 32768 (ntt_global_kernel does the other logn-LOGM stages).
//   forward: canonical output (reduce3, native.rs:238-246)
//   inverse: multiplies by N^-1 (Shoup) when LOGM == logn (native.rs:229-232)
// Register budget 128 VGPRs = 4 waves/SIMD, which is what the LDS footprint allows anyway
// (N = 8192: 68 KiB/workgroup -> 2 workgroups of 8 waves per CU).
// NARROW (forward, whole row, every modulus of the launch below 2^60): see fwd_butterfly_narrow.
// FWD_B0 (forward, NARROW): the loader's values are below FWD_B0 * p -- 1: canonical input (whole rows); 4: this is the
// LDS half of a row larger than LDS, behind ntt_global_kernel's Harvey stages, which leave values below 4p.
// GATHER (inverse only, round 5): the tile is read through the Ntt-domain substitution map.subst_exp (galois_src_index):
// GaloisKey::relinearize's `substitute(c1)` followed by the inverse transform (F/bfv/keys/galois_key.rs:66-70) in one pass --
// per-lane 8-byte gathers inside one row (L2 hits after the first touch) instead of a permutation kernel's write + re-read.
// F64 = HR > 0 (round 6; whole rows only, every modulus of the launch below 2^(53 - HR)): the passes run on doubles
// (zq_f64.hpp) -- `tw` / `ninv` are then the context's F64 tables; canonical u64 words in and out, as always.
template <bool INVERSE, int LOGM, bool NARROW = false, int FWD_B0 = 1, bool GATHER = false, int F64 = 0>
__global__ void __launch_bounds__(ntt_threads_c(LOGM), 4)
    ntt_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map, const DevMod *__restrict__ mods,
               const u64x2 *__restrict__ tw, const u64x2 *__restrict__ ninv, uint32_t logn) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ntt_threads_c(LOGM);
    constexpr int M = 1 << LOGM;
    constexpr int CH = tile_chunks_c(LOGM, T);
    const uint32_t tid = threadIdx.x;
    const uint32_t n = 1u << logn;
    const uint32_t nsub = 1u << (logn - LOGM);
    const uint32_t bid = map.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const uint32_t sub = bid & (nsub - 1);
    const uint32_t rowb = bid >> (logn - LOGM);
    const uint32_t poly = to_sgpr(rowb / map.rows);
    const uint32_t r = map.row_begin + (rowb - poly * map.rows);
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64 *src = rowmap_src(map, in, poly) +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n + (u64)sub * M;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n + (u64)sub * M;
    const u64x2 *twr = tw + (u64)mi * n;

    if constexpr (F64 > 0) {
        const PM pmf = make_pm_f64(md);
        const PF pf = pf_of(pmf);
        auto to_f = [](u64 v) { return bits_of_f64(f64_from_u64(v)); };
        auto to_c = [&](u64 v) { return to_u64_canonical(f64_of_bits(v), pf); };
        if constexpr (!INVERSE) {
            auto ld = [&](uint32_t i, uint32_t) { return to_f(load_last<(FHE_PIPE_NT & 2) != 0>(src + i)); };
            if constexpr (FHE_FWD_DIRECT_STORE && plan_np(LOGM, GMAX) > 1) {
                constexpr int GL = fwd_plan_g<LOGM, GMAX, plan_np(LOGM, GMAX) - 1, false>();
                ntt_fwd_lds<LOGM, T, GMAX, true, true, -F64, decltype(ld), false, 1>(lds, twr, nsub + sub, pmf, tid, ld, 0, [&](uint32_t base, const u64 (&x)[1 << GL]) {
                    u64x2 *d2 = reinterpret_cast<u64x2 *>(dst + base);
#pragma unroll
                    for (int e = 0; e < (1 << GL); e += 2) d2[e / 2] = u64x2{to_c(x[e]), to_c(x[e + 1])};
                });
            } else {
                ntt_fwd_lds<LOGM, T, GMAX, true, true, -F64>(lds, twr, nsub + sub, pmf, tid, ld);
                lds_to_tile<CH, M, T>(lds, dst, tid, to_c);
            }
        } else {
            InvTwFirst<LOGM, T> tw0;
            inv_tw_load(tw0, twr, logn, sub, tid);
            if constexpr (GATHER) {
                const u64 *row = src - (u64)sub * M;
                const uint32_t e = map.subst_exp;
#pragma unroll 8
                for (uint32_t i = tid; i < (uint32_t)M; i += T) lds[padi(i)] = to_f(row[galois_src_index(sub * M + i, e, logn)]);
            } else {
                tile_to_lds<CH, M, T, (FHE_PIPE_NT & 4) != 0>(lds, src, tid, to_f);
            }
            FHE_BARRIER();
            ntt_inv_lds<LOGM, T, 0, 0, false, F64>(lds, twr, logn, sub, pmf, tid, true, ninv[2 * mi], ninv[2 * mi + 1], tw0);
            lds_to_tile<CH, M, T>(lds, dst, tid, to_c);
        }
    } else if constexpr (!INVERSE) {
        // the first pass reads its groups straight from global memory (no tile staging)
        if constexpr (FHE_FWD_DIRECT_STORE && plan_np(LOGM, GMAX) > 1) {
            constexpr int GL = fwd_plan_g<LOGM, GMAX, plan_np(LOGM, GMAX) - 1, false>();
            auto ld = [&](uint32_t i, uint32_t) { return load_last<(FHE_PIPE_NT & 2) != 0>(src + i); };
            const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
            auto canon = [&](u64 v) {
                if constexpr (NARROW) v = csub_n(csub_n(v, p8, np8), p4, np4);   // < 16p -> < 4p
                return csub_n(csub_n(v, p2, pm.np2), p, pm.np);
            };
            ntt_fwd_lds<LOGM, T, GMAX, true, true, (NARROW ? FWD_B0 : 0), decltype(ld), false, 1>(lds, twr, nsub + sub, pm, tid, ld, 0, [&](uint32_t base, const u64 (&x)[1 << GL]) {
                u64x2 *d2 = reinterpret_cast<u64x2 *>(dst + base);
#pragma unroll
                for (int e = 0; e < (1 << GL); e += 2) d2[e / 2] = u64x2{canon(x[e]), canon(x[e + 1])};
            });
            return;
        }
        ntt_fwd_lds<LOGM, T, GMAX, true, true, (NARROW ? FWD_B0 : 0)>(lds, twr, nsub + sub, pm, tid, [&](uint32_t i, uint32_t) { return load_last<(FHE_PIPE_NT & 2) != 0>(src + i); });
        if constexpr (NARROW) {  // < 16p -> canonical
            const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) {
                return csub_n(csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2), p, pm.np);
            });
        } else {
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); });
        }
    } else {
        InvTwFirst<LOGM, T> tw0;
        inv_tw_load(tw0, twr, logn, sub, tid);   // issued ahead of the tile loads: one latency for both
        const bool whole = logn == LOGM;  // ninv[2*mi] = {N^-1, shoup}, ninv[2*mi+1] = {z_last * N^-1, shoup}
        // (feeding the first pass straight from global memory, as the forward transform does, was
        // measured for the inverse: no gain -- its groups are runs of consecutive coefficients)
        if constexpr (GATHER) {
            const u64 *row = src - (u64)sub * M;          // the stored row; this tile holds its elements [sub M, (sub + 1) M)
            const uint32_t e = map.subst_exp;
#pragma unroll 8
            for (uint32_t i = tid; i < (uint32_t)M; i += T) lds[padi(i)] = row[galois_src_index(sub * M + i, e, logn)];
        } else {
            tile_to_lds<CH, M, T, (FHE_PIPE_NT & 4) != 0>(lds, src, tid, [](u64 v) { return v; });
        }
        FHE_BARRIER();
        ntt_inv_lds<LOGM, T, 0, 0, NARROW>(lds, twr, logn, sub, pm, tid, whole, ninv[2 * mi], ninv[2 * mi + 1], tw0);
        if (whole)
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(v, p, pm.np); });
        else
            lds_to_tile<CH, M, T>(lds, dst, tid, [](u64 v) { return v; });  // < 2p, global pass finishes
    }
}

// (Persistent workgroups -- the rows of a launch dealt round-robin to two or four resident workgroups per CU, the next
// row's coefficients prefetched into registers across the epilogue -- were built and measured for the forward
// transform: 17.2-17.9 M against 24.8 M row-NTT/s (60-bit rows), 15.6-17.8 M against 19.8-20.8 M (62-bit): the 32
// prefetch registers on top of a radix-16 pass spill (36-124 B of scratch per lane) and the hardware dispatcher
// already overlaps one workgroup's loads with the other's arithmetic.  profiles/r02_ntt_persist_ab.txt.)

// ------------------------------------------------ fused tensor + inverse NTT ----
// The tensor step of Multiplicator::multiply (F/bfv/ops/mul.rs:198-201) fused into the loader of
// the inverse NTT that Scaler::scale applies next (M/rq/scaler.rs:69-79): the products
//   slot 0: c00*c10   slot 1: c00*c11 + c01*c10   slot 2: c01*c11
// are formed while the row is staged into LDS, so the Ntt-domain tensor never touches HBM.
// Operands: (c00, c01) = extL[b][0..1], (c10, c11) = extR[b][0..1], each [K][N]; rows below
// `ncommon` come straight from the input ciphertexts lhs/rhs [b][2][lrows][N] when given.
// grid = (K rows, nb ciphertext pairs, 3 slots); out is slot-major [3][nb][K][N] PowerBasis.
struct TensorSrc {
    const u64 *extL, *extR, *lhs, *rhs;
    uint32_t ncommon, lrows;
};
// SUB (rows larger than LDS, N = 2^logn > M): a workgroup handles one M-point sub-block -- tensor product
// in the loader, the inverse stages that stay inside the sub-block -- and leaves values below 2p for
// ntt_global_kernel<true, .>, which finishes the transform (so the Ntt-domain tensor never touches HBM
// at N = 32768 / 65536 either).
// A launch covers the rows [row_begin, row_begin + lrows) of the nrows-row extended basis; NARROW (all of them
// below 2^60) selects the inverse passes with tracked bounds (inv_pass): the ciphertext primes of the extended
// basis are 60-bit, the extension primes 62-bit, so bfv_mul issues one launch for each group.
// F64 = HR > 0 (round 6; whole rows, every modulus of the launch below 2^(53 - HR)): the products and the inverse passes
// on doubles (zq_f64.hpp mulmod2_f64: no precomputed quotient, q from h / p); `itw` / `ninv` are the F64 tables.
template <int LOGM, bool SUB = false, bool NARROW = false, int F64 = 0>
__global__ void __launch_bounds__(ntt_threads_c(LOGM), 4)
    tensor_intt_kernel(TensorSrc ts, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                       const u64x2 *__restrict__ itw, const u64x2 *__restrict__ ninv, uint32_t nrows, uint32_t nb,
                       uint32_t logn_arg, uint32_t row_begin, uint32_t lrows, uint32_t reverse) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ntt_threads_c(LOGM);
    constexpr int M = 1 << LOGM;
    constexpr int CH = tile_chunks_c(LOGM, T);
    const uint32_t tid = threadIdx.x;
    const uint32_t logn = SUB ? logn_arg : (uint32_t)LOGM;
    const uint32_t lsub = logn - LOGM;  // log2(sub-blocks per row); 0 unless SUB
    // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (id mod 8), each with its
    // own L2.  The three slots of one (row, ciphertext) pair read the same four operand rows, so their
    // ids are 8 apart: same XCD, dispatched back to back, and the re-reads hit that L2 instead of HBM
    // (a (row, pair, slot) 3-D grid put them nb*K blocks apart: 1.8x the algorithmic HBM traffic).
    // `reverse`: the launch walks its groups from the last one backwards (the extension's forward transform, which
    // wrote the operands, ran ascending).  The grid is a multiple of 8, so (gridDim.x - 1 - id) & 7 = 7 - (id & 7):
    // workgroups of one hardware XCD still share one residue class, i.e. the three slots of a group stay on one L2.
    const uint32_t bid = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const uint32_t t8 = bid >> 3, grp = t8 / 3, slot = t8 - 3 * grp;
    const uint32_t combo = grp * 8 + (bid & 7);
    if (combo >= (lrows * nb) << lsub) return;  // (block-uniform) tail of the rounded-up grid
    const uint32_t sub = combo & ((1u << lsub) - 1), rowb = combo >> lsub;
    const uint32_t b = to_sgpr(rowb / lrows), r = row_begin + (rowb - b * lrows);
    const DevMod md = mods[r];
    const u64 p = md.p;
    const PM pm = F64 ? make_pm_f64(md) : make_pm(md);
    const u64 pk = (u64)nrows << logn;
    const u64 roff = ((u64)r << logn) + (u64)sub * M;  // this tile inside a polynomial
    const u64 *a0, *a1, *b0, *b1;  // rows of c00, c01, c10, c11
    if (ts.lhs && r < ts.ncommon) {
        const u64 pl = (u64)ts.lrows << logn;
        a0 = ts.lhs + (u64)b * 2 * pl + roff;
        a1 = a0 + pl;
        b0 = ts.rhs + (u64)b * 2 * pl + roff;
        b1 = b0 + pl;
    } else {
        a0 = ts.extL + (u64)b * 2 * pk + roff;
        a1 = a0 + pk;
        b0 = ts.extR + (u64)b * 2 * pk + roff;
        b1 = b0 + pk;
    }
    auto prod = [&](u64 x00, u64 x01, u64 x10, u64 x11) -> u64 {
        if constexpr (F64 > 0) {   // (each product below 0.875 p, the double product below 1.75 p: F64_BIN0 = 2 F64_ONE)
            const PF pf = pf_of(pm);
            if (slot == 0) return bits_of_f64(mulmod2_f64(f64_from_u64(x00), f64_from_u64(x10), pf));
            if (slot == 2) return bits_of_f64(mulmod2_f64(f64_from_u64(x01), f64_from_u64(x11), pf));
            return bits_of_f64(mulmod2_f64(f64_from_u64(x00), f64_from_u64(x11), pf) +
                               mulmod2_f64(f64_from_u64(x01), f64_from_u64(x10), pf));
        }
        // (results stay below 2p: the inverse transform's first pass takes that range)
        if (slot == 0) return mul_mod_lazy(x00, x10, md);
        if (slot == 2) return mul_mod_lazy(x01, x11, md);
        // c1 = c00*c11 + c01*c10: one Barrett reduction of the 128-bit sum (< 2p^2 < 2^(2k+1): the
        // quotient estimate is then short by at most 3: below 4p before the conditional subtraction)
        u64 hi, lo;
        mac2_wide62(x00, x11, x01, x10, hi, lo);
        return barrett_reduce_wide_lazy(hi, lo, md);
    };
    // the first inverse pass's per-lane twiddles (56 VGPRs at N = 8192) are requested half-way through the products,
    // when half of the operand registers are free again: their L2 latency then hides behind the remaining
    // products and the barrier instead of following it (FHE_TENSOR_TW_EARLY=0: after the products, as before)
    const u64x2 *twr = itw + ((u64)r << logn);
    InvTwFirst<LOGM, T> tw0;
    if constexpr (CH > 0) {
        if (slot != 1) {
            // c0 = c00*c10 / c2 = c01*c11: two operand rows, all CH chunks of both in flight at once
            const u64x2 *pa = reinterpret_cast<const u64x2 *>(slot == 0 ? a0 : a1);
            const u64x2 *pb = reinterpret_cast<const u64x2 *>(slot == 0 ? b0 : b1);
            u64x2 va[CH], vb[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                va[c] = pa[c * T + tid];
                vb[c] = pb[c * T + tid];
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (FHE_TENSOR_TW_EARLY && CH > 1 && c == CH / 2) {
                    sched_fence();
                    inv_tw_load(tw0, twr, logn, sub, tid);
                }
                const uint32_t i = 2 * (c * T + tid);
                if constexpr (F64 > 0) {
                    const PF pf = pf_of(pm);
                    lds[padi(i)] = bits_of_f64(mulmod2_f64(f64_from_u64(va[c].x), f64_from_u64(vb[c].x), pf));
                    lds[padi(i + 1)] = bits_of_f64(mulmod2_f64(f64_from_u64(va[c].y), f64_from_u64(vb[c].y), pf));
                    continue;
                }
                lds[padi(i)] = mul_mod_lazy(va[c].x, vb[c].x, md);
                lds[padi(i + 1)] = mul_mod_lazy(va[c].y, vb[c].y, md);
            }
        } else {
        constexpr int HALF = CH > 1 ? CH / 2 : 1;  // loads of at most HALF chunks x 4 operands in flight
#pragma unroll
        for (int h = 0; h < CH; h += HALF) {
            u64x2 v00[HALF], v01[HALF], v10[HALF], v11[HALF];
#pragma unroll
            for (int c = 0; c < HALF; c++) {
                const uint32_t ci = (h + c) * T + tid;
                if (slot != 2) v00[c] = reinterpret_cast<const u64x2 *>(a0)[ci];
                if (slot != 0) v01[c] = reinterpret_cast<const u64x2 *>(a1)[ci];
                if (slot != 2) v10[c] = reinterpret_cast<const u64x2 *>(b0)[ci];
                if (slot != 0) v11[c] = reinterpret_cast<const u64x2 *>(b1)[ci];
            }
#pragma unroll
            for (int c = 0; c < HALF; c++) {
                if (FHE_TENSOR_TW_EARLY && CH > 1 && h + HALF >= CH && c == HALF / 2) {   // (last batch, half done)
                    sched_fence();
                    inv_tw_load(tw0, twr, logn, sub, tid);
                }
                const uint32_t i = 2 * ((h + c) * T + tid);
                lds[padi(i)] = prod(v00[c].x, v01[c].x, v10[c].x, v11[c].x);
                lds[padi(i + 1)] = prod(v00[c].y, v01[c].y, v10[c].y, v11[c].y);
            }
            sched_fence();
        }
        }
    } else {
        for (uint32_t i = tid; i < M; i += T) lds[padi(i)] = prod(a0[i], a1[i], b0[i], b1[i]);
    }
    if (!(FHE_TENSOR_TW_EARLY && CH > 1)) inv_tw_load(tw0, twr, logn, sub, tid);   // in flight across the barrier
    FHE_BARRIER();
    u64 *dst = out + ((u64)slot * nb + b) * pk + roff;
    // (a block-uniform branch between the narrow and the general inverse passes inside one kernel was measured:
    // 128 VGPRs, spills and twice the code -- 2 % slower; hence one launch per row group)
    if constexpr (F64 > 0) {
        static_assert(F64 == 0 || (!SUB && !NARROW), "the F64 instances: whole rows");
        const PF pf = pf_of(pm);
        ntt_inv_lds<LOGM, T, 0, 0, false, F64, 2 * F64_ONE>(lds, twr, logn, sub, pm, tid, true, ninv[2 * r], ninv[2 * r + 1], tw0);
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return to_u64_canonical(f64_of_bits(v), pf); });
        return;
    }
    ntt_inv_lds<LOGM, T, 0, 0, NARROW>(lds, twr, logn, sub, pm, tid, !SUB, ninv[2 * r], ninv[2 * r + 1], tw0);
    if constexpr (SUB)
        lds_to_tile<CH, M, T>(lds, dst, tid, [](u64 v) { return v; });  // < 2p, the global pass finishes
    else
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(v, p, pm.np); });
}

// Radix stages that span sub-blocks, done straight on global memory (coalesced along the
// low index).  Forward: stages [0, G0) before the LDS kernel (output < 4p, the LDS kernel's
// loader accepts that range).  Inverse: stages [logm, logn) after it, then N^-1.
// One thread per group of 2^G0 elements {lo + e*M}; grid.x covers npolys*rows*(M/threads).
template <bool INVERSE, int G0>
__global__ void ntt_global_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map,
                                  const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw,
                                  const u64x2 *__restrict__ ninv, uint32_t logn) {
    constexpr uint32_t R = 1u << G0;
    const uint32_t n = 1u << logn, logm = logn - G0, m = 1u << logm;
    const uint32_t chunks = (m + blockDim.x - 1) / blockDim.x;
    const uint32_t rowb = blockIdx.x / chunks;
    const uint32_t lo = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (lo >= m) return;
    const uint32_t poly = rowb / map.rows;
    const uint32_t r = map.row_begin + rowb % map.rows;
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p;
    const PM pm = make_pm(md);
    const u64 *src = rowmap_src(map, in, poly) +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n;
    const u64x2 *twr = tw + (u64)mi * n;
    u64 x[R];
#pragma unroll
    for (uint32_t e = 0; e < R; e++) x[e] = src[lo + e * m];
    if (!INVERSE) {
#pragma unroll
        for (int u = 0; u < G0; u++) {
            const uint32_t half = R >> (u + 1);
#pragma unroll
            for (uint32_t blk = 0; blk < (1u << u); blk++) {
                const u64x2 w = twr[(1u << u) + blk];
#pragma unroll
                for (uint32_t j = 0; j < half; j++) {
                    const uint32_t a = blk * 2 * half + j;
                    fwd_butterfly(x[a], x[a + half], w.x, w.y, pm);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) dst[lo + e * m] = x[e];
    } else {
#pragma unroll
        for (int u = 0; u < G0; u++) {
            const uint32_t v = logm + u;
            const uint32_t nblk = R >> (u + 1);
#pragma unroll
            for (uint32_t blk = 0; blk < nblk; blk++) {
                const u64x2 z = twr[n - (n >> v) + blk];
#pragma unroll
                for (uint32_t j = 0; j < (1u << u); j++) {
                    const uint32_t a = blk * (2u << u) + j;
                    inv_butterfly(x[a], x[a + (1u << u)], z.x, z.y, pm);
                }
            }
        }
        const u64x2 ni = ninv[2 * mi];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) dst[lo + e * m] = mul_shoup(x[e], ni.x, ni.y, p);
    }
}

}  // namespace k
}  // namespace fhe
