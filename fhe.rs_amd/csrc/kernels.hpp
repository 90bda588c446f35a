// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the fhe.rs BFV hot path.
//
// Layout in HBM: every polynomial is `[rows][N]` u64 row-major exactly as rq::Poly
// (M/rq/mod.rs:126-133); batches add outer dimensions.  Row-wise phases (NTT, key-switch) use
// one workgroup per residue row with the row staged in LDS; column-wise phases (RNS scaler,
// modulus switch) use one lane per coefficient so that all row reads/writes are coalesced
// along N -- no transposes anywhere.  No MFMA: this is 64-bit modular integer arithmetic.
//
// Kernel inventory (reference loop each one replaces):
//   ntt_kernel<false/true>  NttOperator::forward / backward        M/ntt/native.rs:77-233
//   ntt_global_kernel<..>   first/last radix stages for N > 16384   (row does not fit LDS)
//   ks_fused_kernel         KeySwitchingKey::key_switch             F/bfv/keys/key_switching_key.rs:241-320
//                           (+ lazy lift M/rq/mod.rs:563-586 + Shoup MAC M/rq/ops.rs:208-245)
//   ks_mac_kernel           unfused MAC step of the same (N > 16384)
//   scale_kernel            RnsScaler::scale per column             M/rns/scaler.rs:249-352, M/rq/scaler.rs:85-94
//   switch_down_kernel      Poly::switch_down                       M/rq/mod.rs:433-492
//   substitute_kernel       Poly::substitute                        M/rq/mod.rs:360-412
//   ew_kernel / tensor_kernel / mul_shoup_kernel                    M/rq/ops.rs:10-245, F/bfv/ops/mul.rs:198-201
//   synth_kernel            synthetic uniform residues (bench/test inputs)
#pragma once
#include "rt.hpp"
#include "zq_dev.hpp"

namespace fhe {
namespace k {

struct u64x2 {
    u64 x, y;
};

// Maps a workgroup index to (polynomial, row) and to source/destination addresses.
// block b -> poly = b / rows, r = row_begin + b % rows;
//   src = in  + poly*src_poly_stride + (src_row_fixed >= 0 ? src_row_fixed : r) * N
//   dst = out + poly*dst_poly_stride + r * N ;  modulus index = mod_offset + r
struct RowMap {
    uint32_t rows;       // rows processed per polynomial
    uint32_t row_begin;  // first row inside the polynomial
    int32_t mod_offset;  // modulus index of row r is mod_offset + r
    int32_t src_row_fixed;
    u64 src_poly_stride, dst_poly_stride;  // in u64 elements
};

enum { PRO_NONE = 0, PRO_REDUCE = 1 };

// LDS padding: one extra u64 every 16 keeps the 16-element-strided accesses of the last
// radix pass (lane stride 128 B) on distinct banks (ds_read_b64: 64 banks x 4 B, conflicts
// are per 32-lane half; 17*l mod 32 is a bijection).
FHE_HD uint32_t padi(uint32_t i) { return i + (i >> 4); }
FHE_HD uint32_t lds_words(uint32_t n) { return n + (n >> 4) + 2; }

constexpr int GMAX = 4;  // radix-16: up to four butterfly stages per LDS round trip

// ---------------------------------------------------------------- forward passes ----
// Stages [s0, s0+G) of the size-(1<<logm) Cooley-Tukey transform held in `lds`.
// A group = 2^G elements {base + e*lo_count}; all G stages stay in registers.
// Twiddle of (stage st, block i) is tw[(kbase << st) + i]  (kbase = 1 for a whole row;
// (2^G0 + sub) when this LDS tile is sub-block `sub` after G0 global stages).
template <int G>
__device__ __forceinline__ void fwd_pass(u64 *lds, uint32_t logm, uint32_t s0, const u64x2 *tw, uint32_t kbase,
                                         u64 p, u64 p2, uint32_t tid, uint32_t nthreads) {
    constexpr uint32_t R = 1u << G;
    const uint32_t lo_bits = logm - s0 - G;
    const uint32_t ngroups = 1u << (logm - G);
    for (uint32_t grp = tid; grp < ngroups; grp += nthreads) {
        const uint32_t lo = grp & ((1u << lo_bits) - 1);
        const uint32_t hi = grp >> lo_bits;
        const uint32_t base = (hi << (logm - s0)) + lo;
        u64 x[R];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) x[e] = lds[padi(base + (e << lo_bits))];
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t half = R >> (u + 1);
            const uint32_t kst = (kbase << (s0 + u)) + (hi << u);
#pragma unroll
            for (uint32_t blk = 0; blk < (1u << u); blk++) {
                const u64x2 w = tw[kst + blk];
#pragma unroll
                for (uint32_t j = 0; j < half; j++) {
                    const uint32_t a = blk * 2 * half + j;
                    fwd_butterfly(x[a], x[a + half], w.x, w.y, p, p2);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) lds[padi(base + (e << lo_bits))] = x[e];
    }
}

// All stages of a size-(1<<logm) forward transform on an LDS tile (values < 4p on exit).
__device__ __forceinline__ void ntt_fwd_lds(u64 *lds, uint32_t logm, const u64x2 *tw, uint32_t kbase, u64 p,
                                            u64 p2, uint32_t tid, uint32_t nthreads) {
    const uint32_t npass = (logm + GMAX - 1) / GMAX;
    const uint32_t basec = logm / npass, rem = logm % npass;
    uint32_t s0 = 0;
    for (uint32_t pass = 0; pass < npass; pass++) {
        const uint32_t g = basec + (pass < rem ? 1 : 0);
        switch (g) {
            case 1: fwd_pass<1>(lds, logm, s0, tw, kbase, p, p2, tid, nthreads); break;
            case 2: fwd_pass<2>(lds, logm, s0, tw, kbase, p, p2, tid, nthreads); break;
            case 3: fwd_pass<3>(lds, logm, s0, tw, kbase, p, p2, tid, nthreads); break;
            default: fwd_pass<4>(lds, logm, s0, tw, kbase, p, p2, tid, nthreads); break;
        }
        s0 += g;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- inverse passes ----
// Stages [v0, v0+G) (half-lengths 2^v0 .. 2^(v0+G-1)) of the Gentleman-Sande transform.
// Twiddle of (stage v, block i) is itw[koff(v) + i], koff(v) = N - (N >> v) + sub*(M >> (v+1)).
template <int G>
__device__ __forceinline__ void inv_pass(u64 *lds, uint32_t logm, uint32_t v0, const u64x2 *itw, uint32_t logn,
                                         uint32_t sub, u64 p, u64 p2, uint32_t tid, uint32_t nthreads) {
    constexpr uint32_t R = 1u << G;
    const uint32_t ngroups = 1u << (logm - G);
    const uint32_t n = 1u << logn;
    for (uint32_t grp = tid; grp < ngroups; grp += nthreads) {
        const uint32_t lo = grp & ((1u << v0) - 1);
        const uint32_t hi = grp >> v0;
        const uint32_t base = (hi << (v0 + G)) + lo;
        u64 x[R];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) x[e] = lds[padi(base + (e << v0))];
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t v = v0 + u;
            const uint32_t nblk = R >> (u + 1);
            const uint32_t kst = n - (n >> v) + (sub << (logm - v - 1)) + hi * nblk;
#pragma unroll
            for (uint32_t blk = 0; blk < nblk; blk++) {
                const u64x2 z = itw[kst + blk];
#pragma unroll
                for (uint32_t j = 0; j < (1u << u); j++) {
                    const uint32_t a = blk * (2u << u) + j;
                    inv_butterfly(x[a], x[a + (1u << u)], z.x, z.y, p, p2);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) lds[padi(base + (e << v0))] = x[e];
    }
}

__device__ __forceinline__ void ntt_inv_lds(u64 *lds, uint32_t logm, const u64x2 *itw, uint32_t logn,
                                            uint32_t sub, u64 p, u64 p2, uint32_t tid, uint32_t nthreads) {
    const uint32_t npass = (logm + GMAX - 1) / GMAX;
    const uint32_t basec = logm / npass, rem = logm % npass;
    uint32_t v0 = 0;
    for (uint32_t pass = 0; pass < npass; pass++) {
        const uint32_t g = basec + (pass < rem ? 1 : 0);
        switch (g) {
            case 1: inv_pass<1>(lds, logm, v0, itw, logn, sub, p, p2, tid, nthreads); break;
            case 2: inv_pass<2>(lds, logm, v0, itw, logn, sub, p, p2, tid, nthreads); break;
            case 3: inv_pass<3>(lds, logm, v0, itw, logn, sub, p, p2, tid, nthreads); break;
            default: inv_pass<4>(lds, logm, v0, itw, logn, sub, p, p2, tid, nthreads); break;
        }
        v0 += g;
        __syncthreads();
    }
}

// ------------------------------------------------------------------- NTT kernel ----
// One workgroup per (row, sub-block).  grid.x = npolys * map.rows * nsub, nsub = 2^(logn-logm).
// logm == logn: whole row in LDS (N <= 16384).  logm < logn: this is the LDS half of the
// two-kernel transform for N = 32768 (ntt_global_kernel does the other logn-logm stages).
//   forward: canonical output (reduce3, native.rs:238-246) unless `lazy_out`
//   inverse: multiplies by N^-1 (Shoup) when logm == logn (native.rs:229-232)
template <bool INVERSE>
__global__ void ntt_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map,
                           const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw,
                           const u64x2 *__restrict__ ninv, uint32_t logn, uint32_t logm, uint32_t prologue) {
    FHE_DYN_SMEM(u64, lds);
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t n = 1u << logn, m = 1u << logm;
    const uint32_t nsub = 1u << (logn - logm);
    const uint32_t sub = blockIdx.x % nsub;
    const uint32_t rowb = blockIdx.x / nsub;
    const uint32_t poly = rowb / map.rows;
    const uint32_t r = map.row_begin + rowb % map.rows;
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p, p2 = md.p2;
    const u64 *src = in + (u64)poly * map.src_poly_stride +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n + (u64)sub * m;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n + (u64)sub * m;
    const u64x2 *twr = tw + (u64)mi * n;

    for (uint32_t i = tid; i < m; i += nthreads) {
        u64 v = src[i];
        if (prologue == PRO_REDUCE) v = reduce_u64(v, md);
        lds[padi(i)] = v;
    }
    __syncthreads();
    if (!INVERSE) {
        ntt_fwd_lds(lds, logm, twr, nsub + sub, p, p2, tid, nthreads);
        for (uint32_t i = tid; i < m; i += nthreads) dst[i] = csub(csub(lds[padi(i)], p2), p);
    } else {
        ntt_inv_lds(lds, logm, twr, logn, sub, p, p2, tid, nthreads);
        if (logm == logn) {
            const u64x2 ni = ninv[mi];
            for (uint32_t i = tid; i < m; i += nthreads) dst[i] = mul_shoup(lds[padi(i)], ni.x, ni.y, p);
        } else {
            for (uint32_t i = tid; i < m; i += nthreads) dst[i] = lds[padi(i)];  // < 2p, finished by global pass
        }
    }
}

// Radix stages that span sub-blocks, done straight on global memory (coalesced along the
// low index).  Forward: stages [0, G0) before the LDS kernel (output < 4p, the LDS kernel's
// loader accepts that range).  Inverse: stages [logm, logn) after it, then N^-1.
// One thread per group of 2^G0 elements {lo + e*M}; grid.x covers npolys*rows*(M/threads).
template <bool INVERSE, int G0>
__global__ void ntt_global_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map,
                                  const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw,
                                  const u64x2 *__restrict__ ninv, uint32_t logn, uint32_t prologue) {
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
    const u64 p = md.p, p2 = md.p2;
    const u64 *src = in + (u64)poly * map.src_poly_stride +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n;
    const u64x2 *twr = tw + (u64)mi * n;
    u64 x[R];
#pragma unroll
    for (uint32_t e = 0; e < R; e++) {
        u64 v = src[lo + e * m];
        if (prologue == PRO_REDUCE) v = reduce_u64(v, md);
        x[e] = v;
    }
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
                    fwd_butterfly(x[a], x[a + half], w.x, w.y, p, p2);
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
                    inv_butterfly(x[a], x[a + (1u << u)], z.x, z.y, p, p2);
                }
            }
        }
        const u64x2 ni = ninv[mi];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) dst[lo + e * m] = mul_shoup(x[e], ni.x, ni.y, p);
    }
}

// ------------------------------------------------------------ fused key switch ----
// (c0, c1)[b][j] (+)= sum_i NTT_j( [p_i]_{q_j} ) (.) (k0, k1)[i][j]   for one (b, j) per workgroup.
// The digit rows p[b][i][:] are lifted (reduced mod q_j), transformed in LDS and multiplied
// into per-thread register accumulators; the key streams from L2/MALL (shared by the batch).
// A non-null addend0/addend1 is added to the respective output (fused relinearisation add,
// F/bfv/ops/mul.rs:224-225; rotation adds substitute(c0) to c0 only); canonical outputs.
// Thread t owns coefficients {t + e*nthreads}, EPT = ceil(N / nthreads) of them.
template <int EPT>
__global__ void ks_fused_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0,
                                u64 *__restrict__ out1, u64 out_poly_stride, const u64 *__restrict__ addend0,
                                const u64 *__restrict__ addend1, u64 addend_poly_stride,
                                const u64 *__restrict__ k0, const u64 *__restrict__ k0s,
                                const u64 *__restrict__ k1, const u64 *__restrict__ k1s,
                                const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw, uint32_t logn,
                                uint32_t ndigits, uint32_t lk, uint32_t digit_shift_bits) {
    FHE_DYN_SMEM(u64, lds);
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t n = 1u << logn;
    const uint32_t j = blockIdx.x % lk, b = blockIdx.x / lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const u64x2 *twr = tw + (u64)j * n;
    u64 acc0[EPT], acc1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) acc0[e] = acc1[e] = 0;
    for (uint32_t i = 0; i < ndigits; i++) {
        // digit_shift_bits == 0: digit i is residue row i of p (RNS decomposition, :256-268).
        // otherwise: base-2^bits digits of the single row 0 (key_switch_decomposition, :323-362).
        const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * n);
        for (uint32_t x = tid; x < n; x += nthreads) {
            u64 v = src[x];
            if (digit_shift_bits) v = (v >> (i * digit_shift_bits)) & ((1ull << digit_shift_bits) - 1);
            lds[padi(x)] = reduce_u64(v, md);
        }
        __syncthreads();
        ntt_fwd_lds(lds, logn, twr, 1, p, p2, tid, nthreads);
        const u64 koff = ((u64)i * lk + j) * n;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const uint32_t x = tid + e * nthreads;
            if (x < n) {
                const u64 v = lds[padi(x)];  // < 4p: Shoup multiplication accepts any u64
                acc0[e] = csub(acc0[e] + mul_shoup_lazy(v, k0[koff + x], k0s[koff + x], p), p2);
                acc1[e] = csub(acc1[e] + mul_shoup_lazy(v, k1[koff + x], k1s[koff + x], p), p2);
            }
        }
        __syncthreads();
    }
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * n;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * n;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const uint32_t x = tid + e * nthreads;
        if (x < n) {
            u64 r0 = csub(acc0[e], p), r1 = csub(acc1[e], p);
            if (addend0) r0 = add_mod(r0, addend0[aoff + x], p);
            if (addend1) r1 = add_mod(r1, addend1[aoff + x], p);
            out0[ooff + x] = r0;
            out1[ooff + x] = r1;
        }
    }
}

// Unfused MAC step (used when the row does not fit LDS): t is NTT(lift(p_i)) [b][lk][N]
// canonical; out (+)= t (.) key_i.  first != 0 initialises the accumulators.
__global__ void ks_mac_kernel(const u64 *__restrict__ t, u64 *__restrict__ out0, u64 *__restrict__ out1,
                              u64 out_poly_stride, const u64 *__restrict__ k0, const u64 *__restrict__ k0s,
                              const u64 *__restrict__ k1, const u64 *__restrict__ k1s,
                              const DevMod *__restrict__ mods, uint32_t logn, uint32_t lk, uint32_t digit,
                              uint32_t first, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const uint32_t x = (uint32_t)(gid & (n - 1));
    const uint32_t j = (uint32_t)((gid >> logn) % lk);
    const u64 b = (gid >> logn) / lk;
    const u64 p = mods[j].p;
    const u64 koff = ((u64)digit * lk + j) * n + x;
    const u64 ooff = b * out_poly_stride + (u64)j * n + x;
    const u64 v = t[gid];
    u64 r0 = mul_shoup(v, k0[koff], k0s[koff], p), r1 = mul_shoup(v, k1[koff], k1s[koff], p);
    if (!first) {
        r0 = add_mod(r0, out0[ooff], p);
        r1 = add_mod(r1, out1[ooff], p);
    }
    out0[ooff] = r0;
    out1[ooff] = r1;
}

// Base-2^bits digit extraction for the unfused decomposition path.
__global__ void digit_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, uint32_t shift, uint32_t bits,
                             u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) out[gid] = (in[gid] >> shift) & ((1ull << bits) - 1);
}

// ------------------------------------------------------------------ RNS scaler ----
struct ScalerDev {
    const u64 *gamma, *gamma_shoup;                      // [nto]
    const u64 *omega, *omega_shoup;                      // [nto][nfrom]
    const u64 *theta_omega_lo, *theta_omega_hi;          // [nfrom]
    const u64 *theta_omega_sign;                         // [nfrom] (0/1)
    const u64 *theta_garner_lo, *theta_garner_hi;        // [nfrom]
    u64 theta_gamma_lo, theta_gamma_hi;
    uint32_t theta_gamma_sign, is_one, shift, nfrom, nto, ncommon;
};

// One lane per coefficient column (RnsScaler::scale, M/rns/scaler.rs:249-352): the 256-bit
// fixed-point sums v and w are reproduced limb for limb; the per-target accumulation
// y = -v*gamma (+/- w) + sum_j r_j*omega_j only matters mod q (the reference ends with
// reduce_u128), so it is kept lazily in [0, 2q) in one word.
// in: [npolys][nfrom][N] PowerBasis; out: rows [ncommon, nto) of [npolys][nto][N].
__global__ void scale_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                             u64 out_poly_stride, ScalerDev s, const DevMod *__restrict__ to_mods, uint32_t logn,
                             u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *rests = in + poly * in_poly_stride + col;

    U256 sum = {0, 0, 0, 0};
    for (uint32_t i = 0; i < s.nfrom; i++)
        u256_mac_64x128(sum, rests[(u64)i * n], s.theta_garner_lo[i], s.theta_garner_hi[i], false);
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
        U256 t = {0, 0, 0, 0};
        for (uint32_t i = 0; i < s.nfrom; i++)
            u256_mac_64x128(t, rests[(u64)i * n], s.theta_omega_lo[i], s.theta_omega_hi[i],
                            s.theta_omega_sign[i] != 0);
        // t -/+= v * theta_gamma  (128 x 128 -> 256 wrapping): low word then high word << 64
        const bool neg = !s.theta_gamma_sign;
        u256_mac_64x128(t, vlo, s.theta_gamma_lo, s.theta_gamma_hi, neg);
        {
            U256 sh = {t.w1, t.w2, t.w3, 0};
            u256_mac_64x128(sh, vhi, s.theta_gamma_lo, s.theta_gamma_hi, neg);
            t.w1 = sh.w0;
            t.w2 = sh.w1;
            t.w3 = sh.w2;
        }
        w_sign = ((t.w2 >> 63) | t.w3) != 0;
        if (w_sign) {
            U256 nt = {~t.w0, ~t.w1, ~t.w2, ~t.w3};
            u256_shr_lo128(nt, 126, wlo, whi);
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
    u64 *o = out + poly * out_poly_stride + col;
    for (uint32_t jt = s.ncommon; jt < s.nto; jt++) {
        const DevMod q = to_mods[jt];
        const u64 *om = s.omega + (u64)jt * s.nfrom, *oms = s.omega_shoup + (u64)jt * s.nfrom;
        u64 y = q.p2 - mul_shoup_lazy(reduce_u128(vhi, vlo, q), s.gamma[jt], s.gamma_shoup[jt], q.p);  // (0, 2q]
        y = csub(y, q.p2);
        if (!s.is_one) {
            const u64 wi = reduce_u128(whi, wlo, q);  // [0, q)
            y = csub(y + (w_sign ? q.p2 - wi : wi), q.p2);
            y = csub(y, q.p2);
        }
        for (uint32_t i = 0; i < s.nfrom; i++)
            y = csub(y + mul_shoup_lazy(rests[(u64)i * n], om[i], oms[i], q.p), q.p2);
        o[(u64)jt * n] = csub(y, q.p);
    }
}

// ----------------------------------------------------------------- switch_down ----
// Poly::switch_down (M/rq/mod.rs:433-492), one lane per coefficient:
// in [npolys][L][N] PowerBasis -> out [npolys][L-1][N].
__global__ void switch_down_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                   u64 out_poly_stride, const DevMod *__restrict__ mods,
                                   const u64x2 *__restrict__ inv_last, uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *src = in + poly * in_poly_stride + col;
    u64 *dst = out + poly * out_poly_stride + col;
    const DevMod ql = mods[nmod - 1];
    const u64 half = ql.p >> 1;
    const u64 last = add_mod(src[(u64)(nmod - 1) * n], half, ql.p);
    for (uint32_t r = 0; r + 1 < nmod; r++) {
        const DevMod qi = mods[r];
        const u64 half_mod = qi.p - reduce_u64(half, qi);       // (0, qi]
        const u64 tmp = reduce_u64(last, qi) + half_mod;        // < 2 qi
        const u64 c = src[(u64)r * n] + 3 * qi.p - tmp;         // < 4 qi
        dst[(u64)r * n] = mul_shoup(c, inv_last[r].x, inv_last[r].y, qi.p);
    }
}

// ------------------------------------------------------------------ substitute ----
// Poly::substitute (M/rq/mod.rs:360-412).  One lane per (row, j).
__global__ void substitute_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                  u64 out_poly_stride, const DevMod *__restrict__ mods, uint32_t nmod,
                                  uint32_t logn, uint32_t exponent, uint32_t repr_is_ntt, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn, mask = n - 1;
    const uint32_t j = (uint32_t)(gid & mask);
    const uint32_t r = (uint32_t)((gid >> logn) % nmod);
    const u64 poly = (gid >> logn) / nmod;
    const u64 *src = in + poly * in_poly_stride + (u64)r * n;
    u64 *dst = out + poly * out_poly_stride + (u64)r * n;
    if (repr_is_ntt) {
        // q[bitrev[j]] = p[bitrev((e-1)/2 + j*e mod N)]; index the gather by destination d = bitrev(j)
        const uint32_t d = j;
        const uint32_t jj = __brev(d) >> (32 - logn);
        const uint32_t srci = (uint32_t)(((u64)(exponent - 1) / 2 + (u64)jj * exponent) & mask);
        dst[d] = src[__brev(srci) >> (32 - logn)];
    } else {
        const u64 power = (u64)j * exponent;
        const u64 v = src[j];
        dst[power & mask] = (power & n) ? neg_mod(v, mods[r].p) : v;
    }
}

// ----------------------------------------------------------- element-wise kernels ----
enum { EW_ADD = 0, EW_SUB = 1, EW_MUL = 2, EW_NEG = 3 };
// a op= b on [rows_total][N]; modulus index = row % nmod (M/rq/ops.rs:10-206, 354-418).
__global__ void ew_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, const DevMod *__restrict__ mods,
                          uint32_t nmod, uint32_t logn, uint32_t op, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const DevMod m = mods[(gid >> logn) % nmod];
    const u64 x = a[gid];
    u64 r;
    switch (op) {
        case EW_ADD: r = add_mod(x, b[gid], m.p); break;
        case EW_SUB: r = sub_mod(x, b[gid], m.p); break;
        case EW_MUL: r = mul_mod(x, b[gid], m); break;
        default: r = neg_mod(x, m.p); break;
    }
    a[gid] = r;
}
__global__ void mul_shoup_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, const u64 *__restrict__ bs,
                                 const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 p = mods[(gid >> logn) % nmod].p;
    a[gid] = mul_shoup(a[gid], b[gid], bs[gid], p);
}
// Tensor step of Multiplicator::multiply (F/bfv/ops/mul.rs:198-201) on extended polys:
// ext [npolys][4][K][N] = (c00, c01, c10, c11) -> t [npolys][3][K][N] = (c00*c10, c00*c11 + c01*c10, c01*c11).
__global__ void tensor_kernel(const u64 *__restrict__ ext, u64 *__restrict__ t, const DevMod *__restrict__ mods,
                              uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 pn = (u64)nmod << logn;  // elements per polynomial
    const u64 b = gid / pn, off = gid % pn;
    const DevMod m = mods[off >> logn];
    const u64 *e = ext + b * 4 * pn + off;
    const u64 c00 = e[0], c01 = e[pn], c10 = e[2 * pn], c11 = e[3 * pn];
    u64 *o = t + b * 3 * pn + off;
    o[0] = mul_mod(c00, c10, m);
    o[pn] = add_mod(mul_mod(c00, c11, m), mul_mod(c01, c10, m), m.p);
    o[2 * pn] = mul_mod(c01, c11, m);
}
// Copies the first `rows` rows of each polynomial: in [npolys][in_rows][N] -> out [npolys][out_rows][N].
__global__ void copy_rows_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                 u64 out_poly_stride, u64 per_poly, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 poly = gid / per_poly, off = gid % per_poly;
    out[poly * out_poly_stride + off] = in[poly * in_poly_stride + off];
}
// x = splitmix64(seed ^ (ct<<40) ^ (part<<36) ^ (row<<28) ^ coeff) mod q_row  
__global__ void synth_kernel(u64 *__restrict__ out, const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn,
                             uint32_t nparts, u64 seed, u64 ct0, u64 part0, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const u64 coeff = gid & (n - 1);
    const u64 rowi = gid >> logn;
    const u64 row = rowi % nmod;
    const u64 part = part0 + (rowi / nmod) % nparts;
    const u64 ct = ct0 + rowi / ((u64)nmod * nparts);
    const u64 v = splitmix64(seed ^ (ct << 40) ^ (part << 36) ^ (row << 28) ^ coeff);
    out[gid] = v % mods[row].p;
}

}  // namespace k
}  // namespace fhe
