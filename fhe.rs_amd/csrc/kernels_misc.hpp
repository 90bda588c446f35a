// kernels_misc.hpp -- the streaming kernels: switch_down, substitute, element-wise ops, tensor (unfused), dot product,
// ct x pt, decrypt phase / tail, wire pack / unpack, oblivious expansion, seeded polynomial, copy, synthetic inputs.
#pragma once
#include "kernels_common.hpp"

namespace fhe {
namespace k {

// ----------------------------------------------------------------- switch_down ----
// Poly::switch_down (M/rq/mod.rs:433-492), one lane per coefficient:
// in [npolys][L][N] PowerBasis -> out [npolys][L-1][N].
// `in` and `out` MAY ALIAS with equal strides (switch_down_to_pb walks the levels in one scratch block): hence no
// __restrict__ on them, and the in-place contract is what the loop below already does -- a lane touches only its own
// column, reads the last row first, and reads row r before it writes row r.
__global__ void switch_down_kernel(const u64 *in, u64 *out, u64 in_poly_stride,
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
        dst[j] = src[galois_src_index(j, exponent, logn)];
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
// Tensor step of Multiplicator::multiply (F/bfv/ops/mul.rs:198-201).  Operand polynomials
// (c00, c01) = extL[b][0..1], (c10, c11) = extR[b][0..1], each [K][N]; rows below `ncommon`
// are read from the original ciphertexts lhs/rhs [b][2][L][N] when those pointers are given
// (the extender copies them verbatim, M/rq/scaler.rs:61-65, so the copy is skipped).
// t is slot-major: t[slot][b][K][N] = (c00*c10, c00*c11 + c01*c10, c01*c11).
__global__ void tensor_kernel(const u64 *__restrict__ extL, const u64 *__restrict__ extR,
                              const u64 *__restrict__ lhs, const u64 *__restrict__ rhs, u64 *__restrict__ t,
                              const DevMod *__restrict__ mods, uint32_t nmod, uint32_t ncommon, uint32_t lrows,
                              uint32_t logn, u64 nb, uint32_t debug_acquire) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (debug_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    // grid: x = chunks of one extended polynomial, y = ciphertext pair (no runtime divisions)
    const u64 pn = (u64)nmod << logn;  // elements per extended polynomial
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pn) return;
    const u64 b = blockIdx.y;
    const uint32_t row = (uint32_t)(off >> logn);
    const DevMod m = mods[row];
    u64 c00, c01, c10, c11;
    if (lhs && row < ncommon) {
        const u64 pl = (u64)lrows << logn;
        c00 = lhs[b * 2 * pl + off];
        c01 = lhs[b * 2 * pl + pl + off];
        c10 = rhs[b * 2 * pl + off];
        c11 = rhs[b * 2 * pl + pl + off];
    } else {
        c00 = extL[b * 2 * pn + off];
        c01 = extL[b * 2 * pn + pn + off];
        c10 = extR[b * 2 * pn + off];
        c11 = extR[b * 2 * pn + pn + off];
    }
    u64 *o = t + b * pn + off;
    if (debug_acquire == 2) {  // developer aid: dump the operands as read
        o[0] = c00;
        o[nb * pn] = c10;
        o[2 * nb * pn] = c01;
        return;
    }
    o[0] = mul_mod(c00, c10, m);
    {
        const u128_t sum = (u128_t)c00 * c11 + (u128_t)c01 * c10;  // one reduction, see tensor_intt_kernel
        o[nb * pn] = barrett_reduce_wide((u64)(sum >> 64), (u64)sum, m);
    }
    o[2 * nb * pn] = mul_mod(c01, c11, m);
}
// dot_product_scalar / rq::dot_product (F/bfv/ops/dot_product.rs:54-180, M/rq/ops.rs:449-570):
// out[b][part][row][c] = sum_k cts[b][k][part][row][c] * pts[b][k][row][c]  mod q_row.
// One lane per pair of coefficients and ALL `NP` parts of the group starting at blockIdx.y*NP
// (each plaintext word is loaded once); exact 128-bit products accumulated in 192 bits and
// reduced once (the reference's periodic reduce_u128 gives the same canonical sum).
// Streaming, HBM bound.
template <int NP>
__global__ void __launch_bounds__(256, 8) dot_kernel(const u64 *__restrict__ cts, u64 ct_batch_stride, const u64 *__restrict__ pts,
                           u64 pt_batch_stride, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                           const u64x2 *__restrict__ pow2 /* {2^64, 2^128} mod q */, uint32_t nparts, uint32_t count,
                           uint32_t logn, u64 pl /* L*N */) {
    // grid: x = pairs of coefficients of one polynomial, y = group of NP parts, z = batch
    const u64 pair = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * pair >= pl) return;
    const u64 off = 2 * pair;
    const uint32_t part0 = blockIdx.y * NP, b = blockIdx.z;
    const uint32_t row = (uint32_t)(off >> logn);
    const DevMod m = mods[row];
    const u64 *cp = cts + (u64)b * ct_batch_stride + (u64)part0 * pl + off;
    const u64 *pp = pts + (u64)b * pt_batch_stride + off;
    Acc192 a0[NP], a1[NP];
#pragma unroll 4   // (2 -> 4: +2 %; 8: no further gain -- the kernel runs at 4.0 TB/s of fabric reads, PMC FETCH_SIZE)
    for (uint32_t k = 0; k < count; k++) {
        // (an operand array that is private to this batch element -- stride != 0 -- is read exactly once: streaming)
        const u64x2 *yp = reinterpret_cast<const u64x2 *>(pp + (u64)k * pl);
        const u64x2 y = pt_batch_stride ? load_stream(yp) : *yp;
#pragma unroll
        for (int q = 0; q < NP; q++) {
            if (part0 + q < nparts) {
                const u64x2 *xp = reinterpret_cast<const u64x2 *>(cp + ((u64)k * nparts + q) * pl);
                const u64x2 x = ct_batch_stride ? load_stream(xp) : *xp;
                mac192(a0[q], x.x, y.x);
                mac192(a1[q], x.y, y.y);
            }
        }
    }
    // value = top * 2^128 + a: reduce a, then add (top mod q) * (2^128 mod q)
    const u64 c128 = pow2[row].y;
    auto fold = [&](const Acc192 &acc) -> u64 {
        u128_t a;
        u64 top;
        acc192_resolve(acc, a, top);
        const u64 r = reduce_u128((u64)(a >> 64), (u64)a, m);
        return top ? add_mod(r, mul_mod(reduce_u64(top, m), c128, m), m.p) : r;
    };
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if (part0 + q < nparts) {
            u64x2 o;
            o.x = fold(a0[q]);
            o.y = fold(a1[q]);
            *reinterpret_cast<u64x2 *>(out + ((u64)b * nparts + part0 + q) * pl + off) = o;
        }
    }
}

// General tensor step of `&ct * &ct` (F/bfv/ops/mod.rs:300-327): out[b][k] = sum_{i+j=k} a[b][i] (.) b[b][j];
// grid = (ceil(pl / block), la + lb - 1, batch).
__global__ void tensor_general_kernel(const u64 *__restrict__ a, const u64 *__restrict__ bb, u64 *__restrict__ out,
                                      const DevMod *__restrict__ mods, uint32_t la, uint32_t lb, uint32_t logn, u64 pl) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t kk = blockIdx.y, b = blockIdx.z;
    const DevMod m = mods[off >> logn];
    const u64 *pa = a + (u64)b * la * pl + off, *pb = bb + (u64)b * lb * pl + off;
    u64 acc = 0;
    for (uint32_t i = 0; i < la; i++) {
        if (kk < i || kk - i >= lb) continue;
        acc = add_mod(acc, mul_mod(pa[(u64)i * pl], pb[(u64)(kk - i) * pl], m), m.p);
    }
    out[((u64)b * (la + lb - 1) + kk) * pl + off] = acc;
}

// `Ciphertext * Plaintext` (F/bfv/ops/mod.rs:229-257): out[b][part] = ct[b][part] (.) pt[b].
// One lane per 16-byte chunk of the polynomial and ALL parts (the plaintext word is loaded once); the ciphertext and
// the result are touched once each: streaming accesses.  grid = (ceil(pl / 2 / block), 1, batch).
// (Round 3's form -- one lane per coefficient and part, 8-byte accesses, the plaintext read once per part -- ran at
// 0.53 of 8 TB/s; profiles/r04_next_rows_ab.txt.)
__global__ void __launch_bounds__(256)
    mul_plain_kernel(const u64 *__restrict__ ct, const u64 *__restrict__ pt, u64 pt_batch_stride, u64 *__restrict__ out,
                     const DevMod *__restrict__ mods, uint32_t nparts, uint32_t logn, u64 pl) {
    const u64 pair = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * pair >= pl) return;
    const u64 off = 2 * pair;
    const uint32_t b = blockIdx.z;
    const DevMod m = mods[off >> logn];
    const u64x2 *pp = reinterpret_cast<const u64x2 *>(pt + (u64)b * pt_batch_stride + off);
    const u64x2 y = pt_batch_stride ? load_stream(pp) : *pp;   // (a plaintext shared by the batch is re-read: cached)
    for (uint32_t part = 0; part < nparts; part++) {
        const u64 idx = ((u64)b * nparts + part) * pl + off;
        const u64x2 x = load_stream(reinterpret_cast<const u64x2 *>(ct + idx));
        store_stream(reinterpret_cast<u64x2 *>(out + idx), u64x2{mul_mod(x.x, y.x, m), mul_mod(x.y, y.y, m)});
    }
}

// SecretKey::try_decrypt (F/bfv/keys/secret_key.rs:205-247).  phase_kernel: out[b] = sum_i
// ct[b][i] (.) s^i by Horner's rule (same canonical value as the reference's running power of s);
// grid = (ceil(L*N / block), batch).  decrypt_tail_kernel: ((d_0 + t) mod q_0) mod t on row 0
// of the scaled polynomial.
__global__ void phase_kernel(const u64 *__restrict__ ct, const u64 *__restrict__ sk, u64 *__restrict__ out,
                             const DevMod *__restrict__ mods, uint32_t nparts, uint32_t logn, u64 pl) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t b = blockIdx.y;
    const DevMod m = mods[off >> logn];
    const u64 *c = ct + (u64)b * nparts * pl + off;
    const u64 sv = sk[off];
    u64 acc = c[(u64)(nparts - 1) * pl];
    for (uint32_t i = nparts - 1; i-- > 0;) acc = add_mod(mul_mod(acc, sv, m), c[(u64)i * pl], m.p);
    out[(u64)b * pl + off] = acc;
}
__global__ void decrypt_tail_kernel(const u64 *__restrict__ d, u64 d_poly_stride, u64 *__restrict__ out, DevMod q0,
                                    DevMod tm, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 b = gid >> logn, x = gid & ((1ull << logn) - 1);
    const u64 w = reduce_u64(d[b * d_poly_stride + x] + tm.p, q0);
    out[gid] = reduce_u64(w, tm);
}

// Rq wire format (crates/fhe-util/src/lib.rs:71-148 via M/zq/mod.rs:783-793): a row is N
// coefficients of nbits = bitlen(p - 1) bits, little-endian bit-packed.  Eight coefficients are
// exactly nbits bytes, so one thread transcodes one such group with the reference's shift
// register; grid = (ceil(N/8 / block), L, npolys).  (Boundary work: byte-granular accesses.)
__device__ __forceinline__ uint32_t wire_bits(u64 p) { return 64u - (uint32_t)__builtin_clzll(p - 1); }
__device__ __forceinline__ u64 wire_row_offset(const DevMod *mods, uint32_t r, uint32_t logn) {
    u64 off = 0;
    for (uint32_t i = 0; i < r; i++) off += (u64)wire_bits(mods[i].p) << (logn - 3);
    return off;
}
__global__ void wire_pack_kernel(const u64 *__restrict__ polys, uint8_t *__restrict__ bytes,
                                 const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << (logn - 3))) return;
    const uint32_t r = blockIdx.y, poly = blockIdx.z;
    const uint32_t nbits = wire_bits(mods[r].p);
    const u64 mask = ~0ull >> (64 - nbits);
    const u64 *src = polys + (((u64)poly * nmod + r) << logn) + 8u * g;
    uint8_t *dst = bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn) + (u64)g * nbits;
    u128_t cur = 0;
    uint32_t have = 0, o = 0;
    for (uint32_t e = 0; e < 8; e++) {
        cur |= (u128_t)(src[e] & mask) << have;
        have += nbits;
        while (have >= 8) {
            dst[o++] = (uint8_t)cur;
            cur >>= 8;
            have -= 8;
        }
    }
}
__global__ void wire_unpack_kernel(const uint8_t *__restrict__ bytes, u64 *__restrict__ polys,
                                   const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << (logn - 3))) return;
    const uint32_t r = blockIdx.y, poly = blockIdx.z;
    const uint32_t nbits = wire_bits(mods[r].p);
    const u64 mask = ~0ull >> (64 - nbits);
    const uint8_t *src = bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn) + (u64)g * nbits;
    u64 *dst = polys + (((u64)poly * nmod + r) << logn) + 8u * g;
    u128_t cur = 0;
    uint32_t have = 0, i = 0;
    for (uint32_t e = 0; e < 8; e++) {
        while (have < nbits) {
            cur |= (u128_t)src[i++] << have;
            have += 8;
        }
        dst[e] = (u64)cur & mask;
        cur >>= nbits;
        have -= nbits;
    }
}

// The same transcoding for rows of 128 coefficients and more (a row is then a whole number of 16-byte words on both
// sides), shaped for the memory system instead of after the reference's shift register (round 4: the kernels above
// wrote / read single bytes at a stride of nbits per lane and ran at 0.06 / 0.11 of the HBM roofline):
// one lane per 16 bytes of OUTPUT, written (pack) or read-combined (unpack) with full-width coalesced accesses; the
// other side is read at word granularity -- neighbouring lanes touch neighbouring or the same words, which the vector
// cache merges, so HBM traffic stays at one pass over both arrays.
// pack: output word j holds stream bits [64 j, 64 j + 64): coefficient c0 = 64 j / nbits from bit 64 j - c0 nbits on,
// then the following coefficients shifted up, until 64 bits are there.
__device__ __forceinline__ u64 wire_pack_word(const u64 *__restrict__ row, uint32_t j, uint32_t nbits, u64 mask, uint32_t n) {
    const uint32_t bit = 64u * j;
    uint32_t c = bit / nbits;
    const uint32_t off = bit - c * nbits;            // bits of coefficient c already consumed by word j - 1
    u64 w = (row[c] & mask) >> off;
    uint32_t have = nbits - off;
    while (have < 64 && ++c < n) {
        w |= (row[c] & mask) << have;                // (bits beyond 64 fall off: they belong to word j + 1)
        have += nbits;
    }
    return w;
}
__global__ void __launch_bounds__(256)
    wire_pack_words_kernel(const u64 *__restrict__ polys, uint8_t *__restrict__ bytes, const DevMod *__restrict__ mods,
                           uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t r = blockIdx.y, poly = blockIdx.z, n = 1u << logn;
    const uint32_t nbits = wire_bits(mods[r].p);
    const uint32_t pairs = (n >> 7) * nbits;         // 16-byte words of the packed row: n * nbits / 128
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pairs) return;
    const u64 mask = ~0ull >> (64 - nbits);
    const u64 *row = polys + (((u64)poly * nmod + r) << logn);
    u64x2 o;
    o.x = wire_pack_word(row, 2 * g, nbits, mask, n);
    o.y = wire_pack_word(row, 2 * g + 1, nbits, mask, n);
    reinterpret_cast<u64x2 *>(bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn))[g] = o;
}
// unpack: coefficient e is stream bits [e nbits, (e + 1) nbits): at most two 64-bit words of the packed row
__device__ __forceinline__ u64 wire_unpack_coeff(const u64 *__restrict__ words, uint32_t e, uint32_t nbits, u64 mask) {
    const u64 bit = (u64)e * nbits;
    const uint32_t w = (uint32_t)(bit >> 6), off = (uint32_t)(bit & 63);
    u64 v = words[w] >> off;
    if (off + nbits > 64) v |= words[w + 1] << (64 - off);     // (only then does the coefficient reach into word w + 1)
    return v & mask;
}
__global__ void __launch_bounds__(256)
    wire_unpack_words_kernel(const uint8_t *__restrict__ bytes, u64 *__restrict__ polys, const DevMod *__restrict__ mods,
                             uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t r = blockIdx.y, poly = blockIdx.z, n = 1u << logn;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * g >= n) return;
    const uint32_t nbits = wire_bits(mods[r].p);
    const u64 mask = ~0ull >> (64 - nbits);
    const u64 *words = reinterpret_cast<const u64 *>(bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn));
    u64x2 o;
    o.x = wire_unpack_coeff(words, 2 * g, nbits, mask);
    o.y = wire_unpack_coeff(words, 2 * g + 1, nbits, mask);
    reinterpret_cast<u64x2 *>(polys + (((u64)poly * nmod + r) << logn))[g] = o;
}

// Oblivious expansion (F/bfv/keys/evaluation_key.rs:233-244).  monomial_kernel writes the
// PowerBasis polynomials -x^(N - 2^l), l < nlev, into a zeroed [nlev][L][N] buffer (the forward
// NTT follows); expand_step_kernel does, per coefficient of the polynomials of the lower half,
// high = (low - sub) (.) monomial  (only the first nhigh polynomials of the upper half exist)
// and low += sub.  grid = (ceil(L*N / block), npolys).
__global__ void monomial_kernel(u64 *__restrict__ buf, const DevMod *__restrict__ mods, uint32_t nlev, uint32_t nmod,
                                uint32_t logn) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nlev * nmod) return;
    const uint32_t lev = gid / nmod, r = gid % nmod, n = 1u << logn;
    buf[((u64)lev * nmod + r) * n + (n - (1u << lev))] = mods[r].p - 1;
}
__global__ void expand_step_kernel(u64 *__restrict__ low, const u64 *__restrict__ sub, u64 *__restrict__ high,
                                   const u64 *__restrict__ mono, const DevMod *__restrict__ mods, uint32_t logn, u64 pl,
                                   uint32_t nhigh) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t poly = blockIdx.y;
    const DevMod m = mods[off >> logn];
    const u64 idx = (u64)poly * pl + off;
    const u64 lo = low[idx], sb = sub[idx];
    if (poly < nhigh) high[idx] = mul_mod(sub_mod(lo, sb, m.p), mono[off], m);
    low[idx] = add_mod(lo, sb, m.p);
}

// ------------------------------------------------------ seeded polynomial (wire c1) ----
// Poly::random_from_seed (M/rq/mod.rs:276-292), the `c1` a received secret-key ciphertext expands from its 32-byte
// seed (F/bfv/ciphertext.rs:287-302): key = SHA-256(seed); one ChaCha8 stream (64-bit block counter from 0, stream
// id 0; a u64 = two consecutive little-endian words); residue row after residue row, `degree` draws each from
// Uniform[0, q_i) by Lemire's widening-multiply rejection: x -> (hi, lo) = x * q, accept hi when
// lo >= (2^64 - q) mod q.  The stream position of a coefficient depends on the rejections before it, so one
// workgroup walks one polynomial: every thread computes one ChaCha block (8 candidates), an exclusive scan of the
// accept counts places the survivors, and the position after the row's last accepted draw starts the next batch.
// SHA-256 and the ChaCha block function are pinned by known-answer tests of the oracle; the generator's layout and
// the sampling rule restate rand_chacha 0.10 / rand 0.10, which are not vendored: PARITY UNPINNED (like psi).
__device__ __forceinline__ uint32_t rotr32(uint32_t v, int c) { return (v >> c) | (v << (32 - c)); }
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
// SHA-256 of exactly 32 bytes (one padded block); digest as 8 big-endian words
__device__ inline void sha256_32(const uint8_t *msg, uint32_t h[8]) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 8; i++)
        w[i] = ((uint32_t)msg[4 * i] << 24) | ((uint32_t)msg[4 * i + 1] << 16) | ((uint32_t)msg[4 * i + 2] << 8) | msg[4 * i + 3];
    w[8] = 0x80000000u;
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 256;   // message length in bits
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c, g = 0x1f83d9ab,
             hh = 0x5be0cd19;
    const uint32_t init[8] = {a, b, c, d, e, f, g, hh};
    for (int i = 0; i < 64; i++) {
        const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
        const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    const uint32_t fin[8] = {a, b, c, d, e, f, g, hh};
    for (int i = 0; i < 8; i++) h[i] = init[i] + fin[i];
}
// One ChaCha8 block: key words (little-endian), 64-bit block counter, stream id 0.
__device__ __forceinline__ void chacha8_block(const uint32_t key[8], u64 counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define FHE_CHACHA_QR(a, b, c, d)                  \
    x[a] += x[b], x[d] = rotl32(x[d] ^ x[a], 16);  \
    x[c] += x[d], x[b] = rotl32(x[b] ^ x[c], 12);  \
    x[a] += x[b], x[d] = rotl32(x[d] ^ x[a], 8);   \
    x[c] += x[d], x[b] = rotl32(x[b] ^ x[c], 7);
#pragma unroll
    for (int r = 0; r < 4; r++) {   // 8 rounds = 4 double rounds
        FHE_CHACHA_QR(0, 4, 8, 12) FHE_CHACHA_QR(1, 5, 9, 13) FHE_CHACHA_QR(2, 6, 10, 14) FHE_CHACHA_QR(3, 7, 11, 15)
        FHE_CHACHA_QR(0, 5, 10, 15) FHE_CHACHA_QR(1, 6, 11, 12) FHE_CHACHA_QR(2, 7, 8, 13) FHE_CHACHA_QR(3, 4, 9, 14)
    }
#undef FHE_CHACHA_QR
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
// grid.x = polynomials; 256 threads; seeds [npolys][32] bytes -> out [npolys][nmod][N].
constexpr int SEED_THREADS = 256;
constexpr size_t SEED_SMEM_BYTES = 8 + 8 * 4 + SEED_THREADS * 4;
__global__ void __launch_bounds__(SEED_THREADS)
    seed_expand_kernel(const uint8_t *__restrict__ seeds, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                       uint32_t nmod, uint32_t logn) {
    FHE_DYN_SMEM(u64, sm);   // SEED_SMEM_BYTES: next position | key[8] | scan[SEED_THREADS]
    u64 &s_next_pos = sm[0];
    uint32_t *const s_key = reinterpret_cast<uint32_t *>(sm + 1);
    uint32_t *const s_scan = s_key + 8;
    const uint32_t tid = threadIdx.x, n = 1u << logn;
    if (tid == 0) {
        uint32_t h[8];
        sha256_32(seeds + (u64)blockIdx.x * 32, h);
        // the digest's bytes (big-endian words) are the seed array; ChaCha reads its key as little-endian words
        for (int i = 0; i < 8; i++) s_key[i] = __builtin_bswap32(h[i]);
    }
    __syncthreads();
    uint32_t key[8];
    for (int i = 0; i < 8; i++) key[i] = s_key[i];
    u64 *dst = out + (u64)blockIdx.x * nmod * n;
    u64 pos = 0;   // index of the next u64 of the stream (uniform)
    for (uint32_t r = 0; r < nmod; r++) {
        const u64 q = mods[r].p;
        const u64 thresh = (0 - q) % q;   // (2^64 - q) mod q
        uint32_t produced = 0;
        while (produced < n) {
            const u64 blk = (pos >> 3) + tid;
            uint32_t w[16];
            chacha8_block(key, blk, w);
            u64 val[8];
            uint32_t accept = 0;   // bit k: candidate k of this block is drawn (not before `pos`) and accepted
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u64 x = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
                const u128_t m = (u128_t)x * q;
                val[k] = (u64)(m >> 64);
                if (8 * blk + k >= pos && (u64)m >= thresh) accept |= 1u << k;
            }
            // exclusive scan of the accept counts over the workgroup (Hillis-Steele in LDS)
            const uint32_t cnt = (uint32_t)__builtin_popcount(accept);
            s_scan[tid] = cnt;
            __syncthreads();
            for (uint32_t off = 1; off < SEED_THREADS; off <<= 1) {
                const uint32_t v = tid >= off ? s_scan[tid - off] : 0;
                __syncthreads();
                s_scan[tid] += v;
                __syncthreads();
            }
            const uint32_t incl = s_scan[tid], total = s_scan[SEED_THREADS - 1];
            uint32_t rank = produced + incl - cnt;
            const uint32_t need = n - produced;   // draws this row still takes
            if (tid == 0) s_next_pos = 8 * ((pos >> 3) + SEED_THREADS);   // all candidates consumed unless the row ends here
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (accept & (1u << k)) {
                    if (rank < n) dst[(u64)r * n + rank] = val[k];
                    if (rank + 1 == n && total >= need) s_next_pos = 8 * blk + k + 1;   // the row's last draw
                    rank++;
                }
            }
            __syncthreads();
            pos = s_next_pos;
            produced = total >= need ? n : produced + total;
            __syncthreads();
        }
    }
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
