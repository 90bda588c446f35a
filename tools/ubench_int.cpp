// ubench_int.cpp -- integer-issue ceiling of the MI355X for the instruction mix of the BFV hot
// path (SURVEY.md §8d asks for it next to the HBM roofline): throughput of 32-bit multiplies
// (v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32), 64-bit adds, the Shoup modular multiply and
// the full Harvey butterfly, all register-resident (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.cpp -o tools/ubench_int
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../fhe.rs_amd/csrc/zq_dev.hpp"
using namespace fhe;

// -p / -2p made opaque so that the compiler keeps the add form (the engine loads them from DevMod)
__device__ __forceinline__ u64 opaque_neg(u64 v) {
    u64 r = 0 - v;
    asm("" : "+s"(r));
    return r;
}
// Hand-written lazy Shoup multiplication a*w - floor(a*ws/2^64)*p (mod 2^64), as a*w + q*np.
// v_mad_u64_u32 wants even-aligned 64-bit addends, so the compiler spends v_movs on zero-extending partial
// words; here a scratch pair keeps its high word at zero / at the carry instead.  Scratch: v[120:125].
__device__ __forceinline__ u64 shoup_asm(u64 a, u64 w, u64 ws, u64 np) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32), n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    u64 t;
    uint32_t h;
    asm volatile(
        "v_mul_hi_u32 v120, %[a0], %[s0]\n\t"
        "v_mov_b32 v121, 0\n\t"
        "v_mad_u64_u32 v[122:123], vcc, %[a0], %[s1], v[120:121]\n\t"   // m = a0*s1 + hi(a0*s0)
        "v_mad_u64_u32 v[122:123], vcc, %[a1], %[s0], v[122:123]\n\t"   // m += a1*s0, carry -> vcc
        "v_mov_b32 v120, v123\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 v121, 0, 1, vcc\n\t"                              // {m.hi, carry}
        "v_mad_u64_u32 v[120:121], vcc, %[a1], %[s1], v[120:121]\n\t"   // q = a1*s1 + (m >> 32)
        "v_mad_u64_u32 %[t], vcc, %[a0], %[w0], 0\n\t"                  // a*w low 64 ...
        "v_mul_lo_u32 v122, %[a0], %[w1]\n\t"
        "v_mul_lo_u32 v123, %[a1], %[w0]\n\t"
        "v_mad_u64_u32 %[t], vcc, v120, %[n0], %[t]\n\t"                // ... + q*np low 64
        "v_mul_lo_u32 v124, v120, %[n1]\n\t"
        "v_mul_lo_u32 v125, v121, %[n0]\n\t"
        "v_add3_u32 v122, v122, v123, v124\n\t"
        "v_add_u32 %[h], v122, v125"
        : [t] "=&v"(t), [h] "=&v"(h)
        : [a0] "v"(a0), [a1] "v"(a1), [w0] "v"(w0), [w1] "v"(w1), [s0] "v"(s0), [s1] "v"(s1), [n0] "v"(n0), [n1] "v"(n1)
        : "vcc", "v120", "v121", "v122", "v123", "v124", "v125");
    return t + ((u64)h << 32);
}
// All-mad Shoup: v_mad_u64_u32 (a 32x32 multiply with a 64-bit addend) issues faster than v_mul_lo_u32 and much
// faster than v_mul_hi_u32 (this file's first three lines of output), but the compiler narrows every product whose
// high half is not demanded to v_mul_lo_u32.  Here the two low products a*w + q*np (mod 2^64) are one chain of six
// mads: the four cross terms accumulate in a 64-bit register that is made opaque (so that all 64 bits count as
// demanded), then enter the sum through one v_lshl_add_u64.
__device__ __forceinline__ u64 opaque64(u64 v) {
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ u64 shoup_allmad(u64 a, u64 w, u64 ws, u64 np) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    const u64 q = mulhi64(a, ws);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
    u64 cross = (u64)a0 * w1;
    cross += (u64)a1 * w0;
    cross += (u64)q0 * n1;
    cross += (u64)q1 * n0;
    cross = opaque64(cross);
    u64 lo = (u64)a0 * w0;
    lo += (u64)q0 * n0;
    return lo + (cross << 32);
}
// the same with the 64x64 -> hi64 product written as mads only (the compiler takes v_mul_hi_u32 for a0*s0)
__device__ __forceinline__ u64 mulhi64_allmad(u64 a, u64 s) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
    const u64 l = opaque64((u64)a0 * s0);
    const u64 m = opaque64((u64)a0 * s1 + (l >> 32));
    const u64 m2 = opaque64((u64)a1 * s0 + (uint32_t)m);
    return (u64)a1 * s1 + (m >> 32) + (m2 >> 32);
}
__device__ __forceinline__ u64 shoup_allmad2(u64 a, u64 w, u64 ws, u64 np) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    const u64 q = mulhi64_allmad(a, ws);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
    u64 cross = (u64)a0 * w1;
    cross += (u64)a1 * w0;
    cross += (u64)q0 * n1;
    cross += (u64)q1 * n0;
    cross = opaque64(cross);
    u64 lo = (u64)a0 * w0;
    lo += (u64)q0 * n0;
    return lo + (cross << 32);
}
// quotient from three partial products (a0*s0 dropped): q' in {q - 1, q}, result below 3p
__device__ __forceinline__ u64 shoup_approx(u64 a, u64 w, u64 ws, u64 np) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
    const uint32_t n0 = (uint32_t)np, n1 = (uint32_t)(np >> 32);
    const u64 m = opaque64((u64)a0 * s1);
    const u64 m2 = opaque64((u64)a1 * s0 + (uint32_t)m);
    const u64 q = (u64)a1 * s1 + (m >> 32) + (m2 >> 32);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32);
    u64 cross = (u64)a0 * w1;
    cross += (u64)a1 * w0;
    cross += (u64)q0 * n1;
    cross += (u64)q1 * n0;
    cross = opaque64(cross);
    u64 lo = (u64)a0 * w0;
    lo += (u64)q0 * n0;
    return lo + (cross << 32);
}
// 64x64 -> hi64 as four mads without any per-use asm: the addend of the first product is a zero the compiler
// cannot see through (one s_mov at kernel start), so it cannot turn a0*s0 >> 32 into the slow v_mul_hi_u32.
__device__ __forceinline__ u64 mulhi64_z(u64 a, u64 s, u64 z) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
    const u64 l = (u64)a0 * s0 + z;
    const u64 m = (u64)a0 * s1 + (l >> 32);
    const u64 m2 = (u64)a1 * s0 + (uint32_t)m;
    return (u64)a1 * s1 + (m >> 32) + (m2 >> 32);
}
__device__ __forceinline__ u64 mulhi64_approx(u64 a, u64 s) {   // a0*s0 dropped: result in {hi - 1, hi}
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
    const u64 m = (u64)a0 * s1;
    const u64 m2 = (u64)a1 * s0 + (uint32_t)m;
    return (u64)a1 * s1 + (m >> 32) + (m2 >> 32);
}
// y * w mod p for p = 2^62 - c, c < 2^28 -- what the reference's generate_prime(62, ...) hands out for the extended
// basis (zq/primes.rs:27-48 walks down from 2^62 in steps of 2N) -- any y < 2^64, w < 2^62; result below 2p (lazy).
// 2^62 = c (mod p), so the 126-bit product folds twice: x = xh 2^62 + xl -> xh c + xl (< 2^92) -> sh c + sl.
// Seven multiply-adds, no quotient and no Shoup companion of w against the Harvey butterfly's ten / eleven multiplies.
// Measured (profiles/r02_ubench_butterflies_carry.jsonl): 2.71 T/s as a bare modular product against 2.51 T/s for the
// Shoup product, but 1.75 / 1.68 T/s inside the forward / inverse butterfly against 1.79 / 1.72 T/s -- the fourteen
// shifts, masks and moves around the folds cost what the three multiplies save.  Not built into the kernels.
__device__ __forceinline__ u64 solinas62_lazy(u64 y, u64 w, uint32_t c) {
#if defined(FHE_HOST_EMULATION)
    if ((w >> 62) || (c >> 28)) __builtin_trap();
#endif
#if FHE_HAVE_MAD_CARRY
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const u64 l = (u64)y0 * w0;
    const u64 m = (u64)y0 * w1 + (l >> 32);          // < 2^62 + 2^32
    u64 cm;
    const u64 n = mad64_carry<false>(y1, w0, m, cm);  // may carry: y1 * w0 < 2^64
    const u64 hi = (u64)y1 * w1 + ((u64)(uint32_t)(n >> 32) | ((u64)carry_bit(cm) << 32));   // < 2^62
    const u64 lo = (u64)(uint32_t)l | (n << 32);
#else
    const u128_t x = (u128_t)y * w;
    const u64 hi = (u64)(x >> 64), lo = (u64)x;
#endif
    const u64 xh = (hi << 2) | (lo >> 62), xl = lo & ((1ull << 62) - 1);
    const u64 a = (u64)(uint32_t)xh * c + xl;              // < 2^60 + 2^62
    const u64 b = (u64)(uint32_t)(xh >> 32) * c + (a >> 32);   // xh c + xl = b 2^32 + (a mod 2^32), b < 2^60 + 2^31
    const u64 sl = (u64)(uint32_t)a | ((b & 0x3FFFFFFFull) << 32);
    return (u64)(uint32_t)(b >> 30) * c + sl;              // (b >> 30) < 2^31: the product is below 2^59
}

constexpr int ILP = 8;
constexpr int ITERS = 4096;

template <int KIND>
__global__ void bench(u64 *out, u64 seed, u64 p, u64 w, u64 ws) {
    u64 x[ILP], y[ILP];
    uint32_t a[ILP], b[ILP];
    for (int i = 0; i < ILP; i++) {
        x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x;
        y[i] = x[i] * 0x9E3779B97F4A7C15ull;
        a[i] = (uint32_t)x[i];
        b[i] = (uint32_t)y[i] | 1;
    }
    const u64 p2 = 2 * p;
    const PM pm{p, p2, opaque_neg(p), opaque_neg(p2)};
    u64 zero = 0;
    asm("" : "+s"(zero));
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (KIND == 0) a[i] = a[i] * b[i] + 1;                                   // v_mul_lo_u32 (+add)
            if (KIND == 1) a[i] = __umulhi(a[i], b[i]) + b[i];                        // v_mul_hi_u32 (+add)
            if (KIND == 2) x[i] = (u64)(uint32_t)x[i] * (uint32_t)y[i] + y[i];         // v_mad_u64_u32
            if (KIND == 3) x[i] = x[i] + y[i] + (u64)it;                               // 64-bit adds
            if (KIND == 4) x[i] = mulhi64(x[i], y[i]) + 1;                             // 64x64 -> hi64
            if (KIND == 5) x[i] = mul_shoup_lazy(x[i], w, ws, p);                      // Shoup modmul
            if (KIND == 6) fwd_butterfly(x[i], y[i], w, ws, pm);                       // Harvey CT butterfly
            if (KIND == 7) inv_butterfly(x[i], y[i], w, ws, pm);                       // Harvey GS butterfly
            if (KIND == 9) {  // CT butterfly without the conditional subtraction (headroom of <= 60-bit moduli)
                const u64 t = mul_shoup_lazy_n(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 12) x[i] = shoup_asm(x[i], w, ws, pm.np);
            if (KIND == 13) {  // narrow CT butterfly on the asm Shoup
                const u64 t = shoup_asm(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 20) x[i] = shoup_allmad(x[i], w, ws, pm.np);
            if (KIND == 21) {
                const u64 t = shoup_allmad(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 22) x[i] = shoup_allmad2(x[i], w, ws, pm.np);
            if (KIND == 23) {
                const u64 t = shoup_allmad2(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 24) x[i] = shoup_approx(x[i], w, ws, pm.np);
            if (KIND == 25) {
                const u64 t = shoup_approx(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 + pm.p - t;
                x[i] = x[i] + t;
            }
            if (KIND == 26) {  // full (wide-modulus) butterfly on the all-mad Shoup
                x[i] = csub_n(x[i], pm.p2, pm.np2);
                const u64 t = shoup_allmad(y[i], w, ws, pm.np);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 27) {  // Gentleman-Sande butterfly on the all-mad Shoup
                const u64 t = x[i];
                x[i] = csub_n(y[i] + t, pm.p2, pm.np2);
                y[i] = shoup_allmad(pm.p2 + t - y[i], w, ws, pm.np);
            }
            if (KIND == 40) fwd_butterfly_narrow<false>(x[i], y[i], w, ws, pm, false);   // round-2 arithmetic (carry-out high products)
            if (KIND == 41) fwd_butterfly<true>(x[i], y[i], w, ws, pm);                  // ... scalar twiddle operands
            if (KIND == 42) inv_butterfly<true>(x[i], y[i], w, ws, pm);
            if (KIND == 43) fwd_butterfly_narrow<true>(x[i], y[i], w, ws, pm, false);
            if (KIND == 50) x[i] = solinas62_lazy(x[i], w, (uint32_t)ws | 1);   // 2^62 - c primes: seven multiply-adds
            if (KIND == 51) {   // wide forward butterfly on it
                x[i] = csub_n(x[i], pm.p2, pm.np2);
                const u64 t = solinas62_lazy(y[i], w, (uint32_t)ws | 1);
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 52) {   // Gentleman-Sande butterfly on it
                const u64 t = x[i];
                x[i] = csub_n(y[i] + t, pm.p2, pm.np2);
                y[i] = solinas62_lazy(pm.p2 + t - y[i], w, (uint32_t)ws | 1);
            }
            if (KIND == 28) x[i] = x[i] * w + mulhi64_z(x[i], ws, zero) * pm.np;
            if (KIND == 29) {
                const u64 t = y[i] * w + mulhi64_z(y[i], ws, zero) * pm.np;
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 30) {
                x[i] = csub_n(x[i], pm.p2, pm.np2);
                const u64 t = y[i] * w + mulhi64_z(y[i], ws, zero) * pm.np;
                y[i] = x[i] + pm.p2 - t;
                x[i] = x[i] + t;
            }
            if (KIND == 31) {
                const u64 t = x[i];
                x[i] = csub_n(y[i] + t, pm.p2, pm.np2);
                const u64 d = pm.p2 + t - y[i];
                y[i] = d * w + mulhi64_z(d, ws, zero) * pm.np;
            }
            if (KIND == 32) x[i] = x[i] * w + mulhi64_approx(x[i], ws) * pm.np;
            if (KIND == 33) {
                const u64 t = y[i] * w + mulhi64_approx(y[i], ws) * pm.np;
                y[i] = x[i] + pm.p2 + pm.p - t;
                x[i] = x[i] + t;
            }
            if (KIND == 34) x[i] = mulhi64_z(x[i], y[i], zero) + 1;
            if (KIND == 8) x[i] = x[i] * y[i] + 1;                                     // 64x64 -> lo64
        }
    }
    u64 acc = 0;
    for (int i = 0; i < ILP; i++) acc += x[i] + y[i] + a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
double run(const char *name, double ops_per_iter) {
    const int blocks = 256 * 8, threads = 256;
    u64 *out;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    const u64 p = 1152921504606830593ull, w = 123456789012345ull;
    const u64 ws = (u64)(((unsigned __int128)w << 64) / p);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1ull, p, w, ws);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 2ull + r, p, w, ws);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * threads * ILP * ITERS * reps * ops_per_iter;
    const double gops = total / (ms * 1e-3) / 1e9;
    printf("{\"kernel\": \"%s\", \"gops\": %.1f, \"ms\": %.3f}\n", name, gops, ms / reps);
    hipFree(out);
    return gops;
}

int main() {
    if (getenv("UB_R02B")) {   // the butterflies as the kernels run them since the carry-out high products
        run<4>("mulhi64 (carry-out form)", 1);
        run<5>("mul_shoup_lazy", 1);
        run<6>("fwd_butterfly (wide)", 1);
        run<41>("fwd_butterfly (wide), scalar twiddle", 1);
        run<7>("inv_butterfly", 1);
        run<42>("inv_butterfly, scalar twiddle", 1);
        run<40>("fwd_butterfly_narrow (approximate quotient)", 1);
        run<43>("fwd_butterfly_narrow, scalar twiddle", 1);
        run<50>("solinas62_lazy (p = 2^62 - c)", 1);
        run<51>("fwd_butterfly (wide) on solinas62_lazy", 1);
        run<52>("inv_butterfly on solinas62_lazy", 1);
        return 0;
    }
    run<0>("v_mul_lo_u32", 1);
    run<1>("v_mul_hi_u32", 1);
    run<2>("v_mad_u64_u32", 1);
    run<3>("add_u64 x2", 2);
    run<4>("mulhi64 (4 mad_u64_u32)", 1);
    run<8>("mullo64", 1);
    run<9>("fwd_butterfly without conditional subtraction (not used: needs moduli <= 60 bits)", 1);
    run<5>("mul_shoup_lazy", 1);
    run<12>("mul_shoup_lazy, hand-written asm", 1);
    run<13>("fwd_butterfly without conditional subtraction, asm Shoup", 1);
    run<20>("mul_shoup_lazy, all-mad low products", 1);
    run<21>("narrow fwd butterfly, all-mad low products", 1);
    run<22>("mul_shoup_lazy, all-mad low products + all-mad mulhi", 1);
    run<23>("narrow fwd butterfly, all-mad low + mulhi", 1);
    run<24>("mul_shoup_lazy, approximate quotient (3 partial products), all-mad", 1);
    run<25>("narrow fwd butterfly, approximate quotient", 1);
    run<26>("fwd_butterfly (wide), all-mad low products", 1);
    run<27>("inv_butterfly, all-mad low products", 1);
    run<34>("mulhi64, four mads (opaque zero addend)", 1);
    run<28>("mul_shoup_lazy, mulhi by four mads", 1);
    run<29>("narrow fwd butterfly, mulhi by four mads", 1);
    run<30>("fwd_butterfly (wide), mulhi by four mads", 1);
    run<31>("inv_butterfly, mulhi by four mads", 1);
    run<32>("mul_shoup_lazy, approximate quotient, compiler's low products", 1);
    run<33>("narrow fwd butterfly, approximate quotient, compiler's low products", 1);
    run<6>("fwd_butterfly", 1);
    run<7>("inv_butterfly", 1);
    return 0;
}
