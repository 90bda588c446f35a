// ubench.hpp -- register-resident integer-issue microbenchmarks behind fhe_ubench_int (include/fhe_hip.h).
//
// SURVEY.md section 8(d) asks for TWO ceilings: the HBM stage model and the integer-issue rate of the instruction
// mix.  These loops run the very butterflies the NTT-type kernels are built from (zq_dev.hpp, same template
// arguments: twiddles in scalar registers) with no memory traffic at all, chip-wide, so bench.py can state "kernel X
// runs at Y % of the same-box register-resident butterfly rate" from numbers measured in the same process.
// Nothing here is on the product path; tools/ubench_int.cpp holds the wider catalogue of round 1-2 experiments.
#pragma once
#include "engine.hpp"

namespace fhe {
namespace ub {

enum Kind : int {
    MAD_U64_U32 = 0,   // v_mad_u64_u32: the multiplier instruction everything below is made of
    MUL_LO_U32 = 1,    // v_mul_lo_u32
    MUL_HI_U32 = 2,    // v_mul_hi_u32
    SHOUP_LAZY = 3,    // one lazy Shoup modular product (mul_shoup_lazy_n)
    FWD_WIDE = 4,      // Harvey forward butterfly, any modulus < 2^62 (fwd_butterfly)
    FWD_NARROW = 5,    // forward butterfly for moduli < 2^60, approximate quotient (fwd_butterfly_narrow)
    INV_WIDE = 6,      // Gentleman-Sande inverse butterfly (inv_butterfly)
    NKINDS = 7
};
constexpr int ILP = 8, ITERS = 2048;

__device__ __forceinline__ u64 opaque_s(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(v));
#endif
    return v;
}

template <int KIND>
__global__ void __launch_bounds__(256) ubench_kernel(u64 *out, u64 seed, u64 p, u64 w, u64 ws) {
    u64 x[ILP], y[ILP];
    uint32_t a[ILP], b[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
        x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x;
        y[i] = x[i] * 0x9E3779B97F4A7C15ull;
        a[i] = (uint32_t)x[i];
        b[i] = (uint32_t)y[i] | 1;
    }
    // the modulus words stay opaque scalars, as the kernels load them from DevMod (so x + (2^64 - p) is not folded
    // back into a subtraction)
    const PM pm{opaque_s(p), opaque_s(2 * p), opaque_s(0 - p), opaque_s(0 - 2 * p)};
    w = opaque_s(w), ws = opaque_s(ws);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (KIND == MAD_U64_U32) x[i] = (u64)(uint32_t)x[i] * (uint32_t)y[i] + y[i];
            if (KIND == MUL_LO_U32) a[i] = a[i] * b[i] + 1;
            if (KIND == MUL_HI_U32) a[i] = (uint32_t)(((u64)a[i] * b[i]) >> 32) + b[i];
            if (KIND == SHOUP_LAZY) x[i] = mul_shoup_lazy_n<true>(x[i], w, ws, pm.np);
            if (KIND == FWD_WIDE) fwd_butterfly<true>(x[i], y[i], w, ws, pm);
            // (values wrap around 2^64 here: the loop measures the instruction stream, not residues; the emulated
            // build's range trap is why its narrow case is spelled out)
#if defined(FHE_HOST_EMULATION)
            if (KIND == FWD_NARROW) {
                const u64 t = y[i] * w + mulhi64_approx<true>(y[i], ws) * pm.np;
                const u64 pk = pm.p2 + pm.p;
                y[i] = x[i] + pk - t;
                x[i] = x[i] + t;
            }
#else
            if (KIND == FWD_NARROW) fwd_butterfly_narrow<true>(x[i], y[i], w, ws, pm, false);
#endif
            if (KIND == INV_WIDE) inv_butterfly<true>(x[i], y[i], w, ws, pm);
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += x[i] + y[i] + a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// Runs kernel `kind` for at least `min_seconds` on `device` (one warm-up launch first); returns operations per second
// chip-wide (one operation = one multiply / one modular product / one butterfly per lane).
inline double run(int device, int kind, double min_seconds) {
    require(kind >= 0 && kind < NKINDS, E_ARG, "unknown microbenchmark");
    require(min_seconds > 0 && min_seconds <= 10, E_ARG, "min_seconds must be in (0, 10]");
    FHE_HIP_CHECK(hipSetDevice(device));
    const unsigned blocks = (unsigned)device_cus(device) * 8, threads = 256;
    DevBuf<u64> out;
    out.alloc((size_t)blocks * threads);
    const u64 p = 1152921504606830593ull, w = 123456789012345ull, ws = shoup(w, p);
    hipStream_t s;
    FHE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto launch = [&](u64 seed) {
        switch (kind) {
#define FHE_UB_CASE(K) \
    case K: hipLaunchKernelGGL(ubench_kernel<K>, dim3(blocks), dim3(threads), 0, s, out.p, seed, p, w, ws); break;
            FHE_UB_CASE(0) FHE_UB_CASE(1) FHE_UB_CASE(2) FHE_UB_CASE(3) FHE_UB_CASE(4) FHE_UB_CASE(5) FHE_UB_CASE(6)
#undef FHE_UB_CASE
        }
    };
    double result = 0;
    try {
        FHE_HIP_CHECK(hipEventCreate(&e0));
        FHE_HIP_CHECK(hipEventCreate(&e1));
        launch(1);
        FHE_HIP_CHECK(hipStreamSynchronize(s));
        double total_ms = 0;
        u64 launches = 0;
        unsigned reps = 4;
        while (total_ms < min_seconds * 1e3) {
            FHE_HIP_CHECK(hipEventRecord(e0, s));
            for (unsigned r = 0; r < reps; r++) launch(2 + launches + r);
            FHE_HIP_CHECK(hipEventRecord(e1, s));
            FHE_HIP_CHECK(hipEventSynchronize(e1));
            FHE_HIP_CHECK(hipGetLastError());
            float ms = 0;
            FHE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            launches += reps;
        }
        result = (double)blocks * threads * ILP * ITERS * (double)launches / (total_ms * 1e-3);
    } catch (...) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
        throw;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    return result;
}

}  // namespace ub
}  // namespace fhe
