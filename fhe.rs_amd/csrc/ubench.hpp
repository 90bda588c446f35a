// ubench.hpp -- register-resident integer-issue microbenchmarks behind fhe_ubench_int (include/fhe_hip.h).
//
// SURVEY.md section 8(d) asks for TWO ceilings: the HBM stage model and the integer-issue rate of the instruction
// mix.  These loops run the very butterflies the NTT-type kernels are built from (zq_dev.hpp, same template
// arguments: twiddles in scalar registers) with no memory traffic at all, chip-wide, so bench.py can state "kernel X
// runs at Y % of the same-box register-resident butterfly rate" from numbers measured in the same process.
// Nothing here is on the product path; tools/ubench_int.cpp holds the wider catalogue of round 1-2 experiments.
#pragma once
#include "engine.hpp"
#include "zq_f64.hpp"

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
    // round 5: the rest of the multiply pipeline's arithmetic, so that the integer ceiling covers the WHOLE operation
    SHOUP_MAC = 7,     // the key switch's accumulate: acc = csub(acc + v * k (Shoup, lazy), 2p)   (kernels_ks.hpp)
    TENSOR_MUL = 8,    // tensor slots 0 / 2: one product of two residues + single-word Barrett, lazy  (mul_mod_lazy)
    TENSOR_MAC2 = 9,   // tensor slot 1: a0 b1 + a1 b0 as one 128-bit sum + one Barrett  (mac2_wide62 + barrett_reduce_wide_lazy)
    // round 6: the FP64-FMA alternative for moduli below 2^50 (zq_f64.hpp), priced before it is built
    F64_FMA = 10,      // v_fma_f64, a dependent chain per lane
    F64_RNDNE = 11,    // v_rndne_f64 (+ one v_add_f64 that keeps the chain alive)
    F64_MULMOD = 12,   // mulmod_f64: one exact lazy modular product, precomputed w / p (scalar twiddles)
    F64_FWD = 13,      // fwd_butterfly_f64 + the amortised reduction (both outputs reduced every fourth stage)
    F64_INV = 14,      // inv_butterfly_f64 + the sum reduced every second stage
    F64_MAC = 15,      // the key switch's accumulate on doubles: acc + v k with the quotient from h (1/p) (no k / p twin), reduced every eighth term
    NKINDS = 16
};
constexpr int ILP = 8, ITERS = 2048;

__device__ __forceinline__ u64 opaque_s(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(v));
#endif
    return v;
}

template <int KIND>
__global__ void __launch_bounds__(256) ubench_kernel(u64 *out, u64 seed, u64 p, u64 w, u64 ws, DevMod md) {
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
            if (KIND == SHOUP_MAC) x[i] = csub_n(mul_shoup_lazy_add_n<true>(x[i], y[i], w, ws, pm.np), pm.p2, pm.np2);
            // (operands masked below 2^60 as the kernels' residues are -- the 4-multiply product needs that -- and made
            // to depend on the previous result so that the loop is a chain per lane, like the butterflies above)
            if (KIND == TENSOR_MUL) x[i] = mul_mod_lazy(x[i] & 0x0FFFFFFFFFFFFFFFull, y[i] & 0x0FFFFFFFFFFFFFFFull, md) + y[i];
            if (KIND == TENSOR_MAC2) {
                u64 hi, lo;
                const u64 xa = x[i] & 0x0FFFFFFFFFFFFFFFull, ya = y[i] & 0x0FFFFFFFFFFFFFFFull;
                mac2_wide62(xa, ya, ya ^ 0x5555, xa ^ 0x3333, hi, lo);
                x[i] = barrett_reduce_wide_lazy(hi & 0x00FFFFFFFFFFFFFFull, lo, md) + y[i];
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += x[i] + y[i] + a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The same loop on doubles holding integers (zq_f64.hpp).  The modulus is a 49-bit prime of the reference's stock
// n = 16384 set (parameters.rs:243-251); values stay exact residues here (the reductions are part of the stream).
template <int KIND>
__global__ void __launch_bounds__(256) ubench_f64_kernel(u64 *out, u64 seed, double p, double ip, double w, double wp) {
    double x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) {
        x[i] = (double)((seed + threadIdx.x * 977 + i * 131 + blockIdx.x) & 0xFFFFFFFFFFFFull);
        y[i] = (double)(((seed + threadIdx.x * 977 + i * 131 + blockIdx.x) * 0x9E3779B97F4A7C15ull) >> 16);
    }
    // (wave-uniform constants kept opaque, as the kernels load them: no folding into immediates)
    union {
        double d;
        u64 u;
    } cp{p}, cip{ip}, cw{w}, cwp{wp};
    cp.u = opaque_s(cp.u), cip.u = opaque_s(cip.u), cw.u = opaque_s(cw.u), cwp.u = opaque_s(cwp.u);
    const PF m{cp.d, cip.d};
    w = cw.d, wp = cwp.d;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (KIND == F64_FMA) x[i] = f64_fma(x[i], w, y[i]);
            if (KIND == F64_RNDNE) x[i] = f64_rint(x[i]) + y[i];
            if (KIND == F64_MULMOD) x[i] = mulmod_f64(x[i], w, wp, m.p);
            if (KIND == F64_FWD) {
                fwd_butterfly_wp_f64(x[i], y[i], w, wp, m);
                if ((it & 3) == 3) x[i] = reduce_f64(x[i], m), y[i] = reduce_f64(y[i], m);
            }
            if (KIND == F64_INV) {
                inv_butterfly_f64(x[i], y[i], w, wp, m);
                if (it & 1) x[i] = reduce_f64(x[i], m);
            }
            if (KIND == F64_MAC) {
                // (the multiplicand changes every iteration -- y + 1, one extra add, counted in -- because with a loop-invariant y
                // the compiler hoists the product and the loop times one add: round 6's first gate figure for this kind,
                // 6.2 T/s, was that.  The products are independent of the accumulator chain, as the kernel's are.)
                y[i] = y[i] + 1.0;
                x[i] = mulmod2_add_f64(x[i], y[i], w, m);
                if ((it & 7) == 7) x[i] = reduce_f64(x[i], m);
            }
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += x[i] + y[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (u64)(long long)acc;
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
    const double pf = 562949951979521.0 /* 0x1fffffff68001 */, wf = 123456789012345.0;
    const ModConsts mc = make_mod_consts(p);
    DevMod md;
    static_assert(sizeof(DevMod) == sizeof(ModConsts), "DevMod layout");
    std::memcpy(&md, &mc, sizeof(md));
    hipStream_t s;
    FHE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto launch = [&](u64 seed) {
        switch (kind) {
#define FHE_UB_CASE(K) \
    case K: hipLaunchKernelGGL(ubench_kernel<K>, dim3(blocks), dim3(threads), 0, s, out.p, seed, p, w, ws, md); break;
            FHE_UB_CASE(0) FHE_UB_CASE(1) FHE_UB_CASE(2) FHE_UB_CASE(3) FHE_UB_CASE(4) FHE_UB_CASE(5) FHE_UB_CASE(6)
            FHE_UB_CASE(7) FHE_UB_CASE(8) FHE_UB_CASE(9)
#undef FHE_UB_CASE
#define FHE_UB_CASE(K) \
    case K: hipLaunchKernelGGL(ubench_f64_kernel<K>, dim3(blocks), dim3(threads), 0, s, out.p, seed, pf, 1.0 / pf, wf, wf / pf); break;
            FHE_UB_CASE(10) FHE_UB_CASE(11) FHE_UB_CASE(12) FHE_UB_CASE(13) FHE_UB_CASE(14) FHE_UB_CASE(15)
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

// A plain streaming copy, 16 bytes per lane, non-temporal loads and stores (what the path's element-wise kernels are made
// of): the box's own streaming rate, read + write bytes per second.  (torch's D2D copy -- hipMemcpyDtoD -- measures
// ~5.1 TB/s on these boxes and is NOT a ceiling: the path's own streaming kernels beat it.)
__global__ void __launch_bounds__(256) ubench_copy_kernel(const k::u64x2 *__restrict__ src, k::u64x2 *__restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) k::store_stream(dst + i, k::load_stream(src + i));
}
inline double run_copy(int device, size_t bytes, double min_seconds) {
    require(min_seconds > 0 && min_seconds <= 10, E_ARG, "min_seconds must be in (0, 10]");
    require(bytes >= 4096 && bytes <= ((size_t)16 << 30), E_ARG, "bytes must be in [4 KiB, 16 GiB]");
    FHE_HIP_CHECK(hipSetDevice(device));
    const size_t n16 = bytes / 16;
    DevBuf<k::u64x2> a, b;
    a.alloc(n16);
    b.alloc(n16);
    FHE_HIP_CHECK(hipMemset(a.p, 1, n16 * 16));
    hipStream_t s;
    FHE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double result = 0;
    try {
        FHE_HIP_CHECK(hipEventCreate(&e0));
        FHE_HIP_CHECK(hipEventCreate(&e1));
        // one 16-byte element per lane (no grid-stride tail): enough workgroups in flight to cover HBM latency
        const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)1 << 30);
        hipLaunchKernelGGL(ubench_copy_kernel, dim3(blocks), dim3(256), 0, s, a.p, b.p, n16);
        FHE_HIP_CHECK(hipStreamSynchronize(s));
        double total_ms = 0;
        u64 launches = 0;
        while (total_ms < min_seconds * 1e3) {
            FHE_HIP_CHECK(hipEventRecord(e0, s));
            for (int r = 0; r < 4; r++) hipLaunchKernelGGL(ubench_copy_kernel, dim3(blocks), dim3(256), 0, s, a.p, b.p, n16);
            FHE_HIP_CHECK(hipEventRecord(e1, s));
            FHE_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            FHE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            launches += 4;
        }
        result = 2.0 * (double)n16 * 16.0 * (double)launches / (total_ms * 1e-3);
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

// The scaler's ceiling: scale_kernel itself -- the very instance that serves `sc` -- over `columns` coefficient
// columns whose polynomial strides are ZERO, i.e. every lane group reads the same nfrom rows (N * nfrom * 8 bytes: L2
// resident) and writes the same rows: the instruction stream of RnsScaler::scale with no HBM traffic.  Returns columns
// per second chip-wide.  (A cache-resident, not a register-resident ceiling: the loads and stores still issue -- and,
// the output strides being zero too, every workgroup stores to the same lines: the rate includes those same-line store
// conflicts, so it is a slightly LOW ceiling.)
inline double run_scaler(const Scaler &sc, double min_seconds) {
    require(min_seconds > 0 && min_seconds <= 10, E_ARG, "min_seconds must be in (0, 10]");
    const Ctx &f = *sc.from, &t = *sc.to;
    f.need_device();
    FHE_HIP_CHECK(hipSetDevice(f.device));
    const size_t npolys = std::max<size_t>(1, ((size_t)1 << 23) / f.n);     // 8 M columns per launch
    DevBuf<u64> in, out;
    in.alloc(f.L * f.n);
    out.alloc(t.L * t.n);
    FHE_HIP_CHECK(hipMemset(in.p, 0x11, f.L * f.n * sizeof(u64)));           // (any bit pattern: residues below 2^61)
    hipStream_t s;
    FHE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double result = 0;
    const Profiler::Suppress no_profile;   // (this thread's launches only; other threads keep profiling)
    try {
        FHE_HIP_CHECK(hipEventCreate(&e0));
        FHE_HIP_CHECK(hipEventCreate(&e1));
        launch_scale(sc, in.p, 0, out.p, 0, npolys, s);
        FHE_HIP_CHECK(hipStreamSynchronize(s));
        double total_ms = 0;
        u64 launches = 0;
        while (total_ms < min_seconds * 1e3) {
            FHE_HIP_CHECK(hipEventRecord(e0, s));
            for (int r = 0; r < 4; r++) launch_scale(sc, in.p, 0, out.p, 0, npolys, s);
            FHE_HIP_CHECK(hipEventRecord(e1, s));
            FHE_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            FHE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            launches += 4;
        }
        result = (double)npolys * f.n * (double)launches / (total_ms * 1e-3);
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
