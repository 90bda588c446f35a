// ubench_int.cpp -- integer-issue ceiling of the MI355X for the instruction mix of the BFV hot
// path (SURVEY.md §8d asks for it next to the HBM roofline): throughput of 32-bit multiplies
// (v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32), 64-bit adds, the Shoup modular multiply and
// the full Harvey butterfly, all register-resident (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.cpp -o tools/ubench_int
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../fhe.rs_amd/csrc/zq_dev.hpp"
using namespace fhe;

// -p / -2p made opaque so that the compiler keeps the add form (the engine loads them from DevMod)
__device__ __forceinline__ u64 opaque_neg(u64 v) {
    u64 r = 0 - v;
    asm("" : "+s"(r));
    return r;
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
    run<0>("v_mul_lo_u32", 1);
    run<1>("v_mul_hi_u32", 1);
    run<2>("v_mad_u64_u32", 1);
    run<3>("add_u64 x2", 2);
    run<4>("mulhi64 (4 mad_u64_u32)", 1);
    run<8>("mullo64", 1);
    run<9>("fwd_butterfly without conditional subtraction (not used: needs moduli <= 60 bits)", 1);
    run<5>("mul_shoup_lazy", 1);
    run<6>("fwd_butterfly", 1);
    run<7>("inv_butterfly", 1);
    return 0;
}
