// ubench_issue.cpp -- LAB TOOL: issue cost of single gfx950 VALU instructions and of two-instruction mixes, each pinned
// by inline asm (the compiler cannot merge, narrow or reorder them), register-resident, 8 independent chains per lane.
// Answers what a "VALU instruction" costs by kind -- the NTT-type kernels' instruction streams are priced with it
// (DESIGN.md section 6, round 3).  Output: one JSON line per stream: lane-operations per second chip-wide and the
// cycles one wave64 instruction occupies a SIMD (= 64 * SIMDs * clock / rate).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.cpp -o tools/_variants/ubench_issue ; run: ubench_issue [waves_per_simd]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2000;

// one "round" = the instruction(s) applied to each of the 8 chains; REP rounds per loop iteration
#define CHAINS8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

enum Kind {
    ADD_U32, ADD3_U32, MOV_B32, CNDMASK, LSHL_ADD_U64, CMP_LE_U64, SUB_CO_PAIR, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32,
    MAD_U64_U32_SGPR_OUT, MAD_PLUS_ADD, MAD_PLUS_LSHL_ADD, MAD_PLUS_2ADD, MUL_LO_PLUS_ADD, MAD_PLUS_CNDMASK, MAD_PLUS_MUL_LO,
    MAD_PLUS_3ADD, CNDMASK_SGPR, CMP_PLUS_CNDMASK, CSUB_CMP_SEL_ADD, CSUB_ADDC_SEL, CSUB_CMP_SGPR_SEL_ADD, CSUB_SUB_MIN32, NKINDS
};
static const char *kind_name[NKINDS] = {
    "v_add_u32", "v_add3_u32", "v_mov_b32", "v_cndmask_b32 (vcc)", "v_lshl_add_u64", "v_cmp_le_u64", "v_sub_co_u32 + v_subb_co_u32",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (carry to vcc)", "v_mad_u64_u32 (carry to an SGPR pair)",
    "v_mad_u64_u32 + v_add_u32", "v_mad_u64_u32 + v_lshl_add_u64", "v_mad_u64_u32 + 2 v_add_u32", "v_mul_lo_u32 + v_add_u32",
    "v_mad_u64_u32 + v_cndmask_b32", "v_mad_u64_u32 + v_mul_lo_u32", "v_mad_u64_u32 + 3 v_add_u32",
    "v_cndmask_b32 (SGPR-pair mask, e64)", "v_cmp_le_u64 vcc + v_cndmask_b32 (vcc)",
    "csub: v_cmp_le_u64 vcc, 2 v_cndmask (0 : nm), v_lshl_add_u64", "csub: v_add_co, v_addc_co, 2 v_cndmask",
    "csub: v_cmp_le_u64 -> SGPR pair, 2 v_cndmask e64, v_lshl_add_u64", "csub: v_lshl_add_u64 (x - m), v_cmp_gt_u64... min via 2 v_cndmask"};
static const int kind_insts[NKINDS] = {1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 2, 2, 3, 2, 2, 2, 4, 1, 2, 4, 4, 4, 4};

template <int KIND>
__global__ void __launch_bounds__(256) issue_kernel(u64 *out, u64 seed) {
    u64 x[8];
    uint32_t a[8], b[8];
    u64 sd;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x;
        a[i] = (uint32_t)(x[i] * 0x9E3779B97F4A7C15ull >> 20);
        b[i] = (uint32_t)x[i] | 1;
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(x[i]));
                if (KIND == CMP_LE_U64) asm volatile("v_cmp_le_u64 vcc, %0, %0" : : "v"(x[i]) : "vcc");
                if (KIND == SUB_CO_PAIR) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %1, %0, vcc" : "+v"(a[i]), "+v"(b[i]) : : "vcc");
                if (KIND == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (KIND == MAD_U64_U32_SGPR_OUT) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(x[i]), "=s"(sd) : "v"(a[i]), "v"(b[i]));
                if (KIND == MAD_PLUS_ADD) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2" : "+v"(x[i]), "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == MAD_PLUS_LSHL_ADD) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_lshl_add_u64 %3, %3, 0, %3" : "+v"(x[i]) : "v"(a[i]), "v"(b[i]), "v"(x[(i + 4) & 7]) : "vcc");
                if (KIND == MAD_PLUS_2ADD) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %1" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : : "vcc");
                if (KIND == MUL_LO_PLUS_ADD) asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(a[i]), "+v"(b[i]));
                if (KIND == MAD_PLUS_CNDMASK) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_cndmask_b32 %1, %1, %2, s[10:11]" : "+v"(x[i]), "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == MAD_PLUS_MUL_LO) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(x[i]), "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == CMP_PLUS_CNDMASK) asm volatile("v_cmp_le_u64 vcc, %2, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(x[i]) : "vcc");
                // the four conditional-subtraction forms: x >= m ? x - m : x with nm = 2^64 - m in s[12:13] / m in s[14:15]
                if (KIND == CSUB_CMP_SEL_ADD) asm volatile("v_cmp_le_u64 vcc, s[14:15], %0\n\tv_cndmask_b32 %1, 0, %3, vcc\n\tv_cndmask_b32 %2, 0, %4, vcc\n\tv_lshl_add_u64 %0, %0, 0, %0" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(b[(i + 1) & 7]) : "vcc");
                if (KIND == CSUB_ADDC_SEL) asm volatile("v_add_co_u32 %1, vcc, s12, %1\n\ts_nop 1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc\n\ts_nop 1\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(b[(i + 1) & 7]) : "vcc");
                if (KIND == CSUB_CMP_SGPR_SEL_ADD) asm volatile("v_cmp_le_u64 s[16:17], s[14:15], %0\n\ts_nop 1\n\tv_cndmask_b32 %1, 0, %3, s[16:17]\n\tv_cndmask_b32 %2, 0, %4, s[16:17]\n\tv_lshl_add_u64 %0, %0, 0, %0" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(b[(i + 1) & 7]) : "s16", "s17");
                if (KIND == CSUB_SUB_MIN32) asm volatile("v_lshl_add_u64 %0, %0, 0, s[12:13]\n\tv_cmp_gt_i32 vcc, 0, %1\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(b[(i + 1) & 7]) : "vcc");
                if (KIND == MAD_PLUS_3ADD) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %1, %1, %2" : "+v"(x[i]), "+v"(a[i]), "+v"(b[i]) : : "vcc");
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += x[i] + a[i] + b[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
static void run_kind(u64 *out, int cus, int waves_per_simd, double clock_hz) {
    // 256 threads = 4 waves = one per SIMD; waves_per_simd workgroups per CU
    const unsigned blocks = (unsigned)(cus * waves_per_simd), threads = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1ull);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 2ull + r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double rounds = (double)blocks * threads * ITERS * 4 * 8;        // asm statements executed (per lane)
    const double insts = rounds * kind_insts[KIND];
    const double rate = insts / (best * 1e-3);
    const double wave_insts_per_simd = (double)waves_per_simd * ITERS * 4 * 8;   // statements per SIMD
    const double cycles_per_stmt = best * 1e-3 * clock_hz / wave_insts_per_simd;
    printf("{\"stream\": \"%s\", \"waves_per_simd\": %d, \"lane_insts_per_s\": %.4g, \"ms\": %.3f, \"simd_cycles_per_statement\": %.2f, "
           "\"insts_per_statement\": %d}\n", kind_name[KIND], waves_per_simd, rate, best, cycles_per_stmt, kind_insts[KIND]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    const double mhz = argc > 2 ? atof(argv[2]) : 2380.0;     // engine clock under this load (rocm-smi), for the cycle column
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u64 *out;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(u64)));
    fprintf(stderr, "%s, %d CUs, %d waves per SIMD, clock assumed %.0f MHz\n", prop.name, cus, wps, mhz);
#define RUN(K) run_kind<K>(out, cus, wps, mhz * 1e6);
    RUN(ADD_U32) RUN(ADD3_U32) RUN(MOV_B32) RUN(CNDMASK) RUN(LSHL_ADD_U64) RUN(CMP_LE_U64) RUN(SUB_CO_PAIR) RUN(MUL_LO_U32)
    RUN(MUL_HI_U32) RUN(MAD_U64_U32) RUN(MAD_U64_U32_SGPR_OUT) RUN(MAD_PLUS_ADD) RUN(MAD_PLUS_LSHL_ADD) RUN(MAD_PLUS_2ADD)
    RUN(MUL_LO_PLUS_ADD) RUN(MAD_PLUS_CNDMASK) RUN(MAD_PLUS_MUL_LO) RUN(MAD_PLUS_3ADD)
    RUN(CNDMASK_SGPR) RUN(CMP_PLUS_CNDMASK) RUN(CSUB_CMP_SEL_ADD) RUN(CSUB_ADDC_SEL) RUN(CSUB_CMP_SGPR_SEL_ADD) RUN(CSUB_SUB_MIN32)
    CHECK(hipFree(out));
    return 0;
}
