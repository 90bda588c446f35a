// phase_timing.hpp -- lab-only instrumentation (see kernels_common.hpp FHE_TS); never part of the release build.
#pragma once
// Phase timing of one wave (diagnostic builds only: -DFHE_LAB -DFHE_PHASE_TIMING=1 or =2; tools/ks_phase_timing.py).
//   mode 1  thread 0 of workgroup FHE_TS_BLOCK adds the shader-clock time since its previous stamp to global slot `k`
//           at every FHE_TS / FHE_TSK: fine-grained (every pass of the transform), but each stamp is a global
//           read-modify-write and costs vector registers -- the N = 16384 key switch (124 of 128 VGPRs) spills under it.
//   mode 2  (round 5) kernel-scope stamps only (FHE_TSK), kept in SCALAR registers: s_memtime lands in an SGPR pair and
//           the per-slot sums are wave-uniform 64-bit scalars, so the instrumented kernel uses the same vector
//           registers as the product; thread 0 of workgroup FHE_TS_BLOCK writes the sums out once, at exit.
//           Every wave of that workgroup reports (row w of the slot table), so the budget is the AVERAGE over the waves,
//           not the view of wave 0 (the oldest wave of its SIMD wins issue arbitration and "waits" at barriers for the rest).
// With a stamp at entry and one at exit the slots telescope: their sum IS the wave's time between the two.
// slots: mode 1 uses [0, 64); mode 2 keeps one row of 32 per wave of the stamped workgroup (up to 16 waves): row w, slot k
// at [32 * w + k], the number of kernel entries seen by wave w at [32 * w + 31]
#define FHE_TS_TOTAL_SLOTS 512
#if defined(FHE_PHASE_TIMING) && !defined(FHE_HOST_EMULATION)
__device__ unsigned long long g_phase_ts[FHE_TS_TOTAL_SLOTS];
__device__ unsigned long long g_phase_last;
#endif
#if defined(FHE_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#ifndef FHE_TS_BLOCK
#define FHE_TS_BLOCK 100
#endif
#if FHE_PHASE_TIMING == 2
#define FHE_TS(k) do { } while (0)
#define FHE_TS_NSLOT 24
#define FHE_TS_BEGIN()                                                      \
    unsigned long long fhe_ts_last_ = __builtin_amdgcn_s_memtime();         \
    unsigned long long fhe_ts_acc_[FHE_TS_NSLOT] = {0}
#define FHE_TSK(k)                                                          \
    do {                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
        fhe_ts_acc_[(k)] += now_ - fhe_ts_last_;                            \
        fhe_ts_last_ = now_;                                                \
    } while (0)
#define FHE_TS_END()                                                        \
    do {                                                                    \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == FHE_TS_BLOCK) {        \
            const unsigned w_ = (threadIdx.x >> 6) & 15;                    \
            for (int k_ = 0; k_ < FHE_TS_NSLOT; k_++) g_phase_ts[32 * w_ + k_] += fhe_ts_acc_[k_]; \
            g_phase_ts[32 * w_ + 31] += 1;                                  \
        }                                                                   \
    } while (0)
#else
#define FHE_TS(k)                                                           \
    do {                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == FHE_TS_BLOCK) {               \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            g_phase_ts[(k)] += now_ - g_phase_last;                         \
            g_phase_last = now_;                                            \
        }                                                                   \
    } while (0)
#define FHE_TSK(k) FHE_TS(k)
#define FHE_TS_BEGIN()                                                      \
    do {                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == FHE_TS_BLOCK) {               \
            g_phase_last = __builtin_amdgcn_s_memtime();                    \
            g_phase_ts[63] += 1;                                            \
        }                                                                   \
    } while (0)
#define FHE_TS_END() do { } while (0)
#endif
#else
#define FHE_TS(k) do { } while (0)
#define FHE_TSK(k) do { } while (0)
#define FHE_TS_BEGIN() do { } while (0)
#define FHE_TS_END() do { } while (0)
#endif
