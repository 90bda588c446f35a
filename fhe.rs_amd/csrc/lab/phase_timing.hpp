// phase_timing.hpp -- lab-only instrumentation (see kernels.hpp FHE_TS); never part of the release build.
#pragma once
// Phase timing of one wave (diagnostic builds only: -DFHE_PHASE_TIMING; tools/ks_phase_timing.py).  Thread 0 of
// workgroup FHE_TS_BLOCK adds the shader-clock time since its previous stamp to slot `k`.
#if defined(FHE_PHASE_TIMING) && !defined(FHE_HOST_EMULATION)
__device__ unsigned long long g_phase_ts[64];
__device__ unsigned long long g_phase_last;
#endif
#if defined(FHE_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#ifndef FHE_TS_BLOCK
#define FHE_TS_BLOCK 777
#endif
#define FHE_TS(k)                                                           \
    do {                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == FHE_TS_BLOCK) {               \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
            g_phase_ts[(k)] += now_ - g_phase_last;                         \
            g_phase_last = now_;                                            \
        }                                                                   \
    } while (0)
#else
#define FHE_TS(k) do { } while (0)
#endif

