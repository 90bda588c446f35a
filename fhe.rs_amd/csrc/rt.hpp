// rt.hpp -- the one place that chooses the runtime the kernels are compiled against.
//
// Product build (hipcc --offload-arch=gfx950): the HIP runtime, nothing else.
// FHE_HOST_EMULATION is defined ONLY by tests/emu/build.sh: it compiles these very kernel
// sources for the host with a fiber-based workgroup emulator so that indexing/barrier logic
// can be checked against the oracle in a container without a GPU.  It is test
// infrastructure, never shipped, never loaded by the package (fhe.rs_amd/_lib.py loads
// libfhe_hip.so only and fails loudly when it is missing).
#pragma once

#ifdef FHE_HOST_EMULATION
#include "emu_rt.hpp"  // tests/emu/emu_rt.hpp (on the include path of the emulation build only)
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: a launch that carries its own start / stop events (profiler)
#define FHE_DYN_SMEM(type, name)                                            \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type *name = reinterpret_cast<type *>(name##_raw)
#endif
