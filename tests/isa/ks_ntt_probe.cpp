// TEST INFRASTRUCTURE (tests/test_isa_guards.py): explicit instantiations of the unfused key switch's stage-A kernel, so
// that its device assembly can be produced in seconds and checked for serialised loads (tools/isa_serial_loads.py).
#include "kernels.hpp"
namespace fhe {
namespace k {
#define FHE_PROBE(LOGM, G0, NW, RNS)                                                                                  \
    template __global__ void ks_ntt_kernel<LOGM, G0, NW, RNS>(const u64 *, u64, u64 *, const DevMod *, const u64x2 *, \
                                                              uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, u64 *, \
                                                              u64, uint32_t, uint32_t);
FHE_PROBE(13, 0, true, false)
FHE_PROBE(13, 0, true, true)
FHE_PROBE(13, 1, true, true)
FHE_PROBE(13, 2, true, false)
}  // namespace k
}  // namespace fhe
