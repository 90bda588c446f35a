// TEST INFRASTRUCTURE (tests/test_isa_guards.py): explicit instantiations of the FUSED key switch's generic (non-RNS)
// instances for rows larger than LDS -- the loaders whose lift mode became a compile-time constant in round 6 -- so that
// their device assembly can be produced in seconds and checked for serialised loads (tools/isa_serial_loads.py).
#include "kernels.hpp"
namespace fhe {
namespace k {
#define FHE_PROBE_F(NW, G0)                                                                                            \
    template __global__ void ks_fused_kernel<14, NW, GM_MIXED, 0, false, G0, false>(                                   \
        const u64 *, u64, u64 *, u64 *, u64, const u64 *, const u64 *, u64, const u64 *, const u64 *, const u64 *,     \
        const u64 *, const DevMod *, const u64x2 *, uint32_t, uint32_t, uint32_t, const u64 *, u64, uint32_t, uint32_t);
FHE_PROBE_F(true, 1)
FHE_PROBE_F(true, 2)
#define FHE_PROBE_S(G0, NW)                                                                                            \
    template __global__ void ks_fused_split_kernel<G0, 13, NW, false>(                                                 \
        const u64 *, u64, u64 *, u64 *, u64, const u64 *, const u64 *, u64, const u64 *, const u64 *, const u64 *,     \
        const u64 *, const DevMod *, const u64x2 *, uint32_t, uint32_t, uint32_t, const u64 *, u64, uint32_t);
FHE_PROBE_S(2, true)
FHE_PROBE_S(3, true)
}  // namespace k
}  // namespace fhe
