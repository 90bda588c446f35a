// TEST INFRASTRUCTURE (tests/test_isa_guards.py): explicit instantiations of the FUSED key switch -- the two hot instances of
// BASELINE configs C3 / C5 (N = 16384 tiles, RNS loader) and the generic (non-RNS) instances for rows larger than LDS -- so
// that their device assembly can be produced in seconds and checked for scratch (spills) and serialised loads.
#include "kernels.hpp"
namespace fhe {
namespace k {
#define FHE_PROBE_F(LOGN, NW, GM, RNS, G0)                                                                             \
    template __global__ void ks_fused_kernel<LOGN, NW, GM, 0, RNS, G0, false>(                                         \
        const u64 *, u64, u64 *, u64 *, u64, const u64 *, const u64 *, u64, const u64 *, const u64 *, const u64 *,     \
        const u64 *, const DevMod *, const u64x2 *, uint32_t, uint32_t, uint32_t, const u64 *, u64, uint32_t, uint32_t);
FHE_PROBE_F(14, true, GM_MIXED, true, 0)    // C3 relinearise / rotate
FHE_PROBE_F(14, true, GM_MIXED, true, 1)    // C5 (N = 32768 as two halves)
FHE_PROBE_F(13, true, KS_GMAX, true, 0)     // C2
FHE_PROBE_F(14, true, GM_MIXED, false, 1)   // generic loaders, rows larger than LDS
FHE_PROBE_F(14, true, GM_MIXED, false, 2)
// round 6: the F64 instances the reference's stock sets run on (radix-8 passes at every size, one-word per-lane twiddles)
#define FHE_PROBE_D(LOGN, GALV, HR, TTV)                                                                               \
    template __global__ void ks_fused_kernel<LOGN, false, KS_GMAX, TTV, true, 0, GALV, HR>(                            \
        const u64 *, u64, u64 *, u64 *, u64, const u64 *, const u64 *, u64, const u64 *, const u64 *, const u64 *,     \
        const u64 *, const DevMod *, const u64x2 *, uint32_t, uint32_t, uint32_t, const u64 *, u64, uint32_t, uint32_t);
FHE_PROBE_D(13, false, 5, 512)   // stock n = 8192, launches of more than one workgroup per CU (512 threads x 16 coefficients)
FHE_PROBE_D(13, false, 5, 0)     // stock n = 8192, smaller launches (1024 threads x 8, resident item loop)
FHE_PROBE_D(14, false, 4, 0)     // stock n = 16384 relinearise
FHE_PROBE_D(14, true, 4, 0)      // stock n = 16384 rotations
#define FHE_PROBE_S(G0, NW)                                                                                            \
    template __global__ void ks_fused_split_kernel<G0, 13, NW, false>(                                                 \
        const u64 *, u64, u64 *, u64 *, u64, const u64 *, const u64 *, u64, const u64 *, const u64 *, const u64 *,     \
        const u64 *, const DevMod *, const u64x2 *, uint32_t, uint32_t, uint32_t, const u64 *, u64, uint32_t);
FHE_PROBE_S(2, true)
FHE_PROBE_S(3, true)
}  // namespace k
}  // namespace fhe
