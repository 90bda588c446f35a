// kernels_common.hpp -- what every kernel family shares: the row map, LDS padding, scalar-register / wave-local
// helpers and the phase-timing hook.  (Part of kernels.hpp, split by family in round 3 for reviewability.)
#pragma once
#include <type_traits>

#include "rt.hpp"
#include "zq_dev.hpp"

namespace fhe {
namespace k {

struct u64x2 {
    u64 x, y;
};

// Streaming accesses: data that ONE lane reads or writes once (a database row of the PIR loop, a ciphertext that is
// multiplied by a plaintext and never looked at again) is marked non-temporal, so that it does not push the operands
// other workgroups re-read (shared queries, keys, tables) out of L2 / the Infinity Cache.  (FHE_STREAM_NT, knobs.hpp)
#if defined(__HIP_DEVICE_COMPILE__) && FHE_STREAM_NT
typedef u64 u64v2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64x2 load_stream(const u64x2 *p) {
    const u64v2_native v = __builtin_nontemporal_load(reinterpret_cast<const u64v2_native *>(p));
    return u64x2{v.x, v.y};
}
__device__ __forceinline__ void store_stream(u64x2 *p, u64x2 v) {
    u64v2_native w;
    w.x = v.x, w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<u64v2_native *>(p));
}
#else
__device__ __forceinline__ u64x2 load_stream(const u64x2 *p) { return *p; }
__device__ __forceinline__ void store_stream(u64x2 *p, u64x2 v) { *p = v; }
#endif

// (last-use loads of the multiply pipeline, knobs.hpp FHE_PIPE_NT)
template <bool NT>
__device__ __forceinline__ u64 load_last(const u64 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <bool NT>
__device__ __forceinline__ u64x2 load_last2(const u64x2 *p) {
#if defined(__HIP_DEVICE_COMPILE__) && FHE_STREAM_NT
    if constexpr (NT) return load_stream(p);
#endif
    return *p;
}

// FHE_TS(k): phase-timing stamps of one wave, compiled to nothing except in -DFHE_LAB -DFHE_PHASE_TIMING builds
// (tools/ks_phase_timing.py).
#if defined(FHE_LAB) && defined(FHE_PHASE_TIMING)
#include "lab/phase_timing.hpp"
#else
#define FHE_TS(k) do { } while (0)
#define FHE_TSK(k) do { } while (0)
#define FHE_TS_BEGIN() do { } while (0)
#define FHE_TS_END() do { } while (0)
#endif

// Maps a workgroup index to (polynomial, row) and to source/destination addresses.
// block b -> poly = b / rows, r = row_begin + b % rows;
//   src = in  + poly*src_poly_stride + (src_row_fixed >= 0 ? src_row_fixed : r) * N
//   dst = out + poly*dst_poly_stride + r * N ;  modulus index = mod_offset + r
struct RowMap {
    uint32_t rows;       // rows processed per polynomial
    uint32_t row_begin;  // first row inside the polynomial
    int32_t mod_offset;  // modulus index of row r is mod_offset + r
    int32_t src_row_fixed;
    u64 src_poly_stride, dst_poly_stride;  // in u64 elements
    // Two source arrays in one launch (the operand extensions of a multiply: lhs and rhs ciphertexts live in separate
    // buffers, their transforms go to ONE scratch array): polynomials [split, ...) are read from
    // in2 + (poly - split) * src_poly_stride.  in2 == nullptr: one source (every other launch).
    const u64 *in2;
    uint32_t split;
    // workgroups walk the launch's tiles from the last one backwards (the producer of `in` ran ascending, so its
    // most recent output -- what still sits in the Infinity Cache -- is at the end)
    uint32_t reverse;
    // ntt_kernel<true, ..., GATHER = true> only (round 5): the source row is read through the Ntt-domain substitution
    // x -> x^subst_exp (Poly::substitute, M/rq/mod.rs:360-412) -- element d of the row the transform works on is element
    // galois_src_index(d) of the stored row -- so that a Galois rotation needs no separate permutation pass
    uint32_t subst_exp;
};
// the source polynomial of workgroup-uniform index `poly`
__device__ __forceinline__ const u64 *rowmap_src(const RowMap &map, const u64 *in, uint32_t poly) {
    return (map.in2 != nullptr && poly >= map.split) ? map.in2 + (u64)(poly - map.split) * map.src_poly_stride
                                                     : in + (u64)poly * map.src_poly_stride;
}


// Ntt-domain substitution x -> x^e as a gather (M/rq/mod.rs:389-402: q[bitrev(j)] = p[bitrev((e-1)/2 + j e mod N)]):
// the source index of destination element d of a row of 2^logn points.
FHE_HD uint32_t bitrev_n(uint32_t v, uint32_t logn) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - logn);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < logn; i++) r |= ((v >> i) & 1u) << (logn - 1 - i);
    return r;
#endif
}
FHE_HD uint32_t galois_src_index(uint32_t d, uint32_t e, uint32_t logn) {
    const uint32_t mask = (1u << logn) - 1;
    const uint32_t jj = bitrev_n(d, logn);
    return bitrev_n((uint32_t)(((u64)(e - 1) / 2 + (u64)jj * e) & mask), logn);
}

// LDS padding: one extra u64 every 16 keeps the 16-element-strided accesses of the last
// radix pass (lane stride 128 B) on distinct banks (ds_read_b64: 64 banks x 4 B, conflicts
// are per 32-lane half; 17*l mod 32 is a bijection).
// Round 2: in a model of 32 bank pairs per 32-lane half this layout is two-way conflicted in EVERY access pattern of
// the passes (a unit-stride half spans 34 words) -- the SQ counters agree: half of all LDS cycles are conflict cycles.
// A layout found by enumeration, i + 3 * (i >> 5) (tools/lds_pad_search.py), is conflict-free in seven of the nine
// patterns; built, bit-exact, and measured in a drift-cancelling ABBA run: every kernel within +-1 %
// (profiles/r02_lds_pad_ab.txt).  LDS time is not on these kernels' critical path; this layout (2 KiB smaller per
// tile) stays and the alternative is not carried in the source.
// Round 6: with the F64 instances the arithmetic of a pass shrank by a third and LDS time came closer to the critical path, so
// the alternative layout is back as a lab knob (FHE_LDS_PAD=1: i + 3 (i >> 5); per-element offsets of a group stay additive,
// (base mod 32) + (offset mod 32) < 32 by the same argument as below).  Measured again (profiles/r06_lds_pad_ab_rejected.jsonl, two lab builds alternating three times, same digest): transforms on 60-bit / 62-bit / F64
// rows, the stock and C2 multiplies, C3 relinearise -- every cell within -2.9 ... +2.4 %, medians within 1 %: still not on the critical path.  Off.
FHE_HD uint32_t padi(uint32_t i) { return FHE_LDS_PAD ? i + 3 * (i >> 5) : i + (i >> 4); }
FHE_HD uint32_t lds_words(uint32_t n) { return FHE_LDS_PAD ? n + 3 * (n >> 5) + 4 : n + (n >> 4) + 2; }

constexpr int GMAX = 4;  // radix-16: up to four butterfly stages per LDS round trip

// A wave-uniform value moved to a scalar register so that table addresses derived from it are
// scalar and the twiddle loads become s_load (no VGPRs, no per-lane address math).
// Compiler scheduling fence: keeps a batch of loads (and the registers they pin) from being
// hoisted across it.  No instruction is emitted.
// (s_setprio 3 from a workgroup's start until its operand loads are issued -- so that a freshly dispatched workgroup
// gets its loads out ahead of its CU neighbour's arithmetic -- was measured: forward NTT unchanged, inverse NTT 11 %
// and tensor+iNTT 5 % slower, profiles/r02_setprio_ab.txt.)
__device__ __forceinline__ void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// Makes a per-lane value opaque to loop-invariant code motion: address arithmetic derived from
// it is recomputed per iteration (a few integer ops) instead of being hoisted and spilled.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
// A block-uniform value the compiler computed on the VALU (integer division has no scalar
// form) stays in a VGPR, and so does all address arithmetic derived from it; the builtin
// readfirstlane is folded away for provably uniform inputs, so this goes through asm.
__device__ __forceinline__ uint32_t to_sgpr(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    // The hazard recognizer does not look inside asm: gfx950 needs a wait state between the VALU
    // write of a VGPR and a readlane of it (leading s_nop), and 5 wait states before a VMEM
    // instruction may use the VALU-written SGPR as an address (trailing s_nop).
    asm("s_nop 1\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
    return r;
#else
    return v;
#endif
}
// Exchange through LDS between lanes of ONE wavefront: LDS instructions of a wave execute in order, so all that
// is needed is that the compiler keeps the reads behind the writes -- no s_barrier, the other waves of the
// workgroup run on.  Used between radix passes whose groups stay inside the wave's own block of the tile
// (wave_local_exchange below).  Host emulation (fibers per thread): the workgroup barrier.
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "wave_sync() / wave_local_exchange() assume 64-lane wavefronts (gfx950)"
#endif
// FHE_BARRIER: the workgroup barrier of the NTT / tensor / key-switch kernels.
#define FHE_BARRIER() __syncthreads()
__device__ __forceinline__ void wave_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#else
    __syncthreads();
#endif
}
// Pass p hands thread t the groups {g0 + t}; a group of pass (skip, G) is {base + (e << skip)} with
// base = (grp >> skip) << (skip + G) | (grp & (2^skip - 1)).  When 2^skip <= 64 the 64 consecutive groups of a
// wave cover the contiguous elements [(g0 + 64 w) 2^G, + 64 * 2^G); two passes with the same G and both skips
// <= 6 therefore read and write the same per-wave ranges, and the exchange between them is wave-local.
constexpr bool wave_local_exchange(int skip_a, int g_a, int skip_b, int g_b) {
    return g_a == g_b && skip_a <= 6 && skip_b <= 6;
}
// Host emulation maps wave_sync() to the workgroup barrier, which would hide a violated invariant; so every pass
// that sits next to a wave-local exchange checks there, element by element, that what a thread touches lies in its
// own wave's block [(g0 + 64 w) 2^G, + 64 * 2^G) of the tile (g0: first group of the pass iteration, w = tid / 64).
template <int G>
__device__ __forceinline__ void wave_block_check(uint32_t g0, uint32_t tid, uint32_t idx) {
#if defined(FHE_HOST_EMULATION)
    const uint32_t lo = (g0 + (tid & ~63u)) << G;
    if (idx < lo || idx >= lo + (64u << G)) __builtin_trap();   // a wave-local exchange would race on the GPU
#endif
}
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}

}  // namespace k
}  // namespace fhe
