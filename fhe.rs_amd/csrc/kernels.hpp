// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the fhe.rs BFV hot path.
//
// Layout in HBM: every polynomial is `[rows][N]` u64 row-major exactly as rq::Poly
// (M/rq/mod.rs:126-133); batches add outer dimensions.  Row-wise phases (NTT, key-switch) use
// one workgroup per residue row with the row staged in LDS; column-wise phases (RNS scaler,
// modulus switch) use one lane per coefficient so that all row reads/writes are coalesced
// along N -- no transposes anywhere.  No MFMA: this is 64-bit modular integer arithmetic.
//
// Kernel inventory (reference loop each one replaces):
//   ntt_kernel<false/true>  NttOperator::forward / backward        M/ntt/native.rs:77-233
//   ntt_global_kernel<..>   first/last radix stages for N > 16384   (row does not fit LDS)
//   ks_fused_kernel         KeySwitchingKey::key_switch             F/bfv/keys/key_switching_key.rs:241-320
//                           (+ lazy lift M/rq/mod.rs:563-586 + Shoup MAC M/rq/ops.rs:208-245)
//   ks_fused_split_kernel   the same for N >= 32768: per 8192-point sub-block, first stages in the loader
//   scale_kernel            RnsScaler::scale per column             M/rns/scaler.rs:249-352, M/rq/scaler.rs:85-94
//   switch_down_kernel      Poly::switch_down                       M/rq/mod.rs:433-492
//   substitute_kernel       Poly::substitute                        M/rq/mod.rs:360-412
//   tensor_intt_kernel      tensor step fused with the following inverse NTT   F/bfv/ops/mul.rs:198-205
//   ew_kernel / tensor_kernel / tensor_general_kernel / mul_shoup_kernel   M/rq/ops.rs:10-245, F/bfv/ops/mul.rs:198-201,
//                                                                   F/bfv/ops/mod.rs:300-327
//   dot_kernel, mul_plain_kernel        dot_product_scalar, ct x pt F/bfv/ops/dot_product.rs:54-180, ops/mod.rs:229-257
//   expand_step_kernel, monomial_kernel EvaluationKey::expands      F/bfv/keys/evaluation_key.rs:192-256
//   phase_kernel, decrypt_tail_kernel   SecretKey::try_decrypt      F/bfv/keys/secret_key.rs:198-247
//   wire_pack_kernel, wire_unpack_kernel  Rq payload bit packing    M/rq/convert.rs:17-99, fhe-util lib.rs:71-148
//   synth_kernel            synthetic uniform residues (bench/test inputs)
// Compile-time knobs live in knobs.hpp (pinned in the release build); rejected kernel variants in lab/ (lab builds only).
#pragma once
#include <type_traits>

#include "rt.hpp"
#include "zq_dev.hpp"

namespace fhe {
namespace k {

struct u64x2 {
    u64 x, y;
};

// FHE_TS(k): phase-timing stamps of one wave, compiled to nothing except in -DFHE_LAB -DFHE_PHASE_TIMING builds
// (tools/ks_phase_timing.py).
#if defined(FHE_LAB) && defined(FHE_PHASE_TIMING)
#include "lab/phase_timing.hpp"
#else
#define FHE_TS(k) do { } while (0)
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
};

enum { PRO_NONE = 0, PRO_REDUCE = 1 };

// LDS padding: one extra u64 every 16 keeps the 16-element-strided accesses of the last
// radix pass (lane stride 128 B) on distinct banks (ds_read_b64: 64 banks x 4 B, conflicts
// are per 32-lane half; 17*l mod 32 is a bijection).
// Round 2: in a model of 32 bank pairs per 32-lane half this layout is two-way conflicted in EVERY access pattern of
// the passes (a unit-stride half spans 34 words) -- the SQ counters agree: half of all LDS cycles are conflict cycles.
// A layout found by enumeration, i + 3 * (i >> 5) (tools/lds_pad_search.py), is conflict-free in seven of the nine
// patterns; built, bit-exact, and measured in a drift-cancelling ABBA run: every kernel within +-1 %
// (profiles/r02_lds_pad_ab.txt).  LDS time is not on these kernels' critical path; this layout (2 KiB smaller per
// tile) stays and the alternative is not carried in the source.
FHE_HD uint32_t padi(uint32_t i) { return i + (i >> 4); }
FHE_HD uint32_t lds_words(uint32_t n) { return n + (n >> 4) + 2; }

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

// ------------------------------------------------------------------ tile geometry ----
// A tile of M = 2^LOGM coefficients is processed by T threads; every radix pass handles groups
// of 2^G coefficients per thread.  NTT: 16 coefficients per thread (one radix-16 group);
// key switch: 8 per thread (leaves registers for the two accumulator sets).
constexpr int ntt_threads_c(int logm) { return (1 << logm) / 16 > 64 ? ((1 << logm) / 16 > 1024 ? 1024 : (1 << logm) / 16) : 64; }
constexpr bool KS_LATE = FHE_KS_LATE;  // key-switch transforms: wider radix passes last (wave-local exchanges)
constexpr int KS_GMAX = 3;  // radix-8 passes inside the key switch: room for the accumulators
constexpr int ks_threads_c(int logn) { return (1 << logn) / 8 > 64 ? ((1 << logn) / 8 > 1024 ? 1024 : (1 << logn) / 8) : 64; }
// 16-byte chunks per thread (0: tile smaller than one chunk per thread -> scalar loop)
constexpr int tile_chunks_c(int logm, int threads) { return (1 << logm) >= 2 * threads ? (1 << logm) / (2 * threads) : 0; }
// fused key switch: the c1 accumulators live in LDS behind the row tile (and the next digit's row is
// prefetched into the registers this frees) whenever the tile leaves room, i.e. up to N = 8192
constexpr bool ks_acc1_in_lds_c(int logn) { return logn <= 13 && tile_chunks_c(logn, ks_threads_c(logn)) > 0; }
// pass plan: NP = ceil(LOGM / GMAX) passes of BASE or BASE+1 stages
constexpr int plan_np(int logm, int gmax) { return (logm + gmax - 1) / gmax; }
constexpr int plan_base(int logm, int gmax) { return logm / plan_np(logm, gmax); }
constexpr int plan_rem(int logm, int gmax) { return logm % plan_np(logm, gmax); }

// ---------------------------------------------------------------- forward passes ----
// Stages [S0, S0+G) of the size-2^LOGM Cooley-Tukey transform held in `lds`.
// A group = 2^G elements {base + e*lo_count}; all G stages stay in registers.
// Twiddle of (stage st, block i) is tw[(kbase << st) + i]  (kbase = 1 for a whole row;
// (2^G0 + sub) when this LDS tile is sub-block `sub` after G0 global stages).
// UNIFORM (64 consecutive groups share the block index, i.e. lo_bits >= 6): the twiddles are
// wave-uniform and come through the scalar cache.  Otherwise all 2^G - 1 twiddles of a group
// are fetched up front, one batch of loads in flight instead of a dependent load per stage.
// `Src`: NoSrc -> the group is read from the LDS tile; otherwise a functor (idx, e) -> coefficient
// idx (= element e of the calling thread's group) that
// feeds the pass straight from global memory / registers (first pass only: the elements of a
// group are 2^lo_bits apart, so consecutive lanes read consecutive coefficients -- coalesced --
// and one LDS round trip plus its barrier disappear).
struct NoSrc {};
// Per-lane twiddles of a non-UNIFORM pass: all 2^G - 1 of every group the thread handles.  They
// are fetched BEFORE the barrier that precedes the pass (they do not depend on the tile), so
// their L2 latency overlaps the barrier instead of following it.
template <int G, int LOGM, int S0, int T>
struct FwdTw {
    static constexpr bool UNIFORM = (LOGM - S0 - G) >= 6;
    static constexpr int NG = (1 << (LOGM - G)) > T ? (1 << (LOGM - G)) / T : 1;  // groups per thread
    u64x2 w[UNIFORM ? 1 : NG][UNIFORM ? 1 : (1 << G) - 1];
};
template <int G, int LOGM, int S0, int T>
__device__ __forceinline__ void fwd_tw_load(FwdTw<G, LOGM, S0, T> &tw_regs, const u64x2 *__restrict__ tw,
                                            uint32_t kbase, uint32_t tid) {
    using W = FwdTw<G, LOGM, S0, T>;
    if constexpr (!W::UNIFORM) {
        constexpr uint32_t lo_bits = LOGM - S0 - G;
        constexpr uint32_t ngroups = 1u << (LOGM - G);
#pragma unroll
        for (int gi = 0; gi < W::NG; gi++) {
            const uint32_t grp = gi * T + tid;
            if (ngroups < T && grp >= ngroups) break;
            const uint32_t hi = grp >> lo_bits;
#pragma unroll
            for (int u = 0; u < G; u++) {
                const uint32_t kst = (kbase << (S0 + u)) + (hi << u);
#pragma unroll
                for (uint32_t blk = 0; blk < (1u << u); blk++) tw_regs.w[gi][(1u << u) - 1 + blk] = tw[kst + blk];
            }
        }
    }
}
// PRE: the per-lane twiddles were fetched ahead into tw_regs; otherwise each group fetches its
// own right before use (fewer live registers).
// NARROW = b0 > 0: moduli below 2^60 and input to stage 0 below b0*p (1: canonical) --
// fwd_butterfly_narrow (zq_dev.hpp); values are below 16p on exit instead of 4p.
// NT > 1: the same pass on NT tiles that lie `tile_words` apart in LDS (the key switch transforms two digits
// under one modulus at once): addresses and twiddles are formed once and serve every tile.
template <int G, int LOGM, int S0, int T, bool PRE, int NARROW = 0, class Src = NoSrc, int NT = 1, bool WLX = false>
__device__ __forceinline__ void fwd_pass(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                         uint32_t tid, const FwdTw<G, LOGM, S0, T> &tw_regs, Src src = Src{},
                                         uint32_t tile_words = 0) {
    constexpr bool DIRECT = !std::is_same<Src, NoSrc>::value;
    constexpr uint32_t R = 1u << G;
    constexpr uint32_t lo_bits = LOGM - S0 - G;
    constexpr uint32_t ngroups = 1u << (LOGM - G);
    constexpr bool UNIFORM = lo_bits >= 6;
#pragma unroll
    for (uint32_t g0 = 0; g0 < ngroups; g0 += T) {
        const uint32_t grp = g0 + tid;
        if (ngroups < T && grp >= ngroups) break;
        const uint32_t lo = grp & ((1u << lo_bits) - 1);
        uint32_t hi = grp >> lo_bits;
        if (UNIFORM) hi = wave_uniform(hi);
        const uint32_t base = ((grp >> lo_bits) << (LOGM - S0)) + lo;
        if constexpr (WLX) {   // (emulation: this pass reads or writes across a wave-local exchange)
            for (uint32_t e = 0; e < R; e++) wave_block_check<G>(g0, tid, base + (e << lo_bits));
        }
        u64x2 w[UNIFORM || PRE ? 1 : R - 1];
        if constexpr (!UNIFORM && !PRE) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                const uint32_t kst = (kbase << (S0 + u)) + (hi << u);
#pragma unroll
                for (uint32_t blk = 0; blk < (1u << u); blk++) w[(1u << u) - 1 + blk] = tw[kst + blk];
            }
        }
        // padi(base + off) = padi(base) + padi(off) for every element of a group (no carry out
        // of the low four bits: a group never straddles a 16-element pad block unless off is a
        // multiple of 16), so the per-element LDS offsets are compile-time constants.
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
        u64 *const g = lds + (NT > 1 ? tl * tile_words : 0u) + padi(base);
        u64 x[R];
        if constexpr (DIRECT) {
#pragma unroll
            for (uint32_t e = 0; e < R; e++) x[e] = src(base + (e << lo_bits), e);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < R; e++) x[e] = g[padi(e << lo_bits)];
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t half = R >> (u + 1);
            const uint32_t kst = (kbase << (S0 + u)) + (hi << u);
#pragma unroll
            for (uint32_t blk = 0; blk < (1u << u); blk++) {
                const u64x2 wv = UNIFORM ? tw[kst + blk]
                                 : PRE   ? tw_regs.w[UNIFORM || !PRE ? 0 : g0 / T][UNIFORM || !PRE ? 0 : (1u << u) - 1 + blk]
                                         : w[UNIFORM || PRE ? 0 : (1u << u) - 1 + blk];
#pragma unroll
                for (uint32_t j = 0; j < half; j++) {
                    const uint32_t a = blk * 2 * half + j;
                    if constexpr (NARROW > 0)
                        fwd_butterfly_narrow<UNIFORM>(x[a], x[a + half], wv.x, wv.y, pm, fwd_narrow_corrects(S0 + u, NARROW));
                    else
                        fwd_butterfly<UNIFORM>(x[a], x[a + half], wv.x, wv.y, pm);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) g[padi(e << lo_bits)] = x[e];
        if constexpr (NT > 1) sched_fence();   // one tile's group in registers at a time
        }
    }
}

// All stages of a size-2^LOGM forward transform on an LDS tile (values < 4p on exit); the
// pass plan is resolved at compile time.  Early passes (scalar twiddles) take the wider radix.
// LATE = false: the wider passes come first (they run on scalar twiddles); LATE = true: last, so that the trailing
// passes share one radix and their exchanges are wave-local (the key switch at N = 8192: 2+2+3+3+3, two workgroup
// barriers per digit transform instead of four).
// GM_MIXED: radix-8 passes as long as a pass's twiddles are wave-uniform (stage offset + 3 <= LOGM - 6: they come
// through scalar registers and cost no VGPRs), radix-4 passes after that (3 per-lane twiddles instead of 7: 16
// VGPRs less than a per-lane radix-8 pass) -- the key switch at N = 16384, whose two accumulator sets leave ~64 VGPRs.
constexpr int GM_MIXED = 32;
constexpr int mixed_plan_g(int logm, int pass) {   // stages of pass `pass` (0 beyond the last pass)
    int s0 = 0;
    for (int q = 0;; q++) {
        if (s0 >= logm) return 0;
        int g = (s0 + 3 <= logm - 6) ? 3 : 2;
        if (g > logm - s0) g = logm - s0;
        if (q == pass) return g;
        s0 += g;
    }
}
constexpr int mixed_plan_np(int logm) {
    int n = 0;
    while (mixed_plan_g(logm, n) > 0) n++;
    return n;
}
constexpr int fwd_np(int logm, int gm) { return gm == GM_MIXED ? mixed_plan_np(logm) : plan_np(logm, gm); }
template <int LOGM, int GM, int PASS, bool LATE = false>
constexpr int fwd_plan_g() {
    if (GM == GM_MIXED) return mixed_plan_g(LOGM, PASS);
    return plan_base(LOGM, GM) +
           ((LATE ? PASS >= plan_np(LOGM, GM) - plan_rem(LOGM, GM) : PASS < plan_rem(LOGM, GM)) ? 1 : 0);
}
constexpr int fwd_plan_g_c(int logm, int gm, int pass, bool late) {
    if (gm == GM_MIXED) return mixed_plan_g(logm, pass);
    return plan_base(logm, gm) + ((late ? pass >= plan_np(logm, gm) - plan_rem(logm, gm) : pass < plan_rem(logm, gm)) ? 1 : 0);
}
// is the exchange between forward passes `pass` and `pass + 1` wave-local?
constexpr bool fwd_wl_after(int logm, int gm, bool late, int pass) {
    if (pass < 0 || pass + 1 >= fwd_np(logm, gm)) return false;
    int s0 = 0;
    for (int q = 0; q < pass; q++) s0 += fwd_plan_g_c(logm, gm, q, late);
    const int g = fwd_plan_g_c(logm, gm, pass, late), gn = fwd_plan_g_c(logm, gm, pass + 1, late);
    return wave_local_exchange(logm - s0 - g, g, logm - s0 - g - gn, gn);
}
// TWPF: fetch the next pass's per-lane twiddles before the barrier (costs their registers across
// it: the key-switch kernels, which also hold accumulators, leave it off).
// FSYNC = false: the caller places the barrier after the last pass itself (it has loads to issue first).
template <int LOGM, int T, int GM, bool TWPF, bool FSYNC, int NARROW, int PASS, int S0, bool LATE, int NT, class W, class Src>
__device__ __forceinline__ void ntt_fwd_lds_rec(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                                uint32_t tid, const W &tw_regs, Src src, uint32_t tile_words) {
    constexpr int G = fwd_plan_g<LOGM, GM, PASS, LATE>();
    constexpr bool WLX = fwd_wl_after(LOGM, GM, LATE, PASS) || fwd_wl_after(LOGM, GM, LATE, PASS - 1);
    static_assert(!WLX || T % 64 == 0, "wave-local exchanges need whole 64-lane waves");
    if constexpr (PASS == 0)
        fwd_pass<G, LOGM, S0, T, TWPF, NARROW, Src, NT, WLX>(lds, tw, kbase, pm, tid, tw_regs, src, tile_words);   // (Src != NoSrc: reads `src`, not LDS)
    else
        fwd_pass<G, LOGM, S0, T, TWPF, NARROW, NoSrc, NT, WLX>(lds, tw, kbase, pm, tid, tw_regs, NoSrc{}, tile_words);
    FHE_TS(8 + 2 * PASS);
    if constexpr (PASS + 1 < fwd_np(LOGM, GM)) {
        constexpr int GN = fwd_plan_g<LOGM, GM, PASS + 1, LATE>();
        FwdTw<GN, LOGM, S0 + G, T> next;
        if constexpr (TWPF) fwd_tw_load(next, tw, kbase, tid);   // in flight across the barrier
        static_assert(wave_local_exchange(LOGM - S0 - G, G, LOGM - S0 - G - GN, GN) == fwd_wl_after(LOGM, GM, LATE, PASS),
                      "pass plan bookkeeping");
        if constexpr (wave_local_exchange(LOGM - S0 - G, G, LOGM - S0 - G - GN, GN))
            wave_sync();
        else
            FHE_BARRIER();
        FHE_TS(9 + 2 * PASS);
        ntt_fwd_lds_rec<LOGM, T, GM, TWPF, FSYNC, NARROW, PASS + 1, S0 + G, LATE, NT>(lds, tw, kbase, pm, tid, next, NoSrc{},
                                                                                      tile_words);
    } else {
        if constexpr (FSYNC) FHE_BARRIER();
    }
}
template <int LOGM, int T, int GM = GMAX, bool TWPF = true, bool FSYNC = true, int NARROW = 0, class Src = NoSrc,
          bool LATE = false, int NT = 1>
__device__ __forceinline__ void ntt_fwd_lds(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                            uint32_t tid, Src src = Src{}, uint32_t tile_words = 0) {
    constexpr int G = fwd_plan_g<LOGM, GM, 0, LATE>();
    FwdTw<G, LOGM, 0, T> first;
    if constexpr (TWPF) fwd_tw_load(first, tw, kbase, tid);
    ntt_fwd_lds_rec<LOGM, T, GM, TWPF, FSYNC, NARROW, 0, 0, LATE, NT>(lds, tw, kbase, pm, tid, first, src, tile_words);
}

// ---------------------------------------------------------------- inverse passes ----
// Stages [V0, V0+G) (half-lengths 2^V0 .. 2^(V0+G-1)) of the Gentleman-Sande transform.
// Twiddle of (stage v, block i) is itw[koff(v) + i], koff(v) = N - (N >> v) + sub*(M >> (v+1)).
// `fold` (only meaningful for the pass that contains the last stage of a whole-row transform):
// the N^-1 scaling of native.rs:229-232 is folded into the last stage -- x' = (x + y) * N^-1,
// y' = (x - y) * (z * N^-1) -- which saves half a Shoup multiplication per coefficient.
// Per-lane twiddles of the non-UNIFORM (early) inverse passes, fetched ahead like FwdTw.
template <int G, int LOGM, int V0, int T>
struct InvTw {
    static constexpr bool UNIFORM = V0 >= 6;
    static constexpr int NG = (1 << (LOGM - G)) > T ? (1 << (LOGM - G)) / T : 1;
    u64x2 z[UNIFORM ? 1 : NG][UNIFORM ? 1 : (1 << G) - 1];
};
template <int G, int LOGM, int V0, int T>
__device__ __forceinline__ void inv_tw_load(InvTw<G, LOGM, V0, T> &tw_regs, const u64x2 *__restrict__ itw,
                                            uint32_t logn, uint32_t sub, uint32_t tid) {
    using W = InvTw<G, LOGM, V0, T>;
    if constexpr (!W::UNIFORM) {
        constexpr uint32_t R = 1u << G;
        constexpr uint32_t ngroups = 1u << (LOGM - G);
        const uint32_t n = 1u << logn;
#pragma unroll
        for (int gi = 0; gi < W::NG; gi++) {
            const uint32_t grp = gi * T + tid;
            if (ngroups < T && grp >= ngroups) break;
            const uint32_t hi = grp >> V0;
#pragma unroll
            for (int u = 0; u < G; u++) {
                const uint32_t nblk = R >> (u + 1);
                const uint32_t kst = n - (n >> (V0 + u)) + (sub << (LOGM - (V0 + u) - 1)) + hi * nblk;
#pragma unroll
                for (uint32_t blk = 0; blk < nblk; blk++) tw_regs.z[gi][R - 2 * nblk + blk] = itw[kst + blk];
            }
        }
    }
}
// NARROW (moduli below 2^60, 16p < 2^64): the sum output of a Gentleman-Sande butterfly is left
// unreduced -- its bound is the sum of the input bounds, the difference output goes through the Shoup
// multiplication and is below 2p again -- and `bnd[]` tracks every register's bound (in units of p)
// through the fully unrolled stages, so the conditional subtractions shrink to the few needed to keep
// sums below 16p and to hand the next pass values below 2p (7 instead of 12 per radix-8 group).
// FHE_APPROX_SHOUP (zq_dev.hpp): the narrow passes also take the three-partial-product quotient -- the product is
// then below 3p instead of 2p, `bnd[]` holds any integer up to 16, conditional subtractions pick the multiple of p
// (8p, 4p, 2p, p) that leaves the smallest bound, and values travel between passes below BIN / BOUT = 4 p (first
// pass in: 2p; last pass out: 2p, through the exact quotient of the folded last stage): one v_mul_hi_u32 less per
// butterfly for one more conditional subtraction per radix-8 group (8 instead of 7; 20 instead of 16 per radix-16).
constexpr int INV_NARROW_MID = FHE_APPROX_SHOUP ? 4 : 2;
constexpr int inv_best_c(int bd) {   // the power of two c <= 8, c < bd, that minimises max(c, bd - c)
    int best = 1, bv = bd - 1;
    for (int c = 2; c <= 8; c *= 2)
        if (c < bd) {
            const int v = c > bd - c ? c : bd - c;
            if (v <= bv) best = c, bv = v;
        }
    return best;
}
// k * p for a compile-time k <= 16 out of the (uniform) p and 2p by shifts and adds: scalar instructions, where a
// 64-bit multiply by k would go through the vector multiplier
__device__ __forceinline__ u64 small_multiple(const PM &pm, int k) {
    u64 r = 0;
    if (k & 1) r += pm.p;
    if (k & 2) r += pm.p2;
    if (k & 4) r += pm.p2 << 1;
    if (k & 8) r += pm.p2 << 2;
    if (k & 16) r += pm.p2 << 3;
    return r;
}
template <int G, int LOGM, int V0, int T, bool NARROW = false, bool WLX = false, int BIN = 2, int BOUT = 2>
__device__ __forceinline__ void inv_pass(u64 *lds, const u64x2 *__restrict__ itw, uint32_t logn, uint32_t sub,
                                         const PM pm, uint32_t tid, bool fold, u64x2 ninv, u64x2 zninv,
                                         const InvTw<G, LOGM, V0, T> &tw_regs) {
    constexpr uint32_t R = 1u << G;
    constexpr uint32_t ngroups = 1u << (LOGM - G);
    constexpr bool UNIFORM = V0 >= 6;
    const uint32_t n = 1u << logn;
#pragma unroll
    for (uint32_t g0 = 0; g0 < ngroups; g0 += T) {
        const uint32_t grp = g0 + tid;
        if (ngroups < T && grp >= ngroups) break;
        const uint32_t lo = grp & ((1u << V0) - 1);
        uint32_t hi = grp >> V0;
        if (UNIFORM) hi = wave_uniform(hi);
        const uint32_t base = ((grp >> V0) << (V0 + G)) + lo;
        if constexpr (WLX) {   // (emulation, see fwd_pass)
            for (uint32_t e = 0; e < R; e++) wave_block_check<G>(g0, tid, base + (e << V0));
        }
        u64 *const g = lds + padi(base);  // see fwd_pass: constant per-element offsets
        u64 x[R];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) x[e] = g[padi(e << V0)];
        int bnd[R];  // NARROW: x[e] < bnd[e] * p (compile-time after unrolling)
#pragma unroll
        for (uint32_t e = 0; e < R; e++) bnd[e] = BIN;
        // x < bd*p -> x < max(c, bd - c)*p by one conditional subtraction of c*p, c in {8, 4, 2, 1}
        // (straight-line code, no loops: everything folds once the stage loops are unrolled)
        auto reduce = [&](u64 &v, int &bd) {
            const int c = inv_best_c(bd);
            const int sh = c == 8 ? 2 : c == 4 ? 1 : 0;
            v = c == 1 ? csub_n(v, pm.p, pm.np) : csub_n(v, pm.p2 << sh, pm.np2 << sh);
            bd = c > bd - c ? c : bd - c;
        };
        auto fit16 = [&](u64 &va, int &ba, u64 &vb, int &bb) {   // keep va + vb and va + bb*p below 16p
#pragma unroll
            for (int it = 0; it < 4; it++)
                if (ba + bb > 16) {
                    if (ba >= bb) reduce(va, ba); else reduce(vb, bb);
                }
        };
#pragma unroll
        for (int u = 0; u < G; u++) {
            const uint32_t nblk = R >> (u + 1);
            const uint32_t kst = n - (n >> (V0 + u)) + (sub << (LOGM - (V0 + u) - 1)) + hi * nblk;
#pragma unroll
            for (uint32_t blk = 0; blk < nblk; blk++) {
                const u64x2 zv = UNIFORM ? itw[kst + blk] : tw_regs.z[UNIFORM ? 0 : g0 / T][UNIFORM ? 0 : R - 2 * nblk + blk];
#pragma unroll
                for (uint32_t j = 0; j < (1u << u); j++) {
                    const uint32_t a = blk * (2u << u) + j, b = a + (1u << u);
                    if constexpr (NARROW) {
                        fit16(x[a], bnd[a], x[b], bnd[b]);
                        const u64 t = x[a], y = x[b];
                        const u64 diff = small_multiple(pm, bnd[b]) + t - y;   // (a compile-time multiple of the uniform p)
#if defined(FHE_HOST_EMULATION)
                        if (y >= small_multiple(pm, bnd[b]) || (bnd[a] < 16 && t >= small_multiple(pm, bnd[a])) || bnd[a] + bnd[b] > 16)
                            __builtin_trap();  // range tracking broken
#endif
                        if (V0 + G == LOGM && u == G - 1 && fold) {  // exact quotients: both outputs below 2p
                            x[a] = mul_shoup_lazy_n<true>(y + t, ninv.x, ninv.y, pm.np);
                            x[b] = mul_shoup_lazy_n<true>(diff, zninv.x, zninv.y, pm.np);
                        } else {
                            x[a] = y + t;
#if FHE_APPROX_SHOUP
                            x[b] = diff * zv.x + mulhi64_approx<UNIFORM>(diff, zv.y) * pm.np;   // below 3p
#else
                            x[b] = mul_shoup_lazy_n<UNIFORM>(diff, zv.x, zv.y, pm.np);
#endif
                        }
                        bnd[a] = bnd[a] + bnd[b];  // (kept independent of the run-time `fold`)
                        bnd[b] = FHE_APPROX_SHOUP ? 3 : 2;
                    } else if (V0 + G == LOGM && u == G - 1 && fold) {
                        const u64 t = x[a], y = x[b];
                        x[a] = mul_shoup_lazy_n<true>(y + t, ninv.x, ninv.y, pm.np);
                        x[b] = mul_shoup_lazy_n<true>(pm.p2 + t - y, zninv.x, zninv.y, pm.np);
                    } else {
                        inv_butterfly<UNIFORM>(x[a], x[b], zv.x, zv.y, pm);
                    }
                }
            }
        }
        if constexpr (NARROW) {  // the next pass (or the epilogue / the global pass) expects values below BOUT * p
            if (!(V0 + G == LOGM && fold)) {
#pragma unroll
                for (uint32_t e = 0; e < R; e++) {
#pragma unroll
                    for (int it = 0; it < 4; it++)
                        if (bnd[e] > BOUT) reduce(x[e], bnd[e]);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) g[padi(e << V0)] = x[e];
    }
}

// Late passes (scalar twiddles) take the wider radix.  (Storing the last pass straight to global
// memory instead of going through the tile once more was measured: no gain -- 8-byte stores.)
// The caller fetches the first pass's twiddles (inv_tw_first) BEFORE it waits for its tile loads.
template <int LOGM, int PASS>
constexpr int inv_plan_g() {
    return plan_base(LOGM, GMAX) + (PASS >= plan_np(LOGM, GMAX) - plan_rem(LOGM, GMAX) ? 1 : 0);
}
constexpr int inv_plan_g_c(int logm, int pass) {
    return plan_base(logm, GMAX) + (pass >= plan_np(logm, GMAX) - plan_rem(logm, GMAX) ? 1 : 0);
}
constexpr bool inv_wl_after(int logm, int pass) {   // exchange between inverse passes `pass` and `pass + 1`
    if (pass < 0 || pass + 1 >= plan_np(logm, GMAX)) return false;
    int v0 = 0;
    for (int q = 0; q < pass; q++) v0 += inv_plan_g_c(logm, q);
    const int g = inv_plan_g_c(logm, pass);
    return wave_local_exchange(v0, g, v0 + g, inv_plan_g_c(logm, pass + 1));
}
template <int LOGM, int T>
using InvTwFirst = InvTw<inv_plan_g<LOGM, 0>(), LOGM, 0, T>;
template <int LOGM, int T, int PASS = 0, int V0 = 0, bool NARROW = false, class W>
__device__ __forceinline__ void ntt_inv_lds(u64 *lds, const u64x2 *__restrict__ itw, uint32_t logn, uint32_t sub,
                                            const PM pm, uint32_t tid, bool fold, u64x2 ninv, u64x2 zninv,
                                            const W &tw_regs) {
    constexpr int G = inv_plan_g<LOGM, PASS>();
    constexpr bool WLX = inv_wl_after(LOGM, PASS) || inv_wl_after(LOGM, PASS - 1);
    static_assert(!WLX || T % 64 == 0, "wave-local exchanges need whole 64-lane waves");
    constexpr bool LAST = PASS + 1 == plan_np(LOGM, GMAX);
    inv_pass<G, LOGM, V0, T, NARROW, WLX, (PASS == 0 ? 2 : INV_NARROW_MID), (LAST ? 2 : INV_NARROW_MID)>(
        lds, itw, logn, sub, pm, tid, fold, ninv, zninv, tw_regs);
    if constexpr (PASS + 1 < plan_np(LOGM, GMAX)) {
        InvTw<inv_plan_g<LOGM, PASS + 1>(), LOGM, V0 + G, T> next;
        inv_tw_load(next, itw, logn, sub, tid);   // in flight across the barrier
        static_assert(wave_local_exchange(V0, G, V0 + G, inv_plan_g<LOGM, PASS + 1>()) == inv_wl_after(LOGM, PASS),
                      "pass plan bookkeeping");
        if constexpr (wave_local_exchange(V0, G, V0 + G, inv_plan_g<LOGM, PASS + 1>()))
            wave_sync();
        else
            FHE_BARRIER();
        ntt_inv_lds<LOGM, T, PASS + 1, V0 + G, NARROW>(lds, itw, logn, sub, pm, tid, fold, ninv, zninv, next);
    } else {
        FHE_BARRIER();
    }
}

// ------------------------------------------------------------- tile load / store ----
// Thread t owns the 16-byte chunks {c*T + t}, c < CH, of the M-element tile (coalesced 16 B
// per lane).  CH > 0: all CH loads are issued before the first LDS write (one latency, not
// CH).  CH == 0: scalar strided loop for tiles smaller than 2*T (tiny test sizes).
template <int CH, int M, int T, class F>
__device__ __forceinline__ void tile_to_lds(u64 *lds, const u64 *__restrict__ src, uint32_t tid, F f) {
    if constexpr (CH > 0) {
        const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src);
        u64x2 v[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = s2[c * T + tid];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i = 2 * (c * T + tid);
            lds[padi(i)] = f(v[c].x);
            lds[padi(i + 1)] = f(v[c].y);
        }
    } else {
        for (uint32_t i = tid; i < M; i += T) lds[padi(i)] = f(src[i]);
    }
}
template <int CH, int M, int T, class F>
__device__ __forceinline__ void lds_to_tile(const u64 *lds, u64 *__restrict__ dst, uint32_t tid, F f) {
    if constexpr (CH > 0) {
        u64x2 *d2 = reinterpret_cast<u64x2 *>(dst);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i = 2 * (c * T + tid);
            u64x2 v;
            v.x = f(lds[padi(i)]);
            v.y = f(lds[padi(i + 1)]);
            d2[c * T + tid] = v;
        }
    } else {
        for (uint32_t i = tid; i < M; i += T) dst[i] = f(lds[padi(i)]);
    }
}

// ------------------------------------------------------------------- NTT kernel ----
// One workgroup (ntt_threads_c(LOGM) threads) per (row, sub-block).
// grid.x = npolys * map.rows * nsub, nsub = 2^(logn - LOGM).
// LOGM == logn: whole row in LDS (N <= 16384).  LOGM < logn: this is the LDS half of the
// two-kernel transform for N >= 32768 (ntt_global_kernel does the other logn-LOGM stages).
//   forward: canonical output (reduce3, native.rs:238-246)
//   inverse: multiplies by N^-1 (Shoup) when LOGM == logn (native.rs:229-232)
// Register budget 128 VGPRs = 4 waves/SIMD, which is what the LDS footprint allows anyway
// (N = 8192: 68 KiB/workgroup -> 2 workgroups of 8 waves per CU).
// NARROW (forward, whole row, every modulus of the launch below 2^60): see fwd_butterfly_narrow.
template <bool INVERSE, int LOGM, bool NARROW = false>
__global__ void __launch_bounds__(ntt_threads_c(LOGM), 4)
    ntt_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map, const DevMod *__restrict__ mods,
               const u64x2 *__restrict__ tw, const u64x2 *__restrict__ ninv, uint32_t logn, uint32_t prologue) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ntt_threads_c(LOGM);
    constexpr int M = 1 << LOGM;
    constexpr int CH = tile_chunks_c(LOGM, T);
    const uint32_t tid = threadIdx.x;
    const uint32_t n = 1u << logn;
    const uint32_t nsub = 1u << (logn - LOGM);
    const uint32_t sub = blockIdx.x & (nsub - 1);
    const uint32_t rowb = blockIdx.x >> (logn - LOGM);
    const uint32_t poly = to_sgpr(rowb / map.rows);
    const uint32_t r = map.row_begin + (rowb - poly * map.rows);
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64 *src = in + (u64)poly * map.src_poly_stride +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n + (u64)sub * M;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n + (u64)sub * M;
    const u64x2 *twr = tw + (u64)mi * n;

    if constexpr (!INVERSE) {
        // the first pass reads its groups straight from global memory (no tile staging)
        const bool red = prologue == PRO_REDUCE;
        ntt_fwd_lds<LOGM, T, GMAX, true, true, (NARROW ? 1 : 0)>(lds, twr, nsub + sub, pm, tid, [&](uint32_t i, uint32_t) {
            const u64 v = src[i];
            return red ? reduce_u64(v, md) : v;
        });
        if constexpr (NARROW) {  // < 16p -> canonical
            const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) {
                return csub_n(csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2), p, pm.np);
            });
        } else {
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); });
        }
    } else {
        InvTwFirst<LOGM, T> tw0;
        inv_tw_load(tw0, twr, logn, sub, tid);   // issued ahead of the tile loads: one latency for both
        const bool whole = logn == LOGM;  // ninv[2*mi] = {N^-1, shoup}, ninv[2*mi+1] = {z_last * N^-1, shoup}
        // (feeding the first pass straight from global memory, as the forward transform does, was
        // measured for the inverse: no gain -- its groups are runs of consecutive coefficients)
        if (prologue == PRO_REDUCE)
            tile_to_lds<CH, M, T>(lds, src, tid, [&](u64 v) { return reduce_u64(v, md); });
        else
            tile_to_lds<CH, M, T>(lds, src, tid, [](u64 v) { return v; });
        FHE_BARRIER();
        ntt_inv_lds<LOGM, T, 0, 0, NARROW>(lds, twr, logn, sub, pm, tid, whole, ninv[2 * mi], ninv[2 * mi + 1], tw0);
        if (whole)
            lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(v, p, pm.np); });
        else
            lds_to_tile<CH, M, T>(lds, dst, tid, [](u64 v) { return v; });  // < 2p, global pass finishes
    }
}

// (Persistent workgroups -- the rows of a launch dealt round-robin to two or four resident workgroups per CU, the next
// row's coefficients prefetched into registers across the epilogue -- were built and measured for the forward
// transform: 17.2-17.9 M against 24.8 M row-NTT/s (60-bit rows), 15.6-17.8 M against 19.8-20.8 M (62-bit): the 32
// prefetch registers on top of a radix-16 pass spill (36-124 B of scratch per lane) and the hardware dispatcher
// already overlaps one workgroup's loads with the other's arithmetic.  profiles/r02_ntt_persist_ab.txt.)

// ------------------------------------------------ fused tensor + inverse NTT ----
// The tensor step of Multiplicator::multiply (F/bfv/ops/mul.rs:198-201) fused into the loader of
// the inverse NTT that Scaler::scale applies next (M/rq/scaler.rs:69-79): the products
//   slot 0: c00*c10   slot 1: c00*c11 + c01*c10   slot 2: c01*c11
// are formed while the row is staged into LDS, so the Ntt-domain tensor never touches HBM.
// Operands: (c00, c01) = extL[b][0..1], (c10, c11) = extR[b][0..1], each [K][N]; rows below
// `ncommon` come straight from the input ciphertexts lhs/rhs [b][2][lrows][N] when given.
// grid = (K rows, nb ciphertext pairs, 3 slots); out is slot-major [3][nb][K][N] PowerBasis.
struct TensorSrc {
    const u64 *extL, *extR, *lhs, *rhs;
    uint32_t ncommon, lrows;
};
// SUB (rows larger than LDS, N = 2^logn > M): a workgroup handles one M-point sub-block -- tensor product
// in the loader, the inverse stages that stay inside the sub-block -- and leaves values below 2p for
// ntt_global_kernel<true, .>, which finishes the transform (so the Ntt-domain tensor never touches HBM
// at N = 32768 / 65536 either).
// A launch covers the rows [row_begin, row_begin + lrows) of the nrows-row extended basis; NARROW (all of them
// below 2^60) selects the inverse passes with tracked bounds (inv_pass): the ciphertext primes of the extended
// basis are 60-bit, the extension primes 62-bit, so bfv_mul issues one launch for each group.
template <int LOGM, bool SUB = false, bool NARROW = false>
__global__ void __launch_bounds__(ntt_threads_c(LOGM), 4)
    tensor_intt_kernel(TensorSrc ts, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                       const u64x2 *__restrict__ itw, const u64x2 *__restrict__ ninv, uint32_t nrows, uint32_t nb,
                       uint32_t logn_arg, uint32_t row_begin, uint32_t lrows) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ntt_threads_c(LOGM);
    constexpr int M = 1 << LOGM;
    constexpr int CH = tile_chunks_c(LOGM, T);
    const uint32_t tid = threadIdx.x;
    const uint32_t logn = SUB ? logn_arg : (uint32_t)LOGM;
    const uint32_t lsub = logn - LOGM;  // log2(sub-blocks per row); 0 unless SUB
    // XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (id mod 8), each with its
    // own L2.  The three slots of one (row, ciphertext) pair read the same four operand rows, so their
    // ids are 8 apart: same XCD, dispatched back to back, and the re-reads hit that L2 instead of HBM
    // (a (row, pair, slot) 3-D grid put them nb*K blocks apart: 1.8x the algorithmic HBM traffic).
    const uint32_t t8 = blockIdx.x >> 3, grp = t8 / 3, slot = t8 - 3 * grp;
    const uint32_t combo = grp * 8 + (blockIdx.x & 7);
    if (combo >= (lrows * nb) << lsub) return;  // (block-uniform) tail of the rounded-up grid
    const uint32_t sub = combo & ((1u << lsub) - 1), rowb = combo >> lsub;
    const uint32_t b = to_sgpr(rowb / lrows), r = row_begin + (rowb - b * lrows);
    const DevMod md = mods[r];
    const u64 p = md.p;
    const PM pm = make_pm(md);
    const u64 pk = (u64)nrows << logn;
    const u64 roff = ((u64)r << logn) + (u64)sub * M;  // this tile inside a polynomial
    const u64 *a0, *a1, *b0, *b1;  // rows of c00, c01, c10, c11
    if (ts.lhs && r < ts.ncommon) {
        const u64 pl = (u64)ts.lrows << logn;
        a0 = ts.lhs + (u64)b * 2 * pl + roff;
        a1 = a0 + pl;
        b0 = ts.rhs + (u64)b * 2 * pl + roff;
        b1 = b0 + pl;
    } else {
        a0 = ts.extL + (u64)b * 2 * pk + roff;
        a1 = a0 + pk;
        b0 = ts.extR + (u64)b * 2 * pk + roff;
        b1 = b0 + pk;
    }
    auto prod = [&](u64 x00, u64 x01, u64 x10, u64 x11) -> u64 {
        // (results stay below 2p: the inverse transform's first pass takes that range)
        if (slot == 0) return mul_mod_lazy(x00, x10, md);
        if (slot == 2) return mul_mod_lazy(x01, x11, md);
        // c1 = c00*c11 + c01*c10: one Barrett reduction of the 128-bit sum (< 2p^2 < 2^(2k+1): the
        // quotient estimate is then short by at most 3: below 4p before the conditional subtraction)
        u64 hi, lo;
        mac2_wide62(x00, x11, x01, x10, hi, lo);
        return barrett_reduce_wide_lazy(hi, lo, md);
    };
    // the first inverse pass's per-lane twiddles (56 VGPRs at N = 8192) are requested half-way through the products,
    // when half of the operand registers are free again: their L2 latency then hides behind the remaining
    // products and the barrier instead of following it (FHE_TENSOR_TW_EARLY=0: after the products, as before)
    const u64x2 *twr = itw + ((u64)r << logn);
    InvTwFirst<LOGM, T> tw0;
    if constexpr (CH > 0) {
        if (slot != 1) {
            // c0 = c00*c10 / c2 = c01*c11: two operand rows, all CH chunks of both in flight at once
            const u64x2 *pa = reinterpret_cast<const u64x2 *>(slot == 0 ? a0 : a1);
            const u64x2 *pb = reinterpret_cast<const u64x2 *>(slot == 0 ? b0 : b1);
            u64x2 va[CH], vb[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                va[c] = pa[c * T + tid];
                vb[c] = pb[c * T + tid];
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (FHE_TENSOR_TW_EARLY && CH > 1 && c == CH / 2) {
                    sched_fence();
                    inv_tw_load(tw0, twr, logn, sub, tid);
                }
                const uint32_t i = 2 * (c * T + tid);
                lds[padi(i)] = mul_mod_lazy(va[c].x, vb[c].x, md);
                lds[padi(i + 1)] = mul_mod_lazy(va[c].y, vb[c].y, md);
            }
        } else {
        constexpr int HALF = CH > 1 ? CH / 2 : 1;  // loads of at most HALF chunks x 4 operands in flight
#pragma unroll
        for (int h = 0; h < CH; h += HALF) {
            u64x2 v00[HALF], v01[HALF], v10[HALF], v11[HALF];
#pragma unroll
            for (int c = 0; c < HALF; c++) {
                const uint32_t ci = (h + c) * T + tid;
                if (slot != 2) v00[c] = reinterpret_cast<const u64x2 *>(a0)[ci];
                if (slot != 0) v01[c] = reinterpret_cast<const u64x2 *>(a1)[ci];
                if (slot != 2) v10[c] = reinterpret_cast<const u64x2 *>(b0)[ci];
                if (slot != 0) v11[c] = reinterpret_cast<const u64x2 *>(b1)[ci];
            }
#pragma unroll
            for (int c = 0; c < HALF; c++) {
                if (FHE_TENSOR_TW_EARLY && CH > 1 && h + HALF >= CH && c == HALF / 2) {   // (last batch, half done)
                    sched_fence();
                    inv_tw_load(tw0, twr, logn, sub, tid);
                }
                const uint32_t i = 2 * ((h + c) * T + tid);
                lds[padi(i)] = prod(v00[c].x, v01[c].x, v10[c].x, v11[c].x);
                lds[padi(i + 1)] = prod(v00[c].y, v01[c].y, v10[c].y, v11[c].y);
            }
            sched_fence();
        }
        }
    } else {
        for (uint32_t i = tid; i < M; i += T) lds[padi(i)] = prod(a0[i], a1[i], b0[i], b1[i]);
    }
    if (!(FHE_TENSOR_TW_EARLY && CH > 1)) inv_tw_load(tw0, twr, logn, sub, tid);   // in flight across the barrier
    FHE_BARRIER();
    u64 *dst = out + ((u64)slot * nb + b) * pk + roff;
    // (a block-uniform branch between the narrow and the general inverse passes inside one kernel was measured:
    // 128 VGPRs, spills and twice the code -- 2 % slower; hence one launch per row group)
    ntt_inv_lds<LOGM, T, 0, 0, NARROW>(lds, twr, logn, sub, pm, tid, !SUB, ninv[2 * r], ninv[2 * r + 1], tw0);
    if constexpr (SUB)
        lds_to_tile<CH, M, T>(lds, dst, tid, [](u64 v) { return v; });  // < 2p, the global pass finishes
    else
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(v, p, pm.np); });
}

// Radix stages that span sub-blocks, done straight on global memory (coalesced along the
// low index).  Forward: stages [0, G0) before the LDS kernel (output < 4p, the LDS kernel's
// loader accepts that range).  Inverse: stages [logm, logn) after it, then N^-1.
// One thread per group of 2^G0 elements {lo + e*M}; grid.x covers npolys*rows*(M/threads).
template <bool INVERSE, int G0>
__global__ void ntt_global_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map,
                                  const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw,
                                  const u64x2 *__restrict__ ninv, uint32_t logn, uint32_t prologue) {
    constexpr uint32_t R = 1u << G0;
    const uint32_t n = 1u << logn, logm = logn - G0, m = 1u << logm;
    const uint32_t chunks = (m + blockDim.x - 1) / blockDim.x;
    const uint32_t rowb = blockIdx.x / chunks;
    const uint32_t lo = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (lo >= m) return;
    const uint32_t poly = rowb / map.rows;
    const uint32_t r = map.row_begin + rowb % map.rows;
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p;
    const PM pm = make_pm(md);
    const u64 *src = in + (u64)poly * map.src_poly_stride +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * n;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * n;
    const u64x2 *twr = tw + (u64)mi * n;
    u64 x[R];
#pragma unroll
    for (uint32_t e = 0; e < R; e++) x[e] = src[lo + e * m];
    if (prologue == PRO_REDUCE) {
#pragma unroll
        for (uint32_t e = 0; e < R; e++) x[e] = reduce_u64(x[e], md);
    }
    if (!INVERSE) {
#pragma unroll
        for (int u = 0; u < G0; u++) {
            const uint32_t half = R >> (u + 1);
#pragma unroll
            for (uint32_t blk = 0; blk < (1u << u); blk++) {
                const u64x2 w = twr[(1u << u) + blk];
#pragma unroll
                for (uint32_t j = 0; j < half; j++) {
                    const uint32_t a = blk * 2 * half + j;
                    fwd_butterfly(x[a], x[a + half], w.x, w.y, pm);
                }
            }
        }
#pragma unroll
        for (uint32_t e = 0; e < R; e++) dst[lo + e * m] = x[e];
    } else {
#pragma unroll
        for (int u = 0; u < G0; u++) {
            const uint32_t v = logm + u;
            const uint32_t nblk = R >> (u + 1);
#pragma unroll
            for (uint32_t blk = 0; blk < nblk; blk++) {
                const u64x2 z = twr[n - (n >> v) + blk];
#pragma unroll
                for (uint32_t j = 0; j < (1u << u); j++) {
                    const uint32_t a = blk * (2u << u) + j;
                    inv_butterfly(x[a], x[a + (1u << u)], z.x, z.y, pm);
                }
            }
        }
        const u64x2 ni = ninv[2 * mi];
#pragma unroll
        for (uint32_t e = 0; e < R; e++) dst[lo + e * m] = mul_shoup(x[e], ni.x, ni.y, p);
    }
}

// ------------------------------------------------------------ fused key switch ----
// (c0, c1)[b][j] (+)= sum_i NTT_j( [p_i]_{q_j} ) (.) (k0, k1)[i][j]   for one (b, j) per workgroup.
// The digit rows p[b][i][:] are lifted (reduced mod q_j), transformed in LDS and multiplied
// into per-thread register accumulators; the key streams from L2/MALL (shared by the batch).
// A non-null addend0/addend1 is added to the respective output (fused relinearisation add,
// F/bfv/ops/mul.rs:224-225; rotation adds substitute(c0) to c0 only); canonical outputs.
// ks_threads_c(LOGN) threads; thread t owns the 16-byte chunks {c*T + t}, c < CH (or the single
// coefficient t when the row is smaller than one chunk per thread).
// (FHE_KS_TWPF=true: the transforms' per-lane twiddles requested one pass ahead at N = 8192 -- 112 VGPRs, no scratch,
// and no change in same-box A/B, profiles/r02_ks_twpf_ab.txt)
// (FHE_KS_PERSIST14=1: resident workgroups at N = 16384 as well -- no row-prefetch registers there, so nothing to
// overlap: C3 relinearise 108.1-109.7 k against 111.0-111.3 k ops/s, profiles/r02_ks_persist_ab.txt)
// TT (threads per workgroup, 0 = ks_threads_c(LOGN)): TT = N / 16 at N = 8192 is the two-workgroups-per-CU cut -- 512
// threads x 16 coefficients, BOTH accumulator sets in registers (64 VGPRs, as at N = 16384), only the 68 KiB row tile
// in LDS, so that a second workgroup is resident and runs its butterflies while this one sits in a barrier.
constexpr int ks_threads_tt(int logn, int tt) { return tt ? tt : ks_threads_c(logn); }
constexpr bool ks_acc1_in_lds_tt(int logn, int tt) { return tt == 0 && ks_acc1_in_lds_c(logn); }
template <int LOGN, bool NARROW = false, int GM = KS_GMAX, int TT = 0>
__global__ void __launch_bounds__(ks_threads_tt(LOGN, TT), 4)
    ks_fused_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0, u64 *__restrict__ out1,
                    u64 out_poly_stride, const u64 *__restrict__ addend0, const u64 *__restrict__ addend1,
                    u64 addend_poly_stride, const u64 *__restrict__ k0, const u64 *__restrict__ k0s,
                    const u64 *__restrict__ k1, const u64 *__restrict__ k1s, const DevMod *__restrict__ mods,
                    const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t lk, uint32_t digit_arg,
                    const u64 *__restrict__ xhat, u64 xhat_poly_stride, uint32_t total) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ks_threads_tt(LOGN, TT);
    constexpr int N = 1 << LOGN;
    constexpr int CH = tile_chunks_c(LOGN, T);
    constexpr int NE = CH > 0 ? 2 * CH : 1;  // coefficients owned by a thread
    // GM: radix (log2) of the LDS passes.  8 everywhere but N = 16384, whose 1024 threads hold both accumulator sets in
    // registers (64 VGPRs): beside a per-lane-twiddle radix-8 pass that spills 24 VGPRs (100 B of scratch per lane,
    // 5 GB of extra HBM traffic per 512-polynomial launch, PMC).  GM_MIXED keeps radix 8 for the passes whose
    // twiddles are scalar and takes radix 4 for the rest (3+3+2+2+2+2 stages, 124 VGPRs, no scratch): C3 relinearise
    // 99.2 k (radix 8) -> 110.5 k (radix 4 throughout) -> 111.7-112.9 k ops/s.
    const uint32_t tid0 = threadIdx.x;
    // (an XCD-aware order that puts the lk workgroups of one polynomial on one L2, as tensor_intt_kernel
    // does, was measured: no change -- this kernel is nowhere near the HBM limit)
    // The (ciphertext, key modulus) items of a launch are dealt round-robin to the gridDim.x workgroups (`total` of
    // them; the host launches one workgroup per item except at N = 8192, where a workgroup owns its CU: there
    // gridDim.x is the number of CUs and, while an item's result is on its way out, the next item's first digit row
    // is already coming in -- neither that load nor a workgroup launch sits between two items).
    u64x2 pre[ks_acc1_in_lds_tt(LOGN, TT) ? CH : 1];
    bool have_pre = false;   // (block-uniform) `pre` already holds this item's first digit row
#if defined(FHE_HOST_EMULATION)
    constexpr bool ITEM_LOOP = true;    // (every size, so that the emulated suite walks the loop)
#else
    constexpr bool ITEM_LOOP = (LOGN == 13 && TT == 0) || (FHE_KS_PERSIST14 && LOGN == 14);
#endif
    uint32_t item = blockIdx.x;
    if (item >= total) return;
    do {
    const uint32_t b = to_sgpr(item / lk), j = item - b * lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64x2 *twr = tw + (u64)j * N;
    // N = 8192: 1024 threads cap a thread at 128 VGPRs, which 2 x 16 accumulators plus a radix-8
    // pass do not fit; the c1 accumulators live in LDS behind the row tile instead (each thread
    // only ever touches its own 16-byte chunks, so no extra barrier).
    constexpr bool ACC1_LDS = ks_acc1_in_lds_tt(LOGN, TT);
    u64 acc0[NE], acc1[ACC1_LDS ? 1 : NE];
    u64x2 *const acc1_lds = reinterpret_cast<u64x2 *>(lds + lds_words(N));
#pragma unroll
    for (int e = 0; e < NE; e++) acc0[e] = 0;
    if constexpr (ACC1_LDS) {
#pragma unroll
        for (int c = 0; c < CH; c++) acc1_lds[c * T + tid0] = u64x2{0, 0};
    } else {
#pragma unroll
        for (int e = 0; e < NE; e++) acc1[e] = 0;
    }
    // `digit_arg` = digit_shift_bits | lift_mode << 8.  lift_mode says how far a source residue can exceed
    // the key moduli (host-side, from the moduli): 1 -> below 2 q_j (one conditional subtraction lifts
    // it), 2 -> below 4 q_j (two), 0 -> anything (Barrett).  RNS digits of same-width moduli are mode 1.
    const uint32_t digit_shift_bits = digit_arg & 0xff, lift_mode = digit_arg >> 8;
    auto lift = [&](u64 v) -> u64 {
        if (lift_mode == 1) return csub_n(v, p, pm.np);
        if (lift_mode == 2) return csub_n(csub_n(v, p2, pm.np2), p, pm.np);
        return reduce_u64(v, md);
    };
    // digit_shift_bits == 0: digit i is residue row i of p (RNS decomposition, :256-268).
    // otherwise: base-2^bits digits of the single row 0 (key_switch_decomposition, :323-362).
    const u64 *const src0 = pin + (u64)b * src_poly_stride;
    const u64 dstride = digit_shift_bits ? 0 : (u64)N;
    const u64 mask = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
    // `xhat` (callers that hold the digit polynomial in Ntt form -- relinearise, Galois, RGSW: [digits][N] per
    // polynomial over the ciphertext moduli, canonical): the RNS digit j reduced mod q_j is row j itself and its
    // transform under key modulus j IS xhat's row j (the ciphertext moduli are a prefix of the key moduli), so that
    // one of the L transforms of this workgroup is not computed: its product initialises the accumulators.
    const bool own = xhat != nullptr && digit_shift_bits == 0 && j < ndigits;   // (block-uniform)
    if (own) {
        const u64 koff = ((u64)j * lk + j) * N;
        const u64 *xr = xhat + (u64)b * xhat_poly_stride + (u64)j * N;
        if constexpr (CH > 0) {
            const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
            const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t ci = c * T + tid0;
                const u64x2 v = reinterpret_cast<const u64x2 *>(xr)[ci];
                const u64x2 q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
                acc0[2 * c] = mul_shoup_lazy_n(v.x, q0.x, q0s.x, pm.np);        // below 2p, like every accumulator value
                acc0[2 * c + 1] = mul_shoup_lazy_n(v.y, q0.y, q0s.y, pm.np);
                const u64x2 a{mul_shoup_lazy_n(v.x, q1.x, q1s.x, pm.np), mul_shoup_lazy_n(v.y, q1.y, q1s.y, pm.np)};
                if constexpr (ACC1_LDS) {
                    acc1_lds[ci] = a;
                } else {
                    acc1[ACC1_LDS ? 0 : 2 * c] = a.x;
                    acc1[ACC1_LDS ? 0 : 2 * c + 1] = a.y;
                }
            }
        } else if (tid0 < N) {
            const u64 v = xr[tid0];
            acc0[0] = mul_shoup_lazy_n(v, k0[koff + tid0], k0s[koff + tid0], pm.np);
            acc1[0] = mul_shoup_lazy_n(v, k1[koff + tid0], k1s[koff + tid0], pm.np);
        }
    }
    const uint32_t nloop = ndigits - (own ? 1u : 0u);          // digits that go through the transform
    auto digit_of = [&](uint32_t ii) -> uint32_t { return ii + ((own && ii >= j) ? 1u : 0u); };
    // The workgroup is alone on its CU (LDS), so nothing else hides the row load: digit i+1's
    // row is fetched into registers while digit i goes through its passes.
    constexpr bool PREFETCH = ks_acc1_in_lds_tt(LOGN, TT);   // (needs the VGPRs the LDS accumulators free)
    // (Feeding the first pass from these registers instead of staging the lifted row in LDS was
    // measured: 2.5 % slower -- the extra register shuffling outweighs the saved barrier.)
    if constexpr (PREFETCH) {
        if (nloop > 0 && !have_pre) {
            const u64x2 *first = reinterpret_cast<const u64x2 *>(src0 + (u64)digit_of(0) * dstride);
#pragma unroll
            for (int c = 0; c < CH; c++) pre[c] = first[c * T + tid0];
        }
    }
    for (uint32_t ii = 0; ii < nloop; ii++) {
        const uint32_t i = digit_of(ii);
        const uint32_t tid = opaque(tid0);
        const uint32_t sh = i * digit_shift_bits;
        FHE_TS(0);
        if constexpr (PREFETCH) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t e = 2 * (c * T + tid);
                lds[padi(e)] = lift((pre[c].x >> sh) & mask);
                lds[padi(e + 1)] = lift((pre[c].y >> sh) & mask);
            }
        } else {
            // (address and mask are recomputed per digit on purpose: hoisted, they cost VGPRs that the
            // N = 16384 variant does not have)
            const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * N);
            const u64 mask_i = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
            tile_to_lds<CH, N, T>(lds, src, tid, [&](u64 v) { return lift((v >> sh) & mask_i); });
        }
        FHE_TS(1);
        FHE_BARRIER();
        FHE_TS(2);
        if constexpr (PREFETCH) {
            if (ii + 1 < nloop) {
                const u64x2 *nx = reinterpret_cast<const u64x2 *>(src0 + (u64)digit_of(ii + 1) * dstride);
#pragma unroll
                for (int c = 0; c < CH; c++) pre[c] = nx[c * T + tid];
            }
        }
        const u64 koff = ((u64)i * lk + j) * N;
        if constexpr (CH > 0) {
            const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
            const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
            // KPF: the key words of the first two chunks are requested before the barrier that ends the
            // transform, so their L2 latency is spent waiting for the other waves, not after them
            constexpr bool KPF = PREFETCH && CH >= 2;
            // (twiddle prefetch measured: no gain here; NARROW: values < 16p on exit, fine for the Shoup MAC)
            ntt_fwd_lds<LOGN, T, GM, FHE_KS_TWPF && LOGN == 13, !KPF, (NARROW ? 1 : 0), NoSrc, KS_LATE>(lds, twr, 1, pm, tid);
            // (all four chunks prefetched -- 118 VGPRs, no scratch: no change; three: 2 % slower.  ABBA runs in
            // profiles/r02_ks_kpf_ab.txt: the key words' latency is not what the MAC waits for)
            constexpr int KPFN = KPF ? (FHE_KS_KPF_CHUNKS < CH ? FHE_KS_KPF_CHUNKS : CH) : 0;   // chunks whose key words are prefetched
            u64x2 kq[KPF ? 4 * KPFN : 1];
            if constexpr (KPF) {
#pragma unroll
                for (int c = 0; c < KPFN; c++) {
                    const uint32_t ci = c * T + tid;
                    kq[4 * c] = a0[ci], kq[4 * c + 1] = a0s[ci], kq[4 * c + 2] = a1[ci], kq[4 * c + 3] = a1s[ci];
                }
                FHE_TS(3);
                FHE_BARRIER();
                FHE_TS(4);
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t ci = c * T + tid;
                u64x2 q0, q0s, q1, q1s;
                if (KPF && c < KPFN) {
                    q0 = kq[KPF ? 4 * c : 0], q0s = kq[KPF ? 4 * c + 1 : 0], q1 = kq[KPF ? 4 * c + 2 : 0], q1s = kq[KPF ? 4 * c + 3 : 0];
                } else {
                    q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
                }
                const u64 vx = lds[padi(2 * ci)], vy = lds[padi(2 * ci + 1)];  // < 4p: Shoup accepts any u64
                acc0[2 * c] = csub_n(acc0[2 * c] + mul_shoup_lazy_n(vx, q0.x, q0s.x, pm.np), p2, pm.np2);
                acc0[2 * c + 1] = csub_n(acc0[2 * c + 1] + mul_shoup_lazy_n(vy, q0.y, q0s.y, pm.np), p2, pm.np2);
                if constexpr (ACC1_LDS) {
                    u64x2 a = acc1_lds[ci];
                    a.x = csub_n(a.x + mul_shoup_lazy_n(vx, q1.x, q1s.x, pm.np), p2, pm.np2);
                    a.y = csub_n(a.y + mul_shoup_lazy_n(vy, q1.y, q1s.y, pm.np), p2, pm.np2);
                    acc1_lds[ci] = a;
                } else {
                    acc1[2 * c] = csub_n(acc1[2 * c] + mul_shoup_lazy_n(vx, q1.x, q1s.x, pm.np), p2, pm.np2);
                    acc1[2 * c + 1] = csub_n(acc1[2 * c + 1] + mul_shoup_lazy_n(vy, q1.y, q1s.y, pm.np), p2, pm.np2);
                }
                if (c & 1) sched_fence();  // at most two chunks of key loads (32 VGPRs) in flight
            }
        } else {
            ntt_fwd_lds<LOGN, T, GM, false, true, (NARROW ? 1 : 0), NoSrc, KS_LATE>(lds, twr, 1, pm, tid);
            if (tid < N) {
            const u64 v = lds[padi(tid)];
            acc0[0] = csub_n(acc0[0] + mul_shoup_lazy_n(v, k0[koff + tid], k0s[koff + tid], pm.np), p2, pm.np2);
            acc1[0] = csub_n(acc1[0] + mul_shoup_lazy_n(v, k1[koff + tid], k1s[koff + tid], pm.np), p2, pm.np2);
            }
        }
        // (wave-contiguous chunk ownership, which makes this barrier and the one before the MAC wave-local as
        // well, was measured: nothing beyond what the late pass plan already gives)
        FHE_TS(5);
        FHE_BARRIER();
        FHE_TS(6);
    }
    // (FHE_DEBUG_KS_NOMEM -- every polynomial aliased to the first: rows, addends and outputs out of L2 -- makes this
    // kernel 11 % faster at C2: what its one workgroup per CU cannot hide.  Requesting the addends during the last
    // digit, into the row-prefetch registers that are idle then, was built: those 16 registers stay live through the
    // last MAC and spill (84-164 B of scratch); not kept.)
    const uint32_t tid = opaque(tid0);  // keeps the epilogue's address arithmetic below the digit loop
    have_pre = false;
    if constexpr (PREFETCH && ITEM_LOOP) {
        const uint32_t nitem = item + gridDim.x;
        if (nitem < total) {   // the next item's first digit row (the same selection as at the top of the loop)
            const uint32_t nb = nitem / lk, nj = nitem - nb * lk;
            const bool nown = xhat != nullptr && digit_shift_bits == 0 && nj < ndigits;
            if (ndigits - (nown ? 1u : 0u) > 0) {
                const uint32_t nd = (nown && nj == 0) ? 1u : 0u;
                const u64x2 *first = reinterpret_cast<const u64x2 *>(pin + (u64)nb * src_poly_stride + (u64)nd * dstride);
#pragma unroll
                for (int c = 0; c < CH; c++) pre[c] = first[c * T + tid];
                have_pre = true;
            }
        }
    }
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * N;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * N;
    if constexpr (CH > 0) {
        u64x2 *o0 = reinterpret_cast<u64x2 *>(out0 + ooff), *o1 = reinterpret_cast<u64x2 *>(out1 + ooff);
        const u64x2 *d0 = reinterpret_cast<const u64x2 *>(addend0 ? addend0 + aoff : nullptr);
        const u64x2 *d1 = reinterpret_cast<const u64x2 *>(addend1 ? addend1 + aoff : nullptr);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            u64x2 r0, r1;
            r0.x = csub_n(acc0[2 * c], p, pm.np);
            r0.y = csub_n(acc0[2 * c + 1], p, pm.np);
            const u64x2 a1 = ACC1_LDS ? acc1_lds[ci] : u64x2{acc1[ACC1_LDS ? 0 : 2 * c], acc1[ACC1_LDS ? 0 : 2 * c + 1]};
            r1.x = csub_n(a1.x, p, pm.np);
            r1.y = csub_n(a1.y, p, pm.np);
            if (d0) {
                const u64x2 a = d0[ci];
                r0.x = add_mod_n(r0.x, a.x, pm);
                r0.y = add_mod_n(r0.y, a.y, pm);
            }
            if (d1) {
                const u64x2 a = d1[ci];
                r1.x = add_mod_n(r1.x, a.x, pm);
                r1.y = add_mod_n(r1.y, a.y, pm);
            }
            o0[ci] = r0;
            o1[ci] = r1;
        }
    } else if (tid < N) {
        u64 r0 = csub(acc0[0], p), r1 = csub(acc1[0], p);
        if (addend0) r0 = add_mod(r0, addend0[aoff + tid], p);
        if (addend1) r1 = add_mod(r1, addend1[aoff + tid], p);
        out0[ooff + tid] = r0;
        out1[ooff + tid] = r1;
    }
    } while (ITEM_LOOP && (item += gridDim.x) < total);
}

// The same for rows that do not fit LDS (N = 2^(13+G0) >= 32768): one workgroup per (ciphertext,
// key modulus j, 8192-point sub-block).  The first G0 Cooley-Tukey stages (native.rs:142-175,
// blocks larger than the tile) are folded into the loader: coefficient e of sub-block `sub`
// depends on the 2^G0 source coefficients e + k*8192 through G0 butterflies of which only the
// branch leading to `sub` is evaluated (2^G0 - 1 Shoup multiplications per coefficient instead
// of G0/2 amortised, but no round trip of the lifted row through HBM); the remaining 13 stages
// run in LDS with twiddle base 2^G0 + sub, exactly like ntt_kernel's sub-block mode.
template <int G0, int LOGM = 13, bool NARROW = false>
__global__ void __launch_bounds__((1 << LOGM) / 8, 4)
    ks_fused_split_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0,
                          u64 *__restrict__ out1, u64 out_poly_stride, const u64 *__restrict__ addend0,
                          const u64 *__restrict__ addend1, u64 addend_poly_stride, const u64 *__restrict__ k0,
                          const u64 *__restrict__ k0s, const u64 *__restrict__ k1, const u64 *__restrict__ k1s,
                          const DevMod *__restrict__ mods, const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t lk,
                          uint32_t digit_arg, const u64 *__restrict__ xhat, u64 xhat_poly_stride) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int M = 1 << LOGM, T = M / 8, CH = M / (2 * T), NS = 1 << G0;
    constexpr u64 N = (u64)M << G0;
    const uint32_t tid0 = threadIdx.x;
    const uint32_t sub = blockIdx.x & (NS - 1);
    const uint32_t bj = blockIdx.x >> G0;
    const uint32_t b = to_sgpr(bj / lk), j = bj - b * lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64x2 *twr = tw + (u64)j * N;
    const uint32_t digit_shift_bits = digit_arg & 0xff, lift_mode = digit_arg >> 8;  // see ks_fused_kernel
    auto lift = [&](u64 v) -> u64 {
        if (lift_mode == 1) return csub_n(v, p, pm.np);
        if (lift_mode == 2) return csub_n(csub_n(v, p2, pm.np2), p, pm.np);
        return reduce_u64(v, md);
    };
    u64 acc0[2 * CH];
    u64x2 *const acc1_lds = reinterpret_cast<u64x2 *>(lds + lds_words(M));
#pragma unroll
    for (int e = 0; e < 2 * CH; e++) acc0[e] = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc1_lds[c * T + tid0] = u64x2{0, 0};
    // (see ks_fused_kernel: digit j under key modulus j is the caller's Ntt-form row j -- no transform)
    const bool own = xhat != nullptr && digit_shift_bits == 0 && j < ndigits;
    if (own) {
        const u64 koff = ((u64)j * lk + j) * N + (u64)sub * M;
        const u64x2 *xr = reinterpret_cast<const u64x2 *>(xhat + (u64)b * xhat_poly_stride + (u64)j * N + (u64)sub * M);
        const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
        const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid0;
            const u64x2 v = xr[ci], q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
            acc0[2 * c] = mul_shoup_lazy_n(v.x, q0.x, q0s.x, pm.np);
            acc0[2 * c + 1] = mul_shoup_lazy_n(v.y, q0.y, q0s.y, pm.np);
            acc1_lds[ci] = u64x2{mul_shoup_lazy_n(v.x, q1.x, q1s.x, pm.np), mul_shoup_lazy_n(v.y, q1.y, q1s.y, pm.np)};
        }
    }
    const uint32_t nloop = ndigits - (own ? 1u : 0u);
    for (uint32_t ii = 0; ii < nloop; ii++) {
        const uint32_t i = ii + ((own && ii >= j) ? 1u : 0u);
        const uint32_t tid = opaque(tid0);
        const u64 *src = pin + (u64)b * src_poly_stride + (digit_shift_bits ? 0 : (u64)i * N);
        const uint32_t sh = i * digit_shift_bits;
        const u64 mask = digit_shift_bits ? ((1ull << digit_shift_bits) - 1) : ~0ull;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            u64x2 v[NS];
#pragma unroll
            for (int k = 0; k < NS; k++) v[k] = reinterpret_cast<const u64x2 *>(src + (u64)k * M)[ci];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                v[k].x = lift((v[k].x >> sh) & mask);
                v[k].y = lift((v[k].y >> sh) & mask);
            }
            // stage s keeps the half of the pairs whose output leads to `sub`
#pragma unroll
            for (int st = 0; st < G0; st++) {
                const int half = NS >> (st + 1);
                const u64x2 w = twr[(1u << st) + (sub >> (G0 - st))];
                const bool minus = (sub >> (G0 - st - 1)) & 1;
#pragma unroll
                for (int m = 0; m < half; m++) {
                    const u64 lx = csub_n(v[m].x, p2, pm.np2), ly = csub_n(v[m].y, p2, pm.np2);
                    const u64 tx = mul_shoup_lazy_n(v[m + half].x, w.x, w.y, pm.np);
                    const u64 ty = mul_shoup_lazy_n(v[m + half].y, w.x, w.y, pm.np);
                    v[m].x = minus ? lx + p2 - tx : lx + tx;
                    v[m].y = minus ? ly + p2 - ty : ly + ty;
                }
            }
            lds[padi(2 * ci)] = v[0].x;
            lds[padi(2 * ci + 1)] = v[0].y;
        }
        FHE_BARRIER();
        // (NARROW: the folded loader stages leave values below 4p)
        ntt_fwd_lds<LOGM, T, KS_GMAX, false, true, (NARROW ? 4 : 0), NoSrc, KS_LATE>(lds, twr, NS + sub, pm, tid);
        const u64 koff = ((u64)i * lk + j) * N + (u64)sub * M;
        const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
        const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t ci = c * T + tid;
            const u64x2 q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
            const u64 vx = lds[padi(2 * ci)], vy = lds[padi(2 * ci + 1)];
            acc0[2 * c] = csub_n(acc0[2 * c] + mul_shoup_lazy_n(vx, q0.x, q0s.x, pm.np), p2, pm.np2);
            acc0[2 * c + 1] = csub_n(acc0[2 * c + 1] + mul_shoup_lazy_n(vy, q0.y, q0s.y, pm.np), p2, pm.np2);
            u64x2 a = acc1_lds[ci];
            a.x = csub_n(a.x + mul_shoup_lazy_n(vx, q1.x, q1s.x, pm.np), p2, pm.np2);
            a.y = csub_n(a.y + mul_shoup_lazy_n(vy, q1.y, q1s.y, pm.np), p2, pm.np2);
            acc1_lds[ci] = a;
            if (c & 1) sched_fence();
        }
        FHE_BARRIER();
    }
    const uint32_t tid = opaque(tid0);
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * N + (u64)sub * M;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * N + (u64)sub * M;
    u64x2 *o0 = reinterpret_cast<u64x2 *>(out0 + ooff), *o1 = reinterpret_cast<u64x2 *>(out1 + ooff);
    const u64x2 *d0 = reinterpret_cast<const u64x2 *>(addend0 ? addend0 + aoff : nullptr);
    const u64x2 *d1 = reinterpret_cast<const u64x2 *>(addend1 ? addend1 + aoff : nullptr);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t ci = c * T + tid;
        u64x2 r0, r1;
        r0.x = csub_n(acc0[2 * c], p, pm.np);
        r0.y = csub_n(acc0[2 * c + 1], p, pm.np);
        const u64x2 a1v = acc1_lds[ci];
        r1.x = csub_n(a1v.x, p, pm.np);
        r1.y = csub_n(a1v.y, p, pm.np);
        if (d0) {
            const u64x2 a = d0[ci];
            r0.x = add_mod_n(r0.x, a.x, pm);
            r0.y = add_mod_n(r0.y, a.y, pm);
        }
        if (d1) {
            const u64x2 a = d1[ci];
            r1.x = add_mod_n(r1.x, a.x, pm);
            r1.y = add_mod_n(r1.y, a.y, pm);
        }
        o0[ci] = r0;
        o1[ci] = r1;
    }
}

// ------------------------------------------------------------------ RNS scaler ----
struct ScalerDev {
    const u64 *gamma_neg;                                // [nto]      (q - gamma) mod q
    const u64 *omega;                                    // [nto][nfrom]
    const u64 *vhi_tab;                                  // [nto][16]  k * 2^64 * gamma_neg mod q
    const u64 *c64_tab;                                  // [nto][16]  k * 2^64  mod q
    const u64 *c128_tab;                                 // [nto][16]  k * 2^128 mod q
    const u64 *theta_omega_lo, *theta_omega_hi;          // [nfrom]
    const u64 *theta_omega_sign;                         // [nfrom] (0/1)
    const u64 *theta_omega_mask;                         // [nfrom] 0 (term added) or ~0 (term subtracted)
    u64 w_const[4];                                      // the constant the one-accumulator form of w subtracts (below)
    const u64 *theta_garner_lo, *theta_garner_hi;        // [nfrom]
    u64 theta_gamma_lo, theta_gamma_hi;
    u64 narrow_mask;  // bit j: the output sum for target modulus j provably stays below 2^(2k_j+1) (see scaler_upload)
    u64 fold_mask;    // bit j: it stays below 2^(2k_j+6): bits >= 2^(2k_j) are folded through fold_tab first
    const u64 *fold_tab;                                 // [nto][64]  i * 2^(2k_j) mod q_j
    uint32_t theta_gamma_sign, is_one, shift, nfrom, nto, ncommon;
    uint32_t v_fits_64;  // v < 2^64 for every input (factor-one scalers over few moduli): no v_hi term
};

// Sum of 64x64-bit products on the device: the four 32x32 partial products of a term go straight
// into three 64-bit column accumulators (weights 2^0, 2^32, 2^64) THROUGH v_mad_u64_u32's addend,
// and each accumulator's carry-out -- which the compiler never uses -- is banked in a 32-bit
// overflow counter by a v_addc.  8 VALU instructions per term and no register shuffling, against
// 14 for the 128-bit formulation below (the multiply needs zero-extended register pairs there).
// The hazard recognizer does not see inside asm: a VALU-written SGPR needs two wait states
// before a VALU reads it as carry-in; the instruction order below provides them.
struct Acc3x64 {
    u64 c0 = 0, c1 = 0, c2 = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0;
};
// x: per-lane value; y: WAVE-UNIFORM constant (scaler tables): its halves are SGPR operands of the multiplies
// (one constant-bus read per instruction), which saves the two copies into VGPRs a "v" constraint costs per term.
FHE_HD void mac3x64(Acc3x64 &a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    u64 s0, s1, s2;  // carry-outs (SGPR pairs)
    asm("v_mad_u64_u32 %[c0], %[s0], %[xl], %[yl], %[c0]\n\t"
        "v_mad_u64_u32 %[c1], %[s1], %[xl], %[yh], %[c1]\n\t"
        "v_mad_u64_u32 %[c2], %[s2], %[xh], %[yh], %[c2]\n\t"
        "v_addc_co_u32 %[o0], vcc, 0, %[o0], %[s0]\n\t"
        "v_mad_u64_u32 %[c1], %[s0], %[xh], %[yl], %[c1]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s1]\n\t"
        "v_addc_co_u32 %[o2], vcc, 0, %[o2], %[s2]\n\t"
        "v_addc_co_u32 %[o1], vcc, 0, %[o1], %[s0]"
        : [c0] "+v"(a.c0), [c1] "+v"(a.c1), [c2] "+v"(a.c2), [o0] "+v"(a.o0), [o1] "+v"(a.o1), [o2] "+v"(a.o2),
          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
        : [xl] "v"(xl), [xh] "v"(xh), [yl] "s"(yl), [yh] "s"(yh)   // y: the wave-uniform constant, straight from SGPRs
        : "vcc");
#else  // host pass / host emulation: the same columns in plain C
    const u64 xl = (uint32_t)x, xh = x >> 32, yl = (uint32_t)y, yh = y >> 32;
    const u64 pr[4] = {xl * yl, xl * yh, xh * yh, xh * yl};
    u64 *const cs[4] = {&a.c0, &a.c1, &a.c2, &a.c1};
    uint32_t *const os[4] = {&a.o0, &a.o1, &a.o2, &a.o1};
    for (int k = 0; k < 4; k++) {
        const u64 t = *cs[k] + pr[k];
        *os[k] += t < *cs[k];
        *cs[k] = t;
    }
#endif
}
// value = (c0 + o0 2^64) + (c1 + o1 2^64) 2^32 + (c2 + o2 2^64) 2^64  ->  low 128 bits and the rest
FHE_HD void acc3x64_resolve(const Acc3x64 &a, u64 extra, u128_t &low, u64 &top) {
    const u128_t l = (u128_t)a.c0 + ((u128_t)a.c1 << 32) + extra;                     // < 2^98
    const u128_t m = (u128_t)a.c2 + a.o0 + ((u128_t)a.o1 << 32) + (l >> 64);         // weight 2^64, < 2^67
    low = (u128_t)(u64)l | (m << 64);
    top = (u64)(m >> 64) + a.o2;
}

// Sum of 64x64-bit products without carry detection: the low and the high 64-bit halves of the
// products are summed separately (each sum of up to 2^32 terms fits 96 bits, so a plain
// zero-extending 128-bit add never overflows and the compiler emits one add/addc chain, no
// compares); value = lo + (hi << 64), resolved once at the end.
struct Acc192 {
    u128_t lo = 0, hi = 0;
};
FHE_HD void mac192(Acc192 &acc, u64 a, u64 b) {
    const u128_t p = (u128_t)a * b;
    acc.lo += (u64)p;
    acc.hi += (u64)(p >> 64);
}
// -> low 128 bits and the bits above them (`top`)
FHE_HD void acc192_resolve(const Acc192 &acc, u128_t &low, u64 &top) {
    const u128_t mid = acc.hi + (acc.lo >> 64);
    low = (u128_t)(u64)acc.lo | (mid << 64);
    top = (u64)(mid >> 64);
}

// One lane per coefficient column (RnsScaler::scale, M/rns/scaler.rs:249-352).  The 256-bit
// fixed-point sums v and w are reproduced limb for limb (they define the rounding).  The
// per-target value y = -v*gamma (+/- w) + sum_j r_j*omega_j only matters mod q (the reference
// ends with reduce_u128), so instead of one Shoup product per term it is accumulated as
// exact 128-bit products in a 192-bit register and reduced ONCE (4 instead of 10 32-bit
// multiplies per term); the few bits of v, w and of the accumulator above 2^64 / 2^128 are
// folded through 16-entry tables (v, |w| < 2^68 and top < 16 for up to 64 source moduli).
// in: [npolys][nfrom][N] PowerBasis; out: rows [ncommon, nto) of [npolys][nto][N].
// NF >= nfrom: the column's residues are loaded once, together, into registers (coalesced
// along N; one batch of loads in flight); all scaler constants are wave-uniform scalar loads.
template <int NF>
__global__ void __launch_bounds__(256, NF <= 4 ? 8 : 1)   // (NF <= 4: 64 VGPRs / 8 waves per SIMD measured 3 % faster)
    scale_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                             u64 out_poly_stride, ScalerDev s, const DevMod *__restrict__ to_mods, uint32_t logn,
                             u64 total) {
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    // Columns are handed out from the LAST polynomial backwards: the kernel that wrote `in` (an inverse NTT, the
    // fused tensor kernel) went through the polynomials in ascending order, so its most recent output is what
    // still sits in the 256 MiB Infinity Cache; and the forward NTT that follows this kernel (ascending again)
    // starts on what was written here last.  Same-box A/B: -1.2 % per ct x ct step, -3 % on that forward NTT.
    gid = total - 1 - gid;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *src = in + poly * in_poly_stride + col;
    u64 rests[NF];
#pragma unroll
    for (int i = 0; i < NF; i++) rests[i] = (uint32_t)i < s.nfrom ? src[(u64)i * n] : 0;

    // (all per-source tables are zero-padded to NF entries by the host, scaler_upload: the term loops run without
    // per-term bounds checks -- a padded term multiplies a zero residue by a zero constant -- so the constants of
    // a sum are fetched together, one scalar wait per sum instead of one per term)
    Cols5 vc;
#pragma unroll
    for (int i = 0; i < NF; i++) cols5_mac_64x128(vc, rests[i], s.theta_garner_lo[i], s.theta_garner_hi[i]);
    const U256 sum = cols_resolve(cols5_to_cols256(vc));
    u64 vlo, vhi;
    u256_shr_lo128(sum, s.shift - 1, vlo, vhi);
    {  // v = div_ceil(v, 2)
        const u64 odd = vlo & 1;
        vlo = (vlo >> 1) | (vhi << 63);
        vhi >>= 1;
        vlo += odd;
        vhi += (vlo < odd);
    }
    u64 wlo = 0, whi = 0;
    bool w_sign = false;
    if (!s.is_one) {
        // t = sum_i +/- r_i * theta_omega_i  -/+  v * theta_gamma  (mod 2^256, scaler.rs:278-301).  ONE accumulator:
        // a subtracted term  -x * theta  is written  (~x) * theta - (2^64 - 1) * theta  (mod 2^256), so every term is
        // an addition of (x ^ mask) * theta with a wave-uniform mask of 0 or ~0, and the constants
        // (2^64 - 1) * theta of the subtracted terms are one 256-bit constant the host summed (ScalerDev::w_const).
        // Round 3: the two-accumulator form (added and subtracted terms summed separately) chose its accumulator by a
        // uniform branch per term, and every merge of the two paths cost a copy of the ten accumulator registers.
        Cols5 acc5;
#pragma unroll
        for (int i = 0; i < NF; i++) {
                // theta_omega_i = 0 whenever the scaled Garner coefficient is an integer -- e.g. every
                // source modulus outside the denominator when scaling Q*P -> Q by t/Q (5 of C2's 9)
                const u64 tlo = s.theta_omega_lo[i], thi = s.theta_omega_hi[i];
                if ((tlo | thi) == 0) continue;
                cols5_mac_64x128(acc5, rests[i] ^ s.theta_omega_mask[i], tlo, thi);
            }
        // v * theta_gamma (128 x 128 -> 256 wrapping): low word of v, then (high word) << 64; subtracted unless
        // theta_gamma_sign
        const u64 gmask = s.theta_gamma_sign ? 0ull : ~0ull;
        cols5_mac_64x128(acc5, vlo ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
        Cols256 acc = cols5_to_cols256(acc5);
        cols_mac_64x128_shl64(acc, vhi ^ gmask, s.theta_gamma_lo, s.theta_gamma_hi);
        const U256 wk{(u128_t)s.w_const[0] | ((u128_t)s.w_const[1] << 64), (u128_t)s.w_const[2] | ((u128_t)s.w_const[3] << 64)};
        const U256 t = u256_sub(cols_resolve(acc), wk);
        w_sign = u256_ge_2_191(t);
        if (w_sign) {
            u256_shr_lo128(u256_not(t), 126, wlo, whi);
            wlo += 1;
            whi += (wlo == 0);
            wlo = (wlo >> 1) | (whi << 63);
            whi >>= 1;
        } else {
            u256_shr_lo128(t, 126, wlo, whi);
            const u64 odd = wlo & 1;
            wlo = (wlo >> 1) | (whi << 63);
            whi >>= 1;
            wlo += odd;
            whi += (wlo < odd);
        }
    }
    const uint32_t vh = (uint32_t)vhi & 15, wh = (uint32_t)whi & 15;
    u64 *o = out + poly * out_poly_stride + col;
    for (uint32_t jt = s.ncommon; jt < s.nto; jt++) {
        const DevMod q = to_mods[jt];
        const u64 *om = s.omega + (u64)jt * NF;   // rows zero-padded to NF
        Acc3x64 a192;
        u128_t extra = 0;                                      // small addends of the sum (< 2^66)
        mac3x64(a192, vlo, s.gamma_neg[jt]);                   // -v_lo * gamma
        // -v_hi * 2^64 * gamma (< q) through a 16-entry table -- a per-lane load, skipped when the host-side bound
        // on v (scaler_upload: v <= sum_i (q_i - 1) + 1) says v_hi is always zero
        u64 small = s.v_fits_64 ? 0 : s.vhi_tab[jt * 16 + vh];
        if (!s.is_one) {
            // +/- w = +/- (w_hi * 2^64 + w_lo): the high part through the table, the low word straight
            // into the 192-bit sum -- as w_lo, or as K - w_lo with K = q * ceil(2^64 / q) = 2^64 + K_lo = 0 (mod q)
            const u64 c = s.c64_tab[jt * 16 + wh];             // w_hi * 2^64 mod q
            small += w_sign ? (c ? q.p - c : 0) : c;           // < 2q
            const u64 k_lo = q.p * (q.brt_hi + 1);             // K mod 2^64 (K >= 2^64 > w_lo)
            extra = w_sign ? ((((u128_t)1 << 64) | k_lo) - wlo) : (u128_t)wlo;
        }
#pragma unroll
        for (int i = 0; i < NF; i++) mac3x64(a192, rests[i], om[i]);
        extra += small;
        // (extra < 2^66 does not fit the u64 parameter: split it)
        u128_t acc;
        u64 top;
        acc3x64_resolve(a192, (u64)extra, acc, top);
        {
            const u128_t hi_extra = (extra >> 64) << 64;       // at most 3 * 2^64
            const bool c = __builtin_add_overflow(acc, hi_extra, &acc);
            top += c ? 1 : 0;
        }
        u64 r;
        if ((s.narrow_mask >> (jt & 63)) & 1) {
            // the whole sum is < 2^(2k+1) (hence top == 0): the single-word Barrett of zq_dev.hpp does it
            r = barrett_reduce_wide((u64)(acc >> 64), (u64)acc, q);
        } else if ((s.fold_mask >> (jt & 63)) & 1) {
            // < 2^(2k+6): replace the bits above 2^(2k) by their residue (64-entry table), which leaves
            // < 2^(2k) + q < 2^(2k+1) for the same single-word Barrett
            const uint32_t f = 2 * q.k;
            const uint32_t idx = (uint32_t)(acc >> f);
            acc = (acc & ((((u128_t)1) << f) - 1)) + s.fold_tab[jt * 64 + idx];
            r = barrett_reduce_wide((u64)(acc >> 64), (u64)acc, q);
        } else {
            r = reduce_u128((u64)(acc >> 64), (u64)acc, q);    // [0, q)
            r = csub_n(r + s.c128_tab[jt * 16 + ((uint32_t)top & 15)], q.p, q.np);
        }
        o[(u64)jt * n] = r;
    }
}

// ----------------------------------------------------------------- switch_down ----
// Poly::switch_down (M/rq/mod.rs:433-492), one lane per coefficient:
// in [npolys][L][N] PowerBasis -> out [npolys][L-1][N].
__global__ void switch_down_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                   u64 out_poly_stride, const DevMod *__restrict__ mods,
                                   const u64x2 *__restrict__ inv_last, uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const uint32_t col = (uint32_t)(gid & (n - 1));
    const u64 poly = gid >> logn;
    const u64 *src = in + poly * in_poly_stride + col;
    u64 *dst = out + poly * out_poly_stride + col;
    const DevMod ql = mods[nmod - 1];
    const u64 half = ql.p >> 1;
    const u64 last = add_mod(src[(u64)(nmod - 1) * n], half, ql.p);
    for (uint32_t r = 0; r + 1 < nmod; r++) {
        const DevMod qi = mods[r];
        const u64 half_mod = qi.p - reduce_u64(half, qi);       // (0, qi]
        const u64 tmp = reduce_u64(last, qi) + half_mod;        // < 2 qi
        const u64 c = src[(u64)r * n] + 3 * qi.p - tmp;         // < 4 qi
        dst[(u64)r * n] = mul_shoup(c, inv_last[r].x, inv_last[r].y, qi.p);
    }
}

// ------------------------------------------------------------------ substitute ----
// Poly::substitute (M/rq/mod.rs:360-412).  One lane per (row, j).
__global__ void substitute_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                  u64 out_poly_stride, const DevMod *__restrict__ mods, uint32_t nmod,
                                  uint32_t logn, uint32_t exponent, uint32_t repr_is_ntt, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn, mask = n - 1;
    const uint32_t j = (uint32_t)(gid & mask);
    const uint32_t r = (uint32_t)((gid >> logn) % nmod);
    const u64 poly = (gid >> logn) / nmod;
    const u64 *src = in + poly * in_poly_stride + (u64)r * n;
    u64 *dst = out + poly * out_poly_stride + (u64)r * n;
    if (repr_is_ntt) {
        // q[bitrev[j]] = p[bitrev((e-1)/2 + j*e mod N)]; index the gather by destination d = bitrev(j)
        const uint32_t d = j;
        const uint32_t jj = __brev(d) >> (32 - logn);
        const uint32_t srci = (uint32_t)(((u64)(exponent - 1) / 2 + (u64)jj * exponent) & mask);
        dst[d] = src[__brev(srci) >> (32 - logn)];
    } else {
        const u64 power = (u64)j * exponent;
        const u64 v = src[j];
        dst[power & mask] = (power & n) ? neg_mod(v, mods[r].p) : v;
    }
}

// ----------------------------------------------------------- element-wise kernels ----
enum { EW_ADD = 0, EW_SUB = 1, EW_MUL = 2, EW_NEG = 3 };
// a op= b on [rows_total][N]; modulus index = row % nmod (M/rq/ops.rs:10-206, 354-418).
__global__ void ew_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, const DevMod *__restrict__ mods,
                          uint32_t nmod, uint32_t logn, uint32_t op, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const DevMod m = mods[(gid >> logn) % nmod];
    const u64 x = a[gid];
    u64 r;
    switch (op) {
        case EW_ADD: r = add_mod(x, b[gid], m.p); break;
        case EW_SUB: r = sub_mod(x, b[gid], m.p); break;
        case EW_MUL: r = mul_mod(x, b[gid], m); break;
        default: r = neg_mod(x, m.p); break;
    }
    a[gid] = r;
}
__global__ void mul_shoup_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, const u64 *__restrict__ bs,
                                 const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 p = mods[(gid >> logn) % nmod].p;
    a[gid] = mul_shoup(a[gid], b[gid], bs[gid], p);
}
// Tensor step of Multiplicator::multiply (F/bfv/ops/mul.rs:198-201).  Operand polynomials
// (c00, c01) = extL[b][0..1], (c10, c11) = extR[b][0..1], each [K][N]; rows below `ncommon`
// are read from the original ciphertexts lhs/rhs [b][2][L][N] when those pointers are given
// (the extender copies them verbatim, M/rq/scaler.rs:61-65, so the copy is skipped).
// t is slot-major: t[slot][b][K][N] = (c00*c10, c00*c11 + c01*c10, c01*c11).
__global__ void tensor_kernel(const u64 *__restrict__ extL, const u64 *__restrict__ extR,
                              const u64 *__restrict__ lhs, const u64 *__restrict__ rhs, u64 *__restrict__ t,
                              const DevMod *__restrict__ mods, uint32_t nmod, uint32_t ncommon, uint32_t lrows,
                              uint32_t logn, u64 nb, uint32_t debug_acquire) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (debug_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    // grid: x = chunks of one extended polynomial, y = ciphertext pair (no runtime divisions)
    const u64 pn = (u64)nmod << logn;  // elements per extended polynomial
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pn) return;
    const u64 b = blockIdx.y;
    const uint32_t row = (uint32_t)(off >> logn);
    const DevMod m = mods[row];
    u64 c00, c01, c10, c11;
    if (lhs && row < ncommon) {
        const u64 pl = (u64)lrows << logn;
        c00 = lhs[b * 2 * pl + off];
        c01 = lhs[b * 2 * pl + pl + off];
        c10 = rhs[b * 2 * pl + off];
        c11 = rhs[b * 2 * pl + pl + off];
    } else {
        c00 = extL[b * 2 * pn + off];
        c01 = extL[b * 2 * pn + pn + off];
        c10 = extR[b * 2 * pn + off];
        c11 = extR[b * 2 * pn + pn + off];
    }
    u64 *o = t + b * pn + off;
    if (debug_acquire == 2) {  // developer aid: dump the operands as read
        o[0] = c00;
        o[nb * pn] = c10;
        o[2 * nb * pn] = c01;
        return;
    }
    o[0] = mul_mod(c00, c10, m);
    {
        const u128_t sum = (u128_t)c00 * c11 + (u128_t)c01 * c10;  // one reduction, see tensor_intt_kernel
        o[nb * pn] = barrett_reduce_wide((u64)(sum >> 64), (u64)sum, m);
    }
    o[2 * nb * pn] = mul_mod(c01, c11, m);
}
// dot_product_scalar / rq::dot_product (F/bfv/ops/dot_product.rs:54-180, M/rq/ops.rs:449-570):
// out[b][part][row][c] = sum_k cts[b][k][part][row][c] * pts[b][k][row][c]  mod q_row.
// One lane per pair of coefficients and ALL `NP` parts of the group starting at blockIdx.y*NP
// (each plaintext word is loaded once); exact 128-bit products accumulated in 192 bits and
// reduced once (the reference's periodic reduce_u128 gives the same canonical sum).
// Streaming, HBM bound.
template <int NP>
__global__ void __launch_bounds__(256, 8) dot_kernel(const u64 *__restrict__ cts, u64 ct_batch_stride, const u64 *__restrict__ pts,
                           u64 pt_batch_stride, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                           const u64x2 *__restrict__ pow2 /* {2^64, 2^128} mod q */, uint32_t nparts, uint32_t count,
                           uint32_t logn, u64 pl /* L*N */) {
    // grid: x = pairs of coefficients of one polynomial, y = group of NP parts, z = batch
    const u64 pair = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * pair >= pl) return;
    const u64 off = 2 * pair;
    const uint32_t part0 = blockIdx.y * NP, b = blockIdx.z;
    const uint32_t row = (uint32_t)(off >> logn);
    const DevMod m = mods[row];
    const u64 *cp = cts + (u64)b * ct_batch_stride + (u64)part0 * pl + off;
    const u64 *pp = pts + (u64)b * pt_batch_stride + off;
    Acc192 a0[NP], a1[NP];
#pragma unroll 4   // (2 -> 4: +2 %; 8: no further gain -- the kernel runs at 4.0 TB/s of fabric reads, PMC FETCH_SIZE)
    for (uint32_t k = 0; k < count; k++) {
        const u64x2 y = *reinterpret_cast<const u64x2 *>(pp + (u64)k * pl);
#pragma unroll
        for (int q = 0; q < NP; q++) {
            if (part0 + q < nparts) {
                const u64x2 x = *reinterpret_cast<const u64x2 *>(cp + ((u64)k * nparts + q) * pl);
                mac192(a0[q], x.x, y.x);
                mac192(a1[q], x.y, y.y);
            }
        }
    }
    // value = top * 2^128 + a: reduce a, then add (top mod q) * (2^128 mod q)
    const u64 c128 = pow2[row].y;
    auto fold = [&](const Acc192 &acc) -> u64 {
        u128_t a;
        u64 top;
        acc192_resolve(acc, a, top);
        const u64 r = reduce_u128((u64)(a >> 64), (u64)a, m);
        return top ? add_mod(r, mul_mod(reduce_u64(top, m), c128, m), m.p) : r;
    };
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if (part0 + q < nparts) {
            u64x2 o;
            o.x = fold(a0[q]);
            o.y = fold(a1[q]);
            *reinterpret_cast<u64x2 *>(out + ((u64)b * nparts + part0 + q) * pl + off) = o;
        }
    }
}

// General tensor step of `&ct * &ct` (F/bfv/ops/mod.rs:300-327): out[b][k] = sum_{i+j=k} a[b][i] (.) b[b][j];
// grid = (ceil(pl / block), la + lb - 1, batch).
__global__ void tensor_general_kernel(const u64 *__restrict__ a, const u64 *__restrict__ bb, u64 *__restrict__ out,
                                      const DevMod *__restrict__ mods, uint32_t la, uint32_t lb, uint32_t logn, u64 pl) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t kk = blockIdx.y, b = blockIdx.z;
    const DevMod m = mods[off >> logn];
    const u64 *pa = a + (u64)b * la * pl + off, *pb = bb + (u64)b * lb * pl + off;
    u64 acc = 0;
    for (uint32_t i = 0; i < la; i++) {
        if (kk < i || kk - i >= lb) continue;
        acc = add_mod(acc, mul_mod(pa[(u64)i * pl], pb[(u64)(kk - i) * pl], m), m.p);
    }
    out[((u64)b * (la + lb - 1) + kk) * pl + off] = acc;
}

// `Ciphertext * Plaintext` (F/bfv/ops/mod.rs:229-257): out[b][part] = ct[b][part] (.) pt[b].
__global__ void mul_plain_kernel(const u64 *__restrict__ ct, const u64 *__restrict__ pt, u64 pt_batch_stride,
                                 u64 *__restrict__ out, const DevMod *__restrict__ mods, uint32_t nparts, uint32_t logn,
                                 u64 pl) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t part = blockIdx.y, b = blockIdx.z;
    const DevMod m = mods[off >> logn];
    const u64 idx = ((u64)b * nparts + part) * pl + off;
    out[idx] = mul_mod(ct[idx], pt[(u64)b * pt_batch_stride + off], m);
}

// SecretKey::try_decrypt (F/bfv/keys/secret_key.rs:205-247).  phase_kernel: out[b] = sum_i
// ct[b][i] (.) s^i by Horner's rule (same canonical value as the reference's running power of s);
// grid = (ceil(L*N / block), batch).  decrypt_tail_kernel: ((d_0 + t) mod q_0) mod t on row 0
// of the scaled polynomial.
__global__ void phase_kernel(const u64 *__restrict__ ct, const u64 *__restrict__ sk, u64 *__restrict__ out,
                             const DevMod *__restrict__ mods, uint32_t nparts, uint32_t logn, u64 pl) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t b = blockIdx.y;
    const DevMod m = mods[off >> logn];
    const u64 *c = ct + (u64)b * nparts * pl + off;
    const u64 sv = sk[off];
    u64 acc = c[(u64)(nparts - 1) * pl];
    for (uint32_t i = nparts - 1; i-- > 0;) acc = add_mod(mul_mod(acc, sv, m), c[(u64)i * pl], m.p);
    out[(u64)b * pl + off] = acc;
}
__global__ void decrypt_tail_kernel(const u64 *__restrict__ d, u64 d_poly_stride, u64 *__restrict__ out, DevMod q0,
                                    DevMod tm, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 b = gid >> logn, x = gid & ((1ull << logn) - 1);
    const u64 w = reduce_u64(d[b * d_poly_stride + x] + tm.p, q0);
    out[gid] = reduce_u64(w, tm);
}

// Rq wire format (crates/fhe-util/src/lib.rs:71-148 via M/zq/mod.rs:783-793): a row is N
// coefficients of nbits = bitlen(p - 1) bits, little-endian bit-packed.  Eight coefficients are
// exactly nbits bytes, so one thread transcodes one such group with the reference's shift
// register; grid = (ceil(N/8 / block), L, npolys).  (Boundary work: byte-granular accesses.)
__device__ __forceinline__ uint32_t wire_bits(u64 p) { return 64u - (uint32_t)__builtin_clzll(p - 1); }
__device__ __forceinline__ u64 wire_row_offset(const DevMod *mods, uint32_t r, uint32_t logn) {
    u64 off = 0;
    for (uint32_t i = 0; i < r; i++) off += (u64)wire_bits(mods[i].p) << (logn - 3);
    return off;
}
__global__ void wire_pack_kernel(const u64 *__restrict__ polys, uint8_t *__restrict__ bytes,
                                 const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << (logn - 3))) return;
    const uint32_t r = blockIdx.y, poly = blockIdx.z;
    const uint32_t nbits = wire_bits(mods[r].p);
    const u64 mask = ~0ull >> (64 - nbits);
    const u64 *src = polys + (((u64)poly * nmod + r) << logn) + 8u * g;
    uint8_t *dst = bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn) + (u64)g * nbits;
    u128_t cur = 0;
    uint32_t have = 0, o = 0;
    for (uint32_t e = 0; e < 8; e++) {
        cur |= (u128_t)(src[e] & mask) << have;
        have += nbits;
        while (have >= 8) {
            dst[o++] = (uint8_t)cur;
            cur >>= 8;
            have -= 8;
        }
    }
}
__global__ void wire_unpack_kernel(const uint8_t *__restrict__ bytes, u64 *__restrict__ polys,
                                   const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn, u64 poly_bytes) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << (logn - 3))) return;
    const uint32_t r = blockIdx.y, poly = blockIdx.z;
    const uint32_t nbits = wire_bits(mods[r].p);
    const u64 mask = ~0ull >> (64 - nbits);
    const uint8_t *src = bytes + (u64)poly * poly_bytes + wire_row_offset(mods, r, logn) + (u64)g * nbits;
    u64 *dst = polys + (((u64)poly * nmod + r) << logn) + 8u * g;
    u128_t cur = 0;
    uint32_t have = 0, i = 0;
    for (uint32_t e = 0; e < 8; e++) {
        while (have < nbits) {
            cur |= (u128_t)src[i++] << have;
            have += 8;
        }
        dst[e] = (u64)cur & mask;
        cur >>= nbits;
        have -= nbits;
    }
}

// Oblivious expansion (F/bfv/keys/evaluation_key.rs:233-244).  monomial_kernel writes the
// PowerBasis polynomials -x^(N - 2^l), l < nlev, into a zeroed [nlev][L][N] buffer (the forward
// NTT follows); expand_step_kernel does, per coefficient of the polynomials of the lower half,
// high = (low - sub) (.) monomial  (only the first nhigh polynomials of the upper half exist)
// and low += sub.  grid = (ceil(L*N / block), npolys).
__global__ void monomial_kernel(u64 *__restrict__ buf, const DevMod *__restrict__ mods, uint32_t nlev, uint32_t nmod,
                                uint32_t logn) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nlev * nmod) return;
    const uint32_t lev = gid / nmod, r = gid % nmod, n = 1u << logn;
    buf[((u64)lev * nmod + r) * n + (n - (1u << lev))] = mods[r].p - 1;
}
__global__ void expand_step_kernel(u64 *__restrict__ low, const u64 *__restrict__ sub, u64 *__restrict__ high,
                                   const u64 *__restrict__ mono, const DevMod *__restrict__ mods, uint32_t logn, u64 pl,
                                   uint32_t nhigh) {
    const u64 off = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= pl) return;
    const uint32_t poly = blockIdx.y;
    const DevMod m = mods[off >> logn];
    const u64 idx = (u64)poly * pl + off;
    const u64 lo = low[idx], sb = sub[idx];
    if (poly < nhigh) high[idx] = mul_mod(sub_mod(lo, sb, m.p), mono[off], m);
    low[idx] = add_mod(lo, sb, m.p);
}

// ------------------------------------------------------ seeded polynomial (wire c1) ----
// Poly::random_from_seed (M/rq/mod.rs:276-292), the `c1` a received secret-key ciphertext expands from its 32-byte
// seed (F/bfv/ciphertext.rs:287-302): key = SHA-256(seed); one ChaCha8 stream (64-bit block counter from 0, stream
// id 0; a u64 = two consecutive little-endian words); residue row after residue row, `degree` draws each from
// Uniform[0, q_i) by Lemire's widening-multiply rejection: x -> (hi, lo) = x * q, accept hi when
// lo >= (2^64 - q) mod q.  The stream position of a coefficient depends on the rejections before it, so one
// workgroup walks one polynomial: every thread computes one ChaCha block (8 candidates), an exclusive scan of the
// accept counts places the survivors, and the position after the row's last accepted draw starts the next batch.
// SHA-256 and the ChaCha block function are pinned by known-answer tests of the oracle; the generator's layout and
// the sampling rule restate rand_chacha 0.10 / rand 0.10, which are not vendored: PARITY UNPINNED (like psi).
__device__ __forceinline__ uint32_t rotr32(uint32_t v, int c) { return (v >> c) | (v << (32 - c)); }
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
// SHA-256 of exactly 32 bytes (one padded block); digest as 8 big-endian words
__device__ inline void sha256_32(const uint8_t *msg, uint32_t h[8]) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 8; i++)
        w[i] = ((uint32_t)msg[4 * i] << 24) | ((uint32_t)msg[4 * i + 1] << 16) | ((uint32_t)msg[4 * i + 2] << 8) | msg[4 * i + 3];
    w[8] = 0x80000000u;
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 256;   // message length in bits
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c, g = 0x1f83d9ab,
             hh = 0x5be0cd19;
    const uint32_t init[8] = {a, b, c, d, e, f, g, hh};
    for (int i = 0; i < 64; i++) {
        const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
        const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    const uint32_t fin[8] = {a, b, c, d, e, f, g, hh};
    for (int i = 0; i < 8; i++) h[i] = init[i] + fin[i];
}
// One ChaCha8 block: key words (little-endian), 64-bit block counter, stream id 0.
__device__ __forceinline__ void chacha8_block(const uint32_t key[8], u64 counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define FHE_CHACHA_QR(a, b, c, d)                  \
    x[a] += x[b], x[d] = rotl32(x[d] ^ x[a], 16);  \
    x[c] += x[d], x[b] = rotl32(x[b] ^ x[c], 12);  \
    x[a] += x[b], x[d] = rotl32(x[d] ^ x[a], 8);   \
    x[c] += x[d], x[b] = rotl32(x[b] ^ x[c], 7);
#pragma unroll
    for (int r = 0; r < 4; r++) {   // 8 rounds = 4 double rounds
        FHE_CHACHA_QR(0, 4, 8, 12) FHE_CHACHA_QR(1, 5, 9, 13) FHE_CHACHA_QR(2, 6, 10, 14) FHE_CHACHA_QR(3, 7, 11, 15)
        FHE_CHACHA_QR(0, 5, 10, 15) FHE_CHACHA_QR(1, 6, 11, 12) FHE_CHACHA_QR(2, 7, 8, 13) FHE_CHACHA_QR(3, 4, 9, 14)
    }
#undef FHE_CHACHA_QR
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
// grid.x = polynomials; 256 threads; seeds [npolys][32] bytes -> out [npolys][nmod][N].
constexpr int SEED_THREADS = 256;
constexpr size_t SEED_SMEM_BYTES = 8 + 8 * 4 + SEED_THREADS * 4;
__global__ void __launch_bounds__(SEED_THREADS)
    seed_expand_kernel(const uint8_t *__restrict__ seeds, u64 *__restrict__ out, const DevMod *__restrict__ mods,
                       uint32_t nmod, uint32_t logn) {
    FHE_DYN_SMEM(u64, sm);   // SEED_SMEM_BYTES: next position | key[8] | scan[SEED_THREADS]
    u64 &s_next_pos = sm[0];
    uint32_t *const s_key = reinterpret_cast<uint32_t *>(sm + 1);
    uint32_t *const s_scan = s_key + 8;
    const uint32_t tid = threadIdx.x, n = 1u << logn;
    if (tid == 0) {
        uint32_t h[8];
        sha256_32(seeds + (u64)blockIdx.x * 32, h);
        // the digest's bytes (big-endian words) are the seed array; ChaCha reads its key as little-endian words
        for (int i = 0; i < 8; i++) s_key[i] = __builtin_bswap32(h[i]);
    }
    __syncthreads();
    uint32_t key[8];
    for (int i = 0; i < 8; i++) key[i] = s_key[i];
    u64 *dst = out + (u64)blockIdx.x * nmod * n;
    u64 pos = 0;   // index of the next u64 of the stream (uniform)
    for (uint32_t r = 0; r < nmod; r++) {
        const u64 q = mods[r].p;
        const u64 thresh = (0 - q) % q;   // (2^64 - q) mod q
        uint32_t produced = 0;
        while (produced < n) {
            const u64 blk = (pos >> 3) + tid;
            uint32_t w[16];
            chacha8_block(key, blk, w);
            u64 val[8];
            uint32_t accept = 0;   // bit k: candidate k of this block is drawn (not before `pos`) and accepted
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u64 x = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
                const u128_t m = (u128_t)x * q;
                val[k] = (u64)(m >> 64);
                if (8 * blk + k >= pos && (u64)m >= thresh) accept |= 1u << k;
            }
            // exclusive scan of the accept counts over the workgroup (Hillis-Steele in LDS)
            const uint32_t cnt = (uint32_t)__builtin_popcount(accept);
            s_scan[tid] = cnt;
            __syncthreads();
            for (uint32_t off = 1; off < SEED_THREADS; off <<= 1) {
                const uint32_t v = tid >= off ? s_scan[tid - off] : 0;
                __syncthreads();
                s_scan[tid] += v;
                __syncthreads();
            }
            const uint32_t incl = s_scan[tid], total = s_scan[SEED_THREADS - 1];
            uint32_t rank = produced + incl - cnt;
            const uint32_t need = n - produced;   // draws this row still takes
            if (tid == 0) s_next_pos = 8 * ((pos >> 3) + SEED_THREADS);   // all candidates consumed unless the row ends here
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (accept & (1u << k)) {
                    if (rank < n) dst[(u64)r * n + rank] = val[k];
                    if (rank + 1 == n && total >= need) s_next_pos = 8 * blk + k + 1;   // the row's last draw
                    rank++;
                }
            }
            __syncthreads();
            pos = s_next_pos;
            produced = total >= need ? n : produced + total;
            __syncthreads();
        }
    }
}

// Copies the first `rows` rows of each polynomial: in [npolys][in_rows][N] -> out [npolys][out_rows][N].
__global__ void copy_rows_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 in_poly_stride,
                                 u64 out_poly_stride, u64 per_poly, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 poly = gid / per_poly, off = gid % per_poly;
    out[poly * out_poly_stride + off] = in[poly * in_poly_stride + off];
}
// x = splitmix64(seed ^ (ct<<40) ^ (part<<36) ^ (row<<28) ^ coeff) mod q_row  
__global__ void synth_kernel(u64 *__restrict__ out, const DevMod *__restrict__ mods, uint32_t nmod, uint32_t logn,
                             uint32_t nparts, u64 seed, u64 ct0, u64 part0, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const uint32_t n = 1u << logn;
    const u64 coeff = gid & (n - 1);
    const u64 rowi = gid >> logn;
    const u64 row = rowi % nmod;
    const u64 part = part0 + (rowi / nmod) % nparts;
    const u64 ct = ct0 + rowi / ((u64)nmod * nparts);
    const u64 v = splitmix64(seed ^ (ct << 40) ^ (part << 36) ^ (row << 28) ^ coeff);
    out[gid] = v % mods[row].p;
}

#if defined(FHE_LAB)
#include "lab/lab_kernels.hpp"   // measured-and-rejected variants: lab builds only
#endif

}  // namespace k
}  // namespace fhe
