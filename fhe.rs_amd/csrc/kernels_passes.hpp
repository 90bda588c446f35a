// kernels_passes.hpp -- tile geometry, the forward / inverse radix passes on an LDS tile and the tile load / store
// helpers: the building blocks of the NTT, tensor+iNTT and key-switch kernels.
#pragma once
#include "kernels_common.hpp"
#include "zq_f64.hpp"

namespace fhe {
namespace k {

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
// NARROW = -HR < 0 (round 6): the F64 form for launches whose moduli are all below 2^(53 - HR) (zq_f64.hpp).  The tile
// holds the bit patterns of doubles (signed representatives), `tw` is the context's F64 twiddle table (w as a double in the
// first word of each pair and w / p in the second, same indexing; -NARROW & 8: per-lane twiddles are read as one word), pm carries {p, 1 / p} (make_pm_f64); input to stage 0 below 2^(53 - HR) in magnitude
// (canonical residues, or digit rows of another modulus of the launch); values leave below f64_fwd_out_bound().
// NT > 1: the same pass on NT tiles that lie `tile_words` apart in LDS (the key switch transforms two digits
// under one modulus at once): addresses and twiddles are formed once and serve every tile.
// `Dst` (last pass only, lab knob FHE_FWD_DIRECT_STORE): a functor (first index, x[]) that takes the group's 2^G
// CONSECUTIVE results straight from the registers (the stride-1 pass: lo_bits == 0) instead of the tile.
template <int G, int LOGM, int S0, int T, bool PRE, int NARROW = 0, class Src = NoSrc, int NT = 1, bool WLX = false,
          class Dst = NoSrc>
__device__ __forceinline__ void fwd_pass(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                         uint32_t tid, const FwdTw<G, LOGM, S0, T> &tw_regs, Src src = Src{},
                                         uint32_t tile_words = 0, Dst dst = Dst{}) {
    constexpr bool DIRECT = !std::is_same<Src, NoSrc>::value;
    constexpr bool DIRECT_OUT = !std::is_same<Dst, NoSrc>::value;
    static_assert(!DIRECT_OUT || (S0 + G == LOGM && NT == 1), "direct stores: the last (stride-1) pass of one tile");
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
                    if constexpr (NARROW < 0) {
                        // -NARROW = HR | (8: per-lane twiddles are read as ONE word).  The one-word form (quotient from
                        // h (1/p)) is for the key switch, whose accumulators leave no registers for {w, w/p} pairs: it lets the
                        // N = 16384 tile run radix-8 passes throughout (stock n = 16384 relinearise -19 %).  Register-resident it is
                        // 8 % slower than the {w, w/p} form, which the transforms proper keep in every pass
                        // (profiles/r06_f64_one_word_ab.jsonl).
                        constexpr int HRV = (-NARROW) & 7;
                        constexpr bool ONE_WORD = ((-NARROW) & 8) != 0;
                        const PF pf = pf_of(pm);
                        double xa = f64_of_bits(x[a]), xb = f64_of_bits(x[a + half]);
                        if (f64_fwd_reduces(S0 + u, HRV)) xa = reduce_f64(xa, pf), xb = reduce_f64(xb, pf);
                        if constexpr (UNIFORM || !ONE_WORD) fwd_butterfly_wp_f64(xa, xb, f64_of_bits(wv.x), f64_of_bits(wv.y), pf);
                        else fwd_butterfly_f64(xa, xb, f64_of_bits(wv.x), pf);
                        x[a] = bits_of_f64(xa), x[a + half] = bits_of_f64(xb);
                    } else if constexpr (NARROW > 0)
                        fwd_butterfly_narrow<UNIFORM>(x[a], x[a + half], wv.x, wv.y, pm, fwd_narrow_corrects(S0 + u, NARROW));
                    else
                        fwd_butterfly<UNIFORM>(x[a], x[a + half], wv.x, wv.y, pm);
                }
            }
        }
        if constexpr (DIRECT_OUT) {
            dst(base, x);
        } else {
#pragma unroll
        for (uint32_t e = 0; e < R; e++) g[padi(e << lo_bits)] = x[e];
        }
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
template <int LOGM, int T, int GM, bool TWPF, bool FSYNC, int NARROW, int PASS, int S0, bool LATE, int NT, class W, class Src,
          class Dst = NoSrc>
__device__ __forceinline__ void ntt_fwd_lds_rec(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                                uint32_t tid, const W &tw_regs, Src src, uint32_t tile_words, Dst dst = Dst{}) {
    constexpr int G = fwd_plan_g<LOGM, GM, PASS, LATE>();
    constexpr bool OUT = !std::is_same<Dst, NoSrc>::value && PASS + 1 == fwd_np(LOGM, GM);   // this pass stores to `dst`
    constexpr bool WLX = fwd_wl_after(LOGM, GM, LATE, PASS) || fwd_wl_after(LOGM, GM, LATE, PASS - 1);
    static_assert(!WLX || T % 64 == 0, "wave-local exchanges need whole 64-lane waves");
    if constexpr (OUT && PASS != 0)
        fwd_pass<G, LOGM, S0, T, TWPF, NARROW, NoSrc, NT, WLX, Dst>(lds, tw, kbase, pm, tid, tw_regs, NoSrc{}, tile_words, dst);
    else if constexpr (PASS == 0)
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
                                                                                      tile_words, dst);
    } else {
        if constexpr (FSYNC && !OUT) FHE_BARRIER();
    }
}
template <int LOGM, int T, int GM = GMAX, bool TWPF = true, bool FSYNC = true, int NARROW = 0, class Src = NoSrc,
          bool LATE = false, int NT = 1, class Dst = NoSrc>
__device__ __forceinline__ void ntt_fwd_lds(u64 *lds, const u64x2 *__restrict__ tw, uint32_t kbase, const PM pm,
                                            uint32_t tid, Src src = Src{}, uint32_t tile_words = 0, Dst dst = Dst{}) {
    constexpr int G = fwd_plan_g<LOGM, GM, 0, LATE>();
    static_assert(std::is_same<Dst, NoSrc>::value || fwd_np(LOGM, GM) > 1, "direct stores need a pass of their own");
    FwdTw<G, LOGM, 0, T> first;
    if constexpr (TWPF) fwd_tw_load(first, tw, kbase, tid);
    ntt_fwd_lds_rec<LOGM, T, GM, TWPF, FSYNC, NARROW, 0, 0, LATE, NT>(lds, tw, kbase, pm, tid, first, src, tile_words, dst);
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
// F64 = HR > 0 (round 6): the F64 form (see fwd_pass; zq_f64.hpp).  BIN / BOUT are then bounds in 1/1024ths of
// 2^(53 - HR): every register's bound is tracked through the unrolled stages as for NARROW, a register is reduced
// (reduce_f64, three operations) only where a sum would leave 2^53, and values travel between passes below BOUT.
template <int G, int LOGM, int V0, int T, bool NARROW = false, bool WLX = false, int BIN = 2, int BOUT = 2, int F64 = 0>
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
                    if constexpr (F64 > 0) {
                        const PF pf = pf_of(pm);
                        double xa = f64_of_bits(x[a]), xb = f64_of_bits(x[b]);
                        // keep xa + xb and xa - xb below 2^53: reduce the larger operand (then the other) if needed
#pragma unroll
                        for (int it = 0; it < 2; it++)
                            if (bnd[a] + bnd[b] >= f64_limit(F64)) {
                                if (bnd[a] >= bnd[b]) xa = reduce_f64(xa, pf), bnd[a] = F64_REDUCED;
                                else xb = reduce_f64(xb, pf), bnd[b] = F64_REDUCED;
                            }
                        const double d = xa - xb, sm = xa + xb;
                        if (V0 + G == LOGM && u == G - 1 && fold) {
                            xa = mulmod_f64(sm, f64_of_bits(ninv.x), f64_of_bits(ninv.y), pf.p);     // (N^-1 pairs: wave-uniform)
                            xb = mulmod_f64(d, f64_of_bits(zninv.x), f64_of_bits(zninv.y), pf.p);
                        } else {
                            xa = sm;
                            xb = mulmod_f64(d, f64_of_bits(zv.x), f64_of_bits(zv.y), pf.p);
                        }
                        const int sum = bnd[a] + bnd[b];
                        bnd[a] = sum;   // (kept independent of the run-time `fold`: conservative for the folded last stage)
                        bnd[b] = f64_product_bound(sum, F64);
                        x[a] = bits_of_f64(xa), x[b] = bits_of_f64(xb);
                    } else if constexpr (NARROW) {
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
                            x[b] = shoup_lo<UNIFORM, false>(0, diff, zv.x, mulhi64_approx<UNIFORM>(diff, zv.y), pm.np);   // below 3p
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
        if constexpr (F64 > 0) {
            if (!(V0 + G == LOGM && fold)) {
                const PF pf = pf_of(pm);
#pragma unroll
                for (uint32_t e = 0; e < R; e++)
                    if (bnd[e] > BOUT) x[e] = bits_of_f64(reduce_f64(f64_of_bits(x[e]), pf));
            }
        } else if constexpr (NARROW) {  // the next pass (or the epilogue / the global pass) expects values below BOUT * p
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
// values between F64 inverse passes: below a quarter of 2^53 (two more doublings fit before the next reduction)
constexpr int f64_inv_mid(int HR) { return f64_limit(HR) / 4; }
// F64_BIN0: bound of the tile the first F64 pass reads (canonical residues: F64_ONE; the tensor kernel's double products:
// twice that)
template <int LOGM, int T, int PASS = 0, int V0 = 0, bool NARROW = false, int F64 = 0, int F64_BIN0 = F64_ONE, class W>
__device__ __forceinline__ void ntt_inv_lds(u64 *lds, const u64x2 *__restrict__ itw, uint32_t logn, uint32_t sub,
                                            const PM pm, uint32_t tid, bool fold, u64x2 ninv, u64x2 zninv,
                                            const W &tw_regs) {
    constexpr int G = inv_plan_g<LOGM, PASS>();
    constexpr bool WLX = inv_wl_after(LOGM, PASS) || inv_wl_after(LOGM, PASS - 1);
    static_assert(!WLX || T % 64 == 0, "wave-local exchanges need whole 64-lane waves");
    constexpr bool LAST = PASS + 1 == plan_np(LOGM, GMAX);
    if constexpr (F64 > 0)
        inv_pass<G, LOGM, V0, T, false, WLX, (PASS == 0 ? F64_BIN0 : f64_inv_mid(F64)), f64_inv_mid(F64), F64>(
            lds, itw, logn, sub, pm, tid, fold, ninv, zninv, tw_regs);
    else
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
        ntt_inv_lds<LOGM, T, PASS + 1, V0 + G, NARROW, F64, F64_BIN0>(lds, itw, logn, sub, pm, tid, fold, ninv, zninv, next);
    } else {
        FHE_BARRIER();
    }
}

// ------------------------------------------------------------- tile load / store ----
// Thread t owns the 16-byte chunks {c*T + t}, c < CH, of the M-element tile (coalesced 16 B
// per lane).  CH > 0: all CH loads are issued before the first LDS write (one latency, not
// CH).  CH == 0: scalar strided loop for tiles smaller than 2*T (tiny test sizes).
template <int CH, int M, int T, bool NT = false, class F>
__device__ __forceinline__ void tile_to_lds(u64 *lds, const u64 *__restrict__ src, uint32_t tid, F f) {
    if constexpr (CH > 0) {
        const u64x2 *s2 = reinterpret_cast<const u64x2 *>(src);
        u64x2 v[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = load_last2<NT>(s2 + c * T + tid);
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

}  // namespace k
}  // namespace fhe
