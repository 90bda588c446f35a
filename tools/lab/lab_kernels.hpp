// lab_kernels.hpp -- measured-and-rejected kernel variants, compiled ONLY into -DFHE_LAB builds (tools/ab_*.sh).
// Nothing here is part of the shipped library: __graft_entry__.build() never defines FHE_LAB and kernels.hpp
// includes this file only under that macro.  Each variant is bit-exact (they were parity-tested when measured) and
// lost to the shipped kernel; numbers in DESIGN.md section 6 and the profiles/r02_* files named below.
#pragma once
// (included from inside namespace fhe::k, after the shipped passes and tile helpers it reuses)

// ------------------------------- forward NTT, 8 coefficients per thread (occupancy experiment) ----
// Same transform as ntt_kernel<false, 13>, cut for twice the resident waves: 1024 threads x 8 coefficients, pass plan
// GM (3: radix 8 throughout, GM_MIXED: radix 8 while the twiddles are scalar, radix 4 after), registers capped at 64
// so that the two workgroups a CU's LDS holds bring 8 waves per SIMD instead of 4.  FHE_NTT_CPT8 = 3 / 32 selects it
// for forward launches over 8192-point rows; numbers in DESIGN.md section 6.
template <bool NARROW, int GM>
__global__ void __launch_bounds__(1024, 8)
    ntt_fwd8_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map, const DevMod *__restrict__ mods,
                    const u64x2 *__restrict__ tw) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int LOGM = 13, T = 1024, M = 1 << LOGM;
    constexpr int CH = tile_chunks_c(LOGM, T);
    const uint32_t tid = threadIdx.x;
    const uint32_t poly = to_sgpr(blockIdx.x / map.rows);
    const uint32_t r = map.row_begin + (blockIdx.x - poly * map.rows);
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64 *src = in + (u64)poly * map.src_poly_stride +
                     (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * M;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * M;
    const u64x2 *twr = tw + (u64)mi * M;
    ntt_fwd_lds<LOGM, T, GM, false, true, (NARROW ? 1 : 0)>(lds, twr, 1, pm, tid, [&](uint32_t i, uint32_t) { return src[i]; });
    if constexpr (NARROW) {
        const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) {
            return csub_n(csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2), p, pm.np);
        });
    } else {
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); });
    }
}

// ------------------------------- forward NTT with in-wave stages by lane exchange ----
// north_star's "wavefront shuffles for the inner radix stages", built and measured against ntt_kernel (DESIGN.md
// section 6; FHE_NTT_SWAP=1 selects it for forward launches over 8192-point rows).  N = 8192, 512 threads x 16
// coefficients, three passes instead of four:
//   pass 1  stages 0-2 (position bits 12..10), strided groups straight from global memory, written to the tile;
//           the ONE workgroup barrier of the transform follows;
//   pass 2  six stages without touching LDS: wave w owns positions [1024 w, 1024 (w + 1)); a thread's 16 registers
//           are position bits 9..6, its lane number bits 5..0.  Stages 3-6 run in registers (wave-uniform scalar
//           twiddles); for stage 7 v_permlane32_swap exchanges register bit 3 with lane bit 5 -- afterwards each
//           lane holds both ends of its butterflies on bit 5 -- and for stage 8 v_permlane16_swap does the same
//           with register bit 2 and lane bit 4: one VALU instruction per 32-bit half instead of an LDS round trip;
//   pass 3  stages 9-12 on 16 consecutive coefficients per thread after a WAVE-LOCAL exchange through the tile
//           (pass 2 and pass 3 touch only the wave's own 1024 positions), then the usual coalesced store.
// LDS round trips 3 (4 in ntt_kernel), workgroup barriers 1 + the store's (2 + 1).
__device__ __forceinline__ void lane_swap64(u64 &a, u64 &b, uint32_t tid, int bit /* 5 or 4 */) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    if (bit == 5) {   // lanes 32..63 of `a` <-> lanes 0..31 of `b`
        const auto lo = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
        al = lo[0], bl = lo[1], ah = hi[0], bh = hi[1];
    } else {          // odd 16-lane rows of `a` <-> even rows of `b`
        const auto lo = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
        al = lo[0], bl = lo[1], ah = hi[0], bh = hi[1];
    }
    a = (u64)al | ((u64)ah << 32);
    b = (u64)bl | ((u64)bh << 32);
#else
    // host emulation (one workgroup at a time, fibers): the same exchange through a scratch array
    static u64 xchg[2][1024];
    xchg[0][tid] = a, xchg[1][tid] = b;
    __syncthreads();
    const uint32_t partner = tid ^ (1u << bit);
    u64 na = a, nb = b;
    if ((tid >> bit) & 1)
        na = xchg[1][partner];   // upper half: a <- partner's b
    else
        nb = xchg[0][partner];   // lower half: b <- partner's a
    __syncthreads();
    a = na, b = nb;
#endif
}
template <bool NARROW>
__global__ void __launch_bounds__(512, 4)
    ntt_fwd_swap_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, RowMap map, const DevMod *__restrict__ mods,
                        const u64x2 *__restrict__ tw) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int LOGM = 13, T = 512, M = 1 << LOGM, CH = tile_chunks_c(LOGM, T);
    constexpr int NB = NARROW ? 1 : 0;
    const uint32_t tid = threadIdx.x;
    const uint32_t rowb = blockIdx.x;
    const uint32_t poly = to_sgpr(rowb / map.rows);
    const uint32_t r = map.row_begin + (rowb - poly * map.rows);
    const uint32_t mi = (uint32_t)(map.mod_offset + (int32_t)r);
    const DevMod md = mods[mi];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64 *src = in + (u64)poly * map.src_poly_stride + (u64)(map.src_row_fixed >= 0 ? (uint32_t)map.src_row_fixed : r) * M;
    u64 *dst = out + (u64)poly * map.dst_poly_stride + (u64)r * M;
    const u64x2 *twr = tw + (u64)mi * M;
    auto bfly = [&](u64 &x, u64 &y, const u64x2 wv, int stage) {
        if constexpr (NARROW)
            fwd_butterfly_narrow(x, y, wv.x, wv.y, pm, fwd_narrow_corrects(stage, NB));
        else
            fwd_butterfly(x, y, wv.x, wv.y, pm);
    };
    // ---- pass 1: stages 0-2 from global memory (two groups of 8 per thread)
    {
            FwdTw<3, LOGM, 0, T> none;
        fwd_pass<3, LOGM, 0, T, true, NB>(lds, twr, 1, pm, tid, none, [&](uint32_t i, uint32_t) { return src[i]; });
    }
    const uint32_t w = wave_uniform(tid >> 6), lane = tid & 63, l5 = lane >> 5, l4 = (lane >> 4) & 1;
    // per-lane twiddles of the two exchange stages, requested before the barrier
    u64x2 tw7[8], tw8[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) tw7[e] = twr[128 + (w << 4) + (l5 << 3) + e];   // 2^7 + (bits 12..6)
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {   // c = (f3 f1 f0): 2^8 + (bits 12..5), bit 5 = f3
        const uint32_t f3 = c >> 2, f10 = c & 3;
        tw8[c] = twr[256 + (w << 5) + (l5 << 4) + (l4 << 3) + (f10 << 1) + f3];
    }
    __syncthreads();
    // ---- pass 2: stages 3-8 in registers and across lanes
    u64 x[16];
    {
        const u64 *g = lds + padi((w << 10) + lane);
#pragma unroll
        for (uint32_t e = 0; e < 16; e++) x[e] = g[padi(e << 6)];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {   // stages 3..6: register bits 3..0 = position bits 9..6
        const uint32_t half = 16u >> (u + 1);
#pragma unroll
        for (uint32_t blk = 0; blk < (1u << u); blk++) {
            const u64x2 wv = twr[(8u << u) + (w << u) + blk];   // 2^(3+u) + (position >> (10 - u)): wave-uniform
#pragma unroll
            for (uint32_t j = 0; j < half; j++) bfly(x[blk * 2 * half + j], x[blk * 2 * half + j + half], wv, 3 + u);
        }
    }
    // stage 7 (position bit 5 = lane bit 5): register bit 3 <-> lane bit 5
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) lane_swap64(x[e], x[e + 8], tid, 5);
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) bfly(x[e], x[e + 8], tw7[e], 7);
    // stage 8 (position bit 4 = lane bit 4): register bit 2 <-> lane bit 4
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t f = ((c >> 2) << 3) | (c & 3);   // register index with bit 2 clear
        lane_swap64(x[f], x[f + 4], tid, 4);
    }
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t f = ((c >> 2) << 3) | (c & 3);
        bfly(x[f], x[f + 4], tw8[c], 8);
    }
    // registers (f3 f2 f1 f0) = position bits (5 4 7 6); lanes (l5 l4 l3..l0) = bits (9 8 3..0)
    {
        const uint32_t base = (w << 10) + (l5 << 9) + (l4 << 8) + (lane & 15);
#pragma unroll
        for (uint32_t f = 0; f < 16; f++) {
            const uint32_t pos = base + ((f & 3) << 6) + ((f >> 3) << 5) + (((f >> 2) & 1) << 4);
            wave_block_check<4>(0, tid, pos);
            lds[padi(pos)] = x[f];
        }
    }
    // ---- pass 3: stages 9-12, 16 consecutive coefficients per thread (wave-local exchange before it)
    FwdTw<4, LOGM, 9, T> tw3;
    fwd_tw_load(tw3, twr, 1, tid);
    wave_sync();
    fwd_pass<4, LOGM, 9, T, true, NB, NoSrc, 1, true>(lds, twr, 1, pm, tid, tw3);
    FHE_BARRIER();
    if constexpr (NARROW) {  // < 16p -> canonical
        const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) {
            return csub_n(csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2), p, pm.np);
        });
    } else {
        lds_to_tile<CH, M, T>(lds, dst, tid, [&](u64 v) { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); });
    }
}

// ------------------------------------------- fused key switch, two digits per round ----
// MEASURED ALTERNATIVE, not part of the default build (compile with -DFHE_KS_EXPERIMENTS, select with
// FHE_KS_VARIANT; tools/ab_ks.sh).  Round 2 rebuilt the key switch around fewer barriers / more work per LDS
// exchange as this kernel; on the MI355X every variant lost to ks_fused_kernel (C2, 512 polynomials per launch):
//   ks_fused_kernel (one digit per round, row prefetch in registers, c1 accumulators in LDS)   0.514 ms
//   two digits per round, 1024 threads x 8 coefficients, both accumulator sets in registers     0.552 ms (13 VGPRs spilled)
//   one digit per round on this kernel's structure (MAC stages the coming row, no prefetch regs) 0.543 ms ( 7 VGPRs spilled)
//   two digits per round, 512 threads x 16 coefficients, radix-16, 236 VGPRs, 2 waves per SIMD  0.707 ms
// (profiles/r02_ks_variants.txt).  What decides is waves per SIMD and the registers the radix pass leaves: 32
// VGPRs of accumulators that stay live through the passes do not fit beside a radix-8 pass under the 128-VGPR
// cap of a 1024-thread workgroup, and trading waves for registers loses outright.
// Same contract as ks_fused_kernel.  The digits of one (ciphertext, key modulus) go through the loop TWO at a time:
// both lifted rows sit in LDS (two tiles), every radix pass runs on both tiles between the same pair of barriers
// with one set of addresses and twiddles (the modulus, hence the twiddles, is the same for all digits), and the
// Shoup MAC folds both into the accumulators.  Per digit this halves the workgroup barriers and the twiddle
// fetches and doubles the independent work a wave has between two LDS exchanges -- the workgroup is alone on its
// CU (LDS), so nothing else covers those gaps.  Both accumulator sets live in registers (2 x 2*CH u64; the tiles
// take the LDS the c1 accumulators had); the MAC loop stages the coming round's lifted rows into the chunks it has
// just consumed.  An odd digit count ends with one single-tile round.
// NARROW (key moduli below 2^60): the accumulators are left unreduced -- each lazy product is below 2p, so up to
// 8 of them fit below 16p < 2^64 -- and are folded back below 2p only every 7 digits (never for <= 8 digits).
template <int V>
struct int_c {
    static constexpr int value = V;
};
constexpr bool ks_pair_ok_c(int logn) { return tile_chunks_c(logn, ks_threads_c(logn)) > 0 && logn <= 13; }
// CPT: coefficients per thread and tile.  8: N/8 threads, radix-8 passes, 128 VGPRs (4 waves per SIMD);
// 16: N/16 threads, radix-16 passes, both accumulator sets and a 16-point group in a 256-VGPR budget (2 waves per SIMD).
constexpr int ks_pair_threads_c(int logn, int cpt) { return cpt == 8 ? ks_threads_c(logn) : ((1 << logn) / 16 > 64 ? (1 << logn) / 16 : 64); }
template <int LOGN, bool NARROW = false, int CPT = 8, int MAXNT = 2>
__global__ void __launch_bounds__(ks_pair_threads_c(LOGN, CPT), CPT == 8 ? 4 : 2)
    ks_pair_kernel(const u64 *__restrict__ pin, u64 src_poly_stride, u64 *__restrict__ out0, u64 *__restrict__ out1,
                   u64 out_poly_stride, const u64 *__restrict__ addend0, const u64 *__restrict__ addend1,
                   u64 addend_poly_stride, const u64 *__restrict__ k0, const u64 *__restrict__ k0s,
                   const u64 *__restrict__ k1, const u64 *__restrict__ k1s, const DevMod *__restrict__ mods,
                   const u64x2 *__restrict__ tw, uint32_t ndigits, uint32_t lk) {
    FHE_DYN_SMEM(u64, lds);
    constexpr int T = ks_pair_threads_c(LOGN, CPT);
    constexpr int GM = CPT == 8 ? KS_GMAX : GMAX;
    constexpr int N = 1 << LOGN;
    constexpr int CH = tile_chunks_c(LOGN, T);
    static_assert(CH > 0, "ks_pair_kernel needs at least one 16-byte chunk per thread");
    constexpr uint32_t TW = N + (N >> 4) + 2;   // u64 words per tile (= lds_words(N))
    const uint32_t tid0 = threadIdx.x;
    const uint32_t b = to_sgpr(blockIdx.x / lk), j = blockIdx.x - b * lk;
    const DevMod md = mods[j];
    const u64 p = md.p, p2 = md.p2;
    const PM pm = make_pm(md);
    const u64x2 *twr = tw + (u64)j * N;
    u64 acc0[2 * CH], acc1[2 * CH];
#pragma unroll
    for (int e = 0; e < 2 * CH; e++) acc0[e] = acc1[e] = 0;
    // RNS digits whose source moduli are below 4 q_j only (lift_mode 1 or 2 of ks_fused_kernel; the host sends
    // everything else -- base-2^k digits, moduli of very different widths -- to that kernel): two conditional
    // subtractions lift a residue, no branch in the loops.
    auto lift = [&](u64 v) -> u64 { return csub_n(csub_n(v, p2, pm.np2), p, pm.np); };
    const u64 *const src0 = pin + (u64)b * src_poly_stride;
    // The lifted rows of the COMING round are staged by the MAC loop of the current one: once a thread has consumed
    // chunk c of tile d it owns that slot (nobody else touches a thread's 16-byte chunks outside the passes), so it
    // writes the lifted chunk of the digit that takes tile d next -- no prefetch registers, the row's load latency
    // hides behind the Shoup MACs, and no barrier is needed between the MAC and the next round's first pass other
    // than the one that closes the staging.
    auto stage = [&](int tile, int c, uint32_t tid, u64x2 raw) {
        const uint32_t e = 2 * (c * T + tid);
        lds[tile * TW + padi(e)] = lift(raw.x);
        lds[tile * TW + padi(e + 1)] = lift(raw.y);
    };
    auto row_ptr = [&](uint32_t digit) { return reinterpret_cast<const u64x2 *>(src0 + (u64)digit * N); };
    {   // first round: straight from global memory
        u64x2 raw[2][CH];
#pragma unroll
        for (int d = 0; d < MAXNT; d++)
            if ((uint32_t)d < ndigits) {
#pragma unroll
                for (int c = 0; c < CH; c++) raw[d][c] = row_ptr(d)[c * T + tid0];
            }
#pragma unroll
        for (int d = 0; d < MAXNT; d++)
            if ((uint32_t)d < ndigits) {
#pragma unroll
                for (int c = 0; c < CH; c++) stage(d, c, tid0, raw[d][c]);
            }
    }
    uint32_t i = 0;        // first digit of the round
    uint32_t since = 0;    // NARROW: accumulators are below 2p * max(since, 1)
    auto round = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;
        const uint32_t tid = opaque(tid0);
        FHE_BARRIER();   // the round's tiles are complete
        if constexpr (NARROW) {
            if (since + NT > 8) {   // (block-uniform) fold the accumulators back below 2p
                const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
#pragma unroll
                for (int e = 0; e < 2 * CH; e++) {
                    acc0[e] = csub_n(csub_n(csub_n(acc0[e], p8, np8), p4, np4), p2, pm.np2);
                    acc1[e] = csub_n(csub_n(csub_n(acc1[e], p8, np8), p4, np4), p2, pm.np2);
                }
                since = 1;
            }
            since += NT;
        }
        ntt_fwd_lds<LOGN, T, GM, false, false, (NARROW ? 1 : 0), NoSrc, KS_LATE, NT>(lds, twr, 1, pm, tid, NoSrc{}, TW);
        // the key words of the first chunk are requested before the barrier that ends the transform
        const u64 koff0 = ((u64)i * lk + j) * N;
        u64x2 kq[4];
        const uint32_t tid_m = opaque(tid0);   // (a fresh opaque copy: the MAC's address arithmetic stays below the passes)
        {
            const uint32_t ci = tid_m;
            kq[0] = reinterpret_cast<const u64x2 *>(k0 + koff0)[ci], kq[1] = reinterpret_cast<const u64x2 *>(k0s + koff0)[ci];
            kq[2] = reinterpret_cast<const u64x2 *>(k1 + koff0)[ci], kq[3] = reinterpret_cast<const u64x2 *>(k1s + koff0)[ci];
        }
        FHE_BARRIER();
        // MAC of tile d; STAGE: the digit `nxt` takes the tile in the coming round and its lifted row is written
        // into every chunk right after the chunk has been consumed
        auto mac_tile = [&](int d, auto stc, uint32_t nxt) {
            constexpr bool STAGE = decltype(stc)::value != 0;
            const u64 koff = ((u64)(i + d) * lk + j) * N;
            const u64x2 *a0 = reinterpret_cast<const u64x2 *>(k0 + koff), *a0s = reinterpret_cast<const u64x2 *>(k0s + koff);
            const u64x2 *a1 = reinterpret_cast<const u64x2 *>(k1 + koff), *a1s = reinterpret_cast<const u64x2 *>(k1s + koff);
            const u64x2 *nrow = row_ptr(STAGE ? nxt : 0);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t ci = c * T + tid_m;
                u64x2 q0, q0s, q1, q1s, raw = u64x2{0, 0};
                if (d == 0 && c == 0) {
                    q0 = kq[0], q0s = kq[1], q1 = kq[2], q1s = kq[3];
                } else {
                    q0 = a0[ci], q0s = a0s[ci], q1 = a1[ci], q1s = a1s[ci];
                }
                if constexpr (STAGE) raw = nrow[ci];
                const u64 vx = lds[d * TW + padi(2 * ci)], vy = lds[d * TW + padi(2 * ci + 1)];  // any u64: fine for Shoup
                if constexpr (NARROW) {
                    acc0[2 * c] += mul_shoup_lazy_n(vx, q0.x, q0s.x, pm.np);
                    acc0[2 * c + 1] += mul_shoup_lazy_n(vy, q0.y, q0s.y, pm.np);
                    acc1[2 * c] += mul_shoup_lazy_n(vx, q1.x, q1s.x, pm.np);
                    acc1[2 * c + 1] += mul_shoup_lazy_n(vy, q1.y, q1s.y, pm.np);
                } else {
                    acc0[2 * c] = csub_n(acc0[2 * c] + mul_shoup_lazy_n(vx, q0.x, q0s.x, pm.np), p2, pm.np2);
                    acc0[2 * c + 1] = csub_n(acc0[2 * c + 1] + mul_shoup_lazy_n(vy, q0.y, q0s.y, pm.np), p2, pm.np2);
                    acc1[2 * c] = csub_n(acc1[2 * c] + mul_shoup_lazy_n(vx, q1.x, q1s.x, pm.np), p2, pm.np2);
                    acc1[2 * c + 1] = csub_n(acc1[2 * c + 1] + mul_shoup_lazy_n(vy, q1.y, q1s.y, pm.np), p2, pm.np2);
                }
                if constexpr (STAGE) stage(d, c, tid_m, raw);
                sched_fence();  // one chunk's loads (key words + the coming row: 20 VGPRs) at a time
            }
        };
#pragma unroll
        for (int d = 0; d < NT; d++) {
            const uint32_t nxt = i + NT + d;          // the digit that takes tile d in the coming round
            if (nxt < ndigits)                        // (block-uniform)
                mac_tile(d, int_c<1>{}, nxt);
            else
                mac_tile(d, int_c<0>{}, 0u);
        }
        i += NT;
    };
    if constexpr (MAXNT >= 2) {
        while (i + 1 < ndigits) round(int_c<2>{});
        if (i < ndigits) round(int_c<1>{});
    } else {
        while (i < ndigits) round(int_c<1>{});   // (MAXNT = 1: one digit per round, tile 1 unused)
    }
    const uint32_t tid = opaque(tid0);  // keeps the epilogue's address arithmetic below the digit loop
    const u64 ooff = (u64)b * out_poly_stride + (u64)j * N;
    const u64 aoff = (u64)b * addend_poly_stride + (u64)j * N;
    u64x2 *o0 = reinterpret_cast<u64x2 *>(out0 + ooff), *o1 = reinterpret_cast<u64x2 *>(out1 + ooff);
    const u64x2 *d0 = reinterpret_cast<const u64x2 *>(addend0 ? addend0 + aoff : nullptr);
    const u64x2 *d1 = reinterpret_cast<const u64x2 *>(addend1 ? addend1 + aoff : nullptr);
    auto canon = [&](u64 v) -> u64 {
        if constexpr (NARROW) {   // below 16p
            const u64 p4 = p2 << 1, p8 = p2 << 2, np4 = pm.np2 << 1, np8 = pm.np2 << 2;
            v = csub_n(csub_n(csub_n(v, p8, np8), p4, np4), p2, pm.np2);
        }
        return csub_n(v, p, pm.np);
    };
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t ci = c * T + tid;
        u64x2 r0, r1;
        r0.x = canon(acc0[2 * c]);
        r0.y = canon(acc0[2 * c + 1]);
        r1.x = canon(acc1[2 * c]);
        r1.y = canon(acc1[2 * c + 1]);
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

