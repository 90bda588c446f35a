// lab_engine.hpp -- host-side dispatch of the rejected kernel variants (lab/lab_kernels.hpp).  Lab builds only:
// engine.hpp includes this file under -DFHE_LAB and the release library contains none of it.
//   FHE_LAB_NTT_SWAP=1        forward NTT at N = 8192 with in-wave stages by lane exchange (ntt_fwd_swap_kernel)
//   FHE_LAB_NTT_CPT8=3 | 32   forward NTT at N = 8192, 8 coefficients per thread (ntt_fwd8_kernel; radix 8 / mixed)
//   FHE_LAB_KS_VARIANT=1|2|3  key switch on ks_pair_kernel (two digits per round / 16 coefficients per thread /
//                             one digit per round on that structure)
#pragma once

namespace fhe {

inline bool lab_try_ntt_fwd(const Ctx &c, unsigned rows_total, bool narrow, const u64 *in, u64 *out, const k::RowMap &map,
                            hipStream_t s) {
    if (c.logn != 13) return false;
    static const bool swap_variant = FHE_LAB_INT("NTT_SWAP", 0) != 0;
    static const int cpt8 = FHE_LAB_INT("NTT_CPT8", 0);
    const size_t lds = k::lds_words(1u << 13) * sizeof(u64);
    if (swap_variant) {
        if (narrow) {
            allow_big_lds((k::ntt_fwd_swap_kernel<true>), lds);
            FHE_LAUNCH("ntt_fwd", (k::ntt_fwd_swap_kernel<true>), dim3(rows_total), dim3(512), lds, s, in, out, map,
                       c.dmods(), c.dtw());
        } else {
            allow_big_lds((k::ntt_fwd_swap_kernel<false>), lds);
            FHE_LAUNCH("ntt_fwd", (k::ntt_fwd_swap_kernel<false>), dim3(rows_total), dim3(512), lds, s, in, out, map,
                       c.dmods(), c.dtw());
        }
        return true;
    }
    if (cpt8) {
#define FHE_NTT8(NW, GMV)                                                                                          \
    do {                                                                                                           \
        allow_big_lds((k::ntt_fwd8_kernel<NW, GMV>), lds);                                                         \
        FHE_LAUNCH("ntt_fwd", (k::ntt_fwd8_kernel<NW, GMV>), dim3(rows_total), dim3(1024), lds, s, in, out, map,   \
                   c.dmods(), c.dtw());                                                                  \
    } while (0)
        if (cpt8 == 3) {
            if (narrow) FHE_NTT8(true, 3); else FHE_NTT8(false, 3);
        } else {
            if (narrow) FHE_NTT8(true, k::GM_MIXED); else FHE_NTT8(false, k::GM_MIXED);
        }
#undef FHE_NTT8
        return true;
    }
    return false;
}

template <int LOGN>
inline bool lab_try_ks_pair(const Ksk &k_, const u64 *p, u64 p_stride, u64 *o0, u64 *o1, u64 out_stride, const u64 *a0,
                            const u64 *a1, u64 a_stride, size_t npolys, hipStream_t s) {
    static const int variant = FHE_LAB_INT("KS_VARIANT", 0);
    if constexpr (k::ks_pair_ok_c(LOGN)) {
        const Ctx &kc = *k_.ksk_ctx;
        const uint32_t lm = k_.lift_mode();   // 1 / 2: RNS digits below 2 / 4 q_j (what the pair kernel lifts)
        if (variant >= 1 && variant <= 3 && k_.ndigits >= 2 && (lm == 1 || lm == 2)) {
            const size_t lds2 = 2 * k::lds_words(1u << LOGN) * sizeof(u64);
            bool nrw = !FHE_LAB_FLAG("NO_NARROW");
            for (u64 q : kc.moduli) nrw = nrw && (q >> 60) == 0;
#define FHE_KS_PAIR_LAUNCH(NW, CPT, ...)                                                                             \
    allow_big_lds((k::ks_pair_kernel<LOGN, NW, CPT, ##__VA_ARGS__>), lds2);                                          \
    FHE_LAUNCH("key_switch_fused", (k::ks_pair_kernel<LOGN, NW, CPT, ##__VA_ARGS__>), dim3((unsigned)(npolys * kc.L)), \
               dim3(k::ks_pair_threads_c(LOGN, CPT)), lds2, s, p, p_stride, o0, o1, out_stride, a0, a1, a_stride,    \
               k_.c0.p, k_.c0s.p, k_.c1.p, k_.c1s.p, kc.dmods(), kc.dtw(), (uint32_t)k_.ndigits, (uint32_t)kc.L)
            if (variant == 3) {
                if (nrw) {
                    FHE_KS_PAIR_LAUNCH(true, 8, 1);
                } else {
                    FHE_KS_PAIR_LAUNCH(false, 8, 1);
                }
            } else if (variant == 2 && LOGN >= 11) {
                if (nrw) {
                    FHE_KS_PAIR_LAUNCH(true, 16);
                } else {
                    FHE_KS_PAIR_LAUNCH(false, 16);
                }
            } else if (nrw) {
                FHE_KS_PAIR_LAUNCH(true, 8);
            } else {
                FHE_KS_PAIR_LAUNCH(false, 8);
            }
#undef FHE_KS_PAIR_LAUNCH
            return true;
        }
    }
    return false;
}

}  // namespace fhe
