// knobs.hpp -- the compile-time switches of measured alternatives, all in one place.
//
// Every knob below selects between EXACT implementations (same residues, different instruction streams); the values
// written here are the ones the measurements of rounds 1-2 picked.  The release build
// (__graft_entry__.build()) pins them: defining any of them on the command line without -DFHE_LAB is a compile error,
// so a -D typo cannot ship a different kernel.  -DFHE_LAB builds (tools/ab_*.sh; never loaded by the package) may
// override them, additionally compile the rejected kernels under tools/lab/ (outside the product tree; tools/build_variant.sh adds -I tools) and read FHE_LAB_* environment switches.
// The release library reads no environment variable at all.
#pragma once

#if !defined(FHE_LAB)
#if defined(FHE_SENS) || defined(FHE_MAD_CARRY) || defined(FHE_MAD_CROSS) || defined(FHE_APPROX_SHOUP) || defined(FHE_KS_LATE) ||             \
    defined(FHE_KS_TWPF) || defined(FHE_KS_PERSIST14) || defined(FHE_KS_KPF_CHUNKS) || defined(FHE_TENSOR_TW_EARLY) ||  \
    defined(FHE_KS_EXPERIMENTS) || defined(FHE_PHASE_TIMING) || defined(FHE_LDS_PAD) || defined(FHE_NO_WAVE_SYNC) || \
    defined(FHE_DIAG_NO_SGPR_ASM) || defined(FHE_KS_HALF13) || defined(FHE_KS_SPLIT_XCD) || defined(FHE_STREAM_NT) || \
    defined(FHE_MUL_DIRFLAGS) || defined(FHE_MUL_MERGED_EXT) || defined(FHE_PIPE_NT) || \
    defined(FHE_KS_HALF15) || defined(FHE_FWD_DIRECT_STORE) || defined(FHE_KS12_F64_T256)
#error "kernel-variant macros are lab-only: add -DFHE_LAB (the release build pins every knob, see knobs.hpp)"
#endif
#endif
#if defined(FHE_SENS)
#error "FHE_SENS (timing-only builds that computed wrong residues) was removed in round 3; see commit db30352"
#endif

// High products through v_mad_u64_u32's carry-out (zq_dev.hpp); 0: the compiler's generic expansion.
#ifndef FHE_MAD_CARRY
#define FHE_MAD_CARRY 1
#endif
// Low word of the lazy Shoup products: the four 32-bit cross products summed by v_mad_u64_u32 used as a 32-bit
// multiply-add (zq_dev.hpp shoup_lo).  Round 3: 4.3-4.7 % fewer VALU instructions in the forward transform, 2-3 % in
// the inverse / tensor kernels -- and no change in kernel time (profiles/r03_shoup_lo_ab.txt): per
// profiles/r03_ubench_issue.jsonl a v_mad_u64_u32 occupies a SIMD for 5.2 cycles against 4.4 for v_mul_lo_u32 and 4.7
// for v_add3_u32, so 4 + 1 of the former cost what 4 + 2 of the latter do; at N = 16384 the key switch spills (C3 -2.5 %).
#ifndef FHE_MAD_CROSS
#define FHE_MAD_CROSS 0
#endif
// Forward transform (ntt_kernel): the last (stride-1) pass stores its canonical results straight from the registers --
// 2^G consecutive coefficients per thread, 16-byte stores at a 64-byte lane stride -- instead of one more trip through the
// tile (LDS write, barrier, coalesced read-back).  Round 6 (VERDICT r05 #4a), measured (profiles/r06_fwd_direct_store_ab_rejected.jsonl, same box, alternating lab builds): 60-bit
// rows 0.3594 -> 0.3661 ms per 8,192 row transforms (+1.9 %), 62-bit rows unchanged, the F64 instances 0.334 -> 0.375 ms (+12 %),
// C2 multiply -0.2 %: the 64-byte lane stride makes four partial-line stores of what the tile read-back writes as one full line
// per lane pair; the saved LDS round trip does not pay for it.  Off.
#ifndef FHE_FWD_DIRECT_STORE
#define FHE_FWD_DIRECT_STORE 0
#endif
// F64 key switch at N = 4096 in the 16-coefficients-per-thread geometry of N = 8192's F64 instance (256 threads, accumulators in
// registers, four workgroups per CU) instead of 512 threads x 8 with the c1 accumulators in LDS.  Measured and rejected
// (profiles/r06_l_f64_geometry_ab.jsonl, release builds alternating, same digests): relinearise of 1,024 / 256 at the stock
// n = 4096 set -2 %, of 64 (192 workgroups, less than one per CU: the slower single workgroup is what shows) +23 %.
#ifndef FHE_KS12_F64_T256
#define FHE_KS12_F64_T256 0
#endif
// LDS tile padding (kernels_common.hpp padi): 0 = one extra word every 16 (two-way conflicted in every pass pattern, 2 KiB
// smaller per tile), 1 = three extra words every 32 (conflict-free in seven of the nine patterns, tools/lds_pad_search.py).
#ifndef FHE_LDS_PAD
#define FHE_LDS_PAD 0
#endif
// Narrow (< 2^60) butterflies take the Shoup quotient from three partial products (zq_dev.hpp).
#ifndef FHE_APPROX_SHOUP
#define FHE_APPROX_SHOUP 1
#endif
// Key-switch transforms: the wider radix passes last, so that the trailing exchanges are wave-local (kernels.hpp).
#ifndef FHE_KS_LATE
#define FHE_KS_LATE true
#endif
// Key switch at N = 8192: per-lane twiddles requested one pass ahead (measured: no change).
#ifndef FHE_KS_TWPF
#define FHE_KS_TWPF false
#endif
// Resident key-switch workgroups at N = 16384 as well (measured: 2 % slower at C3).
#ifndef FHE_KS_PERSIST14
#define FHE_KS_PERSIST14 0
#endif
// Chunks of key words requested before the barrier that ends a digit's transform.
#ifndef FHE_KS_KPF_CHUNKS
#define FHE_KS_KPF_CHUNKS 2
#endif
// Fused tensor + iNTT: first-pass twiddles requested half-way through the products.
#ifndef FHE_TENSOR_TW_EARLY
#define FHE_TENSOR_TW_EARLY 1
#endif
// Non-temporal loads / stores on the streaming operands of the element-wise kernels (kernels_common.hpp load_stream).
#ifndef FHE_STREAM_NT
#define FHE_STREAM_NT 1
#endif
// Multiply pipeline: every kernel starts on what its producer wrote last (tensor+iNTT descending, down-scaler
// ascending, the forward transform of c0, c1 descending); 0: round 3's order (only the scalers run backwards).
// Measured (profiles/r04_merged_ext_dirflags_ab.txt, same box, alternating): no change at any batch size -- off.
#ifndef FHE_MUL_DIRFLAGS
#define FHE_MUL_DIRFLAGS 0
#endif
// Multiply pipeline: the lhs and rhs extensions as one pass of three launches when both use the same extender.
// Measured (same file): batch 1024 unchanged (power-limited), batch 64 -5 %, batch 16 -12.5 % -- on.
#ifndef FHE_MUL_MERGED_EXT
#define FHE_MUL_MERGED_EXT 1
#endif
// Multiply pipeline: non-temporal LOADS where a kernel reads data for the last time (the scalers' input columns, the
// forward transforms' rows, the inverse transform's tile): bit 0 scalers, bit 1 forward NTT, bit 2 inverse NTT.
// Measured (profiles/r04_pipe_nt_ab.txt): mask 3 is 0.6 % ahead of none in six of six same-box comparisons at
// C2 / 1024, mask 4 (the inverse transform of the inputs, which the tensor kernel reads again) is noise -- 3.
#ifndef FHE_PIPE_NT
#define FHE_PIPE_NT 3
#endif
// Key switch at N = 32768 / 65536: 0 ks_fused_split_kernel (8192-point parts: two / three folded stages, 3 / 7 Shoup
// products per coefficient in the loader); 1 / 2 ks_fused_kernel<14> on 16384-point parts with ONE / two folded stages
// (1 / 3 products; generic loader / RNS loader for residue-row digits).  Measured at C5 (N = 32768, 16 moduli,
// profiles/r04_ks_half15_ab.txt, same box, three rounds each): relinearise 1.46 -> 1.29 (1) -> 1.25 ms (2) at batch 16,
// 5.00 -> 4.40 -> 4.23 ms at batch 64; the level-0 multiply step 3.43 -> 3.29 -> 3.23 ms -- 2.
#ifndef FHE_KS_HALF15
#define FHE_KS_HALF15 2
#endif
