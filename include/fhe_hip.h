/* fhe_hip.h -- C ABI of the MI355X-native RNS polynomial engine for fhe.rs' BFV hot path.
 *
 * This is the drop-in boundary: the entry points a `hip` cargo feature of fhe-math / fhe
 * would bind (see INTEGRATION.md for the Rust `extern "C"` block and the call-site patches).
 * fhe.rs has no FFI of its own; each entry point cites the Rust item it replaces
 * (paths relative to the reference repo: M/ = crates/fhe-math/src, F/ = crates/fhe/src).
 *
 * Conventions
 *  - All coefficient buffers are caller-owned, contiguous row-major u64 `[batch][...][L][N]`,
 *    exactly `fhe_math::rq::Poly`'s `Array2<u64>` layout (M/rq/mod.rs:126-133, 189-204) with
 *    outer batch dimensions.  Inputs and outputs are canonical residues in [0, q_i).
 *  - Plain entry points take HOST pointers and are synchronous.  `_dev` twins take DEVICE
 *    pointers plus a `hipStream_t` (passed as `void*`) and are stream-ordered.  Device buffers and
 *    streams come from the "device memory and streams" block below (or from any other HIP
 *    allocator in the process: the engine only sees pointers), so a host written in C, Rust, Go ...
 *    keeps polynomials resident on the GPU between calls without linking HIP itself.
 *  - Handles are opaque, immutable after creation and may be shared by concurrent callers
 *    working on different buffers (matches `Arc<Context>`, M/rq/context.rs:8-19).
 *  - Every function returns 0 (FHE_OK) or a negative `fhe_status`; nothing throws or aborts
 *    across the boundary.  `fhe_last_error()` gives a thread-local message.
 *  - Results are bit-identical to the reference CPU path on the same inputs and tables.
 */
#ifndef FHE_HIP_H
#define FHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t fhe_status;

/* Status codes: 1:1 with the `Result` variants used on the path
 * (M/errors.rs:12-118, F/errors.rs:18-70) plus runtime/argument errors. */
enum {
    FHE_OK = 0,
    FHE_E_ARG = -1,                           /* null pointer, zero size, bad flag               */
    FHE_E_HIP = -2,                           /* HIP runtime error (message in fhe_last_error)   */
    FHE_E_INVALID_MODULUS = -3,               /* Error::InvalidModulus                            */
    FHE_E_INVALID_DEGREE = -4,                /* Error::InvalidPolynomialDegree                   */
    FHE_E_NTT_UNAVAILABLE = -5,               /* Error::NttOperatorUnavailable                    */
    FHE_E_CONTEXT_MISMATCH = -6,              /* Error::PolynomialContextMismatch                 */
    FHE_E_DEGREE_MISMATCH = -7,               /* Error::DegreeMismatch                            */
    FHE_E_NO_MORE_CONTEXT = -8,               /* Error::NoMoreContext                             */
    FHE_E_CONTEXT_NOT_REACHABLE = -9,         /* Error::ContextNotReachable                       */
    FHE_E_INVALID_SUBSTITUTION_EXPONENT = -10,/* Error::InvalidSubstitutionExponent               */
    FHE_E_PARAMETER_MISMATCH = -11,           /* fhe::Error::ParameterMismatch                    */
    FHE_E_INVALID_LEVEL = -12,                /* fhe::Error::InvalidLevel / InvalidContextLevel   */
    FHE_E_MUL_POLY_COUNT = -13,               /* CiphertextError::MultiplicationPolynomialCount   */
    FHE_E_EMPTY_MODULI = -14,                 /* Error::EmptyModuli                               */
    FHE_E_NON_COPRIME = -15,                  /* Error::NonCoprimeModuli                          */
    FHE_E_NOT_ENOUGH_PRIMES = -16,            /* ParametersError::NotEnoughPrimes                 */
    FHE_E_KEYSWITCH_UNSUPPORTED = -17,        /* EvaluationKeyError::KeySwitchingNotSupported     */
    FHE_E_NO_DEVICE = -18,                    /* compute call on a host-only (device = -1) handle */
    FHE_E_EMPTY_DOT_PRODUCT = -19,            /* Error::EmptyDotProduct / DotProductError::EmptyInput */
    FHE_E_INVALID_EXPANSION_SIZE = -20,       /* EvaluationKeyError::InvalidExpansionSize          */
    FHE_E_EXPANSION_UNSUPPORTED = -21         /* EvaluationKeyError::Unsupported{Expansion} / Missing{GaloisKey} */
};

typedef struct fhe_ctx fhe_ctx;       /* == rq::Context on one device   (M/rq/context.rs:9-19)        */
typedef struct fhe_scaler fhe_scaler; /* == rq::scaler::Scaler          (M/rq/scaler.rs:18-23)        */
typedef struct fhe_ksk fhe_ksk;       /* == bfv::KeySwitchingKey        (F/bfv/keys/key_switching_key.rs:22-46) */
typedef struct fhe_mul fhe_mul;       /* == bfv::Multiplicator          (F/bfv/ops/mul.rs:21-32)      */
typedef struct fhe_params fhe_params; /* == bfv::BfvParameters' level tables (F/bfv/parameters.rs:83-117) */

const char *fhe_last_error(void);
const char *fhe_version(void);
/* Number of visible HIP devices (0 without a GPU).  Never fails. */
int fhe_device_count(void);

/* -------------------------------------------------- device memory and streams ---- */
/* What a non-HIP host needs to use the `_dev` entry points: `rq::Poly`'s `coefficients: Array2<u64>`
 * (M/rq/mod.rs:126-133) gets a device-resident shadow that lives across calls -- upload once,
 * chain multiply -> relinearise -> rotate -> switch_down on the device, download lazily (the Rust side of
 * this is `DevicePoly` / `DeviceCiphertext` in rust/fhe-math-hip).
 *  - A stream is a plain `hipStream_t` handed out as `void *` (non-blocking with respect to the null
 *    stream); NULL everywhere means the device's null stream.  Work on one stream runs in order.
 *  - `*_async` calls only enqueue: with pageable host memory HIP stages the bytes itself, with pinned
 *    host memory (fhe_host_alloc) the caller must keep the host buffer untouched until the stream
 *    reaches that point (fhe_stream_sync).  The plain twins additionally wait for the stream.
 *  - fhe_stream_destroy waits for the stream, then frees what the engine kept for it (its internal
 *    second stream and idle scratch blocks).  Handles (contexts, keys ...) are not tied to a stream. */
fhe_status fhe_buf_alloc(int device, size_t bytes, void **out);      /* hipMalloc on `device`           */
fhe_status fhe_buf_free(void *buf);                                   /* NULL is a no-op                  */
/* Stream-ordered twins (hipMallocFromPoolAsync / hipFreeAsync on a memory pool PRIVATE to this library, one per device,
 * told to keep freed blocks -- the device's default pool is left alone): the block may be used by work enqueued on `stream` after the call, and a free takes effect
 * behind the work already enqueued on `stream` -- no device synchronisation, unlike hipMalloc / hipFree.  Using the
 * block on another stream needs the caller's own ordering (events).  fhe_buf_free also accepts such a block (and waits);
 * fhe_buf_free_async is for blocks from fhe_buf_alloc_async only.  fhe_workspace_trim returns the pool's idle blocks. */
fhe_status fhe_buf_alloc_async(int device, size_t bytes, void *stream, void **out);
fhe_status fhe_buf_free_async(void *buf, void *stream);              /* NULL is a no-op                  */
fhe_status fhe_buf_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream);
fhe_status fhe_buf_upload_async(void *dst_dev, const void *src_host, size_t bytes, void *stream);
fhe_status fhe_buf_download(void *dst_host, const void *src_dev, size_t bytes, void *stream);
fhe_status fhe_buf_download_async(void *dst_host, const void *src_dev, size_t bytes, void *stream);
fhe_status fhe_buf_copy_async(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
fhe_status fhe_buf_zero_async(void *buf, size_t bytes, void *stream);  /* e.g. wiping a secret-dependent buffer */
fhe_status fhe_host_alloc(size_t bytes, void **out);                 /* pinned host memory (true async copies) */
fhe_status fhe_host_free(void *p);
fhe_status fhe_stream_create(int device, void **stream_out);
fhe_status fhe_stream_sync(void *stream);
fhe_status fhe_stream_destroy(void *stream);
fhe_status fhe_device_sync(int device);
fhe_status fhe_device_mem_info(int device, size_t *free_bytes, size_t *total_bytes);  /* either may be NULL */

/* ------------------------------------------------------------------ rq::Context ---- */
/* Context::new (M/rq/context.rs:42-92).  `device` >= 0 uploads tables to that GPU; -1 builds
 * a host-only handle (setup / introspection, no compute).  Tables (M/ntt/native.rs:16-26,
 * each `[nmoduli][degree]`, size_inv* `[nmoduli]`) are what the Rust host already holds in
 * its NttOperators; pass all NULL to let the engine derive its own primitive roots
 * (psi = g^((p-1)/2N) for the smallest g >= 2 that is a primitive 2N-th root).
 * The whole `next_context` chain (moduli prefixes) is built eagerly. */
fhe_status fhe_ctx_create(int device, size_t degree, size_t nmoduli, const uint64_t *moduli,
                          const uint64_t *omegas, const uint64_t *omegas_shoup,
                          const uint64_t *zetas_inv, const uint64_t *zetas_inv_shoup,
                          const uint64_t *size_inv, const uint64_t *size_inv_shoup, fhe_ctx **out);
void fhe_ctx_destroy(fhe_ctx *ctx);
/* Context::context_at_level (M/rq/context.rs:143-156): borrowed handle owned by `ctx`. */
fhe_status fhe_ctx_at_level(const fhe_ctx *ctx, size_t level, const fhe_ctx **out);
/* Context::niterations_to (M/rq/context.rs:117-141). */
fhe_status fhe_ctx_niterations_to(const fhe_ctx *from, const fhe_ctx *to, size_t *out);
size_t fhe_ctx_degree(const fhe_ctx *ctx);
size_t fhe_ctx_nmoduli(const fhe_ctx *ctx);
int fhe_ctx_device(const fhe_ctx *ctx);
fhe_status fhe_ctx_moduli(const fhe_ctx *ctx, uint64_t *out /* [nmoduli] */);
/* Introspection (tests / Rust host cross-check): which: 0 omegas, 1 omegas_shoup, 2 zetas_inv,
 * 3 zetas_inv_shoup (each [nmoduli][degree]); 4 size_inv, 5 size_inv_shoup (each [nmoduli]);
 * 6 inv_last_qi_mod_qj, 7 its Shoup twin (each [nmoduli-1], M/rq/context.rs:66-73). */
fhe_status fhe_ctx_get_table(const fhe_ctx *ctx, int which, uint64_t *out);

/* NttOperator::{forward,backward} via Poly::{ntt_forward,ntt_backward} (M/rq/mod.rs:335-354):
 * `polys` is [batch][L][N], transformed in place, canonical outputs.
 * The reference also has a lazy forward transform, NttOperator::forward_vt_lazy (M/ntt/native.rs:142-175, outputs in
 * [0, 2p)); its only caller on the path is the key switch's lift-and-transform (M/rq/mod.rs:563-586), which the
 * engine performs inside fhe_key_switch and its relatives.  No `lazy` flag is exported on purpose (SURVEY 8b's draft
 * signature had one): only canonical residues cross this ABI -- a value in [0, 2p) is not a function of the inputs
 * alone, and bit-exact parity is stated on canonical values. */
fhe_status fhe_ntt_forward(const fhe_ctx *ctx, uint64_t *polys, size_t batch);
fhe_status fhe_ntt_backward(const fhe_ctx *ctx, uint64_t *polys, size_t batch);
fhe_status fhe_ntt_forward_dev(const fhe_ctx *ctx, uint64_t *polys, size_t batch, void *stream);
fhe_status fhe_ntt_backward_dev(const fhe_ctx *ctx, uint64_t *polys, size_t batch, void *stream);

/* AddAssign / SubAssign / MulAssign<&Poly<Ntt>> / Neg (M/rq/ops.rs:10-206, 354-418): a op= b. */
fhe_status fhe_poly_add(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch);
fhe_status fhe_poly_sub(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch);
fhe_status fhe_poly_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch);
fhe_status fhe_poly_neg(const fhe_ctx *ctx, uint64_t *a, size_t batch);
/* MulAssign<&Poly<NttShoup>> (M/rq/ops.rs:208-245); Poly::compute_coefficients_shoup (M/rq/mod.rs:244-258). */
fhe_status fhe_poly_mul_shoup(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, const uint64_t *b_shoup,
                              size_t batch);
fhe_status fhe_poly_shoup(const fhe_ctx *ctx, const uint64_t *a, uint64_t *a_shoup, size_t batch);
fhe_status fhe_poly_add_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch, void *stream);
fhe_status fhe_poly_sub_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch, void *stream);
fhe_status fhe_poly_mul_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t batch, void *stream);
fhe_status fhe_poly_neg_dev(const fhe_ctx *ctx, uint64_t *a, size_t batch, void *stream);
fhe_status fhe_poly_mul_shoup_dev(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, const uint64_t *b_shoup,
                                  size_t batch, void *stream);

/* Poly::substitute (M/rq/mod.rs:360-412) with SubstitutionExponent::new (M/rq/mod.rs:99-121).
 * repr_is_ntt != 0: Ntt-domain permutation; 0: PowerBasis signed scatter.  in != out. */
fhe_status fhe_poly_substitute(const fhe_ctx *ctx, size_t exponent, const uint64_t *in, uint64_t *out,
                               size_t batch, int repr_is_ntt);
fhe_status fhe_poly_substitute_dev(const fhe_ctx *ctx, size_t exponent, const uint64_t *in, uint64_t *out,
                                   size_t batch, int repr_is_ntt, void *stream);

/* Poly::<PowerBasis>::switch_down (M/rq/mod.rs:433-492): in [batch][L][N] -> out [batch][L-1][N]. */
fhe_status fhe_poly_switch_down(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, size_t batch);
fhe_status fhe_poly_switch_down_dev(const fhe_ctx *ctx, const uint64_t *in, uint64_t *out, size_t batch,
                                    void *stream);

/* Poly::<PowerBasis>::switch_down_to (M/rq/mod.rs:498-507): niterations_to(to) applications of switch_down in one
 * call; in [batch][from.L][N] -> out [batch][to.L][N], PowerBasis.  `to` not on `from`'s chain ->
 * FHE_E_CONTEXT_NOT_REACHABLE; from == to copies. */
fhe_status fhe_poly_switch_down_to(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *in, uint64_t *out,
                                   size_t batch);
fhe_status fhe_poly_switch_down_to_dev(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *in, uint64_t *out,
                                       size_t batch, void *stream);

/* Rq wire format (`impl From<&Poly> for Rq` / parse_proto, M/rq/convert.rs:17-44, 46-99; Modulus::
 * serialize_vec / deserialize_vec, M/zq/mod.rs:783-793; fhe-util transcode_{to,from}_bytes,
 * crates/fhe-util/src/lib.rs:71-148): the `coefficients` bytes of the Rq message hold, per residue
 * row i, the N PowerBasis coefficients as bitlen(q_i - 1)-bit little-endian bit-packed integers, rows
 * concatenated (N % 8 == 0, so every row is byte aligned).  The protobuf envelope (representation tag,
 * degree) stays on the host; these calls move its payload.
 *   fhe_poly_serialized_size: bytes per polynomial = sum_i N * bitlen(q_i - 1) / 8.
 *   fhe_poly_serialize:   polys [batch][L][N] -> bytes [batch][size]; from_ntt != 0: the input is in Ntt
 *                         form and is taken to PowerBasis first (the wire is always PowerBasis).
 *   fhe_poly_deserialize: bytes [batch][size] -> polys [batch][L][N], coefficients taken verbatim
 *                         (TryConvertFrom<Vec<u64>>, convert.rs:148-160); to_ntt != 0: followed by
 *                         into_ntt (representation tag Ntt / NttShoup). */
size_t fhe_poly_serialized_size(const fhe_ctx *ctx);
fhe_status fhe_poly_serialize(const fhe_ctx *ctx, const uint64_t *polys, uint8_t *bytes, size_t batch, int from_ntt);
fhe_status fhe_poly_serialize_dev(const fhe_ctx *ctx, const uint64_t *polys, uint8_t *bytes, size_t batch,
                                  int from_ntt, void *stream);
fhe_status fhe_poly_deserialize(const fhe_ctx *ctx, const uint8_t *bytes, uint64_t *polys, size_t batch, int to_ntt);
fhe_status fhe_poly_deserialize_dev(const fhe_ctx *ctx, const uint8_t *bytes, uint64_t *polys, size_t batch,
                                    int to_ntt, void *stream);

/* ------------------------------------------------------------------ rq::Scaler ---- */
/* Scaler::new (M/rq/scaler.rs:27-52) + RnsScaler::new (M/rns/scaler.rs:79-175): the engine
 * derives gamma/omega/theta with its own big-integer code from the ScalingFactor
 * numerator/denominator, given as little-endian u64 limbs (ScalingFactor::new,
 * M/rns/scaler.rs:26-36; numerator == denominator <=> is_one). */
fhe_status fhe_scaler_create(const fhe_ctx *from, const fhe_ctx *to, const uint64_t *numerator,
                             size_t numerator_limbs, const uint64_t *denominator, size_t denominator_limbs,
                             fhe_scaler **out);
/* Same, but every RnsScaler field (M/rns/scaler.rs:52-72) is supplied by the Rust host. */
fhe_status fhe_scaler_create_from_constants(
    const fhe_ctx *from, const fhe_ctx *to, size_t number_common_moduli, int is_one,
    const uint64_t *gamma, const uint64_t *gamma_shoup,   /* [to]       */
    const uint64_t *omega, const uint64_t *omega_shoup,   /* [to][from] */
    uint64_t theta_gamma_lo, uint64_t theta_gamma_hi, int theta_gamma_sign,
    const uint64_t *theta_omega_lo, const uint64_t *theta_omega_hi, const uint8_t *theta_omega_sign, /* [from] */
    const uint64_t *theta_garner_lo, const uint64_t *theta_garner_hi, size_t theta_garner_shift,     /* [from] */
    fhe_scaler **out);
/* Switcher::new (M/rq/switcher.rs:17-22): factor to.modulus / from.modulus. */
fhe_status fhe_switcher_create(const fhe_ctx *from, const fhe_ctx *to, fhe_scaler **out);
void fhe_scaler_destroy(fhe_scaler *s);
size_t fhe_scaler_number_common_moduli(const fhe_scaler *s);
/* Introspection: which: 0 gamma, 1 gamma_shoup ([to]); 2 omega, 3 omega_shoup ([to][from]);
 * 4 theta_omega_lo, 5 theta_omega_hi, 6 theta_omega_sign, 7 theta_garner_lo, 8 theta_garner_hi ([from]);
 * 9 {theta_gamma_lo, theta_gamma_hi, theta_gamma_sign, theta_garner_shift, is_one} ([5]). */
fhe_status fhe_scaler_get_constants(const fhe_scaler *s, int which, uint64_t *out);
/* Scaler::scale (M/rq/scaler.rs:55-127) == Poly::scale / Poly::switch (M/rq/mod.rs:660-680):
 * in [batch][from.L][N] -> out [batch][to.L][N]; repr_is_ntt selects Ntt vs PowerBasis. */
fhe_status fhe_poly_scale(const fhe_scaler *s, const uint64_t *in, uint64_t *out, size_t batch, int repr_is_ntt);
fhe_status fhe_poly_scale_dev(const fhe_scaler *s, const uint64_t *in, uint64_t *out, size_t batch,
                              int repr_is_ntt, void *stream);

/* ------------------------------------------------------------ KeySwitchingKey ---- */
/* KeySwitchingKey{c0,c1: Box<[Poly<NttShoup>]>, ctx_ciphertext, ctx_ksk, log_base}
 * (F/bfv/keys/key_switching_key.rs:22-46).  c0/c1: [ndigits][Lk][N]; Shoup twins may be NULL
 * (then computed as floor(c * 2^64 / q), M/zq/mod.rs:195-199).  log_base != 0 selects the
 * single-modulus decomposition path (:323-362); otherwise ndigits must equal ct_ctx's L. */
fhe_status fhe_ksk_create(const fhe_ctx *ct_ctx, const fhe_ctx *ksk_ctx, size_t ndigits, const uint64_t *c0,
                          const uint64_t *c0_shoup, const uint64_t *c1, const uint64_t *c1_shoup,
                          size_t log_base, fhe_ksk **out);
/* Same with key polynomials already resident on the device (copied device-to-device). */
fhe_status fhe_ksk_create_dev(const fhe_ctx *ct_ctx, const fhe_ctx *ksk_ctx, size_t ndigits, const uint64_t *c0,
                              const uint64_t *c1, size_t log_base, void *stream, fhe_ksk **out);
void fhe_ksk_destroy(fhe_ksk *k);
/* Execution options of every key switch through this handle (key_switch, relinearise, Galois, RGSW, the relinearise
 * step of fhe_bfv_mul), kept on the handle like fhe_mul's: atomics read once per call; they only choose how the sum
 * of KeySwitchingKey::key_switch (F/bfv/keys/key_switching_key.rs:241-320) is evaluated, never its value.
 *   mode     FHE_KS_AUTO (default): the engine picks per shape and launch size; FHE_KS_FUSED: one kernel per
 *            (ciphertext, key modulus) that transforms the digits in LDS and multiplies them into register
 *            accumulators (Shoup twins) -- rows larger than LDS (N >= 32768) as 16384-point parts with the first one /
 *            two stages folded into the loader; FHE_KS_FUSED_SUB: the same on 8192-point sub-blocks for N >= 32768
 *            (twice the workgroups, each 1.7 x shorter: ahead only while a launch does not fill the device);
 *            FHE_KS_UNFUSED: the digit transforms of a launch as one batched NTT into scratch, then a streaming
 *            multiply-accumulate with lazy 128-bit sums (the pattern of F/bfv/ops/dot_product.rs:54-180) that reads
 *            the key without its twins; FHE_KS_UNFUSED_SUB: the same on 8192-point sub-block tiles at N = 16384 too.
 *            Decomposition keys (log_base != 0) always take the fused kernel.
 *   w_budget bytes of transformed digit rows one launch pair may have in flight (0 = default, 4 GiB).
 * What FHE_KS_AUTO does, from measurements on the MI355X (profiles/r04_ks_unfused_ab.txt, r04_ks_half15_ab.txt,
 * r04_final3_ks_modes_ab_c5.jsonl, r04_ks_small_launch_ab.txt, r04_ks_small_batches_all_modes.txt):
 *   - a launch that fills the device takes FHE_KS_FUSED at every size (N = 32768, 16 moduli, 16 polynomials: 1.08 vs
 *     1.23 ms unfused; N = 16384, 8 moduli, 512: 3.5 vs 4.3 ms; N = 8192, 4 moduli, 1024: 0.83 vs 0.97 ms);
 *   - a launch with few fused workgroups (2 x batch x key moduli [x N / 16384] <= compute units; RNS-digit keys with at
 *     least three digits, N >= 4096) takes FHE_KS_UNFUSED, whose first stage has digits x more tiles: one ciphertext at
 *     N = 16384, 8 moduli 0.220 -> 0.055 ms, at N = 32768, 16 moduli 0.32 -> 0.12 ms, at N = 8192, 4 moduli 0.054 -> 0.030;
 *   - other launches at N >= 32768 whose 8192-point sub-blocks fit the device at once (decomposition keys, fewer than
 *     three digits) take FHE_KS_FUSED_SUB. */
enum { FHE_KS_AUTO = 0, FHE_KS_FUSED = 1, FHE_KS_UNFUSED = 2, FHE_KS_UNFUSED_SUB = 3, FHE_KS_FUSED_SUB = 4 };
fhe_status fhe_ksk_set_mode(fhe_ksk *k, int mode, size_t w_budget);
fhe_status fhe_ksk_get_mode(const fhe_ksk *k, int *mode, size_t *w_budget);
/* KeySwitchingKey::key_switch / key_switch_assign (:241-320): p [batch][L][N] PowerBasis over
 * ct_ctx -> c0_out, c1_out [batch][Lk][N] Ntt over ksk_ctx. */
fhe_status fhe_key_switch(const fhe_ksk *k, const uint64_t *p, uint64_t *c0_out, uint64_t *c1_out, size_t batch);
fhe_status fhe_key_switch_dev(const fhe_ksk *k, const uint64_t *p, uint64_t *c0_out, uint64_t *c1_out,
                              size_t batch, void *stream);
/* RelinearizationKey::relinearizes (F/bfv/keys/relinearization_key.rs:69-102):
 * ct3 [batch][3][L][N] Ntt -> out [batch][2][L][N] Ntt (switch_down_to when the key level is higher). */
fhe_status fhe_bfv_relinearize(const fhe_ksk *rk, const uint64_t *ct3, uint64_t *out, size_t batch);
fhe_status fhe_bfv_relinearize_dev(const fhe_ksk *rk, const uint64_t *ct3, uint64_t *out, size_t batch,
                                   void *stream);
/* GaloisKey::relinearize / relinearize_into (F/bfv/keys/galois_key.rs:63-123), i.e.
 * EvaluationKey::rotates_columns_by (exponent 3^i mod 2N) / rotates_rows (exponent 2N-1)
 * (F/bfv/keys/evaluation_key.rs:110-170, 278-286): ct, out [batch][2][L][N] Ntt.
 * From N = 4096 on (key at the ciphertext's level) there is no separate permutation pass: the Ntt-domain substitution is
 * read as a gather inside the inverse transform and the key switch.  `out` may be `ct` itself (the reference's
 * `c1 = ek.rotates_rows(&c1)`): any overlap of `ct` and `out` takes the copying path (every read of `ct` precedes the first write of `out`). */
fhe_status fhe_bfv_galois(const fhe_ksk *gk, size_t exponent, const uint64_t *ct, uint64_t *out, size_t batch);
fhe_status fhe_bfv_galois_dev(const fhe_ksk *gk, size_t exponent, const uint64_t *ct, uint64_t *out,
                              size_t batch, void *stream);
/* Ciphertext::switch_down (F/bfv/ciphertext.rs:148-161): ct [batch][nparts][L][N] Ntt ->
 * out [batch][nparts][L-1][N] Ntt. */
fhe_status fhe_bfv_switch_down(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, uint64_t *out, size_t batch);
fhe_status fhe_bfv_switch_down_dev(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, uint64_t *out,
                                   size_t batch, void *stream);

/* Ciphertext::switch_to_level (F/bfv/ciphertext.rs:164-183) with levels = target_level - level >= 0:
 * ct [batch][nparts][L][N] Ntt -> out [batch][nparts][L-levels][N] Ntt.  The reference loops
 * switch_down (PowerBasis -> divide-and-round -> Ntt per level); here one inverse transform, `levels` divide-and-round
 * passes and one forward transform give the same values (the Ntt round trips in between are the identity).
 * levels beyond the chain -> FHE_E_INVALID_LEVEL; levels == 0 copies. */
fhe_status fhe_bfv_switch_to_level(const fhe_ctx *ctx, size_t levels, size_t nparts, const uint64_t *ct, uint64_t *out,
                                   size_t batch);
fhe_status fhe_bfv_switch_to_level_dev(const fhe_ctx *ctx, size_t levels, size_t nparts, const uint64_t *ct,
                                       uint64_t *out, size_t batch, void *stream);

/* ------------------------------------ PIR / RGSW / inner sum ("next" rows) ---- */
/* fhe_math::rq::dot_product (M/rq/ops.rs:449-570) and bfv::dot_product_scalar
 * (F/bfv/ops/dot_product.rs:54-180): out[b][part] = sum_k cts[b][k][part] (.) pts[b][k], all Ntt.
 * cts: [batch][count][nparts][L][N] (or [count][nparts][L][N] shared by the batch when cts_shared != 0);
 * pts: [batch][count][L][N] (`Plaintext::poly_ntt`; shared when pts_shared != 0); out: [batch][nparts][L][N].
 * nparts = 1 is the polynomial dot product.  count == 0 -> FHE_E_EMPTY_DOT_PRODUCT. */
fhe_status fhe_bfv_dot_product_scalar(const fhe_ctx *ctx, size_t nparts, size_t count, const uint64_t *cts,
                                      int cts_shared, const uint64_t *pts, int pts_shared, uint64_t *out,
                                      size_t batch);
fhe_status fhe_bfv_dot_product_scalar_dev(const fhe_ctx *ctx, size_t nparts, size_t count, const uint64_t *cts,
                                          int cts_shared, const uint64_t *pts, int pts_shared, uint64_t *out,
                                          size_t batch, void *stream);
/* `Ciphertext * Plaintext` (F/bfv/ops/mod.rs:229-257): out[b][part] = ct[b][part] (.) pt[b] (pt shared if pt_shared). */
fhe_status fhe_bfv_mul_plain(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, const uint64_t *pt, int pt_shared,
                             uint64_t *out, size_t batch);
fhe_status fhe_bfv_mul_plain_dev(const fhe_ctx *ctx, size_t nparts, const uint64_t *ct, const uint64_t *pt,
                                 int pt_shared, uint64_t *out, size_t batch, void *stream);
/* Poly::<Ntt>::random_from_seed (M/rq/mod.rs:276-292) as a received ciphertext applies it (F/bfv/ciphertext.rs:
 * 287-302): seeds [batch][32] bytes -> out [batch][L][N], the polynomial c1 of a seeded ciphertext, taken as Ntt form.
 * key = SHA-256(seed); one ChaCha8 stream per polynomial; every residue row draws `degree` values from
 * Uniform[0, q_i).  SHA-256 and the ChaCha block function are public algorithms (known-answer tested); the
 * generator's word layout and the rejection sampler restate rand_chacha 0.10 / rand 0.10, third-party crates the
 * reference does not vendor: PARITY UNPINNED against a real fhe.rs run, exactly like the NTT's psi -- a host that
 * needs certainty expands c1 itself and uploads it. */
fhe_status fhe_poly_from_seed(const fhe_ctx *ctx, const uint8_t *seeds, uint64_t *out, size_t batch);
fhe_status fhe_poly_from_seed_dev(const fhe_ctx *ctx, const uint8_t *seeds, uint64_t *out, size_t batch, void *stream);
/* SecretKey::try_decrypt, small-plaintext branch (F/bfv/keys/secret_key.rs:198-247): phase
 * c0 + c1 s + c2 s^2 + ... over the ciphertext context, PowerBasis, Scaler::scale with the level's
 * cipher-to-plaintext scaler (factor t / q, F/bfv/parameters.rs:636-643), then per coefficient
 * ((d_0 + t) mod q_0) mod t.  `cipher_plain_scaler`: from = the ciphertext context, to = the plaintext
 * context (a prefix of the moduli).  s_ntt [L][N]: the secret key polynomial over the ciphertext
 * context in Ntt form (the host builds it from its ternary coefficients, secret_key.rs:200-203).
 * ct [batch][nparts][L][N] Ntt -> out [batch][N], the plaintext polynomial's coefficients in [0, t).
 * Secret hygiene (the reference wraps s, the phase and the scaled plaintext in Zeroizing, secret_key.rs:198-226):
 * the engine's intermediates (phase, scaled plaintext) are cleared on the stream before their scratch blocks
 * are reused; the host-pointer variant also clears its staged copies of s_ntt and of the result.  The `_dev`
 * variant's s_ntt and out are caller-owned device buffers: clearing them is the caller's job. */
fhe_status fhe_bfv_decrypt(const fhe_scaler *cipher_plain_scaler, uint64_t plaintext_modulus, const uint64_t *s_ntt,
                           const uint64_t *ct, size_t nparts, uint64_t *out, size_t batch);
fhe_status fhe_bfv_decrypt_dev(const fhe_scaler *cipher_plain_scaler, uint64_t plaintext_modulus,
                               const uint64_t *s_ntt, const uint64_t *ct, size_t nparts, uint64_t *out, size_t batch,
                               void *stream);
/* `&Ciphertext * &RGSWCiphertext` (F/bfv/rgsw_ciphertext.rs:122-156), RGSWCiphertext{ksk0, ksk1}:
 * ct, out [batch][2][L][N] Ntt; both keys at the ciphertext level. */
fhe_status fhe_bfv_rgsw_mul(const fhe_ksk *ksk0, const fhe_ksk *ksk1, const uint64_t *ct, uint64_t *out, size_t batch);
fhe_status fhe_bfv_rgsw_mul_dev(const fhe_ksk *ksk0, const fhe_ksk *ksk1, const uint64_t *ct, uint64_t *out,
                                size_t batch, void *stream);
/* EvaluationKey::computes_inner_sum (F/bfv/keys/evaluation_key.rs:56-100): out = ct; for every
 * (gks[i], exponents[i]) in order: out += GaloisKey::relinearize(out).  The caller passes the keys for
 * 3^(2^j) mod 2N, j = 0 .. log2(N/2)-1, then 2N-1 (exactly the reference's sequence). */
fhe_status fhe_bfv_inner_sum(const fhe_ksk *const *gks, const size_t *exponents, size_t ngk, const uint64_t *ct,
                             uint64_t *out, size_t batch);
fhe_status fhe_bfv_inner_sum_dev(const fhe_ksk *const *gks, const size_t *exponents, size_t ngk, const uint64_t *ct,
                                 uint64_t *out, size_t batch, void *stream);
/* EvaluationKey::expands (F/bfv/keys/evaluation_key.rs:192-256), the oblivious expansion of
 * eprint 2019/1483: gks[l] is the Galois key of element (N >> l) + 1, l < nlevels; expanding to
 * `size` outputs needs ceil(log2(size)) of them (fewer -> FHE_E_EXPANSION_UNSUPPORTED; size == 0 or
 * size > N -> FHE_E_INVALID_EXPANSION_SIZE).  The monomials -x^(N - 2^l) of the reference's
 * EvaluationKey are derived by the engine from the context's NTT tables.
 * ct [batch][2][L][N] Ntt -> out [size][batch][2][L][N] Ntt (batch = 1: the reference's Vec<Ciphertext>). */
fhe_status fhe_bfv_expand(const fhe_ksk *const *gks, size_t nlevels, const uint64_t *ct, uint64_t *out, size_t size,
                          size_t batch);
fhe_status fhe_bfv_expand_dev(const fhe_ksk *const *gks, size_t nlevels, const uint64_t *ct, uint64_t *out, size_t size,
                              size_t batch, void *stream);

/* ------------------------------------------------------------- Multiplicator ---- */
/* Multiplicator::new_leveled_internal + enable_relinearization + enable_mod_switching
 * (F/bfv/ops/mul.rs:74-163).  rk may be NULL (3-part output). */
fhe_status fhe_mul_create(const fhe_scaler *extender_lhs, const fhe_scaler *extender_rhs,
                          const fhe_scaler *down_scaler, const fhe_ksk *rk_or_null, int mod_switch,
                          fhe_mul **out);
void fhe_mul_destroy(fhe_mul *m);
/* Output geometry: parts (2 with rk, else 3) and rows per part (L, or L-1 with mod switch). */
fhe_status fhe_mul_out_shape(const fhe_mul *m, size_t *parts, size_t *rows);
/* The multiplication basis (Multiplicator::mul_ctx moduli, mul.rs:84): *count = its length; moduli may be NULL. */
fhe_status fhe_mul_basis(const fhe_mul *m, size_t *count, uint64_t *moduli);
/* Execution options of fhe_bfv_mul(_dev), kept on the handle (there are no process-wide knobs): atomics, read
 * once on entry of every call, so they may be set while other threads use the handle -- calls already running
 * keep the values they started with.  (The reference's Multiplicator has no such state; these only choose how
 * the same values are computed.)
 *   chunk   ciphertext pairs per pipeline pass; 0 (default) = equal chunks under a workspace budget
 *           (3 GiB with one stream: 512 pairs at N = 8192, 4 moduli; 384 MiB but at least 8 pairs with two: 64
 *           pairs -- used when the batch gives four or more such chunks, else the one-stream cut)
 *   streams 2 (default) = the chunks of a batch alternate between the caller's stream and an internal one,
 *           forked from and joined back into the caller's stream with events: stream-ordered for the caller
 *           exactly as with 1, capturable into a hipGraph, bit-identical results, +4.5 % at C2;
 *           1 = the caller's stream only (per-kernel durations do not overlap: what profilers want). */
fhe_status fhe_mul_set_chunk(fhe_mul *m, size_t chunk);
fhe_status fhe_mul_set_streams(fhe_mul *m, size_t streams);
fhe_status fhe_mul_get_options(const fhe_mul *m, size_t *chunk, size_t *streams);
/* Multiplicator::multiply (F/bfv/ops/mul.rs:165-243), the metric's unit of work:
 * lhs, rhs [batch][2][L][N] Ntt -> out [batch][parts][rows][N] Ntt.  lhs == rhs (the same buffer: squaring, the
 * reference's `&c1 * &c1`) extends the operand once; the values are those of the general call.
 * The host-pointer form sends a large batch (>= 3 slices of >= 32 MiB per operand) through in slices whose upload,
 * pipeline and download overlap on internal streams (so do fhe_bfv_relinearize and fhe_bfv_galois); it returns when
 * `out` is complete, like every host-pointer call. */
fhe_status fhe_bfv_mul(const fhe_mul *m, const uint64_t *lhs, const uint64_t *rhs, uint64_t *out, size_t batch);
fhe_status fhe_bfv_mul_dev(const fhe_mul *m, const uint64_t *lhs, const uint64_t *rhs, uint64_t *out,
                           size_t batch, void *stream);
/* `Mul<&Ciphertext> for &Ciphertext` with any number of parts (F/bfv/ops/mod.rs:259-358; the squaring branch
 * computes the same values): extend every part, c[k] = sum_{i+j=k} lhs[i] (.) rhs[j] over the multiplication
 * basis, scale every c[k] down; never relinearises (the handle's key and mod-switch flag are ignored).
 * lhs [batch][lhs_parts][L][N], rhs [batch][rhs_parts][L][N] Ntt -> out [batch][lhs_parts+rhs_parts-1][L][N] Ntt.
 * (fhe_bfv_mul is the 2 x 2 case on its fused pipeline and is what Multiplicator::multiply accepts.) */
fhe_status fhe_bfv_tensor(const fhe_mul *m, size_t lhs_parts, size_t rhs_parts, const uint64_t *lhs, const uint64_t *rhs,
                          uint64_t *out, size_t batch);
fhe_status fhe_bfv_tensor_dev(const fhe_mul *m, size_t lhs_parts, size_t rhs_parts, const uint64_t *lhs,
                              const uint64_t *rhs, uint64_t *out, size_t batch, void *stream);

/* ------------------------------------------------------------- BfvParameters ---- */
/* BfvParametersBuilder::build (F/bfv/parameters.rs:560-738), the part that defines device
 * tables: per-level contexts, the extended multiplication basis (:660-676) and per-level
 * MultiplicationParameters{extender, down_scaler} (:686-700, 793-814).
 * moduli may be generated first with fhe_generate_moduli. */
fhe_status fhe_params_create(int device, size_t degree, size_t nmoduli, const uint64_t *moduli,
                             uint64_t plaintext_modulus, fhe_params **out);
/* NTT tables of the host.  fhe_params_create derives its own primitive root psi per modulus (the smallest
 * generator), so its handles agree only with Ntt-form data produced through THIS engine.  The reference draws
 * psi from ChaCha8Rng::seed_from_u64(0) (M/ntt/native.rs:320-336; third-party RNG, not reproducible here), and
 * every Ntt-form ciphertext or key made by the Rust host is in that evaluation order: a host that brings its own
 * Ntt-form data MUST create the parameter set with fhe_params_create_with_tables.  The engine calls `tables`
 * once for every modulus it builds a context over -- the ciphertext moduli and the 62-bit extension primes of
 * parameters.rs:660-676 -- and the host fills the four [degree] tables and the two scalars of
 * NttOperator::new(modulus, degree) (M/ntt/native.rs:16-26, same meaning as in fhe_ctx_create).  A non-zero
 * return aborts creation with FHE_E_NTT_UNAVAILABLE.  The callback is invoked only while this call runs (the
 * tables are cached per modulus for handles made later, e.g. fhe_mul_create_default's level-specific basis).
 * tables == NULL behaves like fhe_params_create. */
typedef int (*fhe_ntt_tables_fn)(void *user, uint64_t modulus, size_t degree, uint64_t *omegas,
                                 uint64_t *omegas_shoup, uint64_t *zetas_inv, uint64_t *zetas_inv_shoup,
                                 uint64_t *size_inv, uint64_t *size_inv_shoup);
fhe_status fhe_params_create_with_tables(int device, size_t degree, size_t nmoduli, const uint64_t *moduli,
                                         uint64_t plaintext_modulus, fhe_ntt_tables_fn tables, void *user,
                                         fhe_params **out);
void fhe_params_destroy(fhe_params *p);
size_t fhe_params_max_level(const fhe_params *p);
fhe_status fhe_params_ctx(const fhe_params *p, size_t level, const fhe_ctx **out);      /* context_at_level */
fhe_status fhe_params_mul_ctx(const fhe_params *p, size_t level, const fhe_ctx **out);  /* mul_params.to    */
fhe_status fhe_params_extender(const fhe_params *p, size_t level, const fhe_scaler **out);
fhe_status fhe_params_down_scaler(const fhe_params *p, size_t level, const fhe_scaler **out);
/* Multiplicator::default(rk) (F/bfv/ops/mul.rs:101-138) at rk's ciphertext level: the extension primes are
 * the first 62-bit NTT primes that are not moduli OF THAT LEVEL (mul.rs:110-126) -- at level > 0 this can differ
 * from the level's mul_params basis, which skips every top-level modulus (parameters.rs:660-676); the handle then
 * owns its own multiplication context and scalers.  rk == NULL gives the `&ct * &ct` strategy without
 * relinearisation (F/bfv/ops/mod.rs:259-358), which uses the level's mul_params as the reference does. */
fhe_status fhe_mul_create_default(const fhe_params *p, size_t level, const fhe_ksk *rk_or_null, int mod_switch,
                                  fhe_mul **out);

/* ------------------------------------------------- zq::primes (host, no GPU) ---- */
/* generate_prime (M/zq/primes.rs:30-59): returns 0 when none exists. */
uint64_t fhe_generate_prime(size_t num_bits, uint64_t modulo, uint64_t upper_bound);
int fhe_supports_opt(uint64_t p);                                      /* M/zq/primes.rs:10-24 */
int fhe_is_prime(uint64_t p);                                          /* fhe-util/src/lib.rs:16-18 */
/* BfvParametersBuilder::generate_moduli (F/bfv/parameters.rs:391-431). */
fhe_status fhe_generate_moduli(const size_t *sizes, size_t count, size_t degree, uint64_t *out);

/* ---------------------------------------------------------- bench / test aids ---- */
/* Counter-based synthetic residues written on the device:
 * x = splitmix64(seed ^ (ct<<40) ^ (part<<36) ^ (row<<28) ^ coeff) mod q_row for
 * out[b][part_local][row][coeff], ct = ct0 + b, part = part0 + part_local. */
fhe_status fhe_synth_uniform_dev(const fhe_ctx *ctx, uint64_t seed, uint64_t ct0, uint64_t part0, size_t nparts,
                                 uint64_t *out, size_t batch, void *stream);
/* The engine keeps its scratch buffers (grow-only, reused in stream order per device), its internal second
 * streams and a few pooled events between calls; this frees every idle one (call it while no engine call is
 * running) and returns the number of scratch bytes released.  (Up to eight idle internal stream HANDLES stay parked
 * for reuse -- a stream holds no device memory, and the runtime assigns a stream's hardware queue once, at creation:
 * a re-created stream can land on the caller's queue and serialise the two lanes of a multiply.) */
size_t fhe_workspace_trim(void);
/* Bounds on what the engine RETAINS between calls: `per_stream_bytes` for the scratch blocks keyed to one
 * (device, stream), `total_bytes` over all of them; 0 = no bound.  Defaults: no per-stream bound, total = a quarter of
 * the device's memory (at least 8 GiB) -- pass FHE_WORKSPACE_DEFAULT to ask for it again.  Idle blocks beyond a bound are
 * evicted least-recently-used first, when a block is released and before a stream's block grows; blocks in use are
 * never refused (a call that needs more than the bound runs, its blocks are not kept afterwards).
 * Hosts that create and destroy their OWN HIP streams (torch, hip-rs): the engine is never told that such a stream is
 * gone and has no safe way to ask (hipStreamQuery on a destroyed handle crashes in this runtime), so what it kept for
 * it -- scratch blocks, an internal second stream -- stays until it is the least recently used: blocks under the
 * `total_bytes` bound, internal streams beyond 32 live user streams.  Such a host sets `total_bytes` to a few times
 * one stream's footprint (tests: 1,000 short-lived streams stay within 2x of one), or calls fhe_workspace_trim.
 * Scratch blocks come from the library's private stream-ordered SCRATCH pool; a stream's own blocks return to it in
 * stream order (growing does not synchronise the device), other streams' blocks are evicted with hipFree (which waits).
 * `total_bytes` is PER DEVICE (the default a quarter of that device's memory), and it bounds DEVICE MEMORY, not only the
 * engine's bookkeeping: the scratch pool's release threshold is the bound and evictions trim the pool, so another
 * allocator in the process (torch, the host's own hipMalloc) gets the memory back.  fhe_buf_alloc_async has its own pool
 * (BUFFERS), which keeps what is freed into it until fhe_workspace_trim.
 * fhe_workspace_stats: bytes held (idle + in use), bytes in use, blocks, distinct (device, stream) owners, internal
 * second streams alive.  fhe_workspace_pool_stats: what the DRIVER says the two pools of `device` hold -- reserved
 * (backed by device memory) and used (handed out) bytes of each (hipMemPoolAttrReservedMemCurrent / UsedMemCurrent). */
#define FHE_WORKSPACE_DEFAULT ((size_t)-1)
fhe_status fhe_workspace_set_limit(size_t per_stream_bytes, size_t total_bytes);
fhe_status fhe_workspace_get_limit(size_t *per_stream_bytes, size_t *total_bytes);
fhe_status fhe_workspace_stats(size_t *held_bytes, size_t *in_use_bytes, size_t *blocks, size_t *owners,
                               size_t *internal_streams);
fhe_status fhe_workspace_pool_stats(int device, size_t *scratch_reserved_bytes, size_t *scratch_used_bytes,
                                    size_t *buffers_reserved_bytes, size_t *buffers_used_bytes);
/* Integer-issue ceiling ("report both ceilings"): register-resident loops of the instructions /
 * butterflies the NTT-type kernels are made of, chip-wide, no memory traffic, run for at least min_seconds
 * (0 < min_seconds <= 10).  which: 0 v_mad_u64_u32, 1 v_mul_lo_u32, 2 v_mul_hi_u32, 3 lazy Shoup product,
 * 4 forward butterfly (any modulus < 2^62), 5 forward butterfly for moduli < 2^60, 6 inverse butterfly,
 * 7 the key switch's Shoup multiply-accumulate, 8 the tensor product of slots 0 / 2 (one product + single-word Barrett),
 * 9 the tensor product of slot 1 (two products, one 128-bit sum, one Barrett); 10-15 (round 6) the FP64 forms for
 * moduli below 2^50 (csrc/zq_f64.hpp): 10 v_fma_f64, 11 v_rndne_f64, 12 exact lazy modular product, 13 forward and
 * 14 inverse butterfly with their amortised reductions, 15 the key switch's multiply-accumulate;
 * *ops_per_s = lane-operations (multiplies / products / butterflies) per second.  Measurement aid, not on the path.
 * fhe_ubench_scaler: RnsScaler::scale's ceiling -- the scale_kernel instance that serves `scaler`, run over coefficient
 * columns that all alias ONE polynomial (L2 resident: the instruction stream without HBM traffic); *columns_per_s
 * chip-wide.  With these, bench.py prices every kernel family of ct x ct + relinearise against an integer ceiling. */
fhe_status fhe_ubench_int(int device, int which, double min_seconds, double *ops_per_s);
fhe_status fhe_ubench_scaler(const fhe_scaler *scaler, double min_seconds, double *columns_per_s);
/* The box's own streaming rate: a 16-byte-per-lane copy of `bytes` bytes with streaming loads / stores (the path's
 * element-wise kernels are made of the same accesses); *bytes_per_s = read + write bytes per second. */
fhe_status fhe_ubench_copy(int device, size_t bytes, double min_seconds, double *bytes_per_s);
/* Execution option (round 6): rows whose moduli are all below 2^50 -- every modulus of the reference's stock parameter sets,
 * F/bfv/parameters.rs:222-251 -- run their transforms and key-switch accumulations on the FP64 FMA pipe (exact arithmetic on
 * doubles holding integers, csrc/zq_f64.hpp); on = 0 sends them to the integer kernels like every wider modulus.  Results
 * are bit-identical either way (canonical residues are a function of the inputs); default on; process-wide, read per launch. */
void fhe_engine_set_f64(int on);
int fhe_engine_get_f64(void);
/* Per-kernel HIP-event timing (events carried by the launches, on the launching stream).  One entry per (launch label,
 * kernel symbol): fhe_prof_get gives the entry's label -- several entries share a label when several instantiations of a
 * kernel template run under it; sum them for the family -- and fhe_prof_get_symbol the kernel's demangled symbol, the
 * name rocprofv3 lists it under.  (No reference counterpart: fhe.rs times with Criterion on the host.) */
void fhe_prof_enable(int on);
void fhe_prof_reset(void);
size_t fhe_prof_count(void);
fhe_status fhe_prof_get(size_t index, char *name, size_t name_cap, uint64_t *launches, double *total_ms);
fhe_status fhe_prof_get_symbol(size_t index, char *symbol, size_t symbol_cap);

#ifdef __cplusplus
}
#endif
#endif /* FHE_HIP_H */
