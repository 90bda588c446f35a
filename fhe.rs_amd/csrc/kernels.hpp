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
// Compile-time knobs live in knobs.hpp (pinned in the release build); rejected kernel variants in tools/lab/ (lab builds only).
#pragma once
#include "kernels_common.hpp"
#include "kernels_passes.hpp"
#include "kernels_ntt.hpp"
#include "kernels_ks.hpp"
#include "kernels_scaler.hpp"
#include "kernels_misc.hpp"

namespace fhe {
namespace k {

#if defined(FHE_LAB)
#include "lab/lab_kernels.hpp"   // measured-and-rejected variants: lab builds only
#endif

}  // namespace k
}  // namespace fhe
