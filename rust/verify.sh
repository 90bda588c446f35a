#!/bin/bash
# One command from "parity unpinned" to "pinned against fhe.rs itself" -- for a machine that has what the build image lacks:
# a Rust toolchain, network access to fetch the reference, an AMD GPU and a built libfhe_hip.so.
#
#   rust/verify.sh [path-to-an-existing-fhe.rs-checkout]
#
# 1. takes (or clones) tlepoint/fhe.rs at the commit the patches were written against,
# 2. copies rust/fhe-math-hip next to its crates and applies rust/patches/*.patch (insert-only, all behind the `hip` feature),
# 3. runs the reference's OWN test-suites with the feature on (every existing test then exercises the engine), and
# 4. runs the native-versus-engine parity tests of patches 17 / 18 (crates/*/tests/hip_parity.rs), which compute every value
#    twice in one process -- fhe.rs's CPU code under fhe_math_hip::with_native, then the engine with its FP64 kernels on and
#    off -- and compare bit for bit: psi / the NTT tables, Poly::random_from_seed (the seeded sampler), Scaler::scale,
#    substitute, switch_down, and every hot-path Criterion ID of benches/bfv.rs on default_parameters_128(20).
#
# NOT RUN in the repository's CI: the build image has no cargo / rustc and no network (probed every round).  What IS
# checked there: the patches apply to the reference checkout, use only items that exist and are visible, and cover every
# bench ID (tests/test_rust_shim.py).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/.." && pwd)"
REF_COMMIT="${FHE_RS_COMMIT:-}"          # pin here when known; empty = the checkout's HEAD as given
WORK="${1:-}"
if [ -z "$WORK" ]; then
    WORK="$(mktemp -d)/fhe.rs"
    git clone https://github.com/tlepoint/fhe.rs "$WORK"
    [ -n "$REF_COMMIT" ] && git -C "$WORK" checkout "$REF_COMMIT"
fi
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain" >&2; exit 2; }
[ -f "$REPO/fhe.rs_amd/libfhe_hip.so" ] || python3 "$REPO/__graft_entry__.py"
mkdir -p "$WORK/rust"
rm -rf "$WORK/rust/fhe-math-hip"
cp -r "$HERE/fhe-math-hip" "$WORK/rust/fhe-math-hip"
for p in "$HERE"/patches/*.patch; do
    patch -d "$WORK" -p1 -N -s -i "$p"
done
export FHE_HIP_LIB_DIR="$REPO/fhe.rs_amd"
export LD_LIBRARY_PATH="$FHE_HIP_LIB_DIR:${LD_LIBRARY_PATH:-}"
cd "$WORK"
# the reference's own suites on the engine (one thread: the handles are shared, the GPU is one)
cargo test --features hip -p fhe-math -p fhe -- --test-threads 1
# the same suites on the native path of the SAME binary (the feature must not change the CPU results)
FHE_HIP_DISABLE=1 cargo test --features hip -p fhe-math -p fhe
# native versus engine, bit for bit
cargo test --features hip -p fhe-math --test hip_parity -- --test-threads 1 --nocapture
cargo test --features hip -p fhe --test hip_parity -- --test-threads 1 --nocapture
echo "verify.sh: the engine and fhe.rs agree bit for bit on this machine"
