#!/bin/bash
# Builds a LAB variant of the library into tools/_variants/ (never loaded by the package; the A/B scripts copy it over
# the in-tree library for the duration of a measurement and restore the release build afterwards).
#   bash tools/build_variant.sh <name> [extra -D flags ...]      e.g.  build_variant.sh ks_half13 -DFHE_KS_HALF13=1
# -DFHE_LAB is always added: knobs.hpp refuses variant macros without it.  The lab-only sources (rejected kernel variants,
# phase-timing stamps) live in tools/lab/, outside the product tree: -I tools makes `#include "lab/..."` resolve there.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
mkdir -p "$ROOT/tools/_variants"
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DFHE_LAB -I"$ROOT/tools" "$@" \
    "$ROOT/fhe.rs_amd/csrc/fhe_hip.cpp" -o "$ROOT/tools/_variants/libfhe_hip_$NAME.so"
echo "built tools/_variants/libfhe_hip_$NAME.so"
