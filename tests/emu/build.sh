#!/bin/bash
# TEST INFRASTRUCTURE: builds the kernel sources for the HOST with the fiber emulator
# (tests/emu/emu_rt.hpp) -> tests/emu/_build/libfhe_emu.so.  Used only by the CPU test-suite
# to validate kernel indexing / pipelines against the oracle without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
# (into a private name, then renamed: a concurrent pytest-xdist worker never dlopens a half-written library)
TMP="$HERE/_build/.libfhe_emu.$$.so"
g++ -O2 -g -std=c++17 -fPIC -shared -DFHE_HOST_EMULATION -Wall -Wno-unused-function -Wno-unknown-pragmas \
    -I"$HERE" -I"$ROOT/fhe.rs_amd/csrc" -x c++ "$ROOT/fhe.rs_amd/csrc/fhe_hip.cpp" \
    -o "$TMP"
mv -f "$TMP" "$HERE/_build/libfhe_emu.so"
echo "built $HERE/_build/libfhe_emu.so"
