#!/bin/bash
# TEST INFRASTRUCTURE: the host-emulation build of the kernel sources (tests/emu) under UBSan and ASan, then the CPU
# suite against each.  The kernels, the engine and the ABI layer are the SAME sources the gfx950 build compiles; what the
# sanitizers see is every index, shift and buffer bound of the kernels as the fiber emulator walks them.
#   bash tools/sanitize_emu.sh [pytest args, default: tests -q -m "not gpu"]
# ASan needs libstdc++ preloaded next to it (python itself does not link it, and ASan's __cxa_throw interceptor is
# resolved at start-up); swapcontext fibers: detect_stack_use_after_return stays off.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${SAN_OUT:-/tmp/fhe_san}
mkdir -p "$OUT"
GCC_LIB=$(dirname "$(gcc -print-file-name=libasan.so)")
STDCPP=$(gcc -print-file-name=libstdc++.so.6)
COMMON="-O1 -g -std=c++17 -fPIC -shared -DFHE_HOST_EMULATION -Wno-unknown-pragmas -I$ROOT/tests/emu -I$ROOT/fhe.rs_amd/csrc -x c++ $ROOT/fhe.rs_amd/csrc/fhe_hip.cpp"
g++ $COMMON -fsanitize=undefined -fno-sanitize-recover=undefined -o "$OUT/libfhe_emu_ubsan.so"
g++ $COMMON -fsanitize=address -fno-omit-frame-pointer -o "$OUT/libfhe_emu_asan.so"
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tests -q -m "not gpu")
cd "$ROOT"
echo "== UBSan"
LD_PRELOAD="$GCC_LIB/libubsan.so" FHE_EMU_LIB="$OUT/libfhe_emu_ubsan.so" UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest "${ARGS[@]}"
echo "== ASan"
rm -f "$OUT"/asan_report.*
LD_PRELOAD="$GCC_LIB/libasan.so $STDCPP" FHE_EMU_LIB="$OUT/libfhe_emu_asan.so" \
    ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:log_path="$OUT/asan_report" python -m pytest "${ARGS[@]}"
if grep -l "ERROR: AddressSanitizer" "$OUT"/asan_report.* 2>/dev/null; then echo "ASan reports above"; exit 1; fi
echo "sanitizers: clean"
