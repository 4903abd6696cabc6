#!/bin/bash
# Builds the CPU-emulation twin of the C-ABI library from the SAME sources (tests only; see hw_emu.h).
set -e
cd "$(dirname "$0")/../.."
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=tests/emu/libexl2_emu.so
SRCS=$(ls exllamav2_amd/csrc/*.hip)
ARGS=""
for f in $SRCS; do ARGS="$ARGS -x c++ $f"; done
$CXX -std=c++17 -O1 -g -fPIC -shared -pthread -Wno-unused-value -Wno-c99-designator \
    -include tests/emu/hw_emu.h -Iexllamav2_amd/csrc -Itests/emu \
    $ARGS -x c++ tests/emu/emu_runtime.cpp -o $OUT
echo "built $OUT"
