#!/bin/bash
# Compiles the REFERENCE's decode headers, from where they lie under /root/reference (never copied into this repo), with
# a host shim for the CUDA fp16 vocabulary -> oracle/_ref/libqdq_ref.so, libcacheq_ref.so (git-ignored; travels to the GPU box with the
# snapshot).  Used only by tests/test_oracle_ref.py to pin oracle/exl2.py by execution of the reference's own code.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${EXL2_REFERENCE:-/root/reference}/exllamav2/exllamav2_ext
OUT="$HERE/../_ref"
[ -d "$REF/cuda/quant" ] || { echo "reference sources not found under $REF" >&2; exit 3; }
mkdir -p "$OUT"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -I"$HERE" -I"$REF" -include "$HERE/cuda_shim.h" \
    "$HERE/qdq_driver.cpp" -o "$OUT/libqdq_ref.so"
echo "built $OUT/libqdq_ref.so"
# the quantized-KV-cache codec (cuda/cache_q.cuh): 256 logical threads per 512-element block, run as fibers
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -I"$HERE" -I"$REF" \
    "$HERE/cache_q_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/libcacheq_ref.so"
echo "built $OUT/libcacheq_ref.so"
