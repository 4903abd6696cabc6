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
# up to date?  (every output newer than the recipe's files and than the reference sources it compiles)
if [ -z "$EXL2_REF_FORCE" ] && [ -f "$OUT/libqdq_ref.so" ] && [ -f "$OUT/libcacheq_ref.so" ] && [ -f "$OUT/libqmatrix_ref.so" ] && [ -f "$OUT/librope_ref.so" ] && [ -f "$OUT/librmsnorm_ref.so" ] && [ -f "$OUT/libmoe_ref.so" ] && [ -f "$OUT/libactmul_ref.so" ] && [ -f "$OUT/libsampling_ref.so" ] && [ -f "$OUT/reference_py/exllamav2/model.py" ]; then
    OLDEST=$(ls -t "$OUT"/libqdq_ref.so "$OUT"/libcacheq_ref.so "$OUT"/libqmatrix_ref.so "$OUT"/librope_ref.so "$OUT"/librmsnorm_ref.so "$OUT"/libmoe_ref.so "$OUT"/libactmul_ref.so "$OUT"/libsampling_ref.so | tail -1)
    if [ -z "$(find "$HERE" "$REF/cuda/quant" "$REF/cuda/cache_q.cuh" "$REF/cuda/cache.cu" "$REF/cuda/q_matrix.cu" "$REF/cuda/rope.cu" "$REF/cuda/rms_norm.cu" "$REF/cuda/q_mlp_softmax.cuh" "$REF/cuda/q_mlp_activation.cuh" "$REF/cuda/q_gemm_kernel.cuh" "$REF/cuda/q_gemm_kernel_gptq.cuh" "$REF/cuda/matrix_view.cuh" "$REF/config.h" "$REF/cpp/sampling.cpp" "$REF/cpp/sampling_avx2.cpp" -type f -newer "$OLDEST" 2>/dev/null | head -1)" ]; then
        echo "oracle/_ref is up to date"; exit 0
    fi
fi
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -I"$HERE" -I"$REF" -include "$HERE/cuda_shim.h" \
    "$HERE/qdq_driver.cpp" -o "$OUT/libqdq_ref.so"
echo "built $OUT/libqdq_ref.so"
# the quantized-KV-cache codec (cuda/cache_q.cuh): 256 logical threads per 512-element block, run as fibers
CC_="$REF/cuda/cache.cu"
: > "$OUT/cache_kernels.inc"
for KN in fp16_to_q_kv_paged_kernel fp16_to_q_kv_kernel q_to_fp16_kv_paged_kernel q_to_fp16_kv_kernel; do
  echo "template <int wbits_k, int wbits_v>" >> "$OUT/cache_kernels.inc"
  awk "/^__global__ void $KN\$/,/^}/" "$CC_" >> "$OUT/cache_kernels.inc"
done
# + the FP8 codec (compress / decompress + their kernels) and the defragmenter's page rotation
awk '/^__device__ inline uint32_t compress/,/^}/' "$CC_" >> "$OUT/cache_kernels.inc"
awk '/^__device__ inline uint32_t decompress/,/^}/' "$CC_" >> "$OUT/cache_kernels.inc"
awk '/^__global__ void fp16_to_fp8_kernel$/,/^}/' "$CC_" >> "$OUT/cache_kernels.inc"
awk '/^__global__ void fp8_to_fp16_kernel$/,/^}/' "$CC_" >> "$OUT/cache_kernels.inc"
echo "#define NUM_THREADS 512" >> "$OUT/cache_kernels.inc"
echo "#define CEIL_DIVIDE(x, size) (((x) + (size) - 1) / (size))" >> "$OUT/cache_kernels.inc"
awk '/^void cache_rotate_kernel$/,/^}/' "$CC_" | sed '1s/^/__global__ /' >> "$OUT/cache_kernels.inc"
grep -q "cache_seqlens\[y\]" "$OUT/cache_kernels.inc" && grep -q "rotate_len" "$OUT/cache_kernels.inc" && grep -q "0xff000000" "$OUT/cache_kernels.inc" || { echo "cache kernel extraction failed" >&2; exit 4; }
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -I"$HERE" -I"$REF" -I"$OUT" \
    "$HERE/cache_q_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/libcacheq_ref.so"
echo "built $OUT/libcacheq_ref.so"
# q_matrix.cu: the text of shuffle_kernel, reconstruct_kernel, reconstruct_gptq_kernel and make_sequential_kernel, extracted at build time into the git-ignored output
# directory (the file as a whole needs nvcc / hipcc: kernel launch syntax, CUDA host API), then compiled with the headers
# it uses straight from the reference tree
QM="$REF/cuda/q_matrix.cu"
{
  awk '/^__global__ void shuffle_kernel/,/^}/' "$QM"
  awk '/^__global__ void reconstruct_kernel/,/^}/' "$QM"
  awk '/^__global__ void reconstruct_gptq_kernel/,/^}/' "$QM"
  awk '/^__global__ void make_sequential_kernel/,/^}/' "$QM"
} > "$OUT/q_matrix_kernels.inc"
grep -q "shuffle_8bit_4" "$OUT/q_matrix_kernels.inc" && grep -q "b_q_group_map" "$OUT/q_matrix_kernels.inc" || { echo "kernel extraction failed" >&2; exit 4; }
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -Wno-pass-failed -I"$HERE" -I"$HERE/stubs" -I"$REF" -I"$OUT" \
    "$HERE/q_matrix_driver.cpp" "$HERE/gptq_gemm_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/libqmatrix_ref.so"
echo "built $OUT/libqmatrix_ref.so"
# rope.cu: the two __device__ rotation functions (same reason, same treatment)
RP="$REF/cuda/rope.cu"
{
  awk '/^__forceinline__ __device__ void rope_cuda_arr_neox/,/^}/' "$RP"
  awk '/^__forceinline__ __device__ void rope_cuda_arr_gptj/,/^}/' "$RP"
} > "$OUT/rope_kernels.inc"
grep -q "__lowhigh2highlow" "$OUT/rope_kernels.inc" && grep -q "item2_ls" "$OUT/rope_kernels.inc" || { echo "rope extraction failed" >&2; exit 4; }
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -I"$HERE" -I"$HERE/stubs" -I"$REF" -I"$OUT" \
    "$HERE/rope_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/librope_ref.so"
echo "built $OUT/librope_ref.so"
# rms_norm.cu: the rms_norm_kernel template (the driver supplies the `template <int blocks_per_warp>` line)
awk '/^__global__ void rms_norm_kernel$/,/^}/' "$REF/cuda/rms_norm.cu" "$REF/cuda/q_mlp_softmax.cuh" "$REF/cuda/q_mlp_activation.cuh" > "$OUT/rms_norm_kernel.inc"
grep -q "rsqrtf" "$OUT/rms_norm_kernel.inc" || { echo "rms_norm extraction failed" >&2; exit 4; }
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -Wno-pass-failed -I"$HERE" -I"$HERE/stubs" -I"$REF" -I"$OUT" \
    "$HERE/rms_norm_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/librmsnorm_ref.so"
echo "built $OUT/librmsnorm_ref.so"
# q_mlp_softmax.cuh: MoE routing kernels (a header: included directly)
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -I"$HERE" -I"$HERE/stubs" -I"$REF" \
    "$HERE/moe_softmax_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/libmoe_ref.so"
echo "built $OUT/libmoe_ref.so"
# q_mlp_activation.cuh: act_mul_kernel (a header: included directly)
$CXX -std=c++17 -O1 -fPIC -shared -ffp-contract=off -Wno-unused-value -I"$HERE" -I"$HERE/stubs" -I"$REF" \
    "$HERE/act_mul_driver.cpp" "$HERE/simt_host.cpp" -o "$OUT/libactmul_ref.so"
echo "built $OUT/libactmul_ref.so"
# cpp/sampling.cpp (+ its AVX2 twin and the profiling stubs it calls): the CPU sampler, plain C++, compiled as it lies
# (g++: clang++ rejects a template call in sampling.cpp that g++ -- the reference's compiler -- accepts)
${HOSTCXX:-g++} -std=c++17 -O2 -fPIC -shared -Wno-unused-value -I"$REF" -I"$REF/cpp" \
    "$HERE/sampling_driver.cpp" "$REF/cpp/sampling.cpp" "$REF/cpp/sampling_avx2.cpp" "$REF/cpp/profiling.cpp" -o "$OUT/libsampling_ref.so"
echo "built $OUT/libsampling_ref.so"
# The reference's HOST code (its Python package, *.py only -- no kernels, no C++) mirrored for the GPU box, which has no
# /root/reference: tests/test_dropin_reference.py runs that unmodified package on top of dropin/exllamav2_ext.py there.
# Build-time mirror into the git-ignored oracle/_ref (like the kernel text above); nothing of it is committed.
PYOUT="$OUT/reference_py"
rm -rf "$PYOUT"; mkdir -p "$PYOUT"
(cd "$REF/../.." && find exllamav2 -name "*.py" -not -path "*/exllamav2_ext/*" | while read f; do mkdir -p "$PYOUT/$(dirname "$f")"; cp "$f" "$PYOUT/$f"; done)
test -f "$PYOUT/exllamav2/model.py" || { echo "reference host mirror failed" >&2; exit 5; }
echo "mirrored $(find "$PYOUT" -name '*.py' | wc -l) reference host files into $PYOUT"
