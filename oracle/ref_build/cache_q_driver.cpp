// cache_q_driver.cpp -- C entry points over the reference's quantized-KV-cache codec (exllamav2_ext/cuda/cache_q.cuh,
// compiled from /root/reference; see build.sh).  TEST INFRASTRUCTURE ONLY.  One call = one 512-element block handled by
// 256 logical threads (THREADS_Q = BLOCKSIZE_Q / 2, cache.cu:10-13), exactly the reference's device functions
// fp16_to_q<wbits> / q_to_fp16<wbits>.
#include <stdint.h>
#include <stdlib.h>
#include "cuda_shim.h"
#include "simt_host.h"

// the rest of the vocabulary cache_q.cuh needs
static inline half2 __habs2(half2 a) { half2 r = {mk_half(fabs((double)a.x.v)), mk_half(fabs((double)a.y.v))}; return r; }
static inline half __hmax(half a, half b) { return (double)a.v >= (double)b.v ? a : b; }
static inline half2 __h2div(half2 a, half2 b) { half2 r = {mk_half((double)a.x.v / (double)b.x.v), mk_half((double)a.y.v / (double)b.y.v)}; return r; }
static inline int __half2int_rn(half a) { return (int)nearbyint((double)a.v); }          // default mode: ties to even
static inline half2 __float2half2_rn(float f) { half2 r = {mk_half(f), mk_half(f)}; return r; }
static inline int clamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }   // cuda/util.cuh
template <typename T> static inline T __ldg(const T* p) { return *p; }
template <typename T> static inline void __stcg(T* p, T v) { *p = v; }
template <typename T> static inline void __stwb(T* p, T v) { *p = v; }

#include "config.h"
#include <algorithm>
struct int2 { int x, y; };
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r = {x, y, z, w}; return r; }
struct uint4 { unsigned x, y, z, w; };
#define __launch_bounds__(...)
using std::min;
using std::max;
#define BLOCKSIZE_Q Q_CACHE_BLOCKSIZE_Q
#define SUPER_BLOCKSIZE_Q Q_CACHE_SUPER_BLOCKSIZE_Q
#define THREADS_Q (BLOCKSIZE_Q / 2)
#include "cuda/cache_q.cuh"

// the four addressing kernels of cuda/cache.cu (:143-223 pack, :324-400 unpack; paged and contiguous): text extracted by
// build.sh at build time into the git-ignored oracle/_ref/cache_kernels.inc (cache.cu itself has launch syntax)
#include "cache_kernels.inc"

extern "C" {

// in: 512 fp16 (bit patterns); out: 256 bytes (4-bit) or 512 bytes (8-bit) codes; scales: 16 fp16
int ref_cache_fp16_to_q(int wbits, const uint16_t* in, uint8_t* out, uint16_t* scales)
{
    if (wbits != 4 && wbits != 8) return -1;
    alignas(16) static uint16_t a_in[512]; alignas(16) static uint8_t a_out[512]; alignas(16) static uint16_t a_s[16];
    memcpy(a_in, in, sizeof(a_in));
    simt::run_block(256, [&](int t) {
        if (wbits == 4) fp16_to_q<4>(t, (const half*)a_in, a_out, (half*)a_s, 0, 512);
        else            fp16_to_q<8>(t, (const half*)a_in, a_out, (half*)a_s, 0, 512);
    });
    memcpy(out, a_out, wbits == 4 ? 256 : 512);
    memcpy(scales, a_s, sizeof(a_s));
    return 0;
}

int ref_cache_q_to_fp16(int wbits, const uint8_t* in, const uint16_t* scales, uint16_t* out)
{
    if (wbits != 4 && wbits != 8) return -1;
    alignas(16) static uint8_t a_in[512]; alignas(16) static uint16_t a_s[16]; alignas(16) static uint16_t a_out[512];
    memcpy(a_in, in, wbits == 4 ? 256 : 512);
    memcpy(a_s, scales, sizeof(a_s));
    simt::run_block(256, [&](int t) {
        if (wbits == 4) q_to_fp16<4>(t, a_in, (const half*)a_s, (half*)a_out, 0, 512);
        else            q_to_fp16<8>(t, a_in, (const half*)a_s, (half*)a_out, 0, 512);
    });
    memcpy(out, a_out, sizeof(a_out));
    return 0;
}

// array_fp16_to_q_kv_paged_cuda / array_q_to_fp16_kv_paged_cuda (cache.cu:225-260, 402-440): grid (pages_per_seq,
// SUPER_BLOCKSIZE_Q / BLOCKSIZE_Q, 2 * batch), THREADS_Q threads.  dir 0 = fp16 -> q (append q_len tokens at
// cache_seqlens), 1 = q -> fp16 (everything valid).  wbits 4 | 6 | 8 as in the reference (6 = 8-bit keys, 4-bit values).
int ref_cache_paged(int dir, int wbits, void* k_a, void* k_b, void* k_scales, void* v_a, void* v_b, void* v_scales,
                    int batch, int dim, int pages_per_seq, const int* cache_seqlens, const int* block_table, int page_size, int q_len)
{
    if (wbits != 4 && wbits != 6 && wbits != 8) return -1;
    auto launch = [&](auto kernel_call) { simt::run_grid(pages_per_seq, SUPER_BLOCKSIZE_Q / BLOCKSIZE_Q, THREADS_Q, kernel_call, batch * 2); };
    if (dir == 0)
    {
        const half* ki = (const half*)k_a; unsigned char* ko = (unsigned char*)k_b; half* ks = (half*)k_scales;
        const half* vi = (const half*)v_a; unsigned char* vo = (unsigned char*)v_b; half* vs = (half*)v_scales;
        if (wbits == 4) launch([&]() { fp16_to_q_kv_paged_kernel<4, 4>(ki, ko, ks, vi, vo, vs, cache_seqlens, block_table, pages_per_seq, page_size, dim, q_len); });
        else if (wbits == 6) launch([&]() { fp16_to_q_kv_paged_kernel<8, 4>(ki, ko, ks, vi, vo, vs, cache_seqlens, block_table, pages_per_seq, page_size, dim, q_len); });
        else launch([&]() { fp16_to_q_kv_paged_kernel<8, 8>(ki, ko, ks, vi, vo, vs, cache_seqlens, block_table, pages_per_seq, page_size, dim, q_len); });
    }
    else
    {
        const unsigned char* ki = (const unsigned char*)k_a; half* ko = (half*)k_b; const half* ks = (const half*)k_scales;
        const unsigned char* vi = (const unsigned char*)v_a; half* vo = (half*)v_b; const half* vs = (const half*)v_scales;
        if (wbits == 4) launch([&]() { q_to_fp16_kv_paged_kernel<4, 4>(ki, ks, ko, vi, vs, vo, cache_seqlens, block_table, pages_per_seq, page_size, dim); });
        else if (wbits == 6) launch([&]() { q_to_fp16_kv_paged_kernel<8, 4>(ki, ks, ko, vi, vs, vo, cache_seqlens, block_table, pages_per_seq, page_size, dim); });
        else launch([&]() { q_to_fp16_kv_paged_kernel<8, 8>(ki, ks, ko, vi, vs, vo, cache_seqlens, block_table, pages_per_seq, page_size, dim); });
    }
    return 0;
}

// contiguous: fp16_to_q_kv / q_to_fp16_kv's non-paged branch (ext_cache.cpp:139-171: token range widened to whole
// 512-element blocks, then elements) + array_*_cuda (cache.cu:262-322, 442-497): grid (width / 512, batch, 2)
int ref_cache_contiguous(int dir, int wbits, void* k_a, void* k_b, void* k_scales, void* v_a, void* v_b, void* v_scales,
                         int batch, int dim, int seq_tokens, int offset_tokens, int width_tokens)
{
    if (wbits != 4 && wbits != 6 && wbits != 8) return -1;
    int offset = offset_tokens, width = width_tokens;
    if (dim % Q_CACHE_BLOCKSIZE_Q)
    {
        while ((offset * dim) % Q_CACHE_BLOCKSIZE_Q) offset--;
        while ((width * dim) % Q_CACHE_BLOCKSIZE_Q) width++;
    }
    offset *= dim; width *= dim;
    const int stride = seq_tokens * dim;
    auto launch = [&](auto kernel_call) { simt::run_grid(width / BLOCKSIZE_Q, batch, THREADS_Q, kernel_call, 2); };
    if (dir == 0)
    {
        const half* ki = (const half*)k_a; unsigned char* ko = (unsigned char*)k_b; half* ks = (half*)k_scales;
        const half* vi = (const half*)v_a; unsigned char* vo = (unsigned char*)v_b; half* vs = (half*)v_scales;
        if (wbits == 4) launch([&]() { fp16_to_q_kv_kernel<4, 4>(ki, ko, ks, vi, vo, vs, dim, offset, stride); });
        else if (wbits == 6) launch([&]() { fp16_to_q_kv_kernel<8, 4>(ki, ko, ks, vi, vo, vs, dim, offset, stride); });
        else launch([&]() { fp16_to_q_kv_kernel<8, 8>(ki, ko, ks, vi, vo, vs, dim, offset, stride); });
    }
    else
    {
        const unsigned char* ki = (const unsigned char*)k_a; half* ko = (half*)k_b; const half* ks = (const half*)k_scales;
        const unsigned char* vi = (const unsigned char*)v_a; half* vo = (half*)v_b; const half* vs = (const half*)v_scales;
        if (wbits == 4) launch([&]() { q_to_fp16_kv_kernel<4, 4>(ki, ks, ko, vi, vs, vo, dim, offset, stride); });
        else if (wbits == 6) launch([&]() { q_to_fp16_kv_kernel<8, 4>(ki, ks, ko, vi, vs, vo, dim, offset, stride); });
        else launch([&]() { q_to_fp16_kv_kernel<8, 8>(ki, ks, ko, vi, vs, vo, dim, offset, stride); });
    }
    return 0;
}

// FP8 cache codec: array_fp16_to_fp8_cuda / array_fp8_to_fp16_cuda (cache.cu:78-141): range rounded to 8 elements, block 32,
// grid (ceil(range / 8 / 32), height).  dir 0 = fp16 -> fp8.  offset / width in ELEMENTS (the binding multiplies tokens by
// kv_heads * head_dim, ext_cache.cpp:27-31).
int ref_cache_fp8(int dir, const void* in, void* out, int stride, int height, int offset, int width)
{
    int mn = offset, mx = offset + width;
    mn = mn / 8 * 8;
    mx = mn + (mx - mn + 7) / 8 * 8;
    if (mx <= mn) return 0;
    const unsigned gx = ((mx - mn) / 8 + 31) / 32;
    if (dir == 0) simt::run_grid(gx, height, 32, [&]() { fp16_to_fp8_kernel((const half*)in, (unsigned char*)out, stride, height, mn, mx); });
    else          simt::run_grid(gx, height, 32, [&]() { fp8_to_fp16_kernel((const unsigned char*)in, (half*)out, stride, height, mn, mx); });
    return 0;
}

// cache_rotate (cache.cu:548-576): 128 blocks x 512 threads; order = int32 page ids; temp = one page
int ref_cache_rotate(void* cache, const uint32_t* order, void* temp, long long page_bytes, int rotate_len)
{
    simt::run_grid(128, 1, 512, [&]() { cache_rotate_kernel((uint8_t*)cache, order, (uint8_t*)temp, (size_t)page_bytes, (size_t)rotate_len); });
    return 0;
}

}  // extern "C"
