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

#define BLOCKSIZE_Q 512
#include "cuda/cache_q.cuh"

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

}  // extern "C"
