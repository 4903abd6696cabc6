// cuda_shim.h -- just enough of CUDA's fp16 vocabulary, on the host, to compile the REFERENCE's own decode headers
// (exllamav2_ext/cuda/quant/qdq_*.cuh, from where they lie under /root/reference) into oracle/_ref/libqdq_ref.so.
// TEST INFRASTRUCTURE ONLY: the library built from this is the checker that pins oracle/exl2.py by EXECUTION of the
// reference's code; nothing in the product loads it.  fp16 operations are evaluated in double and rounded once
// (sums and products of two fp16 values are exact in double), i.e. IEEE round-to-nearest-even like the GPU intrinsics.
#pragma once
#include <stdint.h>
#include <math.h>

#define __device__
#define __host__
#define __forceinline__ inline

// `half` is a class, as in cuda_fp16.h: construction from an integer is a user-defined conversion, so the reference's
// half_uint16(0xe400 | zero) picks the uint16_t constructor exactly as under nvcc / hipcc.
struct half
{
    _Float16 v;
    half() = default;
    half(float f) : v((_Float16)f) {}
    half(double f) : v((_Float16)f) {}
    half(int i) : v((_Float16)(double)i) {}
    operator float() const { return (float)v; }
};
struct half2 { half x, y; };                              // .x = low 16 bits, like CUDA's half2

static inline half mk_half(double d) { half h; h.v = (_Float16)d; return h; }
static inline half __float2half_rn(float f) { return mk_half(f); }
static inline half __int2half_rn(int i) { return mk_half((double)i); }
static inline half2 __halves2half2(half a, half b) { half2 r = {a, b}; return r; }
static inline half2 __half2half2(half a) { half2 r = {a, a}; return r; }
static inline half __hadd(half a, half b) { return mk_half((double)a.v + (double)b.v); }
static inline half __hsub(half a, half b) { return mk_half((double)a.v - (double)b.v); }
static inline half __hmul(half a, half b) { return mk_half((double)a.v * (double)b.v); }
static inline half __hfma(half a, half b, half c) { return mk_half(fma((double)a.v, (double)b.v, (double)c.v)); }
static inline half2 __hadd2(half2 a, half2 b) { half2 r = {__hadd(a.x, b.x), __hadd(a.y, b.y)}; return r; }
static inline half2 __hsub2(half2 a, half2 b) { half2 r = {__hsub(a.x, b.x), __hsub(a.y, b.y)}; return r; }
static inline half2 __hmul2(half2 a, half2 b) { half2 r = {__hmul(a.x, b.x), __hmul(a.y, b.y)}; return r; }
static inline half2 __hfma2(half2 a, half2 b, half2 c) { half2 r = {__hfma(a.x, b.x, c.x), __hfma(a.y, b.y, c.y)}; return r; }
static inline half __low2half(half2 a) { return a.x; }
static inline half __high2half(half2 a) { return a.y; }
static inline float __half2float(half a) { return (float)a.v; }
static inline float __low2float(half2 a) { return (float)a.x.v; }
static inline float __high2float(half2 a) { return (float)a.y.v; }
struct int4 { int x, y, z, w; };
// the reference's compat.cuh builds atomicAdd(half2*) from a CAS loop around __hadd2; blocks run one after the other on
// the host, so the read-modify-write is simply sequential (one of the orders the GPU may produce)
static inline void atomicAdd(half2* address, half2 val) { *address = __hadd2(*address, val); }
static inline void atomicAdd(half* address, half val) { *address = __hadd(*address, val); }
static inline half2 __hneg2(half2 a) { half2 r = {mk_half(-(double)a.x.v), mk_half(-(double)a.y.v)}; return r; }
static inline half2 __lowhigh2highlow(half2 a) { half2 r = {a.y, a.x}; return r; }
// funnel shift right, clamped: the low 32 bits of (hi:lo) >> min(shift, 32)
static inline uint32_t __funnelshift_rc(uint32_t lo, uint32_t hi, uint32_t shift)
{
    if (shift > 32) shift = 32;
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> shift);
}
