// act_mul_driver.cpp -- runs the reference's act_mul_kernel (exllamav2_ext/cuda/q_mlp_activation.cuh:54-112, a header,
// included as it lies under /root/reference) on the host: x <- silu(x) * y in fp16 steps (negate, hexp, add 1, hrcp,
// multiply, multiply).  TEST INFRASTRUCTURE ONLY.  hexp / hrcp are evaluated as correctly rounded fp16 functions (the GPU
// instructions are approximations within an ulp of them), so this is a YARDSTICK for the tolerance of the activation
// tests, not a bit-level pin.  Launch shape: q_mlp.cu:171-174 -- block (32, 4), grid (width / 32 / 2, ceil(rows / 4)).
#include <stdint.h>
#include <math.h>
#include "cuda_shim.h"
#include "simt_host.h"

static inline half __float2half(float f) { return mk_half(f); }
static inline half __hneg(half a) { return mk_half(-(double)a.v); }
static inline half hexp(half a) { return mk_half(exp((double)a.v)); }
static inline half hrcp(half a) { return mk_half(1.0 / (double)a.v); }
static inline half2 h2exp(half2 a) { half2 r = {hexp(a.x), hexp(a.y)}; return r; }
static inline half2 h2rcp(half2 a) { half2 r = {hrcp(a.x), hrcp(a.y)}; return r; }
static inline half2 __float2half2_rn(float f) { half2 r = {mk_half(f), mk_half(f)}; return r; }
static inline float tanh_opt(float x) { return tanhf(x); }        // compat.cuh's approximation, only used by gelu

#include "cuda/matrix_view.cuh"
const int THREADS_X = 32;
const int THREADS_Y = 4;
#include "cuda/q_mlp_activation.cuh"

extern "C" {

// x, y: fp16 [rows, width] (width a multiple of 64); x is overwritten with silu(x) * y (gelu = 0) or gelu(x) * y
int ref_act_mul(uint16_t* x, const uint16_t* y, int rows, int width, int gelu)
{
    if (width % 64) return -1;
    simt::run_grid(width / THREADS_X / 2, (rows + THREADS_Y - 1) / THREADS_Y, THREADS_X * THREADS_Y, [&]() {
        if (gelu) act_mul_kernel<true, false, true>((half*)x, (const half*)y, rows, width, nullptr, 0);
        else      act_mul_kernel<true, false, false>((half*)x, (const half*)y, rows, width, nullptr, 0);
    }, 1, THREADS_X);
    return 0;
}

}  // extern "C"
