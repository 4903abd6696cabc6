// moe_softmax_driver.cpp -- runs the reference's MoE routing kernels (exllamav2_ext/cuda/q_mlp_softmax.cuh, a header,
// included as it lies under /root/reference) on the host: softmax over the experts' router logits, keep the top-k, renormalise,
// in place.  TEST INFRASTRUCTURE ONLY.  Launch shape: q_mlp.cu:365-383 -- 32 threads, grid (1, ceil(rows / 32)).
#include <stdint.h>
#include <math.h>
#include <float.h>
#define register                        // the header uses the pre-C++17 storage class
#include "cuda_shim.h"
#include "simt_host.h"

struct int2 { int x, y; };
static inline half2 __floats2half2_rn(float a, float b) { half2 r = {mk_half(a), mk_half(b)}; return r; }
#include "cuda/quant/qdq_util.cuh"     // half2_uint32
#include "cuda/q_mlp_softmax.cuh"

extern "C" {

// x: fp16 [rows, experts] router logits -> routing weights in place; experts 4 | 8 | 16
int ref_moe_softmax_topk(uint16_t* x, int rows, int experts, int topk)
{
    auto run = [&](auto kernel_call) { simt::run_grid(1, (rows + WARPSIZE - 1) / WARPSIZE, WARPSIZE, kernel_call); };
    if (experts == 8) run([&]() { softmax8_topk_norm_kernel((half*)x, rows, topk); });
    else if (experts == 4) run([&]() { softmax4_topk_norm_kernel((half*)x, rows, topk); });
    else if (experts == 16) run([&]() { softmax16_topk_norm_kernel((half*)x, rows, topk); });
    else return -1;
    return 0;
}

}  // extern "C"
