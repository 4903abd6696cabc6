// moe.hip -- MoE routing for the fused expert path: router logits and softmax -> top-k -> renormalise.
//
// Reference: QMoEMLP::forward_ (cuda/q_mlp.cu:318-436): cublasHgemm(temp_state, gate^T) then softmax{4,8,16}_topk_norm_kernel
// (cuda/q_mlp_softmax.cuh) rewrite the logits in place as routing weights (zero outside the top-k); the expert GEMVs
// read them through r_weights and skip zero-weight rows / exit when every row's weight is zero.
#include "hw.h"
#include "errors.h"

// logits[r, e] = sum_k x[r, k] * gate[e, k]   (gate = nn.Linear weight [E, hidden]); one workgroup per row, fp32 sums
template <int E>
KERNEL void __launch_bounds__(256) moe_gate_kernel(const f16* x, const f16* gate, f16* logits, int hidden)
{
    SHARED float part[4 * E];
    const int r = bid_x();
    const int t = tid(), lane = lane_id(), wv = wave_id();
    float acc[E];
    #pragma unroll
    for (int e = 0; e < E; e++) acc[e] = 0.0f;
    const f16x8* xr = (const f16x8*)(x + (size_t)r * hidden);
    for (int i = t; i < (hidden >> 3); i += 256)
    {
        const f16x8 xv = xr[i];
        #pragma unroll
        for (int e = 0; e < E; e++)
        {
            const f16x8 gv = ((const f16x8*)(gate + (size_t)e * hidden))[i];
            #pragma unroll
            for (int j = 0; j < 4; j++)
                acc[e] = dot2_f32_f16((f16x2){xv[2 * j], xv[2 * j + 1]}, (f16x2){gv[2 * j], gv[2 * j + 1]}, acc[e]);
        }
    }
    #pragma unroll
    for (int e = 0; e < E; e++)
    {
        const float s = wave_allreduce_add(acc[e]);
        if (lane == 0) part[wv * E + e] = s;
    }
    block_sync();
    if (t < E) logits[(size_t)r * E + t] = (f16)(part[t] + part[E + t] + part[2 * E + t] + part[3 * E + t]);
}

// in place: softmax over E (q_mlp_softmax.cuh arithmetic incl. its epsilon), drop the E - topk smallest, renormalise
template <int E>
KERNEL void __launch_bounds__(64) moe_topk_kernel(f16* x, int rows, int topk)
{
    const int row = bid_x() * 64 + tid();
    if (row >= rows) return;
    f16* p = x + (size_t)row * E;
    float f[E];
    float mx = -3.0e38f;
    #pragma unroll
    for (int i = 0; i < E; i++) { f[i] = (float)p[i]; mx = fmaxf(mx, f[i]); }
    float sum = 0.0f;
    #pragma unroll
    for (int i = 0; i < E; i++) { f[i] = fast_exp(f[i] - mx); sum += f[i]; }
    const float eps = 1e-8f;
    float isum = 1.0f / (sum + E * eps);
    #pragma unroll
    for (int i = 0; i < E; i++) f[i] = f[i] * isum + eps;
    sum = 1.0f;
    for (int d = 0; d < E - topk; d++)
    {
        float mn = 1.0f; int mj = -1;
        #pragma unroll
        for (int j = 0; j < E; j++) if (f[j] > 0.0f && f[j] < mn) { mn = f[j]; mj = j; }
        #pragma unroll
        for (int j = 0; j < E; j++) if (j == mj) { sum -= f[j]; f[j] = 0.0f; }
    }
    isum = 1.0f / sum;
    #pragma unroll
    for (int i = 0; i < E; i++) p[i] = (f16)(f[i] * isum);
}

extern "C" {

int exl2_moe_route(const void* x, const void* gate, void* logits, int rows, int hidden, int num_experts, int topk, void* stream)
{
    EXL2_REQUIRE(x && gate && logits, "moe_route: null argument");
    EXL2_REQUIRE(num_experts == 4 || num_experts == 8 || num_experts == 16,
                 "moe_route: %d experts (the fused path covers 4, 8, 16 like q_mlp.cu:333)", num_experts);
    EXL2_REQUIRE(topk >= 1 && topk <= num_experts, "moe_route: bad top-k %d", topk);
    EXL2_REQUIRE(hidden % 8 == 0, "moe_route: hidden %d must be a multiple of 8", hidden);
    if (rows <= 0) return EXL2_OK;
    const f16* xp = (const f16*)x; const f16* gp = (const f16*)gate; f16* lp = (f16*)logits;
    const dim3 g1((unsigned)rows), g2((unsigned)((rows + 63) / 64));
    switch (num_experts)
    {
        case 4:  LAUNCH((moe_gate_kernel<4>), g1, dim3(256), 0, stream, xp, gp, lp, hidden);
                 LAUNCH((moe_topk_kernel<4>), g2, dim3(64), 0, stream, lp, rows, topk); break;
        case 8:  LAUNCH((moe_gate_kernel<8>), g1, dim3(256), 0, stream, xp, gp, lp, hidden);
                 LAUNCH((moe_topk_kernel<8>), g2, dim3(64), 0, stream, lp, rows, topk); break;
        default: LAUNCH((moe_gate_kernel<16>), g1, dim3(256), 0, stream, xp, gp, lp, hidden);
                 LAUNCH((moe_topk_kernel<16>), g2, dim3(64), 0, stream, lp, rows, topk); break;
    }
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"
