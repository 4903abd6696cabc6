// moe.hip -- MoE routing for the fused expert path: router logits and softmax -> top-k -> renormalise.
//
// Reference: QMoEMLP::forward_ (cuda/q_mlp.cu:318-436): cublasHgemm(temp_state, gate^T) then softmax{4,8,16}_topk_norm_kernel
// (cuda/q_mlp_softmax.cuh) rewrite the logits in place as routing weights (zero outside the top-k); the expert GEMVs
// read them through r_weights and skip zero-weight rows / exit when every row's weight is zero.
#include "hw.h"
#include "errors.h"
#include "moe.h"
#include <string.h>

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

// Round 5: the front of a sparse-MoE block in ONE launch for decode-sized row counts -- rms_norm_kernel (elementwise.hip), the two
// kernels above and the row gather into the experts' packed order were four launches of 4.7-4.8 us each per layer
// (profiles/history/r05_mixtral_b16_before_kernel_stats.csv: 19 us of a 124 us Mixtral layer at one row).  One workgroup per row; the SAME
// arithmetic in the same order as those kernels (sum of squares per thread -> wave -> four partials; logits per thread over the
// normalised fp16 row -> wave -> four partials; the top-k on one thread), so the routing is bit-identical to the unfused route.
// xn: the normalised row (natural order); xg: the same row gathered through `perm` (nullable = no packed copy).
// Round 6 (one row): `cp` -- the argument blocks of the SELECTED experts' lean launches (qgemv_lean.h: LeanGroupPlan) are copied from
// the per-expert tables to the slots the two launches behind this kernel read (ascending expert index).
template <int E>
KERNEL void __launch_bounds__(256) moe_front_kernel(const f16* x, const f16* w, const f16* gate, const u16* perm, f16* xn, f16* xg,
                                                    f16* logits, int hidden, float eps, float r_dim, int topk, const MoeCopy cp)
{
    DYN_SMEM(smem);
    f16* const row_lds = (f16*)smem;                                   // the normalised row
    float* const part = (float*)(smem + (size_t)hidden * 2);           // 4 E partial sums (4 for the norm)
    int* const sel_lds = (int*)(smem + (size_t)hidden * 2 + 4 * 16 * 4);   // the selected experts (cp.n_sel of them)
    const int r = bid_x();
    const int t = tid(), lane = lane_id(), wv = wave_id();
    const int dim8 = hidden >> 3;
    const f16x8* xr = (const f16x8*)(x + (size_t)r * hidden);
    float ss = 0.0f;
    for (int i = t; i < dim8; i += 256)
    {
        const f16x8 v = xr[i];
        #pragma unroll
        for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
    }
    ss = wave_allreduce_add(ss);
    if (lane == 0) part[wv] = ss;
    block_sync();
    ss = part[0] + part[1] + part[2] + part[3];
    const float rmf = fast_rsqrt(ss * r_dim + eps);
    block_sync();
    const f16x8* wr = (const f16x8*)w;
    for (int i = t; i < dim8; i += 256)
    {
        const f16x8 v = xr[i], wv8 = wr[i];
        f16x8 o;
        #pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (f16)(fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)) * (float)wv8[e] * rmf);
        ((f16x8*)row_lds)[i] = o;
        ((f16x8*)(xn + (size_t)r * hidden))[i] = o;
    }
    block_sync();
    float acc[E];
    #pragma unroll
    for (int e = 0; e < E; e++) acc[e] = 0.0f;
    for (int i = t; i < dim8; i += 256)
    {
        const f16x8 xv = ((const f16x8*)row_lds)[i];
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
    if (t == 0)
    {
        // moe_topk_kernel's arithmetic on the fp16-rounded logits
        float f[E];
        float mx = -3.0e38f;
        #pragma unroll
        for (int i = 0; i < E; i++) { f[i] = (float)(f16)(part[i] + part[E + i] + part[2 * E + i] + part[3 * E + i]); mx = fmaxf(mx, f[i]); }
        float sum = 0.0f;
        #pragma unroll
        for (int i = 0; i < E; i++) { f[i] = fast_exp(f[i] - mx); sum += f[i]; }
        const float epsn = 1e-8f;
        float isum = 1.0f / (sum + E * epsn);
        #pragma unroll
        for (int i = 0; i < E; i++) f[i] = f[i] * isum + epsn;
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
        for (int i = 0; i < E; i++) logits[(size_t)r * E + i] = (f16)(f[i] * isum);
        if (cp.n_sel)
        {
            int ns = 0;
            #pragma unroll
            for (int i = 0; i < E; i++) if (f[i] > 0.0f && ns < cp.n_sel) sel_lds[ns++] = i;
            for (; ns < cp.n_sel; ns++) sel_lds[ns] = ns > 0 ? sel_lds[0] : 0;     // (never: the top-k leaves exactly topk weights)
        }
    }
    if (cp.n_sel && r == 0)
    {
        block_sync();
        for (int y = 0; y < cp.n_sel; y++)
        {
            const int e = sel_lds[y];
            #pragma unroll
            for (int k = 0; k < 2; k++)
            {
                const u32x4* const src = cp.src[k] + (size_t)e * cp.units[k];
                u32x4* const dst = cp.dst[k] + (size_t)y * cp.units[k];
                for (int i = t; i < cp.units[k]; i += 256) dst[i] = src[i];
            }
        }
    }
    if (xg)
        for (int i = t; i < hidden; i += 256) xg[(size_t)r * hidden + i] = row_lds[perm ? (int)perm[i] : i];
}

// cp (nullable; one row only): the argument-block copies of the selected experts (moe.h)
int moe_front_launch(const void* x, const void* norm_w, const void* gate, const void* perm, void* xn, void* xg, void* logits,
                     int rows, int hidden, int num_experts, int topk, float eps, const MoeCopy* cp_, void* stream)
{
    EXL2_REQUIRE(x && norm_w && gate && xn && logits, "moe_front: null argument");
    if (rows <= 0) return EXL2_OK;
    if (!(num_experts == 4 || num_experts == 8 || num_experts == 16) || hidden % 8 || hidden > 16384 || topk < 1 || topk > num_experts) return 1;
    MoeCopy cp; memset(&cp, 0, sizeof(cp));
    if (cp_) { if (rows != 1 || cp_->n_sel != topk || cp_->n_sel > MOE_MAX_SEL) return 1; cp = *cp_; }
    const size_t lds = (size_t)hidden * 2 + 4 * 16 * 4 + MOE_MAX_SEL * 4;
    const dim3 g((unsigned)rows);
#define MOE_FRONT(E_) LAUNCH((moe_front_kernel<E_>), g, dim3(256), lds, stream, (const f16*)x, (const f16*)norm_w, (const f16*)gate, \
                             (const u16*)perm, (f16*)xn, (f16*)xg, (f16*)logits, hidden, eps, 1.0f / (float)hidden, topk, cp)
    switch (num_experts) { case 4: MOE_FRONT(4); break; case 8: MOE_FRONT(8); break; default: MOE_FRONT(16); break; }
#undef MOE_FRONT
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

extern "C" {

// rms_norm (fp16 in / out) + exl2_moe_route + the gather of the normalised rows through `perm`, one launch; 0 = launched,
// 1 = shape outside it (the caller runs the separate kernels)
int exl2_moe_front(const void* x, const void* norm_w, const void* gate, const void* perm, void* xn, void* xg, void* logits,
                   int rows, int hidden, int num_experts, int topk, float eps, void* stream)
{
    return moe_front_launch(x, norm_w, gate, perm, xn, xg, logits, rows, hidden, num_experts, topk, eps, nullptr, stream);
}

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
