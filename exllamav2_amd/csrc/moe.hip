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
    const f16x8* wr = (const f16x8*)w;
    // Round 6: everything whose address does not depend on a result is requested at entry -- the row, the norm weight, the router's
    // rows (E x 16 bytes per thread and pass) and the gather permutation: the kernel was five dependent round trips to memory
    // (row -> weight -> router rows -> permutation -> ...) of 1-2 us each on one workgroup (profiles/r08c_mixtral_b1_kernel_stats.csv:
    // 12-16 us).  Rows of <= 4096 elements (two passes of the 256 threads); longer rows take the loops below.  The arithmetic and its
    // order are those of the loops: the routing stays bit-identical to the unfused kernels.
    constexpr int NI = 2;
    const bool pre = dim8 <= NI * 256;
    const bool pvec = (((size_t)perm) & 15) == 0;                        // (the permutation as 16-byte vectors)
    const f16x8 z8 = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
    f16x8 xv_[NI], wv_[NI], gv_[NI][E];
    u32x4 pv_[NI];
    if (pre)
    {
        #pragma unroll
        for (int k = 0; k < NI; k++)
        {
            const int i = t + 256 * k;
            const bool on = i < dim8;
            xv_[k] = on ? xr[i] : z8;
            wv_[k] = on ? wr[i] : z8;
        }
        #pragma unroll
        for (int k = 0; k < NI; k++)
        {
            const int i = t + 256 * k;
            const bool on = i < dim8;
            #pragma unroll
            for (int e = 0; e < E; e++) gv_[k][e] = on ? ((const f16x8*)(gate + (size_t)e * hidden))[i] : z8;
            const u32x4 zero4 = {0u, 0u, 0u, 0u};
            pv_[k] = (on && xg && perm && pvec) ? ((const u32x4*)perm)[i] : zero4;
        }
    }
    float ss = 0.0f;
    if (pre)
    {
        #pragma unroll
        for (int k = 0; k < NI; k++)
            if (t + 256 * k < dim8)
            {
                #pragma unroll
                for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)xv_[k][e], 65504.0f)); ss = fmaf(f, f, ss); }
            }
    }
    else
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
    if (pre)
    {
        #pragma unroll
        for (int k = 0; k < NI; k++)
        {
            const int i = t + 256 * k;
            if (i < dim8)
            {
                f16x8 o;
                #pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (f16)(fmaxf(-65504.0f, fminf((float)xv_[k][e], 65504.0f)) * (float)wv_[k][e] * rmf);
                ((f16x8*)row_lds)[i] = o;
                ((f16x8*)(xn + (size_t)r * hidden))[i] = o;
                xv_[k] = o;
            }
        }
    }
    else
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
    if (pre)
    {
        // (a thread's logit terms come from the row elements it normalised itself: no read-back)
        #pragma unroll
        for (int k = 0; k < NI; k++)
            if (t + 256 * k < dim8)
            {
                #pragma unroll
                for (int e = 0; e < E; e++)
                {
                    #pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[e] = dot2_f32_f16((f16x2){xv_[k][2 * j], xv_[k][2 * j + 1]}, (f16x2){gv_[k][e][2 * j], gv_[k][e][2 * j + 1]}, acc[e]);
                }
            }
    }
    else
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
    // the packed copy of the row does not wait for the routing
    if (xg)
    {
        if (pre && perm && pvec)
        {
            #pragma unroll
            for (int k = 0; k < NI; k++)
            {
                const int i = t + 256 * k;
                if (i < dim8)
                {
                    const u32x4 pv = pv_[k];
                    f16x8 o;
                    o[0] = row_lds[pv.x & 0xFFFFu]; o[1] = row_lds[pv.x >> 16]; o[2] = row_lds[pv.y & 0xFFFFu]; o[3] = row_lds[pv.y >> 16];
                    o[4] = row_lds[pv.z & 0xFFFFu]; o[5] = row_lds[pv.z >> 16]; o[6] = row_lds[pv.w & 0xFFFFu]; o[7] = row_lds[pv.w >> 16];
                    ((f16x8*)(xg + (size_t)r * hidden))[i] = o;
                }
            }
        }
        else
            for (int i = t; i < hidden; i += 256) xg[(size_t)r * hidden + i] = row_lds[perm ? (int)perm[i] : i];
    }
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
        // all loads of the copies before their stores (one round trip, not one per block); blocks are <= 256 16-byte units
        // (moe_front_launch checks), MOE_MAX_SEL experts
        u32x4 v[2][MOE_MAX_SEL];
        #pragma unroll
        for (int k = 0; k < 2; k++)
        {
            // (sum: one block, unit t from the first or the second selected expert)
            const bool second = (t >= cp.b_lo[k] && t < cp.b_hi[k]) || (t >= cp.b2_lo[k] && t < cp.b2_hi[k]);
            #pragma unroll
            for (int y = 0; y < MOE_MAX_SEL; y++)
                if (y < (cp.sum[k] ? 1 : cp.n_sel) && t < cp.units[k])
                    v[k][y] = cp.src[k][(size_t)sel_lds[cp.sum[k] ? (second ? 1 : 0) : y] * cp.units[k] + t];
        }
        #pragma unroll
        for (int k = 0; k < 2; k++)
            #pragma unroll
            for (int y = 0; y < MOE_MAX_SEL; y++)
                if (y < (cp.sum[k] ? 1 : cp.n_sel) && t < cp.units[k]) cp.dst[k][(size_t)y * cp.units[k] + t] = v[k][y];
    }
}

// cp (nullable; one row only): the argument-block copies of the selected experts (moe.h)
int moe_front_launch(const void* x, const void* norm_w, const void* gate, const void* perm, void* xn, void* xg, void* logits,
                     int rows, int hidden, int num_experts, int topk, float eps, const MoeCopy* cp_, void* stream)
{
    EXL2_REQUIRE(x && norm_w && gate && xn && logits, "moe_front: null argument");
    if (rows <= 0) return EXL2_OK;
    if (!(num_experts == 4 || num_experts == 8 || num_experts == 16) || hidden % 8 || hidden > 16384 || topk < 1 || topk > num_experts) return 1;
    MoeCopy cp; memset(&cp, 0, sizeof(cp));
    if (cp_) { if (rows != 1 || cp_->n_sel != topk || cp_->n_sel > MOE_MAX_SEL || cp_->units[0] > 256 || cp_->units[1] > 256 || ((cp_->sum[0] || cp_->sum[1]) && topk != 2)) return 1; cp = *cp_; }
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
