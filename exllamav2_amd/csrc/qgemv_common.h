// qgemv_common.h -- pieces shared by the skinny q_gemm kernels: decode of one super-chunk into MFMA work, activation
// staging (with the RMSNorm / activation prologue fusions).
#pragma once
#include "qmatrix.h"

struct PhaseCtx
{
    const f16* a_lds;       // staged activations of the current phase
    const f16* sc_lds;      // [G][16] scales
    const f16* zp_lds;      // [G][16] zero points (GPTQ)
    const u16* cg_lds;      // [K/32] group of every 32-row chunk
    int a_stride;
    int phase_k0;
    int M;
};

template <int BITS, bool GPTQ, bool FULL>
DEV void gemv_super(const LaneWords<BITS>& lw, const PhaseCtx& ph, int chunk0, int nvalid, int lane, f32x4& acc)
{
    const int c = lane & 15;
    const int j = lane >> 4;

    // scale (and GPTQ zero point) of this lane's column for each chunk's group: LDS tables built in the prologue
    f16 sc[4];
    ZC zc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int ci = (FULL || q < nvalid) ? chunk0 + q : chunk0;     // padded chunks are never multiplied in
        const int g = ph.cg_lds[ci];
        sc[q] = ph.sc_lds[g * 16 + c];
        if constexpr (GPTQ) zc[q] = make_zc(ph.zp_lds[g * 16 + c]);
    }
    if constexpr (!GPTQ)
    {
        const ZC z = make_zc((f16)(float)(1 << (BITS - 1)));
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = z;
    }

    f16x2 p[16];
    dequant_super<BITS>(lw.w, zc, p);

    const int mrow = c;     // A fragment: lane (i = l & 15, j) holds row i, k-slot j
    const f16* arow = ph.a_lds + mrow * ph.a_stride + (chunk0 * 32 - ph.phase_k0) + 8 * j;
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        if (FULL || q < nvalid)
        {
            const f16x2 s2 = h2_dup(sc[q]);
            const f16x2 b0 = p[4 * q + 0] * s2, b1 = p[4 * q + 1] * s2, b2 = p[4 * q + 2] * s2, b3 = p[4 * q + 3] * s2;
            const f16x8 b = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
            f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (mrow < ph.M) a = *(const f16x8*)(arow + q * 32);
            acc = mfma_16x16x32_f16(a, b, acc);
        }
    }
}

DEV f16 clamp_h(f16 r)
{
    r = r > (f16)65504.0f ? (f16)65504.0f : r;
    return r < (f16)-65504.0f ? (f16)-65504.0f : r;
}
DEV f16 act_h(f16 g, bool gelu)
{
    // mlp.py:486-494 / q_mlp_activation.cuh: act in fp32, rounded to fp16
    const float x = (float)g;
    if (gelu) return (f16)(0.5f * x * (1.0f + tanhf(0.797884560803f * (x + 0.044715f * x * x * x))));
    return (f16)(x / (1.0f + fast_exp(-x)));
}

template <int MODE>
DEV void stage_rows(const GemvJob& job, const QMatDev& m, const f16* a, const f16* a2, f16* a_lds, const float* rmf_lds,
                    int k0, int oct, int M, int t, int nt)
{
    for (int idx = t; idx < M * oct; idx += nt)
    {
        const int r = idx / oct, o = idx - r * oct;
        const int kk = o * 8;
        u32 src[8];
        if (m.perm)
        {
            const u32x4 pv = *(const u32x4*)(m.perm + k0 + kk);
            src[0] = pv.x & 0xFFFF; src[1] = pv.x >> 16; src[2] = pv.y & 0xFFFF; src[3] = pv.y >> 16;
            src[4] = pv.z & 0xFFFF; src[5] = pv.z >> 16; src[6] = pv.w & 0xFFFF; src[7] = pv.w >> 16;
        }
        else
        {
            #pragma unroll
            for (int e = 0; e < 8; e++) src[e] = (u32)(k0 + kk + e);
        }
        const f16* arow = a + (size_t)r * job.lda;
        f16 x[8], y[8], nw8[8];
        #pragma unroll
        for (int e = 0; e < 8; e++) x[e] = arow[src[e]];
        if constexpr (MODE == A_SILU_MUL || MODE == A_GELU_MUL)
        {
            const f16* brow = a2 + (size_t)r * job.lda;
            #pragma unroll
            for (int e = 0; e < 8; e++) y[e] = brow[src[e]];
        }
        if constexpr (MODE == A_RMSNORM)
        {
            #pragma unroll
            for (int e = 0; e < 8; e++) nw8[e] = job.norm_w[src[e]];
        }
        f16x8 v;
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            f16 xv = x[e];
            if constexpr (MODE == A_RMSNORM)
            {
                const float f = fmaxf(-65504.0f, fminf((float)xv, 65504.0f));
                xv = (f16)((f * (float)nw8[e]) * rmf_lds[r]);
            }
            else if constexpr (MODE == A_SILU_MUL) xv = clamp_h(act_h(xv, false) * y[e]);
            else if constexpr (MODE == A_GELU_MUL) xv = clamp_h(act_h(xv, true) * y[e]);
            else if constexpr (MODE == A_SILU) xv = act_h(xv, false);
            else if constexpr (MODE == A_GELU) xv = act_h(xv, true);
            v[e] = xv;
        }
        *(f16x8*)(a_lds + r * job.a_stride + kk) = v;
    }
}

