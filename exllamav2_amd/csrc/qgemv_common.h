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

// One super-chunk (4 chunks of 32 K-rows x 16 columns) of one tile: decode to exact (code - zero) halves, multiply with
// the staged activations on the matrix core, apply the group scale to the fp32 partial sum.
//
// Scale after the dot product, not per weight: a chunk lies inside one quantization group, and a lane's four
// accumulators (rows 4j..4j+3 of column c) all belong to ITS column, so sum_k a_k (q_k - z) s == s * sum_k a_k (q_k - z).
// The B operand is then an exact small integer in fp16 and the only rounding left is the fp32 accumulation -- closer to
// the real-valued product than the reference's per-weight fp16 rounding (q_gemm_kernel.cuh:35-60), and 16 packed
// multiplies per super-chunk cheaper.  When all four chunks share a group (group size >= 128, the common case) the four
// MFMAs chain into one partial sum and the scale is applied once.
// Rows >= M of the A operand are never zeroed: MFMA row i of A only feeds row i of D, and rows >= M are never stored.
template <int BITS, bool GPTQ, bool FULL>
DEV void gemv_super(const LaneWords<BITS>& lw, const PhaseCtx& ph, int chunk0, int nvalid, int lane, f32x4& acc)
{
    const int c = lane & 15;
    const int j = lane >> 4;

    int g[4];
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int ci = (FULL || q < nvalid) ? chunk0 + q : chunk0;     // padded chunks are never multiplied in
        g[q] = uniform((int)ph.cg_lds[ci]);
    }
    const int mrow = c < ph.M ? c : ph.M - 1;
    const f16* arow = ph.a_lds + mrow * ph.a_stride + (chunk0 * 32 - ph.phase_k0) + 8 * j;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    if (FULL && g[0] == g[1] && g[0] == g[2] && g[0] == g[3])
    {
        const float s = (float)ph.sc_lds[g[0] * 16 + c];
        ZC zc[4];
        if constexpr (GPTQ) zc[0] = make_zc(ph.zp_lds[g[0] * 16 + c]);
        else zc[0] = make_zc((f16)(float)(1 << (BITS - 1)));
        zc[1] = zc[0]; zc[2] = zc[0]; zc[3] = zc[0];
        f16x2 p[16];
        dequant_super<BITS>(lw.w, zc, p);
        f32x4 part = zero4;
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y,
                             p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
            const f16x8 a = *(const f16x8*)(arow + q * 32);
            part = mfma_16x16x32_f16(a, b, part);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = fmaf(s, part[i], acc[i]);
        return;
    }

    float s[4];
    ZC zc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        s[q] = (float)ph.sc_lds[g[q] * 16 + c];
        if constexpr (GPTQ) zc[q] = make_zc(ph.zp_lds[g[q] * 16 + c]);
    }
    if constexpr (!GPTQ)
    {
        const ZC z = make_zc((f16)(float)(1 << (BITS - 1)));
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = z;
    }
    f16x2 p[16];
    dequant_super<BITS>(lw.w, zc, p);
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        if (FULL || q < nvalid)
        {
            const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y,
                             p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
            const f16x8 a = *(const f16x8*)(arow + q * 32);
            const f32x4 part = mfma_16x16x32_f16(a, b, zero4);
            #pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = fmaf(s[q], part[i], acc[i]);
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

// ---- staging through LDS: ONE global round trip -----------------------------------------------------------------------
// stage_rows costs three dependent trips when the data is cold (row for the RMS sum -> q_perm -> gather of x / norm
// weight), ~1-2 us each, all on the critical path of a ~10 us kernel.  The streaming kernel instead copies x, the second
// operand and q_perm contiguously into LDS (asynchronous LDS-DMA, issued in the first cycles of the kernel) and permutes
// LDS -> LDS.
struct StageLds
{
    f16* rawx;          // [M][K]  x (or gate) in original order
    f16* raw2;          // A_RMSNORM: norm weight [K]; A_*_MUL: up [M][K]
    u16* perm;          // [K] (only when the matrix has a permutation)
    float* rms;         // [16]
};

// 1 / rms of every row (rmsnorm.py / rms_norm.cu numerics: fp32 sum of squares of the clamped row), all waves working:
// wave w sums octets w, w + nw, ... of every row, the per-wave partial sums meet in LDS (fixed order).
// L.rms must have room for 16 + 16 * nw floats.  Contains one block_sync_lds.
DEV void stage_rms_lds(const StageLds& L, int K, float eps, int M, int lane, int wv, int nw)
{
    const int oct = K >> 3;
    float* part = L.rms + 16;
    for (int r = 0; r < M; r++)
    {
        float ss = 0.0f;
        for (int i = wv * 64 + lane; i < oct; i += nw * 64)
        {
            const f16x8 v = ((const f16x8*)L.rawx)[r * oct + i];
            #pragma unroll
            for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
        }
        ss = wave_allreduce_add(ss);
        part[r * nw + wv] = ss;                                  // every lane stores the same value
    }
    block_sync_lds();
    if (wv == 0 && lane < M)
    {
        float ss = 0.0f;
        for (int w = 0; w < nw; w++) ss += part[lane * nw + w];
        L.rms[lane] = fast_rsqrt(ss * (1.0f / (float)K) + eps);
    }
}

template <int MODE>
DEV void stage_shuffle_lds(const StageLds& L, bool has_perm, f16* a_lds, int a_stride, int K, int M, int t, int nt)
{
    const int oct = K >> 3;
    for (int idx = t; idx < M * oct; idx += nt)
    {
        const int r = idx / oct, o = idx - r * oct;
        const f16* xrow = L.rawx + (size_t)r * K;
        f16x8 x, y = {0, 0, 0, 0, 0, 0, 0, 0};
        if (has_perm)
        {
            const u32x4 pv = ((const u32x4*)L.perm)[o];
            u32 src[8];
            src[0] = pv.x & 0xFFFF; src[1] = pv.x >> 16; src[2] = pv.y & 0xFFFF; src[3] = pv.y >> 16;
            src[4] = pv.z & 0xFFFF; src[5] = pv.z >> 16; src[6] = pv.w & 0xFFFF; src[7] = pv.w >> 16;
            #pragma unroll
            for (int e = 0; e < 8; e++) x[e] = xrow[src[e]];
            if constexpr (MODE == A_RMSNORM)
            {
                #pragma unroll
                for (int e = 0; e < 8; e++) y[e] = L.raw2[src[e]];
            }
            else if constexpr (MODE == A_SILU_MUL || MODE == A_GELU_MUL)
            {
                #pragma unroll
                for (int e = 0; e < 8; e++) y[e] = L.raw2[(size_t)r * K + src[e]];
            }
        }
        else
        {
            // the producer already wrote the row in this matrix' packed order (GemvJob::c_invperm), or no act-order
            x = ((const f16x8*)xrow)[o];
            if constexpr (MODE == A_RMSNORM) y = ((const f16x8*)L.raw2)[o];
            else if constexpr (MODE == A_SILU_MUL || MODE == A_GELU_MUL) y = ((const f16x8*)(L.raw2 + (size_t)r * K))[o];
        }
        f16x8 v;
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            f16 xv = x[e];
            if constexpr (MODE == A_RMSNORM)
            {
                const float f = fmaxf(-65504.0f, fminf((float)xv, 65504.0f));
                xv = (f16)((f * (float)y[e]) * L.rms[r]);
            }
            else if constexpr (MODE == A_SILU_MUL) xv = clamp_h(act_h(xv, false) * y[e]);
            else if constexpr (MODE == A_GELU_MUL) xv = clamp_h(act_h(xv, true) * y[e]);
            else if constexpr (MODE == A_SILU) xv = act_h(xv, false);
            else if constexpr (MODE == A_GELU) xv = act_h(xv, true);
            v[e] = xv;
        }
        *(f16x8*)(a_lds + r * a_stride + o * 8) = v;
    }
}

// ---- streaming one wave's slice of one run ---------------------------------------------------------------------------

template <int BITS> DEV void ring_load(LaneWords<BITS>& b, const u32* p, int lane) { load_lane_words<BITS>(p, lane, b); }

// items [0, n) at ptr0 + i * 64 * BITS words, chunk index chunk0 + 4 i, streamed through a D-deep register ring.
// `b` may already hold items 0..D-1 (clamped, see ring_fill).  EVERY load is unconditional (indices past the end are
// clamped to the last item, a line that is in flight anyway): the compiler can then count -- s_waitcnt vmcnt(D-1) before
// the first decode instead of vmcnt(0) -- and a wave with n <= D has its whole slice in flight from the first cycle.
template <int BITS, int D, int LO = 0, int HI = D>
DEV void ring_fill(LaneWords<BITS> (&b)[D], const u32* ptr0, int n, int lane)
{
    constexpr size_t STEP = 64 * BITS;
    const int last = n > 0 ? n - 1 : 0;
    #pragma unroll
    for (int u = LO; u < HI; u++) ring_load<BITS>(b[u], ptr0 + (size_t)(u < last ? u : last) * STEP, lane);
}

template <int BITS, bool GPTQ, int D>
DEV void stream_items(const u32* ptr0, int n, int chunk0, const PhaseCtx& ph, int lane, f32x4& acc,
                      LaneWords<BITS> (&b)[D], bool preloaded)
{
    constexpr size_t STEP = 64 * BITS;
    if (n <= 0) return;
    if (!preloaded) ring_fill<BITS, D>(b, ptr0, n, lane);
    const int last = n - 1;
    int i = 0;
    while (i + 2 * D <= n)
    {
        #pragma unroll
        for (int u = 0; u < D; u++)
        {
            gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
            ring_load<BITS>(b[u], ptr0 + (size_t)(i + u + D) * STEP, lane);
        }
        i += D;
    }
    // drain: fewer than 2 D items left, the ring holds items i .. i + D - 1 (clamped)
    if (i + D < n)
    {
        #pragma unroll
        for (int u = 0; u < D; u++)
        {
            gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
            const int nx = i + u + D;
            ring_load<BITS>(b[u], ptr0 + (size_t)(nx < last ? nx : last) * STEP, lane);
        }
        i += D;
    }
    #pragma unroll
    for (int u = 0; u < D; u++)
        if (i + u < n) gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
}


// ring depth: the main run keeps up to 8 KB per wave in flight (the whole slice when the split gives <= MAIN_DEPTH items
// per wave); the small leading sections of a mixed-width matrix use a shallow ring (register budget: 128 VGPRs)
#define MINOR_DEPTH 4
#ifndef MAIN_DEPTH_NARROW
#define MAIN_DEPTH_NARROW 4
#endif
template <int MB> struct MainDepth { static constexpr int v = MB <= 4 ? MAIN_DEPTH_NARROW : (MB <= 6 ? 4 : 3); };


// number of vector loads one ring fill issues (what may stay in flight when the prologue data has landed)
template <int MB> struct RingLoads
{
    static constexpr int per_item = (MB == 8 || MB == 6 || MB == 5) ? 2 : (MB == 3 ? 3 : 1);
    static constexpr int v = MB ? MainDepth<(MB ? MB : 4)>::v * per_item : 0;
};


// contiguous global -> LDS copy of `units` 16-byte units, all waves of the workgroup; unit u of the copy comes from
// src_of(u).  LDS destination dst + 16 u.
// `first`: the wave that takes units [0, 64) -- small copies start on different waves so that no wave issues them all.
// AGENT: the source was written by a launch that may still be running (overlapped chain): agent-scope loads
template <bool AGENT = false, typename F>
DEV void dma_units16(F src_of, void* dst, int units, int wv, int nw, int lane, int first = 0)
{
    int vw = wv - first; if (vw < 0) vw += nw;
    for (int base = vw * 64; base < units; base += nw * 64)
    {
        const int u = base + lane;
        if (u < units) { if constexpr (AGENT) dma_to_lds16_agent(src_of(u), (char*)dst + (size_t)base * 16); else dma_to_lds16(src_of(u), (char*)dst + (size_t)base * 16); }
    }
}
template <typename F>
DEV void dma_units4(F src_of, void* dst, int units, int wv, int nw, int lane, int first = 0)
{
    int vw = wv - first; if (vw < 0) vw += nw;
    for (int base = vw * 64; base < units; base += nw * 64)
    {
        const int u = base + lane;
        if (u < units) dma_to_lds4(src_of(u), (char*)dst + (size_t)base * 4);
    }
}

