// qgemv_lean.hip -- decode q_gemm for a CHAIN of modules, round 3: the launch shape the MI355X measurements asked for.
//
// Replaces gemm_half_q_half_kernel (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565) + rms_norm_kernel (rms_norm.cu:33-175)
// + act_mul_kernel (q_mlp_activation.cuh:54-112) on the decode path, as composed by QAttn::forward_cuda_1 / _2
// (q_attn.cu:153-345) and QMLP::forward_run_ (q_mlp.cu:153-236).  Same host interface as qgemv_flat.hip (FlatIn); that
// kernel remains the route for what this one declines (> LEAN_MAX_M rows, grouped MoE launches, shares too big for the
// register stream).
//
// Why another shape (profiles/r02_trace_flat.txt, profiles/r03_lean_probe.txt, profiles/r03_trace_lean_v2.txt): the round-2
// kernel is ONE 1024-thread workgroup per CU whose 16 waves share a scalar unit, plan their split on the device
// (1.3-1.6 us), copy tables workgroup-wide and meet at a barrier before the first weight is decoded (4.5-7.8 us into a
// 9-16 us launch).  A skeleton with the real decode but none of that runs the four launches of a layer in 26.9 us
// against ~50 us.  So here:
//   * workgroup = ONE 16-column tile split over 8 or 16 waves (or a gate / up tile pair, or two tiles, 8 waves each):
//     several workgroups share a CU and are never in the same phase;
//   * the split is made on the HOST at launch-build time and travels in the kernel arguments: per (matrix, wave) one
//     64-byte record (segments of the tile's K range, x range, scale rows, LDS offset).  The launch geometry (waves per
//     tile, tiles per workgroup, pairing) is a template parameter and the matrix is blockIdx.y, so the addresses of the
//     header, the matrix block and the wave record depend on nothing but built-in ids: ONE batch of scalar loads, then
//     the first weight request;
//   * every wave requests its WHOLE share at entry into registers (<= LeanDepth items, no ring, no loop: what does not
//     fit is declined);
//   * the prologue is wave-private: a wave copies only ITS K slice of the activations (LDS-DMA; RMSNorm applied in
//     registers on the way when the producer left the residual stream un-normalised) and ITS rows of the scale table
//     into its own LDS area -- no workgroup barrier before the decode;
//   * partial sums of a tile meet in LDS (one barrier, fixed order = deterministic); cross-workgroup combines through
//     memory were measured and lose 2-10 us (r03_lean_probe.txt, c1 rows).
#include "qgemv_common.h"
#include "qgemv_flat.h"
#include "qgemv_lean.h"
#include "errors.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include <mutex>

#define LEAN_SEGS 3
// tuning switches (A/B builds through tools/build_variant.sh; the defaults are what the MI355X runs picked)
#ifndef LEAN_NORM_LDS
#define LEAN_NORM_LDS 1               // 1: RMSNorm of the slice in place in LDS after it has landed; 0: in registers on its way in (one row, <= 64 units)
#endif
#ifndef LEAN_PRESCALE
#define LEAN_PRESCALE 1               // 1: general items scale the weights (fp16, reconstruct()'s rounding); 0: four scales on four partial sums
#endif
#ifndef LEAN_LOWBITS
#define LEAN_LOWBITS 1                // 0: no 2 / 3-bit register stream (code-size experiment; such segments would be wrong)
#endif
#define LEAN_MAX_WAVES 16
#define LEAN_RECORDS 32               // wave records in the argument block: matrices x waves per tile
#define LEAN_MAX_PART 256             // partial sums of squares per row a chain-out launch may publish (consumer side: 4 per lane)
#define LEAN_LDS_BUDGET (40u * 1024u)  // per 8 waves: four 8-wave / two 16-wave workgroups stay resident on a CU

// one contiguous run of items (super-chunks) of one bit width inside a tile's K range
struct LeanSeg
{
    u32 off;                          // word offset of the first item for tile 0 (in qw, or in tail)
    u32 tstride;                      // words between consecutive tiles
    u32 meta;                         // n (0..9) | bits (10..13) | nvalid_last (14..16) | in_tail (17) | gshift (18..20) | gphase (21..30) | uniform (31)
    u32 place;                        // first 32-row chunk (0..15) | scale-table row of that chunk relative to the wave's first row (16..31)
};
// what one wave of a workgroup does for its tile: <= LEAN_SEGS segments (the first one is requested into registers and holds
// full items only, the others are staged in LDS), one contiguous range of the activation row, one contiguous range of
// scale-table rows, its private LDS area.  64 bytes = one scalar load.
struct alignas(64) LeanWave
{
    u32 xr;                           // first chunk (0..15) | number of chunks (16..31) of the activation slice
    u32 gr;                           // first scale-table row (0..15) | number of rows (16..31)
    u32 lds_off;                      // byte offset of the wave's LDS area inside its slot's area
    u32 pad;
    LeanSeg seg[LEAN_SEGS];
};
// per matrix: 64 bytes; the first 40 are what a wave needs at entry, the rest is read by the finalising waves
struct alignas(64) LeanMat
{
    const u32* qw; const u32* tail;
    const f16* sc_tab; const f16* zp_tab;
    int G, n_tiles; const f16* bias;
    f16* c; const u16* c_invperm;
};
// what every wave needs before anything else: the first 64 bytes; the rest is read by the finalising waves when they get there
struct alignas(64) LeanHdr
{
    const f16* a; const f16* norm_w;
    const float* ss; float eps; int M;
    int K, lda, npart; u32 flags;     // flags: a_mode (0) | gelu (3) | c_accum (4) | any_bias (5)
    u32 slot_bytes, red_off;          // LDS bytes of one slot's waves; offset of the partial sums
    u64* trace;                       // EXL2_TRACE build: [matrix][workgroup][wave][8] realtime stamps (tools/trace_lean.py)
    f16* xp_out; const u16* xp_invperm;
    float* ss_out; int ldxp, wgs;
    int ldc[FLAT_MAX_MATS];
};
struct LeanArgs
{
    LeanHdr hdr;
    LeanMat mat[FLAT_MAX_MATS];
    LeanWave wave[LEAN_RECORDS];      // [matrix][wave of the tile]
};
#define LF_NORM 1u
#define LF_GELU 8u
#define LF_ACCUM 16u
#define LF_BIAS 32u

// items of one bit width a wave may hold in registers (<= 25 dwords per lane in flight)
template <int BITS> struct LeanDepth { static constexpr int v = BITS == 8 ? 3 : BITS == 6 ? 4 : BITS == 5 ? 5 : BITS == 4 ? 6 : 8; };
static int lean_depth(int bits) { return bits == 8 ? 3 : bits == 6 ? 4 : bits == 5 ? 5 : bits == 4 ? 6 : 8; }

struct LeanCtx
{
    const f16* x_lds;                 // the wave's activation slice: row r at x_lds + r * x_stride, element 0 = first element of chunk xc0
    const f16* sc_lds;                // the wave's scale rows [rows][16]
    const f16* zp_lds;
    int x_stride, xc0, M;
};

// One FULL item whose four chunks share a group (group size >= 128 rows, aligned): exact (code - zero) halves -> four chained
// MFMAs against the staged activations -> the group scale on the fp32 partial sum.  Same arithmetic as gemv_super
// (qgemv_common.h).
template <int BITS, bool GPTQ>
DEV void lean_item_uniform(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int g, int lane, f32x4& acc)
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
    const float s = (float)cx.sc_lds[g * 16 + c];
    ZC zc[4];
    if constexpr (GPTQ) zc[0] = make_zc(cx.zp_lds[g * 16 + c]);
    else zc[0] = make_zc((f16)(float)(1 << (BITS - 1)));
    zc[1] = zc[0]; zc[2] = zc[0]; zc[3] = zc[0];
    f16x2 p[16];
    dequant_super<BITS>(lw.w, zc, p);
    f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y, p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
        const f16x8 a = *(const f16x8*)(arow + q * 32);
        part = mfma_16x16x32_f16(a, b, part);
    }
    #pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = fmaf(s, part[i], acc[i]);
}

// One item in general: a group per chunk (group sizes 32 / 64, unaligned groups), nvalid <= 4 chunks.  The group of a chunk is
// affine in the chunk index inside a segment: row = g0 + ((cs + q + gphase) >> gshift) (the host checks that against the
// matrix' chunk -> group map).  The scale goes onto the WEIGHTS here -- fp16 (code - zero) * fp16 scale, one rounding: exactly
// reconstruct()'s value (q_matrix.cu:328-553) -- so the four chunks still accumulate into one fp32 chain and the item
// needs no more registers than the uniform form (four scales on the partial sums cost 13 more, i.e. a workgroup per CU).
#if LEAN_PRESCALE
template <int BITS, bool GPTQ>
DEV void lean_item_general(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int cs, int g0, int gshift, int gphase, int nvalid,
                           int lane, f32x4& acc)
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
    ZC zc[4];
    if constexpr (GPTQ)
    {
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = make_zc(cx.zp_lds[(g0 + ((cs + (q < nvalid ? q : 0) + gphase) >> gshift)) * 16 + c]);
    }
    else
    {
        const ZC z = make_zc((f16)(float)(1 << (BITS - 1)));
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = z;
    }
    f16x2 p[16];
    dequant_super<BITS>(lw.w, zc, p);
    f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        if (q < nvalid)
        {
            const f16x2 s2 = h2_dup(cx.sc_lds[(g0 + ((cs + q + gphase) >> gshift)) * 16 + c]);
            const f16x2 b0 = p[4 * q] * s2, b1 = p[4 * q + 1] * s2, b2 = p[4 * q + 2] * s2, b3 = p[4 * q + 3] * s2;
            const f16x8 b = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
            const f16x8 a = *(const f16x8*)(arow + q * 32);
            part = mfma_16x16x32_f16(a, b, part);
        }
    }
    #pragma unroll
    for (int i = 0; i < 4; i++) acc[i] += part[i];
}
#else
template <int BITS, bool GPTQ>
DEV void lean_item_general(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int cs, int g0, int gshift, int gphase, int nvalid,
                           int lane, f32x4& acc)
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    float s[4];
    ZC zc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int g = g0 + ((cs + (q < nvalid ? q : 0) + gphase) >> gshift);
        s[q] = (float)cx.sc_lds[g * 16 + c];
        if constexpr (GPTQ) zc[q] = make_zc(cx.zp_lds[g * 16 + c]);
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
        if (q < nvalid)
        {
            const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y, p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
            const f16x8 a = *(const f16x8*)(arow + q * 32);
            const f32x4 part = mfma_16x16x32_f16(a, b, zero4);
            #pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = fmaf(s[q], part[i], acc[i]);
        }
    }
}
#endif

// a lane's words of an item that sits in LDS in its memory layout ([piece][lane][words], qlayout.h)
template <int BITS> DEV void lean_lds_words(const u32* slot, int lane, LaneWords<BITS>& r)
{
    if constexpr (BITS == 4)
    {
        const u32x4 v = ((const u32x4*)slot)[lane];
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    }
    else if constexpr (BITS == 8)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        const u32x4 v1 = ((const u32x4*)(slot + 256))[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y; r.w[6] = v1.z; r.w[7] = v1.w;
    }
    else if constexpr (BITS == 6)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        const u32x2 v1 = ((const u32x2*)(slot + 256))[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y;
    }
    else if constexpr (BITS == 5)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = slot[256 + lane];
    }
    else if constexpr (BITS == 3)
    {
        const u32* p = slot + lane * 3;
        r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2];
    }
    else
    {
        const u32x2 v = ((const u32x2*)slot)[lane];
        r.w[0] = v.x; r.w[1] = v.y;
    }
}

// copy one item (16 * bits units of 16 bytes) global -> LDS, non-temporal
DEV void lean_item_to_lds(const u32* src, u8* slot, int bits, int lane)
{
    const int units = 16 * bits;
    for (int base = 0; base < units; base += 64)
        if (base + lane < units) dma_to_lds16_nt(src + (size_t)(base + lane) * 4, slot + (size_t)base * 16);
}

struct LeanSegV { const u32* ptr; int n, bits, nvalid, chunk0, g0, gshift, gphase; bool uni; };
DEV const void* ptr_of(u32 lo, u32 hi) { return (const void*)(((u64)hi << 32) | lo); }

// Geometry (template): S = waves per tile (8 / 16), NSLOTS = tiles per workgroup (1 / 2), PAIR = the two tiles are tile u of
// matrix 0 (gate) and of matrix 1 (up) and the epilogue writes act(gate) * up.  Otherwise blockIdx.y = matrix.
// OCC = waves per SIMD the register allocation leaves room for.
#ifndef EXL2_EMU
#define LEAN_BOUNDS(T, OCC) __launch_bounds__(T, OCC)
#else
#define LEAN_BOUNDS(T, OCC)
#endif
template <bool GPTQ, int S, int NSLOTS, bool PAIR, int OCC>
KERNEL void LEAN_BOUNDS(S * NSLOTS * 64, OCC) qgemv_lean_kernel(const LeanArgs args)
{
    DYN_SMEM(smem);
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int u = bid_x();
    const int slot = wv / S, r = wv % S;
    const int mj = PAIR ? slot : bid_y();
    // ---- arguments: header, matrix block, wave record -- addresses from built-in ids only: one batch of scalar loads ----------
    const u32x4* hb = (const u32x4*)&args.hdr;
    const u32x4* mb = (const u32x4*)&args.mat[mj];
    const u32x4* wb = (const u32x4*)&args.wave[mj * S + r];
    const u32x4 h0 = hb[0], h1 = hb[1], h2 = hb[2], h3 = hb[3];
    const u32x4 m0 = mb[0], m1 = mb[1];
    const u32x2 m2 = *(const u32x2*)(mb + 2);
    const u32x4 w0 = wb[0], w1 = wb[1], w2 = wb[2], w3 = wb[3];
#ifdef EXL2_TRACE
    u64* const trace = (u64*)ptr_of(h3.z, h3.w);
#define LTRACE(i) do { if (trace && lane_id() == 0 && bid_x() < 2048) trace[(((size_t)bid_y() * 2048 + bid_x()) * LEAN_MAX_WAVES + wave_id()) * 8 + (i)] = realtime_stamp(); } while (0)
#else
#define LTRACE(i) do { } while (0)
#endif
    LTRACE(0);
    const f16* const in_a = (const f16*)ptr_of(h0.x, h0.y);
    const f16* const in_nw = (const f16*)ptr_of(h0.z, h0.w);
    const float* const in_ss = (const float*)ptr_of(h1.x, h1.y);
    const float eps = as_f32(h1.z);
    const int M = (int)h1.w, K = (int)h2.x, lda = (int)h2.y, npart = (int)h2.z;
    const u32 flags = h2.w;
    const u32 slot_bytes = h3.x, red_off = h3.y;
    const bool norm_mode = (flags & LF_NORM) != 0;
    const int oct = K >> 3;
    const u32* const qw = (const u32*)ptr_of(m0.x, m0.y); const u32* const tl = (const u32*)ptr_of(m0.z, m0.w);
    const f16* const sc_tab = (const f16*)ptr_of(m1.x, m1.y); const f16* const zp_tab = (const f16*)ptr_of(m1.z, m1.w);
    const int G = (int)m2.x, n_tiles = (int)m2.y;

    // ---- this wave's tile ---------------------------------------------------------------------------------------------------
    const int tile = PAIR ? u : u * NSLOTS + slot;
    if (!PAIR && u * NSLOTS >= n_tiles) return;                          // (matrices of one launch may have different widths)
    const bool active = tile < n_tiles;
    const u32 xr = active ? w0.x : 0u, gr = active ? w0.y : 0u;
    const int xc0 = (int)(xr & 0xFFFFu), xchunks = (int)(xr >> 16);
    const int gw0 = (int)(gr & 0xFFFFu), ng = (int)(gr >> 16);
    auto seg_of = [&](const u32x4& sg) -> LeanSegV {
        const u32 meta = active ? sg.z : 0u, place = sg.w;
        LeanSegV v;
        v.n = (int)(meta & 0x3FFu); v.bits = (int)((meta >> 10) & 0xFu); v.nvalid = (int)((meta >> 14) & 0x7u);
        v.gshift = (int)((meta >> 18) & 0x7u); v.gphase = (int)((meta >> 21) & 0x3FFu); v.uni = (meta >> 31) != 0;
        v.chunk0 = (int)(place & 0xFFFFu); v.g0 = (int)(place >> 16);
        v.ptr = (((meta >> 17) & 1u) ? tl : qw) + sg.x + (size_t)(active ? tile : 0) * sg.y;
        return v;
    };
    const LeanSegV s0 = seg_of(w1), s1 = seg_of(w2), s2 = seg_of(w3);

    // the wave's LDS area: [M rows of the activation slice][norm weight slice (norm mode)][scale rows][zero-point rows][staged items]
    const int x_stride = xchunks * 32 + 8;
    u8* const wbase = smem + (size_t)slot * slot_bytes + w0.z;
    f16* const x_lds = (f16*)wbase;
    const u32 off_nw = ((u32)M * (u32)x_stride * 2u + 15u) & ~15u;
    f16* const nw_lds = (f16*)(wbase + off_nw);
    const u32 off_sc = off_nw + (norm_mode ? (u32)xchunks * 64u : 0u);
    const u32 sc_bytes = ((u32)ng * 32u + 15u) & ~15u;
    f16* const sc_lds = (f16*)(wbase + off_sc);
    f16* const zp_lds = (f16*)(wbase + off_sc + sc_bytes);
    u8* const minor_lds = wbase + off_sc + (GPTQ ? 2u : 1u) * sc_bytes;
    float* const red = (float*)(smem + red_off);
    LTRACE(1);

    // ---- prologue requests: scale rows, activation slice ----------------------------------------------------------------
    if (active)
    {
        const int units = 2 * ng;                                           // 16-byte units: a row = 16 halfs
        const f16* st = sc_tab + ((size_t)tile * G + gw0) * 16;
        for (int base = 0; base < units; base += 64)
            if (base + lane < units) dma_to_lds16(st + (size_t)(base + lane) * 8, (u8*)sc_lds + (size_t)base * 16);
        if constexpr (GPTQ)
        {
            const f16* zt = zp_tab + ((size_t)tile * G + gw0) * 16;
            for (int base = 0; base < units; base += 64)
                if (base + lane < units) dma_to_lds16(zt + (size_t)(base + lane) * 8, (u8*)zp_lds + (size_t)base * 16);
        }
    }
    const int xunits = xchunks * 4;                                         // 16-byte units of the slice (a chunk = 32 halfs)
    const int xu0 = xc0 * 4;
    // the rows of the slice (raw: in norm mode they are normalised in place once they have landed), the norm weight slice
#if !LEAN_NORM_LDS
    f16x8 nx = {0, 0, 0, 0, 0, 0, 0, 0}, nw = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool norm_fast = norm_mode && M == 1 && xunits <= 64 && npart <= 256;
    if (norm_fast && lane < xunits && xu0 + lane < oct)
    {
        nx = *(const f16x8*)(in_a + (size_t)(xu0 + lane) * 8);
        nw = *(const f16x8*)(in_nw + (size_t)(xu0 + lane) * 8);
    }
#else
    const bool norm_fast = false;
#endif
    if (!norm_fast)
        for (int rr = 0; rr < M; rr++)
            for (int base = 0; base < xunits; base += 64)
                if (base + lane < xunits && xu0 + base + lane < oct)
                    dma_to_lds16(in_a + (size_t)rr * lda + (size_t)(xu0 + base + lane) * 8, (u8*)(x_lds + (size_t)rr * x_stride) + (size_t)base * 16);
    float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool ss_early = norm_mode && M == 1 && npart <= 256;            // the partial sums of squares of the one row: requested now
    if (norm_mode)
    {
        if (!norm_fast)
            for (int base = 0; base < xunits; base += 64)
                if (base + lane < xunits && xu0 + base + lane < oct)
                    dma_to_lds16(in_nw + (size_t)(xu0 + base + lane) * 8, (u8*)nw_lds + (size_t)base * 16);
        if (ss_early)
        {
            #pragma unroll
            for (int i = 0; i < 4; i++) if (lane + 64 * i < npart) ssp[i] = in_ss[lane + 64 * i];
        }
    }

    // ---- the other segments of this wave (other bit widths, partial super-chunks): staged in LDS, decoded after the stream ----
    {
        u32 off = 0;
        #pragma nounroll
        for (int i = 0; i < s1.n; i++) { lean_item_to_lds(s1.ptr + (size_t)i * (64u * s1.bits), minor_lds + off, s1.bits, lane); off += 256u * s1.bits; }
        #pragma nounroll
        for (int i = 0; i < s2.n; i++) { lean_item_to_lds(s2.ptr + (size_t)i * (64u * s2.bits), minor_lds + off, s2.bits, lane); off += 256u * s2.bits; }
    }
    LTRACE(2);

    // RMSNorm of the wave's slice, in place once the raw rows and the weight slice have landed: x * w * rsqrt(mean(x^2) + eps),
    // the producer left the partial sums of squares.  (LDS -> LDS with transient registers: holding the slice in registers
    // across the weight requests costs 15 VGPRs = one workgroup per CU fewer.)
    auto norm_prologue = [&]() {
        if (!norm_mode) return;
#if !LEAN_NORM_LDS
        if (norm_fast)
        {
            float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
            ss = wave_allreduce_add(ss);
            const float rms = fast_rsqrt(ss * (1.0f / (float)K) + eps);
            if (lane < xunits && xu0 + lane < oct)
            {
                f16x8 v;
                #pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    const float f = fmaxf(-65504.0f, fminf((float)nx[e], 65504.0f));
                    v[e] = (f16)((f * (float)nw[e]) * rms);
                }
                *(f16x8*)(x_lds + (size_t)lane * 8) = v;
            }
            return;
        }
#endif
        for (int rr = 0; rr < M; rr++)
        {
            float ss = 0.0f;
            if (ss_early) ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
            else
            {
                const float* sp = in_ss + (size_t)rr * npart;
                for (int i = lane; i < npart; i += 64) ss += sp[i];
            }
            ss = wave_allreduce_add(ss);
            const float rms = fast_rsqrt(ss * (1.0f / (float)K) + eps);
            for (int uu = lane; uu < xunits; uu += 64)
            {
                if (xu0 + uu >= oct) continue;
                f16* xp_ = x_lds + (size_t)rr * x_stride + (size_t)uu * 8;
                const f16x8 x = *(const f16x8*)xp_;
                const f16x8 w = *(const f16x8*)(nw_lds + (size_t)uu * 8);
                f16x8 v;
                #pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    const float f = fmaxf(-65504.0f, fminf((float)x[e], 65504.0f));
                    v[e] = (f16)((f * (float)w[e]) * rms);
                }
                *(f16x8*)xp_ = v;
            }
        }
    };

    LeanCtx cx;
    cx.x_lds = x_lds; cx.sc_lds = sc_lds; cx.zp_lds = zp_lds; cx.x_stride = x_stride; cx.xc0 = xc0; cx.M = M;

    // ---- the first segment (full items only, <= LeanDepth of them): everything requested at once, decoded item by item ----------
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto head = [&](auto bits_tag) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int D = LeanDepth<BITS>::v;
        constexpr size_t STEP = 64 * BITS;
        LaneWords<BITS> b[D];
        const int n = s0.n;
        #pragma unroll
        for (int q = 0; q < D; q++) if (q < n) load_lane_words<BITS>(s0.ptr + (size_t)q * STEP, lane, b[q]);
        LTRACE(3);
        wait_vmcnt_le<0>();                  // LDS-DMA copies landed (and the weights: measured free, profiles/r03_lean_probe.txt x0 vs x3)
        wave_converge();
        norm_prologue();
        wait_lds_reads();
        wave_converge();
        LTRACE(4);
        if (s0.uni)
        {
            #pragma unroll
            for (int q = 0; q < D; q++)
                if (q < n) { lean_item_uniform<BITS, GPTQ>(b[q], cx, s0.chunk0 + 4 * q, s0.g0 + ((4 * q + s0.gphase) >> s0.gshift), lane, acc); sched_fence(); }
        }
        else
        {
            #pragma unroll
            for (int q = 0; q < D; q++)
                if (q < n) { lean_item_general<BITS, GPTQ>(b[q], cx, s0.chunk0 + 4 * q, 4 * q, s0.g0, s0.gshift, s0.gphase, 4, lane, acc); sched_fence(); }
        }
    };
    if (s0.n == 0)
    {
        LTRACE(3);
        wait_vmcnt_le<0>();
        wave_converge();
        norm_prologue();
        wait_lds_reads();
        wave_converge();
        LTRACE(4);
    }
    else if constexpr (GPTQ) head(std::integral_constant<int, 4>());
    else switch (s0.bits)
    {
        case 4: head(std::integral_constant<int, 4>()); break;
        case 8: head(std::integral_constant<int, 8>()); break;
        case 6: head(std::integral_constant<int, 6>()); break;
        case 5: head(std::integral_constant<int, 5>()); break;
#if LEAN_LOWBITS
        case 3: head(std::integral_constant<int, 3>()); break;
        default: head(std::integral_constant<int, 2>()); break;
#else
        default: break;
#endif
    }

    // ---- the staged segments, from LDS ------------------------------------------------------------------------------------
    if (s1.n > 0)
    {
        u32 off = 0;
        auto staged = [&](const LeanSegV& sg) {
            #pragma nounroll
            for (int i = 0; i < sg.n; i++)
            {
                const u32* slot_ptr = (const u32*)(minor_lds + off);
                off += 256u * sg.bits;
                const int nv = (i == sg.n - 1) ? sg.nvalid : 4;
                auto consume = [&](auto bits_tag) {
                    constexpr int BITS = decltype(bits_tag)::value;
                    LaneWords<BITS> w;
                    lean_lds_words<BITS>(slot_ptr, lane, w);
                    lean_item_general<BITS, GPTQ>(w, cx, sg.chunk0 + 4 * i, 4 * i, sg.g0, sg.gshift, sg.gphase, nv, lane, acc);
                };
                if constexpr (GPTQ) consume(std::integral_constant<int, 4>());
                else switch (sg.bits)
                {
                    case 4: consume(std::integral_constant<int, 4>()); break;
                    case 8: consume(std::integral_constant<int, 8>()); break;
                    case 6: consume(std::integral_constant<int, 6>()); break;
                    case 5: consume(std::integral_constant<int, 5>()); break;
                    case 3: consume(std::integral_constant<int, 3>()); break;
                    default: consume(std::integral_constant<int, 2>()); break;
                }
            }
        };
        staged(s1);
        staged(s2);
    }

    // ---- partial sums meet in LDS ---------------------------------------------------------------------------------------------
    {
        const int c = lane & 15, j4 = lane >> 4;
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int row = j4 * 4 + q;
            if (row < M) red[(wv * M + row) * 16 + c] = acc[q];
        }
    }
    // the finalising waves (wave `row` finalises row `row`; they sit in slot 0, so the block of THEIR matrix is the output's):
    // lane -> output slot lane >> 4 (pair: the one act(gate) * up output), column lane & 15.  What the epilogue needs from
    // memory is requested before the barrier.
    const int row = wv;
    constexpr int N_OUT = PAIR ? 1 : NSLOTS;
    const int ep_slot = lane >> 4, ep_c = lane & 15;
    const int ep_tile = tile + (PAIR ? 0 : ep_slot);                    // (finalising waves sit in slot 0)
    const bool ep_on = wv < M && ep_slot < N_OUT && ep_tile < n_tiles;
    const int ep_n = ep_tile * 16 + ep_c;
    f16* cp = nullptr;
    f16 c_old = (f16)0.0f;
    int xp_idx = ep_n;
    f16* const xp_out = wv < M ? args.hdr.xp_out : nullptr;
    if (ep_on)
    {
        const u16* const c_invperm = args.mat[mj].c_invperm;
        const u16* const xp_invperm = args.hdr.xp_invperm;
        const int c_idx = c_invperm ? (int)c_invperm[ep_n] : ep_n;
        if (xp_out && xp_invperm) xp_idx = (int)xp_invperm[ep_n];
        cp = args.mat[mj].c + (size_t)row * args.hdr.ldc[mj] + c_idx;
        if (flags & LF_ACCUM) c_old = *cp;
    }
    LTRACE(5);
    block_sync_lds();
    LTRACE(6);
    if (wv >= M) return;

    // ---- combine (fixed order) + epilogue ------------------------------------------------------------------------------------
    auto slot_sum = [&](int s) -> float {
        float v = 0.0f;
        for (int w = s * S; w < s * S + S; w++) v += red[(w * M + row) * 16 + ep_c];
        return v;
    };
    float sq = 0.0f;
    if (ep_on)
    {
        f16 y;
        if constexpr (PAIR)
        {
            float gv = slot_sum(0), uv = slot_sum(1);
            if (flags & LF_BIAS)
            {
                if (args.mat[0].bias) gv += (float)args.mat[0].bias[ep_n];
                if (args.mat[1].bias) uv += (float)args.mat[1].bias[ep_n];
            }
            y = clamp_h(act_h((f16)gv, (flags & LF_GELU) != 0) * (f16)uv);
        }
        else
        {
            float v = slot_sum(ep_slot);
            if (flags & LF_BIAS) { const f16* bias = args.mat[mj].bias; if (bias) v += (float)bias[ep_n]; }
            if (flags & LF_ACCUM) v += (float)c_old;
            y = (f16)v;
        }
        *cp = y;
        if (xp_out)
        {
            xp_out[(size_t)row * args.hdr.ldxp + xp_idx] = y;
            const float f = fmaxf(-65504.0f, fminf((float)y, 65504.0f));
            sq = f * f;
        }
    }
    float* const ss_out = args.hdr.ss_out;
    if (ss_out)
    {
        sq = wave_allreduce_add(sq);
        if (lane == 0) ss_out[(size_t)row * args.hdr.wgs + u] = sq;
    }
    LTRACE(7);
}

#ifdef EXL2_TRACE
static u64* g_ltrace_buf = nullptr;
static int g_ltrace_which = 0, g_ltrace_count = 0;
extern "C" void exl2_debug_set_lean_trace(void* p, int which) { g_ltrace_buf = (u64*)p; g_ltrace_which = which; g_ltrace_count = 0; }
#endif

// ---- host: the split --------------------------------------------------------------------------------------------------------

static inline u32 al16(u32 x) { return (x + 15u) & ~15u; }

struct LeanItemRun { int F; int bits; int nvalid; int chunk0; u32 off, tstride; int in_tail; };

// Splits one tile's items (all runs, K order) over S waves in contiguous ranges of about equal byte cost; fills wave[0 .. S)
// (segments, activation slice, scale rows, LDS offsets for M rows).  Returns the LDS bytes of the S waves together, 0 when
// the matrix is not covered with S waves (more segments per wave than the record carries, more items of the register
// segment than a wave holds, a chunk -> group map that is not affine inside a segment, ...).
static u32 lean_plan_matrix(const QMatrix* qm, int S, int M, bool norm, LeanWave* wave)
{
    const QMatDev& d = qm->dev;
    if (d.n_runs <= 0 || !qm->cg_host || !d.sc_tab || (qm->is_gptq && !d.zp_tab)) return 0;
    std::vector<LeanItemRun> runs;
    long long total = 0;
    for (int i = 0; i < d.n_runs; i++)
    {
        const QRun& r = d.runs[i];
        LeanItemRun t;
        t.F = r.n_super; t.bits = r.bits; t.nvalid = r.nvalid_last; t.chunk0 = (int)r.k_base >> 5;
        t.off = r.base_word; t.tstride = r.tile_stride; t.in_tail = r.in_tail;
        runs.push_back(t);
        total += (long long)r.n_super * r.bits;
    }
    if (total <= 0) return 0;
    const int n_chunks = d.K / 32;
    u32 lds_total = 0;
    // walk the items in K order; wave w takes items until the running cost reaches (w + 1) / S of the total
    size_t ri = 0; int ii = 0; long long done = 0;
    for (int w = 0; w < S; w++)
    {
        LeanWave& lw = wave[w];
        memset(&lw, 0, sizeof(lw));
        lw.lds_off = lds_total;
        const long long goal = total * (w + 1) / S;
        int nseg = 0;
        struct Tmp { LeanItemRun r; int i0, n; } segs[LEAN_SEGS];
        while (ri < runs.size() && (done < goal || w == S - 1))
        {
            const LeanItemRun& r = runs[ri];
            // items of this run the wave takes: up to the goal, rounded to the nearest item (at least one); the last wave takes the rest
            long long want = (goal - done + r.bits / 2) / r.bits;
            if (want < 1) want = 1;
            if (w == S - 1) want = r.F - ii;
            const int take = (int)(want < (long long)(r.F - ii) ? want : (long long)(r.F - ii));
            if (nseg >= LEAN_SEGS) return 0;
            segs[nseg].r = r; segs[nseg].i0 = ii; segs[nseg].n = take; nseg++;
            done += (long long)take * r.bits;
            ii += take;
            if (ii >= r.F) { ri++; ii = 0; }
            else break;                                   // the run goes on: the next wave continues it
        }
        if (nseg == 0) continue;
        // segment 0 is requested into registers: the largest one made of full items; a partial super-chunk is always staged
        auto full = [&](const Tmp& t) { return !(t.i0 + t.n == t.r.F && t.r.nvalid != 4); };
        int big = -1;
        for (int q = 0; q < nseg; q++)
            if (full(segs[q]) && (big < 0 || (long long)segs[q].n * segs[q].r.bits > (long long)segs[big].n * segs[big].r.bits)) big = q;
        if (big >= 0 && segs[big].n > lean_depth(segs[big].r.bits)) return 0;
        Tmp ord[LEAN_SEGS]; int no = 0;
        const bool has0 = big >= 0;
        if (has0) ord[no++] = segs[big]; else no = 1;
        for (int q = 0; q < nseg; q++) if (q != big) { if (no >= LEAN_SEGS) return 0; ord[no++] = segs[q]; }
        const int first = has0 ? 0 : 1;
        // ranges of chunks and groups the wave touches
        int c_lo = 0x7fffffff, c_end = -1, g_lo = 0x7fffffff, g_hi = -1;
        for (int q = first; q < no; q++)
        {
            const Tmp& t = ord[q];
            const bool last_of_run = t.i0 + t.n == t.r.F;
            const int c0 = t.r.chunk0 + 4 * t.i0;
            const int nch = 4 * (t.n - 1) + (last_of_run ? t.r.nvalid : 4);
            if (c0 + nch > n_chunks) return 0;
            if (c0 < c_lo) c_lo = c0;
            if (c0 + 4 * t.n > c_end) c_end = c0 + 4 * t.n;      // a partial item reads 4 chunks' worth of activations (clamped to the row)
            for (int cc = c0; cc < c0 + nch; cc++)
            {
                const int g = qm->cg_host[cc];
                if (g < g_lo) g_lo = g;
                if (g > g_hi) g_hi = g;
            }
        }
        if (c_lo > 0xFFFF || c_end - c_lo > 0xFFFF || g_lo > 0xFFFF || g_hi - g_lo + 1 > 0xFFFF) return 0;
        lw.xr = (u32)c_lo | ((u32)(c_end - c_lo) << 16);
        lw.gr = (u32)g_lo | ((u32)(g_hi - g_lo + 1) << 16);
        u32 minor = 0;
        for (int q = first; q < no; q++)
        {
            const Tmp& t = ord[q];
            const bool last_of_run = t.i0 + t.n == t.r.F;
            const int c0 = t.r.chunk0 + 4 * t.i0;
            const int nch = 4 * (t.n - 1) + (last_of_run ? t.r.nvalid : 4);
            // affine group map inside the segment: row(chunk) = g0 + ((chunk - c0 + phase) >> shift)
            const int g0 = qm->cg_host[c0];
            int phase = 0; while (c0 - phase - 1 >= 0 && qm->cg_host[c0 - phase - 1] == g0 && phase < 1023) phase++;
            int shift = -1;
            for (int sh = 0; sh <= 7 && shift < 0; sh++)
            {
                if ((phase >> sh) != 0) continue;                              // the phase lies inside the first group
                bool ok = true;
                for (int cc = 0; cc < nch && ok; cc++) ok = qm->cg_host[c0 + cc] == g0 + ((cc + phase) >> sh);
                if (ok) shift = sh;
            }
            if (shift < 0 || t.n > 0x3FF) return 0;
            const bool uni = shift >= 2 && (phase & 3) == 0;                   // the four chunks of every item share a group
            LeanSeg& o = lw.seg[q];
            o.off = t.r.off + (u32)t.i0 * 64u * (u32)t.r.bits;
            o.tstride = t.r.tstride;
            o.meta = (u32)t.n | ((u32)t.r.bits << 10) | ((u32)(last_of_run ? t.r.nvalid : 4) << 14) | ((u32)(t.r.in_tail ? 1 : 0) << 17) |
                     ((u32)shift << 18) | ((u32)phase << 21) | (uni ? 0x80000000u : 0u);
            o.place = (u32)c0 | ((u32)(g0 - g_lo) << 16);
            if (q > 0) minor += (u32)t.n * 256u * (u32)t.r.bits;
        }
        const u32 x_stride = (u32)(c_end - c_lo) * 32u + 8u;
        lds_total += al16((u32)M * x_stride * 2u) + (norm ? (u32)(c_end - c_lo) * 64u : 0u) + al16((u32)(g_hi - g_lo + 1) * 32u) * (qm->is_gptq ? 2u : 1u) + al16(minor);
    }
    if (ri < runs.size()) return 0;
    return lds_total ? lds_total : 16u;
}

#define LEAN_FOR_EACH_GEOMETRY(X, OCC) X(8, 1, false, OCC) X(16, 1, false, OCC) X(8, 2, false, OCC) X(8, 2, true, OCC)
// register budgets built side by side (EXL2_LEAN_OCC = 4 / 6 / 8 waves per SIMD; measured on the MI355X): the
// default is what the A/B runs picked
#ifndef LEAN_OCC_DEFAULT
#define LEAN_OCC_DEFAULT 6
#endif

static void lean_attrs()
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (!exl2_first_on_device(attr)) return;
#define LEAN_ATTR(S, NS, P, OCC) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, S, NS, P, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, 4) LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, 6) LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, 8)
#undef LEAN_ATTR
#define LEAN_ATTR(S, NS, P, OCC) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, S, NS, P, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, 6)
#undef LEAN_ATTR
}

// 0: launched; 1: shape not covered (the caller takes the round-2 kernel).  *wgs_out = grid size = partial sums per row a
// chain-out launch publishes.
int qgemv_lean_launch(const FlatIn& in, void* stream, int* wgs_out)
{
    if (in.n_mats < 1 || in.n_mats > FLAT_MAX_MATS || in.M < 1 || in.M > LEAN_MAX_M) return 1;
    if (in.sync_wait || in.sync_signal || in.sync_arrive) return 1;
    const QMatrix* q0 = in.qm[0];
    const int K = q0->height;
    if ((K & 7) || (((size_t)in.a) & 15) || (in.lda & 7)) return 1;
    if (in.a_mode == A_NORM_PRE && ((((size_t)in.norm_w) & 15) || !in.ss || in.npart < 1)) return 1;
    if (in.pair && (in.n_mats != 2 || in.qm[0]->width != in.qm[1]->width)) return 1;
    static LeanArgs args_store;                                    // (large: off the stack; the launch copies it)
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    LeanArgs& a = args_store;
    memset(&a, 0, sizeof(a));
    LeanHdr& h = a.hdr;
    int max_tiles = 0;
    bool any_bias = false;
    for (int j = 0; j < in.n_mats; j++)
    {
        const QMatrix* qm = in.qm[j];
        if (qm->height != K || qm->is_gptq != q0->is_gptq) return 1;
        if (qm->width / TILE_N > max_tiles) max_tiles = qm->width / TILE_N;
        if (qm->dev.bias) any_bias = true;
    }
    // geometry: a pair -> two tiles x 8 waves; a chain-out launch that would publish more partial sums of squares than its
    // consumer reads -> two tiles x 8 waves; else one tile x 8 waves, or x 16 when 8 waves cannot hold their share in registers
    int S = 8, nslots = in.pair ? 2 : 1;
    if (!in.pair && in.n_mats == 1 && in.ss_out && max_tiles > LEAN_MAX_PART) nslots = 2;
    if (const char* e = getenv("EXL2_LEAN_TPW")) { const int v = atoi(e); if (!in.pair && in.n_mats == 1 && (v == 1 || v == 2)) nslots = v; }
    if (const char* e = getenv("EXL2_LEAN_S16")) { if (atoi(e) && nslots == 1 && in.n_mats <= 2) S = 16; }
    u32 slot_bytes = 0;
    for (int attempt = 0; attempt < 2; attempt++)
    {
        if (in.n_mats * S > LEAN_RECORDS) return 1;
        bool ok = true;
        slot_bytes = 0;
        for (int j = 0; j < in.n_mats && ok; j++)
        {
            const u32 b = lean_plan_matrix(in.qm[j], S, in.M, in.a_mode == A_NORM_PRE, a.wave + j * S);
            if (!b) ok = false;
            if (b > slot_bytes) slot_bytes = b;
        }
        if (ok && (u32)nslots * slot_bytes + (u32)(S * nslots) * (u32)in.M * 64u <= LEAN_LDS_BUDGET * (u32)(S * nslots / 8)) break;
        if (attempt == 1 || nslots != 1 || S == 16) return 1;
        S = 16;                                                       // finer split of the tile
    }
    const int wgs = in.pair ? max_tiles : (max_tiles + nslots - 1) / nslots;
    if (in.ss_out && wgs > LEAN_MAX_PART) return 1;
    for (int j = 0; j < in.n_mats; j++)
    {
        const QMatDev& d = in.qm[j]->dev;
        LeanMat& m = a.mat[j];
        m.qw = d.qw; m.tail = d.tail; m.sc_tab = d.sc_tab; m.zp_tab = d.zp_tab;
        m.c = in.c[j]; m.c_invperm = in.c_invperm[j]; m.bias = d.bias;
        m.G = d.G; m.n_tiles = d.N / TILE_N; h.ldc[j] = in.ldc[j];
    }
    h.a = in.a; h.norm_w = in.norm_w; h.ss = in.ss; h.xp_out = in.xp_out; h.xp_invperm = in.xp_invperm; h.ss_out = in.ss_out;
    h.eps = in.eps; h.M = in.M; h.K = K; h.lda = in.lda; h.ldxp = in.ldxp; h.npart = in.npart; h.wgs = wgs;
    h.flags = (in.a_mode == A_NORM_PRE ? LF_NORM : 0u) | (in.act_gelu ? LF_GELU : 0u) | (in.c_mode == C_ACCUM ? LF_ACCUM : 0u) | (any_bias ? LF_BIAS : 0u);
    h.slot_bytes = slot_bytes; h.red_off = slot_bytes * (u32)nslots;
    const int waves = S * nslots;
    const u32 lds = h.red_off + (u32)waves * (u32)in.M * 16u * 4u;
    lean_attrs();
#ifdef EXL2_TRACE
    h.trace = (g_ltrace_buf && g_ltrace_count++ == g_ltrace_which) ? g_ltrace_buf : nullptr;
#endif
    if (getenv("EXL2_LEAN_TRACE"))
    {
        fprintf(stderr, "[lean] M=%d K=%d mats=%d pair=%d S=%d slots=%d wgs=%d lds=%u mode=%d\n", in.M, K, in.n_mats, in.pair, S, nslots, wgs, lds, in.a_mode);
        for (int j = 0; j < in.n_mats; j++)
            for (int w = 0; w < S; w++)
            {
                const LeanWave& lw = a.wave[j * S + w];
                fprintf(stderr, "[lean]  mat %d wave %d: lds+%u, x chunks %u+%u, sc rows %u+%u;", j, w, lw.lds_off, lw.xr & 0xFFFF, lw.xr >> 16, lw.gr & 0xFFFF, lw.gr >> 16);
                for (int q = 0; q < LEAN_SEGS; q++)
                    if (lw.seg[q].meta & 0x3FF)
                        fprintf(stderr, "  %d:[%u x %ub nv%u%s chunk %u row+%u sh%u ph%u%s]", q, lw.seg[q].meta & 0x3FF, (lw.seg[q].meta >> 10) & 0xF, (lw.seg[q].meta >> 14) & 7,
                                ((lw.seg[q].meta >> 17) & 1) ? " tail" : "", lw.seg[q].place & 0xFFFF, lw.seg[q].place >> 16, (lw.seg[q].meta >> 18) & 7,
                                (lw.seg[q].meta >> 21) & 0x3FF, (lw.seg[q].meta >> 31) ? " uni" : "");
                fprintf(stderr, "\n");
            }
    }
    dim3 grid((unsigned)wgs, (unsigned)(in.pair ? 1 : in.n_mats), 1), block((unsigned)waves * 64, 1, 1);
    const bool gptq = q0->is_gptq;
    int occ = LEAN_OCC_DEFAULT;
    if (const char* e = getenv("EXL2_LEAN_OCC")) { const int v = atoi(e); if (v == 4 || v == 6 || v == 8) occ = v; }
#define LEAN_GO(SS, NS, P, OCC) \
    if (!gptq && occ == OCC && S == SS && nslots == NS && (in.pair != 0) == P) LAUNCH((qgemv_lean_kernel<false, SS, NS, P, OCC>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_GEOMETRY(LEAN_GO, 4) LEAN_FOR_EACH_GEOMETRY(LEAN_GO, 6) LEAN_FOR_EACH_GEOMETRY(LEAN_GO, 8)
#undef LEAN_GO
#define LEAN_GO(SS, NS, P, OCC) \
    if (gptq && S == SS && nslots == NS && (in.pair != 0) == P) LAUNCH((qgemv_lean_kernel<true, SS, NS, P, OCC>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_GEOMETRY(LEAN_GO, 6)
#undef LEAN_GO
    if (wgs_out) *wgs_out = wgs;
    return 0;
}
