// qgemv_lean.hip -- decode q_gemm for a CHAIN of modules, round 3: the launch shape the MI355X measurements asked for.
//
// Replaces gemm_half_q_half_kernel (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565) + rms_norm_kernel (rms_norm.cu:33-175)
// + act_mul_kernel (q_mlp_activation.cuh:54-112) on the decode path, as composed by QAttn::forward_cuda_1 / _2
// (q_attn.cu:153-345) and QMLP::forward_run_ (q_mlp.cu:153-236).  Same host interface as qgemv_flat.hip (FlatIn); that
// kernel remains the route for what this one declines (> LEAN_MAX_M rows, grouped MoE launches, shares too big for the
// register stream).
//
// Why another shape (profiles/r02_trace_flat.txt, profiles/history/r03_lean_probe.txt, profiles/history/r03_trace_lean_v2.txt): the round-2
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
#include <stddef.h>
#include <type_traits>
#include <vector>
#include <mutex>

// tuning switches (A/B builds through tools/build_variant.sh; the defaults are what the MI355X runs picked)
#ifndef LEAN_PRESCALE
#define LEAN_PRESCALE 1               // 1: general items scale the weights (fp16, reconstruct()'s rounding); 0: four scales on four partial sums
#endif
#ifndef LEAN_KILL
#define LEAN_KILL 0                   // instruction-count / timing experiments (results are WRONG): 1 no decode (the loads vanish too), 4 no epilogue, 8 loads kept, decode = xor
#endif
#ifndef LEAN_REPEAT
#define LEAN_REPEAT 0                 // timing experiment (results are WRONG): > 0 = the decode of a wave's share runs that many times
#endif
#ifndef LEAN_LOWBITS
#define LEAN_LOWBITS 1                // 0: no 2 / 3-bit register stream (code-size experiment; such segments would be wrong)
#endif
#ifndef LEAN_RAW2
#define LEAN_RAW2 1                   // 4-bit items: the matrix cores get the UN-subtracted codes (half2 1024 + q / 64 + q: one v_and_or_b32 per pair, no
#endif                                // v_pk_add / v_pk_fma) and the constant part leaves through a second MFMA per chunk against a constant B fragment
                                      // -(1024 + z) / -(64 + z): 20 vector instructions per item instead of 50, no pre-pass (round 6; A/B in profiles/r06_raw2_ab.txt)
#ifndef LEAN_DUO
#define LEAN_DUO 0                    // 2 / 3-bit items whose groups are chunk pairs (group size 64): one scale per pair on the pair's fp32 sum (lean_item_general).
                                      // Built and measured in round 6 (profiles/r09e_ab_*_duo.txt, same box): 16 packed multiplies + 2 scale reads fewer per
                                      // item move the 70B gate|up launch 48.5 -> 48.2 us (the launch is NOT bound by instruction issue), Mixtral bs=1 +0.8 %,
                                      // and the larger code costs the LOADS instantiation 6 spilled registers (70B down 33.8 -> 36.5 us, 70B -1.7 %): off
#endif
#ifndef LEAN_ITEM_FENCE
#define LEAN_ITEM_FENCE 1             // a scheduling fence behind every item's decode (register pressure); 0: the compiler may interleave items (A/B experiment)
#endif
#ifndef LEAN_BUF_DMA
#define LEAN_BUF_DMA 1                // the wave-private staging copies in the buffer form (hw.h: dma_buf_to_lds16): counted waits stay exact
#endif
#if LEAN_BUF_DMA
#define LEAN_DMA(base, off_bytes, lds) dma_buf_to_lds16(base, (u32)(off_bytes), lds)
#define LEAN_DMA_X(base, off_bytes, lds) dma_buf_to_lds16_agent(base, (u32)(off_bytes), lds)     // activations: another launch wrote them
#else
#define LEAN_DMA_X(base, off_bytes, lds) dma_to_lds16((const u8*)(base) + (size_t)(off_bytes), lds)
#define LEAN_DMA(base, off_bytes, lds) dma_to_lds16((const u8*)(base) + (size_t)(off_bytes), lds)
#endif
#define LEAN_MARK(id) ASM_MARK(id)     // (hw.h: a numbered comment in the gfx950 code; the emulation twin's hw_emu.h makes it nothing)
#ifndef LEAN_XMEM_AHEAD
#define LEAN_XMEM_AHEAD 3             // XMEM form: items whose A operands are in registers or in flight (16 registers each)
#endif
#define LEAN_X_PIECES_BASE 4           // 1 KB copy instructions per activation row of a wave's slice (stage_copies; lean_plan_matrix declines longer slices)
#define LEAN_X_PIECES_PAIR_LOADS 6     // ... of the pair geometry with several register loads per wave (its own instantiation)
#define LEAN_MAX_WAVES 16
#define LEAN_RECORDS 48               // wave records in the argument block: matrices x waves per tile (q|k|v at 16 waves)
#define LEAN_MAX_PASSES 4             // 16-wave geometry: a wave's share may be this many register loads (qgemv_lean_kernel, further passes)
#ifndef LEAN_S8_PASSES
#define LEAN_S8_PASSES 2              // 8-wave geometry, launches of several matrices (q|k|v): register loads a share may take (1 = round-4 behaviour)
#endif
#ifndef LEAN_S8_PASS_BITS
#define LEAN_S8_PASS_BITS 4           // ... for items of at most this many bits; also the widest items decoded as a RING in either geometry (wider decoders + a full
                                      // ring of requests do not fit 80 registers: spills, and the 5-bit pipelined region of the 16-wave kernel lost its counted wait)
#endif
#ifndef LEAN_PASS_RING
#define LEAN_PASS_RING 1              // shares of several register loads: item q of the NEXT load is requested as soon as item q of this one is decoded
#endif
#define LEAN_MAX_PART 512             // partial sums of squares per row a chain-out launch may publish (hidden 8192 = 512 tiles; the
                                      // host's ss buffers are [rows, 512]; the round-2 kernel reads at most 256 and declines more)
#define LEAN_LDS_BUDGET (40u * 1024u)  // per 8 waves: four 8-wave / two 16-wave workgroups stay resident on a CU

// What one wave of a workgroup does for its tile: ONE run of full items of one bit width (requested into registers, <= LeanDepth
// of them) plus, when the run ends in a partial super-chunk and this wave holds its end, that partial item (it lives in the
// matrix' side buffer); one contiguous range of the activation row, one contiguous range of scale-table rows, its private LDS
// area.  64 bytes = one scalar load.
struct alignas(64) LeanWave
{
    u32 w_off, w_tstride;             // full items: word offset of the first one for tile 0 in qw, words between consecutive tiles
    u32 t_off, t_tstride;             // the partial item: the same in the side buffer
    u32 meta;                         // n (0..7) | bits (8..11) | nvalid of the partial item, 0 = none (12..14) | uniform (15) | gshift (16..18) | gphase (19..28) | pipelined form off (29) | launch of an overlapped chain (30)
    u32 xr;                           // first chunk (0..15) | number of chunks (16..31) of the activation slice
    u32 gr;                           // first scale-table row (0..15) | number of rows (16..31)
    u32 lds_off;                      // byte offset of the wave's LDS area inside its slot's area
    u32 place;                        // first 32-row chunk of the run's part (0..15) | scale-table row of that chunk relative to the wave's first row (16..31)
    u32 off_sc, off_zp;               // byte offsets of the wave's scale / zero-point rows inside its LDS area (the activation rows come first)
    u32 x_stride;                     // halfs between the activation rows of the wave's LDS area (made at plan time: the launch's M is known)
    u32 pad[4];
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
    const f16* a; const f16* xp_w;    // xp_w: chain-out, the next consumer's norm weight in its packed order (nullable = 1)
    const float* ss; float eps; int M;
    int K, lda, npart; u32 flags;     // flags: a_mode (0) | gelu (3) | c_accum (4) | any_bias (5)
    u32 slot_bytes, red_off;          // LDS bytes of one slot's waves; offset of the partial sums
    u64* trace;                       // EXL2_TRACE build: [matrix][workgroup][wave][8] realtime stamps (tools/trace_lean.py)
    f16* xp_out; const u16* xp_invperm;
    float* ss_out; int ldxp, wgs;
    int ldc[FLAT_MAX_MATS];
    // overlapped chain (chain_sync.h; flags & LF_DEP): the producer launch's block ("go" is polled before the activations are staged,
    // they and the partial sums / residual are read at agent scope), this launch's block (every workgroup reports on entry; outputs
    // are agent-scope stores; every finalising wave arrives, the last publishes "go"), workgroups of the grid
    const u32* sync_wait; u32* sync_signal; u32 sync_wgs, sync_pad;
    const f16* out_scale;             // MOE instantiations only: the expert's routing weight of row 0 (q_mlp.cu:373-384); nullable
    int r_stride;                     // ... of row r at out_scale[r * r_stride]; != 0 = a launch over ALL experts (2-4 rows): a workgroup whose expert
    u32 moe_mul;                      //     no row is routed to leaves at entry (q_gemm_kernel.cuh:189-200).  moe_mul: the finished sums are multiplied by the weight
};
static_assert(sizeof(LeanHdr) == 192, "LeanHdr: the dense kernels' offsets must not move");
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
#define LF_DEP 64u
#define LF_ATILED 128u                // the activations are in the MFMA-tiled layout (qgemv_flat.h: FlatIn.a_tiled)
#define LF_CTILED 256u                // the output is written in that layout (FlatIn.c_tiled)
#define LF_XPTILED 512u               // chain-out: xp_out in that layout (FlatIn.xp_tiled)

// items of one bit width a wave may hold in registers (<= 25 dwords per lane in flight)
// (S = 4: the 8-wave gate|up workgroup, built for 6 waves per SIMD = 80 registers -- a wave there holds 8 items of <= 4 bits)
template <int BITS, int S> struct LeanDepth
{
    static constexpr int v = S == 4 ? (BITS == 8 ? 4 : BITS == 6 ? 5 : BITS == 5 ? 6 : BITS == 2 ? 10 : 8)
                                    : (BITS == 8 ? 3 : BITS == 6 ? 4 : BITS == 5 ? 5 : BITS == 4 ? 6 : BITS == 2 ? 10 : 8);
};
// how many of those (the LAST ones of a share) are requested behind the fence load and decoded one by one as they land
// (qgemv_lean_kernel: head); 0 = the round-3 form only.  LEAN_PIPE: 0 off, 1 = D / 2, 2 = D - 2 (at least 1)
#ifndef LEAN_PIPE
#define LEAN_PIPE 2
#endif
template <int BITS, int S> struct LeanTail
{
    static constexpr int D = LeanDepth<BITS, S>::v;
    static constexpr int v = LEAN_PIPE == 0 ? 0 : LEAN_PIPE == 1 ? D / 2 : (D - 2 > 1 ? D - 2 : 1);
};
static int lean_depth(int bits, int S)
{
    return S == 4 ? (bits == 8 ? 4 : bits == 6 ? 5 : bits == 5 ? 6 : bits == 2 ? 10 : 8)
                  : (bits == 8 ? 3 : bits == 6 ? 4 : bits == 5 ? 5 : bits == 4 ? 6 : bits == 2 ? 10 : 8);
}

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
// PRE (the XMEM form of the kernel): the four A operands were requested from MEMORY ahead of time and arrive in `pre`.
#if LEAN_RAW2
// B fragment of one 32-row chunk straight from its packed dword: pairs (e = 0, 1) and (e = 4, 5) as 1024 + q, (2, 3) and (6, 7) as 64 + q
// (the nibble sits where the half's unit bit is: 0x6400 = 1024, ulp 1; 0x5400 = 64, ulp 1 / 16).  Plain C so that the compiler sees
// the VALU write in front of the MFMA (an inline-asm and_or feeding an MFMA hides the hazard: profiles/history/r05_raw4_experiment.txt).
DEV f16x8 raw4_b(u32 x, u32 m_lo, u32 m_hi, u32 k_lo, u32 k_hi)
{
    const u32 y = x >> 8;
    const u32x4 bw = {(x & m_lo) | k_lo, (x & m_hi) | k_hi, (y & m_lo) | k_lo, (y & m_hi) | k_hi};
    return __builtin_bit_cast(f16x8, bw);
}
// the constant B fragment that takes the bias back out: -(1024 + z) where the code was fed as 1024 + q, -(64 + z) where as 64 + q
DEV f16x8 raw4_negc(f16 z)
{
    const f16 a = -((f16)1024.0f + z), b = -((f16)64.0f + z);
    return (f16x8){a, a, b, b, a, a, b, b};
}
struct Raw4K { u32 m_lo, m_hi, k_lo, k_hi; };
DEV Raw4K raw4_consts()
{
    Raw4K k = {0x000F000Fu, 0x00F000F0u, 0x64006400u, 0x54005400u};
    pin_scalar(k.m_lo); pin_scalar(k.m_hi); pin_vector(k.k_lo); pin_vector(k.k_hi);
    return k;
}
#endif

template <int BITS, bool GPTQ, bool PRE = false>
DEV void lean_item_uniform(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int g, int lane, f32x4& acc, const f16x8* pre = nullptr)
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
    const float s = (float)cx.sc_lds[g * 16 + c];
#if LEAN_RAW2
    if constexpr (BITS == 4)
    {
        const Raw4K k = raw4_consts();
        f16 z = (f16)8.0f;
        if constexpr (GPTQ) z = cx.zp_lds[g * 16 + c];
        const f16x8 negc = raw4_negc(z);
        f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const f16x8 b = raw4_b(lw.w[q], k.m_lo, k.m_hi, k.k_lo, k.k_hi);
            f16x8 a;
            if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
            part = mfma_16x16x32_f16(a, b, part);
            part = mfma_16x16x32_f16(a, negc, part);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = fmaf(s, part[i], acc[i]);
        return;
    }
#endif
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
        f16x8 a;
        if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
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
template <int BITS, bool GPTQ, bool PRE = false>
DEV void lean_item_general(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int cs, int g0, int gshift, int gphase, int nvalid,
                           int lane, f32x4& acc, const f16x8* pre = nullptr)
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
#if LEAN_RAW2
    if constexpr (BITS == 4)
    {
        // a scale per chunk: each chunk's two MFMAs start from zero and the scale goes onto the fp32 sums (the same value
        // reconstruct() rounds to for a one-hot row: fp16 scale x small integer is exact in fp32)
        const Raw4K k = raw4_consts();
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            if (q < nvalid)
            {
                const int gi = (g0 + ((cs + q + gphase) >> gshift)) * 16 + c;
                f16 z = (f16)8.0f;
                if constexpr (GPTQ) z = cx.zp_lds[gi];
                const float sq = (float)cx.sc_lds[gi];
                const f16x8 b = raw4_b(lw.w[q], k.m_lo, k.m_hi, k.k_lo, k.k_hi);
                f16x8 a;
                if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
                f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
                part = mfma_16x16x32_f16(a, b, part);
                part = mfma_16x16x32_f16(a, raw4_negc(z), part);
                #pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = fmaf(sq, part[i], acc[i]);
            }
        }
        return;
    }
#endif
#if LEAN_DUO
    if constexpr (BITS <= 3 && !GPTQ)
    {
        // round 6: groups of 64 rows on chunk-pair boundaries (2 / 3-bit items of the 2.5 / 3.5 bpw recipes: q, k, o, gate, up of the 70B) --
        // the two chunks of a group accumulate into one fp32 chain and the group's scale goes onto that sum, as the uniform form does
        // for a whole item (for a one-hot row still reconstruct()'s value: fp16 scale x small integer is exact in fp32): 8 fp32
        // multiply-adds and two scale reads per item instead of 16 packed multiplies and four reads.  At these widths the launch is bound
        // by instruction issue (70B gate|up: ~520 cycles per item and SIMD).
        if (gshift == 1 && (gphase & 1) == 0 && nvalid == 4)                  // (cs is a multiple of 4; uniform over the wave)
        {
            const ZC z = make_zc((f16)(float)(1 << (BITS - 1)));
            ZC zc4[4] = {z, z, z, z};
            f16x2 p[16];
            dequant_super<BITS>(lw.w, zc4, p);
            const int gi = (g0 + ((cs + gphase) >> 1)) * 16 + c;
            const float s01 = (float)cx.sc_lds[gi], s23 = (float)cx.sc_lds[gi + 16];
            #pragma unroll
            for (int h = 0; h < 2; h++)
            {
                f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
                #pragma unroll
                for (int q = 2 * h; q < 2 * h + 2; q++)
                {
                    const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y, p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
                    f16x8 a;
                    if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
                    part = mfma_16x16x32_f16(a, b, part);
                }
                const float sh = h ? s23 : s01;
                #pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = fmaf(sh, part[i], acc[i]);
            }
            return;
        }
    }
#endif
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
            f16x8 a;
            if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
            part = mfma_16x16x32_f16(a, b, part);
        }
    }
    #pragma unroll
    for (int i = 0; i < 4; i++) acc[i] += part[i];
}
#else
template <int BITS, bool GPTQ, bool PRE = false>
DEV void lean_item_general(const LaneWords<BITS>& lw, const LeanCtx& cx, int chunk, int cs, int g0, int gshift, int gphase, int nvalid,
                           int lane, f32x4& acc, const f16x8* pre = nullptr)
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
            f16x8 a;
            if constexpr (PRE) a = pre[q]; else a = *(const f16x8*)(arow + q * 32);
            const f32x4 part = mfma_16x16x32_f16(a, b, zero4);
            #pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = fmaf(s[q], part[i], acc[i]);
        }
    }
}
#endif

DEV const void* ptr_of(u32 lo, u32 hi) { return (const void*)global_ptr_of(lo, hi); }

// XMEM: the A operands of one item (4 chunks x 16 rows x 32 halfs) requested from memory -- cx.x_lds is the activations' GLOBAL
// base there, x_stride their row stride: per chunk lane (c, j) reads the 16 bytes at row min(c, M - 1), column 32 chunk + 8 j
DEV void lean_xmem_request(const LeanCtx& cx, int chunk, int nvalid, int lane, f16x8 (&v)[4])
{
    const int c = lane & 15, j = lane >> 4;
    const int mrow = c < cx.M ? c : cx.M - 1;
    const f16* arow = cx.x_lds + (size_t)mrow * cx.x_stride + (chunk - cx.xc0) * 32 + 8 * j;
    #pragma unroll
    for (int q = 0; q < 4; q++) if (q < nvalid) v[q] = *(const f16x8*)(arow + q * 32);
}

// Geometry (template): S = waves per tile (8 / 16), NSLOTS = tiles per workgroup (1 / 2), PAIR = the two tiles are tile u of
// matrix 0 (gate) and of matrix 1 (up) and the epilogue writes act(gate) * up.  Otherwise blockIdx.y = matrix.
// OCC = waves per SIMD the register allocation leaves room for.
#define LEAN_BOUNDS(T, OCC) __launch_bounds__(T, OCC)
// ROWS (5 .. 16 rows; round 4): the M x K activations no longer fit as wave-private slices next to each other, so the
// workgroup stages the WHOLE rows once, cooperatively (every wave copies a share; one barrier), into an area all its tiles
// read: the pair geometry's gate and up tiles -- and their 16 waves -- share one copy.  One workgroup per CU then (128 KB at
// 16 x 4096); no pipelined form (the barrier sits between the requests and the decode anyway), every finalising wave takes
// several rows.  Replaces the row pre-pass + K-phased kernels (qgemv_stream.hip / qgemm_prefill.hip) on the chained path
// wherever M x (K + 8) x 2 bytes fit in LDS; larger K is served as row groups of <= 4 rows by the host (model.py).
// WALK (ROWS only): the workgroup takes units u, u + grid, ... with its one staged copy of the rows (two-tile geometries: a 7B gate|up
// launch is 688 tile pairs on 256 CUs); without it one unit per workgroup (fewer registers: the 16-wave geometry of down_proj's row
// groups spills otherwise).
// DEP: a launch of the overlapped chain (chain_sync.h; EXPERIMENTAL).  A template parameter, not a run-time flag: the run-time form
// cost the ordinary launches 2-4 % (a few scalar loads and branches on the request path and in the finalising wave's tail:
// profiles/history/r04_bisect.txt).
// XMEM (5 .. 16 rows; round 4, second form): no staged activations at all -- a wave requests the A operands of its items from
// memory (the L2 holds the M x K activations), LEAN_XMEM_AHEAD items ahead of the one it decodes, next to its weights.  For
// launches whose rows do not fit the LDS as a whole (down_proj at K = 11008: 16 rows = 352 KB) this replaces the host's row
// GROUPS (every group re-read all weights: three launches of 12 + 12 + 10 us) by one launch; LDS holds scale rows and the
// partial sums only, so several workgroups share a CU again.
// The kernel's body is a function of WHERE the argument block lies: the kernel arguments (qgemv_lean_kernel: every launch of the dense
// models) or a table in device memory indexed by blockIdx.y (qgemv_lean_moe_kernel: the launch's y-th SELECTED expert, whose block
// the MoE front kernel copied there -- the graph's launch is fixed, the experts are not).  MOE: the output may carry the routing
// weight (hdr.out_scale); `by` = the matrix index of a launch over several matrices.
// MOE_SUM (the down projections of the TWO selected experts of a one-row MoE step as one pair launch): what is not known when the
// experts' blocks are planned -- the residual stream the sum goes to and the hand-off for the next consumer -- comes as kernel arguments
struct LeanDyn { f16* c; f16* xp_out; const u16* xp_invperm; const f16* xp_w; float* ss_out; int ldc, ldxp; };

// LOADS (ROWS + WALK, two tiles x 8 waves; round 6): ONE row of a long K whose tiles outnumber the CUs (70B down_proj: K = 28672, 512
// tiles) -- the plain form is a 16-wave workgroup per tile, alone on its CU with its 57 KB copy of the row + 28 KB of scale rows:
// two rounds of 256.  Here a workgroup takes two tiles, its 16 waves stage the row ONCE (the ROWS form's shared copy), every wave's
// share is up to LEAN_MAX_PASSES register loads (the ring), up to 128 scale rows: one round.
template <bool GPTQ, int S, int NSLOTS, bool PAIR, int OCC, bool ROWS = false, bool WALK = false, bool DEP = false, bool XMEM = false, bool MOE = false, bool LOADS = false>
DEV void lean_body(const LeanArgs& args, const int by, const LeanDyn* const dyn = nullptr)
{
    static_assert(!LOADS || (ROWS && WALK && S == 8 && NSLOTS == 2 && !PAIR && !GPTQ && !DEP && !XMEM), "LOADS: the two-tile ROWS geometry only");
    // the pair = tile u of two experts' down projections: slot s reads ITS expert's activations (pointer in its wave records' spare
    // words, next to the pointer to its routing weight), the epilogue is x += fp16(w0 sum0) + fp16(w1 sum1) -- what the separate
    // launches + moe_combine_kernel (modules.hip) compute, in their order
    constexpr bool MOE_SUM = MOE && PAIR && WALK && S == 8;
    DYN_SMEM(smem);
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    int u = bid_x();                                                      // the workgroup's unit (tile / tile pair); ROWS: its first one
    const int slot = wv / S, r = wv % S;
    const int mj = PAIR ? slot : by;
    // ---- arguments: header, matrix block, wave record -- addresses from built-in ids only: one batch of scalar loads.
    // What the weight requests need (matrix pointers, the record's run) is unpacked first; everything else is unpacked AFTER
    // the requests have been issued: the waves of a CU share one scalar unit, so every scalar instruction ahead of the
    // requests delays the requests of all of them (profiles/history/r03_trace_lean_v4.txt: 1-3 us before this ordering).
    const u32x4* hb = (const u32x4*)&args.hdr;
    const u32x4* mb = (const u32x4*)&args.mat[mj];
    const u32x4* wb = (const u32x4*)&args.wave[mj * S + r];
    const u32x4 m0v = mb[0];
    const u32x2 m2v = *(const u32x2*)(mb + 2);
    const u32x4 w0v = wb[0];
    u32 meta_ = ((const u32*)wb)[4];
    // (pinned: the eleven words arrive as ONE batch of scalar loads here, not one dependent load per early-exit test)
    u32 m0x = m0v.x, m0y = m0v.y, m0z = m0v.z, m0w = m0v.w, m2x = m2v.x, m2y = m2v.y, w0x = w0v.x, w0y = w0v.y, w0z = w0v.z, w0w = w0v.w;
    pin_scalar(m0x); pin_scalar(m0y); pin_scalar(m0z); pin_scalar(m0w); pin_scalar(m2x); pin_scalar(m2y);
    pin_scalar(w0x); pin_scalar(w0y); pin_scalar(w0z); pin_scalar(w0w); pin_scalar(meta_);
    const u32x4 m0 = {m0x, m0y, m0z, m0w}; const u32x2 m2 = {m2x, m2y}; const u32x4 w0 = {w0x, w0y, w0z, w0w};
    if constexpr (MOE && !(PAIR && WALK && S == 8))
    {
        // a launch over all experts (2-4 rows): nothing to do for an expert no row is routed to
        const int rst = args.hdr.r_stride;
        if (rst)
        {
            const f16* const rw = args.hdr.out_scale;
            const int rows_ = args.hdr.M;
            u32 any = 0;
            for (int rr = 0; rr < rows_; rr++) any |= (u32)as_u16(rw[(size_t)rr * rst]);
            if (!any) return;
        }
    }
#ifdef EXL2_TRACE
    u64* const trace = args.hdr.trace;
#define LTRACE(i) do { if (trace && lane_id() == 0 && bid_x() < 2048) trace[(((size_t)bid_y() * 2048 + bid_x()) * LEAN_MAX_WAVES + wave_id()) * 8 + (i)] = realtime_stamp(); } while (0)
#else
#define LTRACE(i) do { } while (0)
#endif
    LTRACE(0);
    // overlapped chain: "this workgroup holds its slot" (before anything may leave: the next launch's gate counts the whole grid)
    // (bit 30 of every wave record says so: the header's sync words sit in a cache line of their own, and a scalar load + wait
    // in front of the weight requests costs every launch ~0.3 us -- measured: profiles/history/r04_states_ab.txt)
    // The workgroup's linear index is computed only THERE too: gridDim comes from the dispatch packet -- another scalar load the
    // requests would wait for (-4.5 % on the whole decode step when it sat here unconditionally: profiles/history/r04_bisect.txt).
    // (kernel-argument loads are speculatable: without the laundered pointer the compiler hoists these two into the first batch.)
    u32* sync_signal_ = nullptr;
    auto lin_wg_of = [&]() -> u32 { return (u32)by * (u32)gdim_x() + (u32)bid_x(); };
    if constexpr (DEP)
    {
        u32 opaque0 = 0;
        pin_scalar(opaque0);
        const LeanHdr* hp = (const LeanHdr*)((const u8*)&args.hdr + opaque0);
        sync_signal_ = hp->sync_signal;
        if (hp->sync_pad && wv == 0 && lane == 0) sync_report_entry(sync_signal_, lin_wg_of());       // (sync_pad: a gate counts the entries)
    }
    u32* const sync_signal = sync_signal_;
    const int n_tiles = (int)m2.y;
    // ROWS: a workgroup walks units u, u + grid, ... with ONE staged copy of the rows (the host sizes the grid to the CUs: one
    // workgroup per CU fits anyway); every other form: one unit, one pass
    for (bool first_unit = true;; first_unit = false)
    {
    const int tile = PAIR ? u : u * NSLOTS + slot;
    if (!PAIR && u * NSLOTS >= n_tiles)                                  // (matrices of one launch may have different widths)
    {
        // overlapped chain: the launch's arrival count covers the whole grid -- a workgroup without a tile arrives empty-handed
        if (sync_signal)
        {
            const u32 m_rows = (u32)args.hdr.M, fin = (u32)(S * NSLOTS);
            const u32 per_wg = ROWS ? (m_rows < fin ? m_rows : fin) : m_rows;
            if ((u32)wv < per_wg) sync_arrive_publish_sharded(sync_signal, lin_wg_of(), per_wg, args.hdr.sync_wgs, args.hdr.sync_wait);
        }
        return;
    }
    const bool active = tile < n_tiles;
    const u32 meta = active ? meta_ : 0u;
    const int n = (int)(meta & 0xFFu), bits = (int)((meta >> 8) & 0xFu), tail_nv = (int)((meta >> 12) & 0x7u);
    const bool pipe_on = ((meta_ >> 29) & 1u) == 0;                       // (bit 29 of every record: the host's EXL2_LEAN_PIPE=0 switch;
                                                                          //  the host also sets it for launches of an overlapped chain)
    // overlapped chain: the producer's "go" -- one wave polls, the workgroup meets behind it; the weight requests are out by then
    auto await_producer = [&]() {
        const u32* const w = args.hdr.sync_wait;
        if (sync_signal)                                                  // (uniform over the launch)
        {
            if (w && wv == 0) sync_wait_go(w, (int)(lin_wg_of() & 7u));
            block_sync_lds();
        }
    };
    const int t_ = active ? tile : 0;
    const u32* const wptr = (const u32*)ptr_of(m0.x, m0.y) + w0.x + (size_t)t_ * w0.y;
    const u32* const tptr = (const u32*)ptr_of(m0.z, m0.w) + w0.z + (size_t)t_ * w0.w;
    LTRACE(1);

    // ---- everything else a wave needs before it can decode -------------------------------------------------------------------
    // stage_copies: the LDS-DMA copies of the wave's activation slice and scale rows, with as few instructions in front of them as
    // possible (offsets and strides were made by the host) -- the requests of the wave's LAST items are issued behind them;
    // rest_ctx: everything else the decode needs, unpacked behind ALL requests.
    struct Rest { LeanCtx cx; float* red; int M; u32 flags; int chunk0, g0, gshift, gphase; bool uni; };
    struct Staged { u8* wbase; u32 off_sc, off_zp; int x_stride, xc0, M; };
    // (six pieces in every instantiation: the 7B step 0.6 % slower, profiles/r07d_ab_7b_ring_in_every_instantiation.txt -- code in front of the last requests)
    constexpr int LEAN_X_PIECES = (S == 4 && PAIR && !ROWS && WALK) ? LEAN_X_PIECES_PAIR_LOADS : LEAN_X_PIECES_BASE;   // (K = 28672 over 16 waves: 1792 rows = 224 16-byte units per slice; K = 8192 over the 4 waves of a pair's tile with a 10 % / 90 % bit mix: 77 chunks = 308)
    auto stage_copies = [&](Staged& P, auto tag) {
        // (a distinct marker per instantiation: identical copies of this code in two instantiations of `head` get merged by the
        // compiler otherwise, and then the registers of EITHER instantiation's pending requests count as pending here -- the
        // compiler then waits for them (a wait that also drains the copies it does not see) before it reuses one)
        LEAN_MARK(decltype(tag)::value);
        constexpr bool CAN_DEP = (decltype(tag)::value % 1000) / 10 == 0;     // (tag = 1000 bits + 10 NB [+ k]: the non-pipelined forms)
        const u32x4 h0 = hb[0], h1 = hb[1], h2 = hb[2];
        const u32x2 h3 = *(const u32x2*)(hb + 3);
        const u32x4 m1 = mb[1];
        const u32x4 w1 = wb[1];
        const u32x4 w2 = wb[2];
        const f16* in_a_ = (const f16*)ptr_of(h0.x, h0.y);
        if constexpr (MOE_SUM) { const u32x4 w3 = wb[3]; in_a_ = (const f16*)ptr_of(w3.z, w3.w); }
        const f16* const in_a = in_a_;
        const int M = (int)h1.w, K = (int)h2.x, lda = (int)h2.y;
        const int oct = K >> 3;
        const f16* const sc_tab = (const f16*)ptr_of(m1.x, m1.y); const f16* const zp_tab = (const f16*)ptr_of(m1.z, m1.w);
        const int G = (int)m2.x;
        const u32 xr = active ? w1.y : 0u, gr = active ? w1.z : 0u;
        const int xc0 = (int)(xr & 0xFFFFu), xchunks = (int)(xr >> 16);
        const int gw0 = (int)(gr & 0xFFFFu), ng = (int)(gr >> 16);
        // the wave's LDS area: [M rows of the activation slice][scale rows][zero-point rows]
        // (ROWS: the workgroup's shared rows come first, x_stride = K + 8; a wave's own area holds its scale rows only)
        const int x_stride = ROWS ? K + 8 : XMEM ? lda : (int)w2.w;
        const u32 rows_bytes = ROWS ? (((u32)M * (u32)x_stride * 2u + 15u) & ~15u) : 0u;
        u8* const wbase = smem + rows_bytes + (size_t)slot * h3.x + w1.w;
        f16* const x_lds = (f16*)wbase;
        u8* const sc_lds = wbase + w2.y;
        u8* const zp_lds = wbase + w2.z;
        P.wbase = wbase; P.off_sc = w2.y; P.off_zp = w2.z; P.x_stride = x_stride; P.xc0 = (ROWS || XMEM) ? 0 : xc0; P.M = M;
        // requests: scale rows, activation slice.  The common case -- one row, <= 64 units (16 bytes) of each -- is straight-line
        // code, one LDS-DMA instruction per table (the compiler's loop skeletons around run-time trip counts cost more
        // instructions per wave than the decode of an item)
        const int xunits = xchunks * 4;                                   // 16-byte units of the slice (a chunk = 32 halfs)
        const int xu0 = xc0 * 4;
        const int sc_units = 2 * ng;                                      // 16-byte units of the scale rows (a row = 16 halfs)
        const f16* const st = sc_tab + ((size_t)t_ * G + gw0) * 16;
        LEAN_MARK(decltype(tag)::value + 1);
        if (!ROWS && !XMEM && M == 1 && xunits <= (S == 4 ? 128 : 64) && sc_units <= 64)
        {
            // the headline case -- one row, one copy instruction per table (two for the 4-wave geometry's 128-unit slices) -- with
            // nothing but those in front of the requests of the wave's last items (1.5 % of the whole decode step against the
            // general form below: profiles/history/r04_states_ab_final.txt)
            if (lane < sc_units) LEAN_DMA(st, lane * 16, sc_lds);
            if constexpr (GPTQ) { if (lane < sc_units) LEAN_DMA(zp_tab + ((size_t)t_ * G + gw0) * 16, lane * 16, zp_lds); }
            if constexpr (CAN_DEP && DEP)
            {
                if (lane < xunits && xu0 + lane < oct) LEAN_DMA_X(in_a, (xu0 + lane) * 16, (u8*)x_lds);
                if constexpr (S == 4) { if (64 + lane < xunits && xu0 + 64 + lane < oct) LEAN_DMA_X(in_a, (xu0 + 64 + lane) * 16, (u8*)x_lds + 1024); }
            }
            else
            {
                if (lane < xunits && xu0 + lane < oct) LEAN_DMA(in_a, (xu0 + lane) * 16, (u8*)x_lds);
                if constexpr (S == 4) { if (64 + lane < xunits && xu0 + 64 + lane < oct) LEAN_DMA(in_a, (xu0 + 64 + lane) * 16, (u8*)x_lds + 1024); }
            }
        }
        else
        {
            // straight-line code: <= 2 copy instructions per scale table, <= LEAN_X_PIECES per activation row (M <= 4 rows,
            // slices of <= 64 * LEAN_X_PIECES 16-byte units, <= 64 scale rows: what the host plans)
            // (the second and later pieces sit behind a SCALAR test: a lane-wise test alone costs ~8 instructions per skipped copy, in
            // front of the requests of the wave's last items)
            if (lane < sc_units) LEAN_DMA(st, lane * 16, sc_lds);
            if (sc_units > 64) { if (64 + lane < sc_units) LEAN_DMA(st, (64 + lane) * 16, sc_lds + 1024); }
            if constexpr (LOADS)
            {
                if (sc_units > 128) { if (128 + lane < sc_units) LEAN_DMA(st, (128 + lane) * 16, sc_lds + 2048); }
                if (sc_units > 192) { if (192 + lane < sc_units) LEAN_DMA(st, (192 + lane) * 16, sc_lds + 3072); }
            }
            if constexpr (GPTQ)
            {
                const f16* zt = zp_tab + ((size_t)t_ * G + gw0) * 16;
                if (lane < sc_units) LEAN_DMA(zt, lane * 16, zp_lds);
                if (sc_units > 64) { if (64 + lane < sc_units) LEAN_DMA(zt, (64 + lane) * 16, zp_lds + 1024); }
            }
            if constexpr (XMEM) { }                                     // (the activations are read from memory, item by item)
            else if constexpr (!ROWS)
            {
                #pragma unroll
                for (int rr = 0; rr < 4; rr++)
                {
                    if (rr < M)
                    {
                        const f16* const row = in_a + (size_t)rr * lda;
                        u8* const dst = (u8*)(x_lds + (size_t)rr * x_stride);
                        #pragma unroll
                        for (int u = 0; u < LEAN_X_PIECES; u++)
                        if (u == 0 || u * 64 < xunits)
                        {
                            // (a launch of an overlapped chain reads its producer's rows at agent scope; it never takes the pipelined form)
                            const bool on = u * 64 + lane < xunits && xu0 + u * 64 + lane < oct;
                            if constexpr (CAN_DEP && DEP) { if (on) LEAN_DMA_X(row, (xu0 + u * 64 + lane) * 16, dst + u * 1024); }
                            else if (on) LEAN_DMA(row, (xu0 + u * 64 + lane) * 16, dst + u * 1024);
                        }
                    }
                }
            }
            else
            {
                // the whole rows, dealt out over the workgroup's waves in 1 KB pieces (piece p of row rr -> wave (p + rr) mod waves);
                // once per workgroup: its later units read the same copy
                constexpr int WAVES = S * NSLOTS;
                const int pieces = (oct + 63) >> 6;
                #pragma nounroll
                for (int rr = 0; rr < (first_unit ? M : 0); rr++)
                {
                    const f16* const row = in_a + (size_t)rr * lda;
                    u8* const dst = smem + (size_t)rr * x_stride * 2;
                    #pragma nounroll
                    for (int pc = (wv + WAVES - (rr % WAVES)) % WAVES; pc < pieces; pc += WAVES)
                    {
                        const bool on = pc * 64 + lane < oct;
                        if constexpr (CAN_DEP && DEP) { if (on) LEAN_DMA_X(row, (pc * 64 + lane) * 16, dst + (size_t)pc * 1024); }
                        else if (on) LEAN_DMA(row, (pc * 64 + lane) * 16, dst + (size_t)pc * 1024);
                    }
                }
            }
        }
        // (larger slices are declined by the host, lean_plan_matrix: loops around copy instructions are not an option here -- behind
        // a loop that issues vector-memory operations the compiler's count of the requests in flight is a guess, and it drains
        // them in front of the next register it reuses: i.e. in front of the requests that are meant to stay in flight)
    };
    auto rest_ctx = [&](Rest& R, const Staged& P) {
        const u32x4 h2 = hb[2];
        const u32x2 h3 = *(const u32x2*)(hb + 3);
        const u32 w2x = ((const u32*)wb)[8];
        R.M = P.M; R.flags = h2.w;
        R.uni = ((meta >> 15) & 1u) != 0;
        R.gshift = (int)((meta >> 16) & 0x7u); R.gphase = (int)((meta >> 19) & 0x3FFu);
        R.chunk0 = (int)(w2x & 0xFFFFu); R.g0 = (int)(w2x >> 16);
        R.red = (float*)(smem + h3.y);
        if constexpr (XMEM) { const u32x4 h0 = hb[0]; R.cx.x_lds = (const f16*)ptr_of(h0.x, h0.y); }
        else R.cx.x_lds = ROWS ? (const f16*)smem : (const f16*)P.wbase;
        R.cx.sc_lds = (const f16*)(P.wbase + P.off_sc); R.cx.zp_lds = (const f16*)(P.wbase + P.off_zp);
        R.cx.x_stride = P.x_stride; R.cx.xc0 = P.xc0; R.cx.M = P.M;
    };

    // ---- the wave's items: everything requested at once into registers, decoded item by item ------------------------------------
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    Rest R;
    // NB = the LAST NB full items of the wave's share are requested unconditionally, behind the LDS-DMA copies and the fence load
    // (hw.h): the wave waits for the fence load only -- i.e. for its first n - NB items, its activation slice and its scale rows
    // -- and decodes those while the last NB items are still in flight, then each of them when IT has landed (the compiler's own
    // counted waits: unconditional loads in straight-line code).  NB = 0 is the round-3 form: one wait for everything.  The
    // conditional requests (q < nA) must come first: the compiler has to assume the smallest count behind them.
    auto head = [&](auto bits_tag, auto nb_tag) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int NB = decltype(nb_tag)::value;
        constexpr int D = LeanDepth<BITS, S>::v;
        constexpr int DA = D - NB;
        constexpr size_t STEP = 64 * BITS;
        // (WALK without ROWS = the pair geometry whose 4 waves per tile take their share in up to three register loads: K = 8192 at 2-3 bits,
        // the 70B gate|up.  Its own instantiation: with the ring code inside, the 7B gate|up launch measured 0.6 % slower)
        constexpr bool PAIR_LOADS = (S == 4 || S == 8) && PAIR && !ROWS && WALK;
        constexpr bool PASSES = !XMEM && (S == 16 || (S == 8 && NSLOTS == 1 && !PAIR && !ROWS && LEAN_S8_PASSES > 1 && BITS <= LEAN_S8_PASS_BITS) ||
                                          (PAIR_LOADS && BITS <= LEAN_S8_PASS_BITS) || LOADS);   // (what lean_plan_matrix plans)
        constexpr bool RING = PASSES && LEAN_PASS_RING && BITS <= LEAN_S8_PASS_BITS;      // (wider items: a whole load at a time, the round-4 loop)
        constexpr bool MANY_LOADS = S == 16 || PAIR_LOADS || LOADS;                          // shares of more than two register loads (the 70B pair: up to three)
        LaneWords<BITS> a[DA > 0 ? DA : 1], b[NB > 0 ? NB : 1], bt;
        const int nA = n - NB;
        u32 xvoff = 0;                                                 // XMEM: the lane's byte offset into the activations (row, 8 j)
        if constexpr (XMEM)
        {
            const u32x4 h1 = hb[1], h2 = hb[2];
            const int M_ = (int)h1.w, lda_ = (int)h2.y, c = lane & 15, j = lane >> 4;
            // (tiled: lane c + 16 j reads row c of 8-column block j of a chunk = bytes [16 lane, 16 lane + 16) of the chunk's 1 KB)
            xvoff = (h2.w & LF_ATILED) ? (u32)lane * 16u : ((u32)(c < M_ ? c : M_ - 1) * (u32)lda_ + 8u * (u32)j) * 2u;
            pin_vector(xvoff);
        }
        #pragma unroll
        for (int q = 0; q < DA; q++) if (q < nA) load_lane_words<BITS>(wptr + (size_t)q * STEP, lane, a[q]);
        if (tail_nv) load_lane_words<BITS>(tptr, lane, bt);
        LTRACE(2);
        if constexpr (NB == 0) await_producer();        // (the pipelined form is never taken inside an overlapped chain)
        Staged P;
        stage_copies(P, std::integral_constant<int, 1000 * BITS + 10 * NB>());
        LEAN_MARK(1000 * BITS + 10 * NB + 2);
        const u32 fence = fence_load(wptr);
        sched_fence();
        if constexpr (NB > 0)
        {
            const u32* const bptr = wptr + (size_t)nA * STEP;
            #pragma unroll
            for (int q = 0; q < NB; q++) load_lane_words<BITS>(bptr + (size_t)q * STEP, lane, b[q]);
            sched_fence();
        }
        LTRACE(3);
        rest_ctx(R, P);
        // XMEM: a ring of XL items' A operands, the first XL requested here -- behind the weights, in front of the wait for them.
        // EVERY request is unconditional (an item the wave does not have requests one 16-byte line, the same for all lanes):
        // the compiler's count of the requests in flight stays exact, so item q waits for ITS operands only while those of
        // items q + 1 .. q + XL - 1 are on their way
        constexpr int XL = !XMEM ? 1 : (LEAN_XMEM_AHEAD < D ? LEAN_XMEM_AHEAD : D);
        f16x8 xa[XL][4];
        // (address = a UNIFORM base -- the activations + the item's columns, scalar arithmetic -- + the lane's byte offset `xvoff`,
        // made in front of the weight requests: no vector instruction per request, nothing that could wait for a register)
        const u32 xcs = (R.flags & LF_ATILED) ? 1024u : 64u;          // bytes between the 32-column chunks of the activations
        auto xrequest = [&](f16x8 (&v)[4], int q) {
            const u8* const xb = (const u8*)R.cx.x_lds + (q < n ? (size_t)(R.chunk0 + 4 * q) * xcs : 0);
            #pragma unroll
            for (int e = 0; e < 4; e++) v[e] = *(const f16x8*)(xb + xvoff + (q < n ? e * xcs : 0u));
        };
        if constexpr (XMEM)
        {
            LEAN_MARK(1000 * BITS + 10 * NB + 3);
            #pragma unroll
            for (int q = 0; q < XL; q++) xrequest(xa[q], q);
            sched_fence();
            LEAN_MARK(1000 * BITS + 10 * NB + 4);
        }
        fence_load_use(fence);               // the first part, the activation slice and the scale rows have landed
        if constexpr (ROWS) { wait_vmcnt_le<0>(); block_sync_lds(); }      // every wave's share of the rows has landed
        wave_converge();
        LTRACE(4);
        auto item = [&](const LaneWords<BITS>& w, int q) {
            if (R.uni) lean_item_uniform<BITS, GPTQ>(w, R.cx, R.chunk0 + 4 * q, R.g0 + ((4 * q + R.gphase) >> R.gshift), lane, acc);
            else lean_item_general<BITS, GPTQ>(w, R.cx, R.chunk0 + 4 * q, 4 * q, R.g0, R.gshift, R.gphase, 4, lane, acc);
            if (LEAN_ITEM_FENCE) sched_fence();
        };
        // (shares of several register loads; returns true when it took the share)
        auto ring_passes = [&](auto on_tag) -> bool {
            if constexpr (decltype(on_tag)::value)
            {
                if (n <= D) return false;

                // more items than the wave's registers hold (K = 28672 split over 16 waves: 14-15 items of 3 bits; the v part of a
                // 70B q|k|v launch on 8 waves: 9 of 3 bits): the registers are a ring -- as soon as item q of one register load is
                // decoded, item q of the NEXT one is requested into the same registers, so a later load's round trip runs under the
                // decode of the current one (round 4 requested a whole load, waited for all of it, decoded it: one exposed round trip
                // per pass).  EVERY request is unconditional -- indices past the end are clamped to the last item, a line that is in
                // flight anyway -- in the first load's straight-line code and in the loop body alike, same order: the compiler's
                // count of the requests in flight is exact on both edges into the loop (tests/test_lean_isa.py).
                const u32* const lastp = wptr + (size_t)(n - 1) * STEP;
                // (a request past the end is lane 0's words of the last item for every lane: one 16-byte line, not another 64 x 4 BITS bytes)
                auto request = [&](LaneWords<BITS>& w, int q) { load_lane_words<BITS>(q < n ? wptr + (size_t)q * STEP : lastp, q < n ? lane : 0, w); };
                #pragma unroll
                for (int q = 0; q < D; q++) { item(a[q], q); request(a[q], D + q); }
                // (the second load in straight-line code as well: around a loop the register allocator leaves copies of the
                // 3-bit items' register triples on the back edge, and a copy waits for ITS load -- vmcnt(0) once per trip, which
                // is the exposed round trip again; 15 three-bit items on 16 waves, 9 on 8 waves end here)
                #pragma unroll
                for (int q = 0; q < D; q++)
                {
                    if (D + q < n) item(a[q], D + q);
                    if constexpr (MANY_LOADS) request(a[q], 2 * D + q);
                }
                // (8 waves: two loads at most, lean_plan_matrix)
                #pragma nounroll
                for (int q0 = 2 * D; MANY_LOADS && q0 < n; q0 += D)
                {
                    #pragma unroll
                    for (int q = 0; q < D; q++)
                    {
                        if (q0 + q < n) item(a[q], q0 + q);
                        request(a[q], q0 + D + q);
                    }
                }
                return true;
            }
            else return false;
        };
        if (LEAN_KILL & 8)
        {
            // timing experiment: the loads stay, the decode is an xor (what does the launch cost without the decode's VALU work?)
            u32 xr = 0;
            #pragma unroll
            for (int q = 0; q < DA; q++) if (q < nA) { for (int e = 0; e < BITS; e++) xr ^= a[q].w[e]; }
            #pragma unroll
            for (int q = 0; q < NB; q++) { for (int e = 0; e < BITS; e++) xr ^= b[q].w[e]; }
            if (tail_nv) for (int e = 0; e < BITS; e++) xr ^= bt.w[e];
            acc[0] = __builtin_bit_cast(float, xr & 0x3fffffffu) + (float)R.cx.x_lds[lane] + (float)R.cx.sc_lds[lane & 15];
        }
        else if (LEAN_KILL & 1) { }
        else if constexpr (XMEM)
        {
            // a ring of LEAN_XMEM_AHEAD items' A operands.  EVERY request is unconditional (an item the wave does not have
            // requests one 16-byte line, the same for all lanes): the compiler's count of the requests in flight stays exact, so
            // item q waits for ITS operands only while those of items q + 1 .. q + AHEAD - 1 are on their way
            #pragma unroll
            for (int q = 0; q < D; q++)
            {
                if (q < n)
                {
                    if (R.uni) lean_item_uniform<BITS, GPTQ, true>(a[q], R.cx, R.chunk0 + 4 * q, R.g0 + ((4 * q + R.gphase) >> R.gshift), lane, acc, xa[q % XL]);
                    else lean_item_general<BITS, GPTQ, true>(a[q], R.cx, R.chunk0 + 4 * q, 4 * q, R.g0, R.gshift, R.gphase, 4, lane, acc, xa[q % XL]);
                }
                sched_fence();
                if (q + XL < D) { xrequest(xa[q % XL], q + XL); sched_fence(); }
            }
        }
        else if (ring_passes(std::integral_constant<bool, RING && NB == 0>())) { }
        else
        {
#if LEAN_REPEAT
            // timing experiment (results are WRONG: every sum is taken LEAN_REPEAT times): the decode of the wave's share runs again
            // on the same registers -- the further passes find their code in the instruction cache and their data landed, so what they
            // add is the decode's pure execution time (profiles/history/r05_repeat_experiment.txt)
            u32 reps_ = LEAN_REPEAT; pin_scalar(reps_);
            #pragma nounroll
            for (u32 rep_ = 0; rep_ < reps_; rep_++)
            {
#endif
            #pragma unroll
            for (int q = 0; q < DA; q++) if (q < nA) item(a[q], q);
            if constexpr (NB > 0)
            {
                #pragma unroll
                for (int q = 0; q < NB; q++) item(b[q], nA + q);
            }
#if LEAN_REPEAT
            }
#endif
        }
        // more items than the wave's registers hold (K = 28672 split over 16 waves: 14-15 items of 3 bits): further passes of D
        // items, each its own round trip -- only the 16-wave geometry, the last one the host tries, plans such shares
        if constexpr (PASSES && NB == 0 && !RING)
        {
            if (!(LEAN_KILL & 9))
            {
                #pragma nounroll
                for (int q0 = D; q0 < n; q0 += D)
                {
                    #pragma unroll
                    for (int q = 0; q < D; q++) if (q0 + q < n) load_lane_words<BITS>(wptr + (size_t)(q0 + q) * STEP, lane, a[q]);
                    #pragma unroll
                    for (int q = 0; q < D; q++) if (q0 + q < n) item(a[q], q0 + q);
                }
            }
        }
        if constexpr (XMEM)
        {
            if (tail_nv)
            {
                const f16x8 z8 = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
                f16x8 xt[4] = {z8, z8, z8, z8};
                const u8* const xb = (const u8*)R.cx.x_lds + (size_t)(R.chunk0 + 4 * n) * xcs;
                #pragma unroll
                for (int e = 0; e < 4; e++) if (e < tail_nv) xt[e] = *(const f16x8*)(xb + xvoff + e * xcs);
                lean_item_general<BITS, GPTQ, true>(bt, R.cx, R.chunk0 + 4 * n, 4 * n, R.g0, R.gshift, R.gphase, tail_nv, lane, acc, xt);
            }
        }
        else if (tail_nv) lean_item_general<BITS, GPTQ>(bt, R.cx, R.chunk0 + 4 * n, 4 * n, R.g0, R.gshift, R.gphase, tail_nv, lane, acc);
    };
    // a share of at least NB and at most D items takes the pipelined form
    auto head_bits = [&](auto bits_tag) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int NB = LeanTail<BITS, S>::v;
        if constexpr (NB > 0 && !ROWS && !XMEM)
        {
            if (n >= NB && n <= LeanDepth<BITS, S>::v && pipe_on) { head(bits_tag, std::integral_constant<int, NB>()); return; }
        }
        head(bits_tag, std::integral_constant<int, 0>());
    };
    if (n == 0 && tail_nv == 0)
    {
        LTRACE(2);
        await_producer();
        Staged P;
        stage_copies(P, std::integral_constant<int, 99000>());
        LTRACE(3);
        rest_ctx(R, P);
        wait_vmcnt_le<0>();
        if constexpr (ROWS) block_sync_lds();
        wave_converge();
        LTRACE(4);
    }
    else if constexpr (GPTQ) head_bits(std::integral_constant<int, 4>());
    else switch (bits)
    {
        case 4: head_bits(std::integral_constant<int, 4>()); break;
        case 8: head_bits(std::integral_constant<int, 8>()); break;
        case 6: head_bits(std::integral_constant<int, 6>()); break;
        case 5: head_bits(std::integral_constant<int, 5>()); break;
#if LEAN_LOWBITS
        case 3: head_bits(std::integral_constant<int, 3>()); break;
        default: head_bits(std::integral_constant<int, 2>()); break;
#else
        default: break;
#endif
    }
    const int M = R.M;
    const u32 flags = R.flags;
    float* const red = R.red;
    const int n_tiles_ep = n_tiles;
    (void)n_tiles_ep;

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
    // the finalising waves (wave `row` finalises row `row` -- ROWS: rows row, row + waves, ...; they sit in slot 0, so the block of
    // THEIR matrix is the output's): lane -> output slot lane >> 4 (pair: the one act(gate) * up output), column lane & 15.
    constexpr int N_OUT = PAIR ? 1 : NSLOTS;
    const int ep_slot = lane >> 4, ep_c = lane & 15;
    const int ep_tile = (PAIR ? u : u * NSLOTS) + (PAIR ? 0 : ep_slot);  // (ROWS: a finalising wave may sit in any slot -- the unit's first tile, not its own)
    const int ep_mj = PAIR ? 0 : mj;                                     // (pair: the one output goes through matrix 0's c / c_invperm)
    const bool ep_on = ep_slot < N_OUT && ep_tile < n_tiles;
    const int ep_n = ep_tile * 16 + ep_c;
    f16* xp_out_ = nullptr;
    if constexpr (MOE_SUM) xp_out_ = dyn->xp_out; else xp_out_ = args.hdr.xp_out;
    f16* const xp_out = xp_out_;
    struct EpIn { f16* cp; f16 c_old; int xp_idx; f16 xw_next; float ssq; };
    // what a row's epilogue needs from memory
    // (two round trips to memory, not one per dependent load: first everything whose address is known -- the two permutation
    // entries and the row's partial sums of squares, <= 256 of them in four independent loads per lane -- then what those address:
    // the next consumer's norm weight and the residual.  The finalising wave runs this between its last item and the workgroup's
    // barrier: one load + wait per loop trip was 4-7 round trips of ~0.5 us at the end of every launch)
    auto ep_inputs = [&](int row, EpIn& e) {
        e.cp = nullptr; e.c_old = (f16)0.0f; e.xp_idx = ep_n; e.xw_next = (f16)1.0f; e.ssq = 0.0f;
        const u16* const c_invperm = ep_on ? args.mat[ep_mj].c_invperm : nullptr;
        const u16* xp_invperm_ = nullptr;
        if constexpr (MOE_SUM) xp_invperm_ = (ep_on && xp_out) ? dyn->xp_invperm : nullptr; else xp_invperm_ = (ep_on && xp_out) ? args.hdr.xp_invperm : nullptr;
        const u16* const xp_invperm = xp_invperm_;
        int c_idx = ep_n;
        if (c_invperm) c_idx = (int)c_invperm[ep_n];
        if (xp_invperm) e.xp_idx = (int)xp_invperm[ep_n];
        // A_NORM_PRE: the activations were x * w (qgemv_flat.h); 1 / rms(x) multiplies the finished sum.  The partial sums of
        // squares of this row (fixed order: a lane adds its entries lane, lane + 64, ... in that order)
        float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* const sp = args.hdr.ss + (size_t)row * args.hdr.npart;
        const int npart = args.hdr.npart;
        if (flags & LF_NORM)
        {
            #pragma unroll
            for (int k = 0; k < 4; k++) if (lane + 64 * k < npart) sv[k] = DEP ? load_agent_f32(sp + lane + 64 * k) : sp[lane + 64 * k];
        }
        if (ep_on)
        {
            if constexpr (MOE_SUM)
            {
                if (xp_out && dyn->xp_w) e.xw_next = dyn->xp_w[e.xp_idx];
                e.cp = dyn->c + (size_t)row * dyn->ldc + c_idx;
            }
            else
            {
            if (xp_out && args.hdr.xp_w) e.xw_next = args.hdr.xp_w[e.xp_idx];
            e.cp = args.mat[ep_mj].c + ((flags & LF_CTILED) ? ((size_t)(c_idx >> 3) * 16 + row) * 8 + (c_idx & 7)
                                                           : (size_t)row * args.hdr.ldc[ep_mj] + c_idx);
            }
            if (flags & LF_ACCUM) e.c_old = DEP ? load_agent_f16(e.cp) : *e.cp;
        }
        if (flags & LF_NORM)
        {
            #pragma unroll
            for (int k = 0; k < 4; k++) e.ssq += sv[k];
            #pragma nounroll
            for (int i = lane + 256; i < npart; i += 64) e.ssq += DEP ? load_agent_f32(sp + i) : sp[i];
        }
    };
    // combine (fixed order) + epilogue of one row
    auto ep_finish = [&](int row, const EpIn& e) {
        float rms = 1.0f;
        if (flags & LF_NORM)
        {
            const float ssq = wave_allreduce_add(e.ssq);
            rms = fast_rsqrt(ssq * (1.0f / (float)args.hdr.K) + args.hdr.eps);
        }
        auto slot_sum = [&](int sl) -> float {
            float v = 0.0f;
            for (int w = sl * S; w < sl * S + S; w++) v += red[(w * M + row) * 16 + ep_c];
            return v;
        };
        float sq = 0.0f;
        if (ep_on && !(LEAN_KILL & 4))
        {
            f16 y;
            if constexpr (MOE_SUM)
            {
                const u32x4 p0 = ((const u32x4*)&args.wave[0])[3], p1 = ((const u32x4*)&args.wave[S])[3];
                const f16 w0 = *(const f16*)ptr_of(p0.x, p0.y), w1 = *(const f16*)ptr_of(p1.x, p1.y);
                float v = (float)e.c_old;                             // (LF_ACCUM: the plan sets it)
                if (as_u16(w0) != 0) v += (float)(f16)(slot_sum(0) * (float)w0);
                if (as_u16(w1) != 0) v += (float)(f16)(slot_sum(1) * (float)w1);
                y = (f16)v;
            }
            else if constexpr (PAIR)
            {
                float gv = slot_sum(0) * rms, uv = slot_sum(1) * rms;
                if (flags & LF_BIAS)
                {
                    if (args.mat[0].bias) gv += (float)args.mat[0].bias[ep_n];
                    if (args.mat[1].bias) uv += (float)args.mat[1].bias[ep_n];
                }
                y = clamp_h(act_h((f16)gv, (flags & LF_GELU) != 0) * (f16)uv);
            }
            else
            {
                float v = slot_sum(ep_slot) * rms;
                if (flags & LF_BIAS) { const f16* bias = args.mat[ep_mj].bias; if (bias) v += (float)bias[ep_n]; }
                if constexpr (MOE) { const f16* const osc = args.hdr.out_scale; if (osc && args.hdr.moe_mul) v *= (float)osc[(size_t)row * args.hdr.r_stride]; }
                if (flags & LF_ACCUM) v += (float)e.c_old;
                y = (f16)v;
            }
            if constexpr (DEP) store_agent_f16(e.cp, y); else *e.cp = y;
            if (xp_out)
            {
                // chain-out: x for the next consumer = x * ITS norm weight (one rounding, saturated), in its packed order; the sum
                // of squares is x's own
                const float f = fmaxf(-65504.0f, fminf((float)y, 65504.0f));
                const f16 xw = (f16)fmaxf(-65504.0f, fminf(f * (float)e.xw_next, 65504.0f));
                size_t xo_;
                if constexpr (MOE_SUM) xo_ = (size_t)row * dyn->ldxp + e.xp_idx;
                else xo_ = (flags & LF_XPTILED) ? ((size_t)(e.xp_idx >> 3) * 16 + row) * 8 + (e.xp_idx & 7)
                                                : (size_t)row * args.hdr.ldxp + e.xp_idx;
                const size_t xo = xo_;
                if constexpr (DEP) store_agent_f16(xp_out + xo, xw);
                else xp_out[xo] = xw;
                sq = f * f;
            }
        }
        float* ss_out_ = nullptr;
        if constexpr (MOE_SUM) ss_out_ = dyn->ss_out; else ss_out_ = args.hdr.ss_out;
        float* const ss_out = ss_out_;
        if (ss_out)
        {
            sq = wave_allreduce_add(sq);
            if (lane == 0) { if constexpr (DEP) store_agent_f32(ss_out + (size_t)row * args.hdr.wgs + u, sq); else ss_out[(size_t)row * args.hdr.wgs + u] = sq; }
        }
    };
    // overlapped chain: this finalising wave's outputs have completed -> it arrives; the launch's last arrival publishes "go"
    constexpr u32 FIN_WAVES = (u32)(S * NSLOTS);
    auto signal_done = [&]() {
        if (sync_signal)
            sync_arrive_publish_sharded(sync_signal, lin_wg_of(), ROWS ? ((u32)M < FIN_WAVES ? (u32)M : FIN_WAVES) : (u32)M, args.hdr.sync_wgs, args.hdr.sync_wait);
    };
    if constexpr (!ROWS && !XMEM)
    {
        if (wv >= M) { LTRACE(5); block_sync_lds(); return; }
        // what the epilogue needs from memory is requested before the barrier
        EpIn e;
        ep_inputs(wv, e);
        LTRACE(5);
        block_sync_lds();
        LTRACE(6);
        ep_finish(wv, e);
        signal_done();
    }
    else
    {
        LTRACE(5);
        block_sync_lds();
        LTRACE(6);
        #pragma nounroll
        for (int row = wv; row < M; row += S * NSLOTS)
        {
            EpIn e;
            ep_inputs(row, e);
            ep_finish(row, e);
        }
        if (wv < M) signal_done();
    }
    LTRACE(7);
    if constexpr (!(ROWS && WALK)) break;
    else
    {
        u += gdim_x();
        if (u >= args.hdr.wgs) break;                                     // (hdr.wgs = units of the launch = partial sums per row it publishes)
    }
    }
}

template <bool GPTQ, int S, int NSLOTS, bool PAIR, int OCC, bool ROWS = false, bool WALK = false, bool DEP = false, bool XMEM = false, bool LOADS = false>
KERNEL void LEAN_BOUNDS(S * NSLOTS * 64, OCC) qgemv_lean_kernel(const LeanArgs args)
{
    lean_body<GPTQ, S, NSLOTS, PAIR, OCC, ROWS, WALK, DEP, XMEM, false, LOADS>(args, bid_y());
}

// grouped-expert launch (sparse MoE at one row): blockIdx.y = the y-th selected expert; its argument block was planned at load time
// (qgemv_lean_plan_export) and copied to table[y] by the MoE front kernel of this step
template <int S, int NSLOTS, bool PAIR, int OCC, bool WALK>
KERNEL void LEAN_BOUNDS(S * NSLOTS * 64, OCC) qgemv_lean_moe_kernel(const LeanArgs* __restrict__ table, const LeanDyn dyn)
{
    lean_body<false, S, NSLOTS, PAIR, OCC, false, WALK, false, false, true>(table[bid_y()], 0, &dyn);
}

#ifdef EXL2_TRACE
static u64* g_ltrace_buf = nullptr;
static int g_ltrace_which = 0, g_ltrace_count = 0;
extern "C" void exl2_debug_set_lean_trace(void* p, int which) { g_ltrace_buf = (u64*)p; g_ltrace_which = which; g_ltrace_count = 0; }
#endif

// ---- host: the split --------------------------------------------------------------------------------------------------------

static inline u32 al16(u32 x) { return (x + 15u) & ~15u; }

// one bit-width section of a tile's K range: F full items + (optionally) a partial last item in the side buffer
struct LeanRun { int F, bits, chunk0; u32 off, tstride; int tail_nv; u32 t_off, t_tstride; };

// Splits one tile's items over S waves: every wave gets a contiguous part of ONE run (a wave never mixes bit widths: one
// decoder per wave, everything in registers); the waves are dealt out to the runs in proportion to their bytes.  Fills
// wave[0 .. S) and returns the LDS bytes of the S waves together, 0 when the matrix is not covered with S waves (more runs
// than waves, more items than a wave's registers hold, a chunk -> group map that is not affine inside a part, ...).
static u32 lean_plan_matrix(const QMatrix* qm, int S, int M, bool norm, LeanWave* wave, bool rows_mode = false, bool xmem = false, int passes = 0, int sc_rows_max = 64)
{
    if (passes <= 0) passes = S == 16 ? LEAN_MAX_PASSES : 1;            // register loads a wave's share may take (S = 8: items of <= LEAN_S8_PASS_BITS bits only)
    const QMatDev& d = qm->dev;
    if (d.n_runs <= 0 || !qm->cg_host || !d.sc_tab || (qm->is_gptq && !d.zp_tab)) return 0;
    // QRun list (K order): a full run, optionally followed by its partial super-chunk -> logical runs
    std::vector<LeanRun> runs;
    for (int i = 0; i < d.n_runs; i++)
    {
        const QRun& r = d.runs[i];
        const int c0 = (int)r.k_base >> 5;
        if (!r.in_tail)
        {
            if (r.nvalid_last != 4) return 0;
            LeanRun t; memset(&t, 0, sizeof(t));
            t.F = r.n_super; t.bits = r.bits; t.chunk0 = c0; t.off = r.base_word; t.tstride = r.tile_stride;
            runs.push_back(t);
        }
        else
        {
            if (r.n_super != 1 || r.nvalid_last < 1 || r.nvalid_last > 3) return 0;
            if (!runs.empty() && runs.back().bits == r.bits && runs.back().tail_nv == 0 && runs.back().chunk0 + 4 * runs.back().F == c0)
            {
                runs.back().tail_nv = r.nvalid_last; runs.back().t_off = r.base_word; runs.back().t_tstride = r.tile_stride;
            }
            else
            {
                LeanRun t; memset(&t, 0, sizeof(t));
                t.F = 0; t.bits = r.bits; t.chunk0 = c0; t.tail_nv = r.nvalid_last; t.t_off = r.base_word; t.t_tstride = r.tile_stride;
                runs.push_back(t);
            }
        }
    }
    const int R = (int)runs.size();
    if (R < 1 || R > S) return 0;
    // waves per run: proportional to bytes, at least one, largest remainders first
    std::vector<int> nw(R, 1);
    long long total = 0;
    std::vector<long long> cost(R);
    for (int i = 0; i < R; i++) { cost[i] = (long long)runs[i].bits * (runs[i].F + (runs[i].tail_nv ? 1 : 0)); total += cost[i]; }
    for (int left = S - R; left > 0; left--)
    {
        int best = -1; double worst = 0.0;
        for (int i = 0; i < R; i++)
        {
            if (nw[i] >= runs[i].F) continue;                          // (a wave needs at least one full item to be worth adding)
            const double per = (double)cost[i] / nw[i];
            if (per > worst) { worst = per; best = i; }
        }
        if (best < 0) break;
        nw[best]++;
    }
    const int n_chunks = d.K / 32;
    u32 lds_total = 0;
    int w = 0;
    for (int s2 = 0; s2 < S; s2++) memset(&wave[s2], 0, sizeof(LeanWave));
    for (int i = 0; i < R; i++)
    {
        const LeanRun& r = runs[i];
        int i0 = 0;
        for (int k = 0; k < nw[i]; k++, w++)
        {
            const int n = (r.F - i0 + (nw[i] - k) - 1) / (nw[i] - k);        // even split, larger parts first
            const bool last = k == nw[i] - 1;
            const int tail_nv = last ? r.tail_nv : 0;
            if (n > lean_depth(r.bits, S) * ((S == 16 || r.bits <= LEAN_S8_PASS_BITS || sc_rows_max > 64) ? passes : 1) || n > 255) return 0;      // (sc_rows_max > 64: the LOADS form -- any width)
            LeanWave& lw = wave[w];
            lw.lds_off = lds_total;
            const int c0 = r.chunk0 + 4 * i0;
            const int nch = 4 * n + tail_nv;                                   // chunks this wave multiplies
            if (nch <= 0) continue;
            if (c0 + nch > n_chunks) return 0;
            const int c_end = c0 + 4 * n + (tail_nv ? 4 : 0);                  // a partial item reads 4 chunks' worth of activations (clamped to the row)
            int g_lo = 0x7fffffff, g_hi = -1;
            for (int cc = c0; cc < c0 + nch; cc++) { const int g = qm->cg_host[cc]; if (g < g_lo) g_lo = g; if (g > g_hi) g_hi = g; }
            // affine group map: row(chunk) = g0 + ((chunk - c0 + phase) >> shift)
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
            if (shift < 0 || c0 > 0xFFFF || c_end - c0 > 0xFFFF || g_lo > 0xFFFF || g_hi - g_lo + 1 > 0xFFFF) return 0;
            // what the kernel's straight-line staging copies (stage_copies); ROWS: the rows are staged by the workgroup, any length
            // (XMEM: no staged activations -- any M <= 16, any slice; one pass only: the ring of A operands covers LeanDepth items)
            if ((g_hi - g_lo + 1) > sc_rows_max || (!rows_mode && !xmem && ((c_end - c0) * 4 > 64 * ((S == 4 && passes > 1) ? LEAN_X_PIECES_PAIR_LOADS : LEAN_X_PIECES_BASE) || M > 4))) return 0;
            if (xmem && n > lean_depth(r.bits, S)) return 0;
            const bool uni = shift >= 2 && (phase & 3) == 0;                   // the four chunks of every full item share a group
            lw.w_off = r.off + (u32)i0 * 64u * (u32)r.bits; lw.w_tstride = r.tstride;
            lw.t_off = r.t_off; lw.t_tstride = r.t_tstride;
            static const u32 pipe_off = []() { const char* e = getenv("EXL2_LEAN_PIPE"); return (e && atoi(e) == 0) ? 1u : 0u; }();   // A/B switch
            lw.meta = (u32)n | ((u32)r.bits << 8) | ((u32)tail_nv << 12) | (uni ? 1u << 15 : 0u) | ((u32)shift << 16) | ((u32)phase << 19) | (pipe_off << 29);
            lw.xr = (u32)c0 | ((u32)(c_end - c0) << 16);
            lw.gr = (u32)g_lo | ((u32)(g_hi - g_lo + 1) << 16);
            lw.place = (u32)c0 | ((u32)(g0 - g_lo) << 16);
            const u32 x_stride = (u32)(c_end - c0) * 32u + 8u;
            (void)norm;
            lw.x_stride = x_stride;
            lw.off_sc = (rows_mode || xmem) ? 0u : al16((u32)M * x_stride * 2u);
            lw.off_zp = lw.off_sc + al16((u32)(g_hi - g_lo + 1) * 32u);
            lds_total += lw.off_sc + al16((u32)(g_hi - g_lo + 1) * 32u) * (qm->is_gptq ? 2u : 1u);
            i0 += n;
        }
        if (i0 != r.F) return 0;
    }
    return lds_total ? lds_total : 16u;
}

#define LEAN_FOR_EACH_GEOMETRY(X, OCC) X(8, 1, false, OCC) X(16, 1, false, OCC) X(8, 2, false, OCC) X(8, 2, true, OCC) X(4, 2, true, OCC)
#define LEAN_FOR_EACH_ROWS_GEOMETRY(X) X(8, 1, false, false) X(16, 1, false, false) X(8, 2, false, true) X(8, 2, true, true)
#define LEAN_FOR_EACH_XMEM_GEOMETRY(X) X(8, 1, false) X(16, 1, false) X(8, 2, false) X(8, 2, true)
// register budget: 6 waves per SIMD (80 registers: no spills on the common paths, three 8-wave workgroups per CU).  4 and 8
// were built side by side during the round and measured (4 slower; 8 equal within noise once gate|up used 8-wave workgroups,
// with spills): tools/build_variant.sh -DLEAN_OCC_DEFAULT=... rebuilds them
#ifndef LEAN_OCC_DEFAULT
#define LEAN_OCC_DEFAULT 6
#endif
// the several-loads pair (70B gate|up): 80 registers as well -- three 8-wave workgroups per CU (53 KB of LDS each); at 128 registers
// (two per CU) the launch takes what the 16-wave geometry took (59 us: gpurun r08a)
#ifndef LEAN_PAIR_LOADS_OCC
#define LEAN_PAIR_LOADS_OCC LEAN_OCC_DEFAULT
#endif

static void lean_attrs()
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (!exl2_first_on_device(attr)) return;
#define LEAN_ATTR(S, NS, P, OCC) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, S, NS, P, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, S, NS, P, OCC, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, LEAN_OCC_DEFAULT)
#undef LEAN_ATTR
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, 4, 2, true, LEAN_PAIR_LOADS_OCC, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, 4, 2, true, LEAN_PAIR_LOADS_OCC, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define LEAN_ATTR(S, NS, P, OCC) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, S, NS, P, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, S, NS, P, OCC, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_GEOMETRY(LEAN_ATTR, 6)
#undef LEAN_ATTR
#define LEAN_ATTR(S, NS, P, W) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, S, NS, P, 4, true, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, S, NS, P, 4, true, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_ROWS_GEOMETRY(LEAN_ATTR)
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, 8, 2, false, 4, true, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#undef LEAN_ATTR
#define LEAN_ATTR(S, NS, P) \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<false, S, NS, P, 4, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)qgemv_lean_kernel<true, S, NS, P, 4, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LEAN_FOR_EACH_XMEM_GEOMETRY(LEAN_ATTR)
#undef LEAN_ATTR
}

// what qgemv_lean_launch hands out instead of launching (FlatIn.lean_export): the argument block and the geometry it chose
struct LeanExport { LeanArgs args; int S, nslots, pair, walk, grid_x; u32 lds; bool plain; };

static int lean_cus()
{
    static int cus[EXL2_MAX_DEVICES] = {0};
    const int dev = exl2_current_device();
    if (!cus[dev]) { hipDeviceProp_t prop; cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    if (const char* e = getenv("EXL2_LEAN_CUS")) { const int v = atoi(e); if (v > 0) return v; }      // (tests: the walking forms on small shapes)
    return cus[dev];
}

// 0: launched; 1: shape not covered (the caller takes the round-2 kernel).  *wgs_out = grid size = partial sums per row a
// chain-out launch publishes.
int qgemv_lean_launch(const FlatIn& in, void* stream, int* wgs_out)
{
    if (in.n_mats < 1 || in.n_mats > FLAT_MAX_MATS || in.M < 1 || in.M > LEAN_MAX_ROWS) return 1;
    static const int rows_on = []() { const char* e = getenv("EXL2_LEAN_ROWS"); return e ? atoi(e) : 1; }();
    bool rows_mode = in.M > LEAN_MAX_M;
    bool rows_loads = false;                                        // the LOADS form (lean_body): one row of a long K, two tiles per workgroup
    if (rows_mode && !rows_on) return 1;
    const bool dep = in.sync_signal != nullptr;                     // a launch of an overlapped chain (chain_sync.h): <= 4 rows only
    if (!dep && (in.sync_wait || in.sync_arrive)) return 1;
    if (dep && rows_mode) return 1;
    if ((in.a_tiled || in.c_tiled || in.xp_tiled) && (dep || in.c_mode == C_ACCUM && in.c_tiled || (in.c_tiled && !in.pair && in.n_mats != 1))) return 1;
    if (const char* e = getenv("EXL2_LEAN_DECLINE_M")) { if (atoi(e) == in.M) return 1; }      // test hook: row groups on different kernels
    const QMatrix* q0 = in.qm[0];
    const int K = q0->height;
    if ((K & 7) || (((size_t)in.a) & 15) || (in.lda & 7)) return 1;
    if (in.a_mode == A_NORM_PRE && (!in.ss || in.npart < 1)) return 1;
    if (in.pair && (in.n_mats != 2 || in.qm[0]->width != in.qm[1]->width)) return 1;
    if (in.pair_sum && (!in.lean_export || !in.pair || in.M != 1 || dep || in.a_tiled || in.c_tiled || in.xp_tiled)) return 1;
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
    // candidates, in order: a one-row pair tries 4 waves per tile first (8-wave workgroups: three per CU at 6 waves per SIMD, so
    // the 688 workgroups of a 7B gate|up launch are resident at once; with 16-wave workgroups a quarter of them starts when
    // the first ones have finished: profiles/history/r03_trace_lean_v6.txt, workgroup entry p90 9.5 us)
    int cand[3] = {0, 0, 0}, n_cand = 0;
    static const int pair4 = []() { const char* e = getenv("EXL2_LEAN_PAIR4"); return e ? atoi(e) : 1; }();
    int cand_passes[3] = {0, 0, 0};                                    // (0 = the geometry's default: lean_plan_matrix)
    u32 cand_budget[3] = {0, 0, 0};                                    // (0 = LEAN_LDS_BUDGET per 8 waves)
    if (in.pair && in.M == 1 && pair4) cand[n_cand++] = 4;
    // a pair whose K is too long for ONE register load of its 4 waves per tile (70B: K = 8192 at 2-3 bits = 19 items per wave): two
    // loads (the ring of the 8-wave q|k|v launches) and a 64 KB budget -- two 8-wave workgroups per CU, 512 of the 1792 resident
    // at once -- before the 16-wave workgroup that is alone on its CU (seven rounds of 256: profiles/history/r05_70b_kernel_stats.csv)
    static const int pair4_passes = []() { const char* e = getenv("EXL2_LEAN_PAIR4_PASSES"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > LEAN_MAX_PASSES ? LEAN_MAX_PASSES : v); }();
    static const u32 pair4_budget = []() { const char* e = getenv("EXL2_LEAN_PAIR4_BUDGET_KB"); const int v = e ? atoi(e) : 64; return (u32)(v < 16 ? 16 : (v > 150 ? 150 : v)) * 1024u; }();
    if (in.pair && in.M == 1 && pair4 && pair4_passes > 1 && !in.a_tiled && !dep) { cand_passes[n_cand] = pair4_passes; cand_budget[n_cand] = pair4_budget; cand[n_cand++] = 4; }
    cand[n_cand++] = S;
    // several matrices in one launch (q|k|v) whose shares 8 waves cannot hold in ONE register load: two loads on 8 waves before 16
    // waves -- a 16-wave workgroup is alone on its CU (80 registers x 4 waves per SIMD), so the 640 tiles of a 70B q|k|v launch ran
    // as three rounds of 256; 8-wave workgroups are three per CU and all resident at once
    static const int s8_passes = []() { const char* e = getenv("EXL2_LEAN_S8_PASSES"); const int v = e ? atoi(e) : LEAN_S8_PASSES; return v < 1 ? 1 : (v > LEAN_S8_PASSES ? LEAN_S8_PASSES : v); }();
    const int s8_single = []() { const char* e = getenv("EXL2_LEAN_S8_SINGLE"); return e ? atoi(e) : 0; }();      // (A/B: the two-load 8-wave form for ONE matrix too -- down_proj)
    if (!in.pair && nslots == 1 && S == 8 && (in.n_mats >= 2 || s8_single) && s8_passes > 1 && !rows_mode && !in.a_tiled) { cand_passes[n_cand] = s8_passes; cand[n_cand++] = 8; }
    if (!in.pair && nslots == 1 && S == 8) cand[n_cand++] = 16;        // finer split of the tile
    // ROWS geometries, in order: two tiles x 8 waves sharing the staged rows (16 waves per CU), one tile x 8, one tile x 16; pair (8 + 8)
    // XMEM geometries (the rows do not fit / EXL2_LEAN_XMEM=2): one tile x 8, one tile x 16; pair (8 + 8)
    // the weighted sum of two experts' down projections (MOE_SUM): 8 waves per expert's tile, up to LEAN_MAX_PASSES ring loads each,
    // the 16-wave workgroup alone on its CU (two activation slices + two sets of scale rows: 86 KB at Mixtral's K = 14336)
    if (in.pair_sum) { n_cand = 1; cand[0] = 8; cand_passes[0] = LEAN_MAX_PASSES; cand_budget[0] = 150u * 1024u; }
    int cand_slots[3] = {nslots, nslots, nslots};
    const int xmem_on = []() { const char* e = getenv("EXL2_LEAN_XMEM"); return e ? atoi(e) : 1; }();     // (read per launch build: tests switch it)
    bool xmem = false;
    u32 rows_bytes = 0, slot_bytes = 0;
    bool planned = false;
    int planned_passes = 0;
    const int nslots0 = nslots;
    const int plain_cand[3] = {cand[0], cand[1], cand[2]};
    const int plain_passes[3] = {cand_passes[0], cand_passes[1], cand_passes[2]};
    const u32 plain_budget[3] = {cand_budget[0], cand_budget[1], cand_budget[2]};
    const int plain_n = n_cand;
    auto plan = [&](int form) {                                      // 0: <= 4 rows; 1: ROWS; 2: XMEM
        n_cand = plain_n;
        for (int i = 0; i < 3; i++) { cand[i] = plain_cand[i]; cand_slots[i] = nslots0; cand_passes[i] = form ? 0 : plain_passes[i]; cand_budget[i] = form ? 0u : plain_budget[i]; }
        if (form == 3)                                              // LOADS: two tiles x 8 waves, shares of up to LEAN_MAX_PASSES register loads
        {
            n_cand = 1; cand[0] = 8; cand_slots[0] = 2; cand_passes[0] = LEAN_MAX_PASSES;
        }
        else if (form == 1)
        {
            n_cand = 0;
            if (in.pair) { cand[n_cand] = 8; cand_slots[n_cand++] = 2; }
            else
            {
                // (up to 8 rows the staged rows leave room for two 8-wave workgroups per CU: one tile per workgroup, the whole grid
                // resident; from 9 rows up one workgroup per CU anyway: two tiles per unit, the workgroup walks -- profiles/history/r04_rows_sweep*.txt)
                if (in.M >= 9 && !(in.ss_out && (max_tiles + 1) / 2 > LEAN_MAX_PART) && max_tiles >= 2) { cand[n_cand] = 8; cand_slots[n_cand++] = 2; }
                cand[n_cand] = 8; cand_slots[n_cand++] = 1;
                cand[n_cand] = 16; cand_slots[n_cand++] = 1;               // (K = 11008: a tile's share does not fit 8 waves' registers)
            }
        }
        else if (form == 2)
        {
            n_cand = 0;
            if (in.pair) { cand[n_cand] = 8; cand_slots[n_cand++] = 2; }
            else if (in.ss_out && max_tiles > LEAN_MAX_PART) { cand[n_cand] = 8; cand_slots[n_cand++] = 2; }
            else
            {
                cand[n_cand] = 8; cand_slots[n_cand++] = 1;
                cand[n_cand] = 16; cand_slots[n_cand++] = 1;
            }
        }
        // ROWS: the workgroup's shared copy of the M rows, in front of the waves' own (scale-row) areas
        rows_bytes = (form == 1 || form == 3) ? al16((u32)in.M * (u32)(K + 8) * 2u) : 0u;
        // (second round: a split that fits only with the whole LDS of a CU -- 2-4 rows of a K = 11008 matrix -- is still better
        // than leaving the chain: the decoder would fall back to the module-by-module route for EVERY launch)
        for (int ci = 0; ci < 2 * n_cand && !planned; ci++)
        {
            S = cand[ci % n_cand];
            if (form) nslots = cand_slots[ci % n_cand];
            const u32 budget = (form == 1 || form == 3) ? 158u * 1024u : (ci < n_cand ? (cand_budget[ci] ? cand_budget[ci] : LEAN_LDS_BUDGET * (u32)((S * nslots + 7) / 8)) : 150u * 1024u);
            if (in.n_mats * S > LEAN_RECORDS) continue;
            bool ok = true;
            slot_bytes = 0;
            for (int j = 0; j < in.n_mats && ok; j++)
            {
                const u32 b = lean_plan_matrix(in.qm[j], S, in.M, in.a_mode == A_NORM_PRE, a.wave + j * S, form == 1 || form == 3, form == 2, cand_passes[ci % n_cand], form == 3 ? 128 : 64);
                if (!b) ok = false;
                if (b > slot_bytes) slot_bytes = b;
            }
            planned = ok && rows_bytes + (u32)nslots * slot_bytes + (u32)(S * nslots) * (u32)in.M * 64u <= budget;
            if (planned) planned_passes = cand_passes[ci % n_cand];
        }
        xmem = planned && form == 2;
    };
    if (in.a_tiled) plan(2);                                       // (only the XMEM form reads that layout)
    else if (!rows_mode) plan(0);
    else if (xmem_on >= 2) { plan(2); if (!planned) plan(1); }
    else { plan(1); if (!planned && xmem_on >= 1) plan(2); }
    if (!planned) return 1;
    // one row, one matrix, a 16-wave workgroup per tile and more tiles than CUs (two rounds of workgroups that are alone on their CU):
    // the LOADS form -- two tiles per workgroup over ONE staged copy of the row, one round -- where it plans
    static const int rows1_on = []() { const char* e = getenv("EXL2_LEAN_ROWS1"); return e ? atoi(e) : 1; }();
    if (rows1_on && !rows_mode && !xmem && !dep && !in.pair && !in.pair_sum && in.n_mats == 1 && in.M == 1 && S == 16 && !q0->is_gptq && !in.a_tiled && !in.c_tiled && !in.xp_tiled &&
        max_tiles > lean_cus() && !(in.ss_out && (max_tiles + 1) / 2 > LEAN_MAX_PART))
    {
        planned = false;
        plan(3);
        if (planned) { rows_mode = true; rows_loads = true; }
        else { plan(0); if (!planned) return 1; }
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
    h.a = in.a; h.xp_w = in.xp_w; h.ss = in.ss; h.xp_out = in.xp_out; h.xp_invperm = in.xp_invperm; h.ss_out = in.ss_out;
    h.eps = in.eps; h.M = in.M; h.K = K; h.lda = in.lda; h.ldxp = in.ldxp; h.npart = in.npart; h.wgs = wgs;
    h.flags = (in.a_mode == A_NORM_PRE ? LF_NORM : 0u) | (in.act_gelu ? LF_GELU : 0u) | (in.c_mode == C_ACCUM ? LF_ACCUM : 0u) | (any_bias ? LF_BIAS : 0u)
            | (dep ? LF_DEP : 0u) | (in.a_tiled ? LF_ATILED : 0u) | (in.c_tiled ? LF_CTILED : 0u) | (in.xp_tiled ? LF_XPTILED : 0u);
    if (dep)
    {
        h.sync_wait = in.sync_wait; h.sync_signal = in.sync_signal; h.sync_wgs = (u32)wgs * (u32)(in.pair ? 1 : in.n_mats);
        h.sync_pad = in.sync_arrive ? 1u : 0u;
        // (no pipelined form: the staging copies wait for the producer, the weight requests must not)
        for (int j = 0; j < in.n_mats; j++) for (int w = 0; w < S; w++) a.wave[j * S + w].meta |= (1u << 29) | (1u << 30);
    }
    h.slot_bytes = slot_bytes; h.red_off = rows_bytes + slot_bytes * (u32)nslots;
    const int waves = S * nslots;
    const u32 lds = h.red_off + (u32)waves * (u32)in.M * 16u * 4u;
    lean_attrs();
#ifdef EXL2_TRACE
    h.trace = (g_ltrace_buf && g_ltrace_count++ == g_ltrace_which) ? g_ltrace_buf : nullptr;
#endif
    if (getenv("EXL2_LEAN_TRACE"))
    {
        fprintf(stderr, "[lean] M=%d K=%d mats=%d pair=%d S=%d slots=%d wgs=%d lds=%u mode=%d form=%s a_tiled=%d c_tiled=%d\n", in.M, K, in.n_mats, in.pair, S, nslots, wgs, lds,
                in.a_mode, xmem ? "xmem" : rows_mode ? "rows" : "plain", in.a_tiled, in.c_tiled);
        for (int j = 0; j < in.n_mats; j++)
            for (int w = 0; w < S; w++)
            {
                const LeanWave& lw = a.wave[j * S + w];
                fprintf(stderr, "[lean]  mat %d wave %d: lds+%u, x chunks %u+%u, sc rows %u+%u; %u x %ub%s chunk %u row+%u sh%u ph%u%s\n", j, w, lw.lds_off,
                        lw.xr & 0xFFFF, lw.xr >> 16, lw.gr & 0xFFFF, lw.gr >> 16, lw.meta & 0xFF, (lw.meta >> 8) & 0xF,
                        ((lw.meta >> 12) & 7) ? " + partial" : "", lw.place & 0xFFFF, lw.place >> 16, (lw.meta >> 16) & 7, (lw.meta >> 19) & 0x3FF,
                        ((lw.meta >> 15) & 1) ? " uni" : "");
            }
    }
    // ROWS: one workgroup per CU fits (the staged rows fill the LDS), so the grid is sized to the CUs and a workgroup walks its
    // units with one staged copy; inside an overlapped chain every workgroup takes one unit (its arrival counts are per unit)
    int grid_x = wgs;
    if (rows_mode && !xmem && !dep && nslots == 2)                   // (the WALK instantiations: the two-tile geometries)
    {
        int cap = lean_cus() / (in.pair ? 1 : in.n_mats);
        if (const char* e = getenv("EXL2_LEAN_ROWS_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }      // (tests: force the walk on small shapes)
        if (cap < 1) cap = 1;
        if (grid_x > cap) grid_x = cap;
    }
    dim3 grid((unsigned)grid_x, (unsigned)(in.pair ? 1 : in.n_mats), 1), block((unsigned)waves * 64, 1, 1);
    if (in.lean_export)
    {
        LeanExport* const ex = (LeanExport*)in.lean_export;
        ex->args = a; ex->S = S; ex->nslots = nslots; ex->pair = in.pair; ex->grid_x = grid_x; ex->lds = lds;
        ex->walk = in.pair && ((S == 4 && planned_passes > 1) || in.pair_sum);
        ex->plain = !rows_mode && !xmem && !dep && !q0->is_gptq && (in.pair || in.n_mats == 1);
        if (wgs_out) *wgs_out = wgs;
        return 0;
    }
    if (in.plan_only) { if (wgs_out) *wgs_out = wgs; return 0; }                      // a caller asking whether this shape is taken (modules.hip)
    if (getenv("EXL2_LEAN_PLAN_ONLY")) { if (wgs_out) *wgs_out = wgs; return 0; }     // test hook: the host plan without the launch (results undefined)
    int launched = 0;                                                                  // a plan no instantiation below matches must not pass for a launch
    const bool gptq = q0->is_gptq;
    const int occ = LEAN_OCC_DEFAULT;
#define LEAN_LAUNCH(...) do { launched++; LAUNCH(__VA_ARGS__); } while (0)
    if (rows_loads) LEAN_LAUNCH((qgemv_lean_kernel<false, 8, 2, false, 4, true, true, false, false, true>), grid, block, lds, stream, a);
#define LEAN_GO(SS, NS, P, W) \
    if (!rows_loads && rows_mode && !xmem && !gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<false, SS, NS, P, 4, true, W>), grid, block, lds, stream, a); \
    if (rows_mode && !xmem && gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<true, SS, NS, P, 4, true, W>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_ROWS_GEOMETRY(LEAN_GO)
#undef LEAN_GO
#define LEAN_GO(SS, NS, P) \
    if (xmem && !gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<false, SS, NS, P, 4, false, false, false, true>), grid, block, lds, stream, a); \
    if (xmem && gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<true, SS, NS, P, 4, false, false, false, true>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_XMEM_GEOMETRY(LEAN_GO)
#undef LEAN_GO
    const bool pair_loads = in.pair && S == 4 && planned_passes > 1 && !rows_mode && !xmem && !dep;      // (the several-register-loads pair: its own instantiation)
    if (pair_loads && !gptq) LEAN_LAUNCH((qgemv_lean_kernel<false, 4, 2, true, LEAN_PAIR_LOADS_OCC, false, true>), grid, block, lds, stream, a);
    if (pair_loads && gptq) LEAN_LAUNCH((qgemv_lean_kernel<true, 4, 2, true, LEAN_PAIR_LOADS_OCC, false, true>), grid, block, lds, stream, a);
#define LEAN_GO(SS, NS, P, OCC) \
    if (!pair_loads && !rows_mode && !xmem && !dep && !gptq && occ == OCC && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<false, SS, NS, P, OCC>), grid, block, lds, stream, a); \
    if (!rows_mode && dep && !gptq && occ == OCC && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<false, SS, NS, P, OCC, false, false, true>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_GEOMETRY(LEAN_GO, LEAN_OCC_DEFAULT)
#undef LEAN_GO
#define LEAN_GO(SS, NS, P, OCC) \
    if (!pair_loads && !rows_mode && !xmem && !dep && gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<true, SS, NS, P, OCC>), grid, block, lds, stream, a); \
    if (!rows_mode && dep && gptq && S == SS && nslots == NS && (in.pair != 0) == P) LEAN_LAUNCH((qgemv_lean_kernel<true, SS, NS, P, OCC, false, false, true>), grid, block, lds, stream, a);
    LEAN_FOR_EACH_GEOMETRY(LEAN_GO, 6)
#undef LEAN_GO
#undef LEAN_LAUNCH
    if (launched != 1)
    {
        fprintf(stderr, "[exl2] qgemv_lean_launch: plan (S=%d slots=%d pair=%d rows=%d xmem=%d dep=%d gptq=%d) matched %d instantiations\n",
                S, nslots, in.pair, (int)rows_mode, (int)xmem, (int)dep, (int)gptq, launched);
        return launched ? -3 : 1;                   // nothing launched: declined (the caller's fallback runs); two: a dispatch bug
    }
    if (wgs_out) *wgs_out = wgs;
    return 0;
}

// ---- grouped-expert launches (qgemv_lean.h) ---------------------------------------------------------------------------------------
#define LEAN_FOR_EACH_MOE_GEOMETRY(X) X(4, 2, true, LEAN_OCC_DEFAULT, false) X(4, 2, true, LEAN_PAIR_LOADS_OCC, true) X(8, 2, true, LEAN_OCC_DEFAULT, false) \
                                      X(8, 2, true, LEAN_OCC_DEFAULT, true) \
                                      X(8, 1, false, LEAN_OCC_DEFAULT, false) X(16, 1, false, LEAN_OCC_DEFAULT, false)

int qgemv_lean_group_plan(const FlatIn* ins, int n_groups, const f16* const* out_scale, int n_sel, LeanGroupPlan* gp, int r_stride, int mul)
{
    if (!ins || !gp || n_groups < 1 || n_sel < 1 || n_sel > n_groups) return 1;
    memset(gp, 0, sizeof(*gp));
    std::vector<LeanArgs> blocks((size_t)n_groups);
    static LeanExport ex;                                            // (large: off the stack)
    static std::mutex gmtx;
    std::lock_guard<std::mutex> lock(gmtx);
    for (int g = 0; g < n_groups; g++)
    {
        FlatIn in = ins[g];
        in.lean_export = &ex; in.plan_only = 0;
        if (in.M < 1 || in.M > LEAN_MAX_M || (in.pair_sum && in.M != 1) || in.sync_signal || in.sync_wait || in.sync_arrive || in.a_tiled || in.c_tiled || in.xp_tiled) return 1;
        int wgs = 0;
        const int rc = qgemv_lean_launch(in, nullptr, &wgs);
        if (rc != 0 || !ex.plain || (in.pair_sum && (wgs > LEAN_MAX_PART || 2 * ex.S > LEAN_RECORDS))) return 1;
        if (g == 0) { gp->S = ex.S; gp->nslots = ex.nslots; gp->pair = ex.pair; gp->walk = ex.walk ? 1 : 0; gp->grid_x = ex.grid_x; gp->lds = ex.lds; }
        else
        {
            if (gp->S != ex.S || gp->nslots != ex.nslots || gp->pair != ex.pair || gp->walk != (ex.walk ? 1 : 0) || gp->grid_x != ex.grid_x) return 1;
            if (ex.lds > gp->lds) gp->lds = ex.lds;
        }
        ex.args.hdr.out_scale = out_scale ? out_scale[g] : nullptr;
        ex.args.hdr.r_stride = r_stride; ex.args.hdr.moe_mul = mul ? 1u : 0u;
        if (in.pair_sum)
        {
            // both slots planned from THIS expert (the front kernel of a step takes slot 1's matrix block and wave records from the
            // second selected expert's table entry): every record carries the expert's routing-weight and activation pointers
            if (!out_scale || !out_scale[g]) return 1;
            for (int i = 0; i < 2 * ex.S; i++)
            {
                const u64 p0 = (u64)(size_t)out_scale[g], p1 = (u64)(size_t)in.a;
                u32* const pad = ex.args.wave[i].pad;
                pad[0] = (u32)p0; pad[1] = (u32)(p0 >> 32); pad[2] = (u32)p1; pad[3] = (u32)(p1 >> 32);
            }
            ex.args.hdr.out_scale = nullptr;
        }
        blocks[(size_t)g] = ex.args;
    }
    if (ins[0].pair_sum)
    {
        // a step's block mixes two experts' records: one slot size, one place for the partial sums in all of them
        u32 slot_bytes = 0;
        for (int g = 0; g < n_groups; g++) if (blocks[(size_t)g].hdr.slot_bytes > slot_bytes) slot_bytes = blocks[(size_t)g].hdr.slot_bytes;
        for (int g = 0; g < n_groups; g++)
        {
            LeanHdr& h = blocks[(size_t)g].hdr;
            h.slot_bytes = slot_bytes; h.red_off = slot_bytes * (u32)gp->nslots;
        }
        gp->lds = slot_bytes * (u32)gp->nslots + (u32)(gp->S * gp->nslots) * 16u * 4u;
        if (gp->lds > 160u * 1024u) return 1;
    }
    bool inst = false;
#define LEAN_HAS(SS, NS, P, OCC, W) if (gp->S == SS && gp->nslots == NS && (gp->pair != 0) == P && (gp->walk != 0) == W) inst = true;
    LEAN_FOR_EACH_MOE_GEOMETRY(LEAN_HAS)
#undef LEAN_HAS
    if (!inst) return 1;
    gp->n_groups = n_groups; gp->n_sel = n_sel; gp->block_bytes = (int)sizeof(LeanArgs);
    gp->pair_sum = ins[0].pair_sum ? 1 : 0;
    gp->all_groups = r_stride != 0 ? 1 : 0;
    // (pair_sum: the ONE block a step's launch reads = the first selected expert's block with units [b_lo, b_hi) and [b2_lo, b2_hi)
    // -- slot 1's matrix block and wave records -- taken from the second selected expert's)
    gp->b_lo = (int)(offsetof(LeanArgs, mat) + sizeof(LeanMat)) / 16; gp->b_hi = gp->b_lo + (int)sizeof(LeanMat) / 16;
    gp->b2_lo = (int)(offsetof(LeanArgs, wave) + (size_t)gp->S * sizeof(LeanWave)) / 16; gp->b2_hi = gp->b2_lo + gp->S * (int)sizeof(LeanWave) / 16;
    if (hipMalloc(&gp->table_src, (size_t)n_groups * sizeof(LeanArgs)) != hipSuccess) { gp->table_src = nullptr; return 1; }
    if (hipMalloc(&gp->table_sel, (size_t)n_sel * sizeof(LeanArgs)) != hipSuccess) { (void)hipFree(gp->table_src); gp->table_src = gp->table_sel = nullptr; return 1; }
    if (hipMemcpy(gp->table_src, blocks.data(), (size_t)n_groups * sizeof(LeanArgs), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(gp->table_sel, blocks.data(), (size_t)n_sel * sizeof(LeanArgs), hipMemcpyHostToDevice) != hipSuccess)    // (valid blocks until the first front kernel runs)
    { qgemv_lean_group_free(gp); return 1; }
    return 0;
}

void qgemv_lean_group_free(LeanGroupPlan* gp)
{
    if (!gp) return;
    if (gp->table_src) (void)hipFree(gp->table_src);
    if (gp->table_sel) (void)hipFree(gp->table_sel);
    gp->table_src = gp->table_sel = nullptr;
}

int qgemv_lean_group_launch(const LeanGroupPlan* gp, void* stream, const LeanGroupDyn* dyn_)
{
    if (!gp || !gp->table_sel) return 1;
    LeanDyn dyn; memset(&dyn, 0, sizeof(dyn));
    if (dyn_) { dyn.c = (f16*)dyn_->c; dyn.xp_out = (f16*)dyn_->xp_out; dyn.xp_invperm = (const u16*)dyn_->xp_invperm; dyn.xp_w = (const f16*)dyn_->xp_w;
                dyn.ss_out = dyn_->ss_out; dyn.ldc = dyn_->ldc; dyn.ldxp = dyn_->ldxp; }
    if ((gp->pair_sum != 0) != (dyn_ != nullptr && dyn_->c != nullptr)) return 1;
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
    {
#define LEAN_ATTR(SS, NS, P, OCC, W) (void)hipFuncSetAttribute((const void*)qgemv_lean_moe_kernel<SS, NS, P, OCC, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        LEAN_FOR_EACH_MOE_GEOMETRY(LEAN_ATTR)
#undef LEAN_ATTR
    }
    // (all_groups: blockIdx.y = expert, the blocks as planned -- workgroups of experts without a row leave at entry)
    const dim3 grid((unsigned)gp->grid_x, (unsigned)(gp->pair_sum ? 1 : (gp->all_groups ? gp->n_groups : gp->n_sel)), 1), block((unsigned)(gp->S * gp->nslots) * 64, 1, 1);
    const LeanArgs* const table = (const LeanArgs*)(gp->all_groups ? gp->table_src : gp->table_sel);
    int launched = 0;
#define LEAN_GO(SS, NS, P, OCC, W) \
    if (gp->S == SS && gp->nslots == NS && (gp->pair != 0) == P && (gp->walk != 0) == W) { launched++; LAUNCH((qgemv_lean_moe_kernel<SS, NS, P, OCC, W>), grid, block, gp->lds, stream, table, dyn); }
    LEAN_FOR_EACH_MOE_GEOMETRY(LEAN_GO)
#undef LEAN_GO
    return launched == 1 ? 0 : 1;
}
