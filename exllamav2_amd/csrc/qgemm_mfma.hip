// qgemm_mfma.hip -- prefill-shaped q_matrix x fp16 GEMM, second generation: the packed weights are decoded ONCE per workgroup
// into LDS (qgemm_prefill.hip decodes them in registers, twice per wave column, and holds 256 VGPRs for it).
//
// Replaces the reference's M > 32 path (cuda/q_gemm.cu:243-263: reconstruct_kernel writes the fp16 [K, N] matrix to HBM,
// cuBLAS reads it back) and the library GEMM this package's host policy used above 64 rows in round 1.
//
//   * block tile 32 MT (rows of C) x 256 (columns), K step 64; 8 waves = 2 (rows) x 4 (columns), wave tile 16 MT x 64,
//     MT x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (MT = 8: 128 VGPRs).  The WEIGHTS are the matrix core's A operand and
//     the activations its B operand (C^T = W^T X^T): a lane then owns four CONSECUTIVE columns of one row of C -- 8-byte stores
//     in the epilogue instead of 2-byte ones;
//   * W stage in LDS = [16 column tiles][2 chunks][64 lanes][16 bytes]: exactly the MFMA fragments, lane-linear, written by
//     ds_write_b128 and read by ds_read_b128 without bank conflicts.  A wave owns two of the 16 tiles: it loads their packed
//     super-chunk (128 K rows; qlayout.h) one K step ahead, decodes chunks {0, 1} for the first K step of the super-chunk and
//     {2, 3} for the second (same decoders, same half(q - z) * half(scale) rounding as reconstruct(): gemm(I) == reconstruct()
//     bit for bit), so a weight is decoded once per workgroup instead of once per wave row;
//   * X stage = [rows][8 units of 16 bytes], unit u of row r at position u ^ ((r >> 1) & 7): filled by LDS-DMA (the lane's
//     GLOBAL address carries the swizzle, the LDS side is lane-linear), read conflict-free by ds_read_b128;
//   * two stages; ONE barrier per K step: everything a step reads was written / requested during the previous step.  The two
//     waves that share a SIMD (wave w and w + 4) take the decode in opposite order on the decode-heavy step -- one multiplies
//     while the other decodes -- so the matrix pipe has work while VALU decodes;
//   * workgroup -> tile map is XCD-aware: the 32 workgroups resident on one XCD (= one L2) at a time cover a 4 x 8 patch of
//     tiles and walk K in step, so an X row block is fetched into that L2 once per 8 column tiles.
// MFMA-bound by design: per K step a wave issues 16 MT MFMAs (MT = 8: 128 x 16 cycles) against ~60 VALU of decode.
#include "qgemm_prefill.h"
#include "errors.h"
#include <stdlib.h>
#include <string.h>

#ifndef MF_KILL
#define MF_KILL 0                        // timing experiments (results WRONG): 1 no X copies, 2 no weight decode, 4 no multiply
#endif
#ifdef MF_NOFENCE
#define MF_FENCE() do { } while (0)
#else
#define MF_FENCE() sched_fence()
#endif
#define MF_BN 256
#ifndef MF_WPRE_MIN_ROWS
#define MF_WPRE_MIN_ROWS 129             // rows from which the weights are decoded once per call (wfrag_kernel): every call this file
                                        // serves -- it wins from 192 rows up (profiles/history/r03_prefill_wpre_threshold.txt)
#endif
#define MF_THREADS 512
#define MF_W_STAGE 32768                        // 16 tiles x 2 chunks x 64 lanes x 16 bytes
#define MF_X_STAGE(MT) ((MT) * 32 * 128)        // rows x 64 halves
#define MF_MAX_CHUNKS 2048                      // chunk -> group map in LDS: K <= 65536

// what a wave keeps of its two tiles between the load and the two decodes of a super-chunk
struct TileRaw { u32 w[2][8]; f16x2 sc[2][2]; f16x2 zp[2][2]; };  // packed words; scale / (GPTQ) zero point of chunks {0,1}, {2,3}

// packed words of super-chunk `sc_ptr` (of tile 0; tiles are tile_stride words apart) and the scales of its 4 chunks
template <int BITS, bool GPTQ>
DEV void load_raw(TileRaw& R, const QMatDev& m, const u32* sc_ptr, u32 tile_stride, int chunk0, int nvalid,
                  const int (&tile)[2], const u16* cg_lds, int lane)
{
    #pragma unroll
    for (int t = 0; t < 2; t++)
    {
        LaneWords<BITS> lw;
        load_lane_words<BITS>(sc_ptr + (size_t)tile[t] * tile_stride, lane, lw);
        #pragma unroll
        for (int i = 0; i < BITS; i++) R.w[t][i] = lw.w[i];
    }
    int g[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) g[q] = uniform((int)cg_lds[q < nvalid ? chunk0 + q : chunk0]);     // padded chunks are never multiplied in
    #pragma unroll
    for (int t = 0; t < 2; t++)
    {
        const f16* st = m.sc_tab + (size_t)tile[t] * m.G * 16 + (lane & 15);
        const f16 s0 = st[g[0] * 16], s1 = st[g[1] * 16], s2 = st[g[2] * 16], s3 = st[g[3] * 16];
        R.sc[t][0] = (f16x2){s0, s1}; R.sc[t][1] = (f16x2){s2, s3};
        if constexpr (GPTQ)
        {
            const f16* zt = m.zp_tab + (size_t)tile[t] * m.G * 16 + (lane & 15);
            const f16 z0 = zt[g[0] * 16], z1 = zt[g[1] * 16], z2 = zt[g[2] * 16], z3 = zt[g[3] * 16];
            R.zp[t][0] = (f16x2){z0, z1}; R.zp[t][1] = (f16x2){z2, z3};
        }
    }
}

// chunks {2 H, 2 H + 1} of both tiles -> W stage (fragment layout); the other half of dequant_super's work is dead code
template <int BITS, bool GPTQ, int H>
DEV void decode_half(const TileRaw& R, u8* w_stage, int wv, int lane)
{
    if (MF_KILL & 2) { if (lane == 99) *(u32*)(w_stage + wv * 64) = R.w[0][0] ^ R.w[1][0]; return; }
    #pragma unroll
    for (int t = 0; t < 2; t++)
    {
        ZC zc[4];
        if constexpr (GPTQ)
        {
            #pragma unroll
            for (int q = 0; q < 4; q++) zc[q] = make_zc((q & 1) ? R.zp[t][q >> 1].y : R.zp[t][q >> 1].x);
        }
        else
        {
            const ZC z = make_zc((f16)(float)(1 << (BITS - 1)));
            #pragma unroll
            for (int q = 0; q < 4; q++) zc[q] = z;
        }
        f16x2 p[16];
        dequant_super<BITS>(R.w[t], zc, p);
        #pragma unroll
        for (int qq = 0; qq < 2; qq++)
        {
            const int q = 2 * H + qq;
            const f16x2 s2 = h2_dup(qq ? R.sc[t][H].y : R.sc[t][H].x);
            const f16x2 b0 = p[4 * q + 0] * s2, b1 = p[4 * q + 1] * s2, b2 = p[4 * q + 2] * s2, b3 = p[4 * q + 3] * s2;
            const f16x8 b = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
            *(f16x8*)(w_stage + ((size_t)(((2 * wv + t) * 2 + qq) * 64 + lane)) * 16) = b;
        }
        MF_FENCE();             // one tile's decode temporaries at a time (register budget: the accumulators stay live)
    }
}

// The bit width is uniform per section; the switch sits around the load and the decode ONLY -- neither touches the
// accumulators, so the K loop exists once and the 128 accumulator registers never cross a control-flow merge (with one
// instantiation of the loop per width, or the multiply duplicated in two branches, hipcc spills hundreds of registers).
template <bool GPTQ>
DEV void load_raw_sw(int bits, TileRaw& R, const QMatDev& m, const u32* base, int s, u32 tile_stride, int chunk0, int nvalid,
                     const int (&tile)[2], const u16* cg_lds, int lane)
{
    switch (GPTQ ? 4 : bits)
    {
        case 4: load_raw<4, GPTQ>(R, m, base + (size_t)s * (64 * 4), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
        case 8: load_raw<8, GPTQ>(R, m, base + (size_t)s * (64 * 8), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
        case 6: load_raw<6, GPTQ>(R, m, base + (size_t)s * (64 * 6), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
        case 5: load_raw<5, GPTQ>(R, m, base + (size_t)s * (64 * 5), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
        case 3: load_raw<3, GPTQ>(R, m, base + (size_t)s * (64 * 3), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
        default: load_raw<2, GPTQ>(R, m, base + (size_t)s * (64 * 2), tile_stride, chunk0, nvalid, tile, cg_lds, lane); break;
    }
}
template <bool GPTQ, int H>
DEV void decode_half_sw(int bits, const TileRaw& R, u8* w_stage, int wv, int lane)
{
    switch (GPTQ ? 4 : bits)
    {
        case 4: decode_half<4, GPTQ, H>(R, w_stage, wv, lane); break;
        case 8: decode_half<8, GPTQ, H>(R, w_stage, wv, lane); break;
        case 6: decode_half<6, GPTQ, H>(R, w_stage, wv, lane); break;
        case 5: decode_half<5, GPTQ, H>(R, w_stage, wv, lane); break;
        case 3: decode_half<3, GPTQ, H>(R, w_stage, wv, lane); break;
        default: decode_half<2, GPTQ, H>(R, w_stage, wv, lane); break;
    }
}

// `nv` (0..2) chunks of 32 K rows from one stage: MT x 4 MFMAs per chunk
template <int MT, bool PIPE = false>
DEV void multiply_stage(const u8* x_stage, const u8* w_stage, int nv, int wm, int wn, int lane, f32x4 (&acc)[MT][4])
{
    const int i16 = lane & 15, j4 = lane >> 4, sw = (lane >> 1) & 7;
    if (MF_KILL & 4) return;
    if constexpr (!PIPE)
    {
        // (the kernels that decode inside the K loop have no registers to spare: fragments of one chunk at a time)
        #pragma unroll
        for (int kq = 0; kq < 2; kq++)
        {
            if (kq < nv)
            {
                f16x8 wf[4];
                #pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    wf[nt] = *(const f16x8*)(w_stage + ((size_t)(((wn * 4 + nt) * 2 + kq) * 64 + lane)) * 16);
                const u8* xr = x_stage + (size_t)(wm * MT * 16 + i16) * 128 + (((4 * kq + j4) ^ sw) * 16);
                #pragma unroll
                for (int mt = 0; mt < MT; mt++)
                {
                    const f16x8 xf = *(const f16x8*)(xr + mt * 16 * 128);
                    #pragma unroll
                    for (int nt = 0; nt < 4; nt++) acc[mt][nt] = mfma_16x16x32_f16(wf[nt], xf, acc[mt][nt]);
                }
            }
            MF_FENCE();
        }
        return;
    }
    // Round 5 (PIPE: the kernels whose weights were decoded by the pre-pass): the NEXT fragments are on their way while the current ones are multiplied -- the X fragment of row tile mt + 1 behind the
    // four MFMAs of row tile mt, the four W fragments of chunk 1 behind chunk 0's first row tiles.  Before, the compiler had every pair
    // of `ds_read_b128` directly in front of its eight MFMAs with `lgkmcnt(0)` between them (an exposed LDS round trip per 128 cycles
    // of MFMA: 61 % matrix-core busy at best); the 24 registers this needs were freed by the buffer-form stage fills (222 instead
    // of 256 + spills).
    auto w_frag = [&](int kq, int nt) -> f16x8 { return *(const f16x8*)(w_stage + ((size_t)(((wn * 4 + nt) * 2 + kq) * 64 + lane)) * 16); };
    auto x_frag = [&](int kq, int mt) -> f16x8 { return *(const f16x8*)(x_stage + (size_t)(wm * MT * 16 + i16 + mt * 16) * 128 + (((4 * kq + j4) ^ sw) * 16)); };
    if (nv <= 0) return;
    f16x8 wf[2][4], xf[2];
    #pragma unroll
    for (int nt = 0; nt < 4; nt++) wf[0][nt] = w_frag(0, nt);
    xf[0] = x_frag(0, 0);
    #pragma unroll
    for (int kq = 0; kq < 2; kq++)
    {
        if (kq < nv)
        {
            const bool next_chunk = kq == 0 && nv > 1;
            #pragma unroll
            for (int mt = 0; mt < MT; mt++)
            {
                // requests first: the next X fragment, and -- spread over the first row tiles -- the next chunk's W fragments
                if (mt + 1 < MT) xf[(mt + 1) & 1] = x_frag(kq, mt + 1);
                else if (next_chunk) xf[(mt + 1) & 1] = x_frag(1, 0);
                if (kq == 0 && mt < 4 && next_chunk) wf[1][mt] = w_frag(1, mt);
                MF_FENCE();
                #pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[mt][nt] = mfma_16x16x32_f16(wf[kq][nt], xf[mt & 1], acc[mt][nt]);
                MF_FENCE();
            }
        }
    }
}

// what the K walk needs besides the accumulators
struct MfCtx
{
    const QMatDev* m;
    u8* x_st[2]; u8* w_st[2];
    const u16* cg_lds;
    const f16* x_base;          // the staged activations (packed K order, row stride K): base of the buffer descriptor
    u32 x_wave;                 // byte offset of this wave's first X piece (wave-uniform)
    u32 x_off0, x_off1;         // the lane's BYTE offset inside an even / odd piece
    int x_u0;                   // the unit the lane holds in even pieces (odd pieces: ^ 4)
    int tile[2];
    int K, lane, wv, wm, wn;
};

// this wave's share of one K step of X (units < 4 nv of every row) -> stage, by LDS-DMA
template <int MT>
DEV void issue_x(const MfCtx& x, int k0, int nv, u8* stage)
{
    constexpr int PIECES = MT / 2;
    if (MF_KILL & 1) return;
    #pragma unroll
    for (int i = 0; i < PIECES; i++)
    {
        const int u = (i & 1) ? (x.x_u0 ^ 4) : x.x_u0;
        // (buffer form: the lane's offset is one of two registers kept for the whole kernel, everything else is scalar)
        if (u < 4 * nv) dma_buf_to_lds16_so(x.x_base, (i & 1) ? x.x_off1 : x.x_off0, x.x_wave + (u32)((i * 8) * x.K + k0) * 2u,
                                            stage + (size_t)(x.wv * PIECES + i) * 1024);
    }
}

// One bit-width section (QRun / QDesc): F super-chunks of 128 K rows from packed row k_base, the last with nvl valid chunks.
// Every super-chunk is two K steps of 64; ONE barrier per step: during step h the LDS-DMA of step h + 1's X tile is in
// flight into the other stage and the other stage's weights are decoded (chunks {2, 3} of this super-chunk during step 0,
// chunks {0, 1} of the next one during step 1; the packed words arrive one step before they are decoded).  The first
// super-chunk of a section starts cold (one exposed round trip per section).  Stage 0 may be refilled as soon as the barrier of step 1 has been passed, which is why the next section's
// prologue needs no barrier of its own.
template <bool GPTQ, int MT>
DEV void run_section(const MfCtx& x, int bits, const u32* base, u32 tile_stride, int F, int k_base, int nvl, f32x4 (&acc)[MT][4])
{
    TileRaw raw;
    issue_x<MT>(x, k_base, F == 1 ? min(2, nvl) : 2, x.x_st[0]);
    load_raw_sw<GPTQ>(bits, raw, *x.m, base, 0, tile_stride, k_base >> 5, F == 1 ? nvl : 4, x.tile, x.cg_lds, x.lane);
    decode_half_sw<GPTQ, 0>(bits, raw, x.w_st[0], x.wv, x.lane);
    for (int s = 0; s < F; s++)
    {
        const int k0 = k_base + s * SUPER_ROWS;
        const int nvalid = (s == F - 1) ? nvl : 4;
        const int nv0 = min(2, nvalid), nv1 = nvalid - nv0;
        const bool more = s + 1 < F;
        const int nvalid_next = (s + 1 == F - 1) ? nvl : 4;

        // ---- K step 0: multiply stage 0; stage 1 <- second half of X and of the decoded weights --------------------------
        wait_vmcnt_le<0>();
        block_sync();
        issue_x<MT>(x, k0 + 64, nv1, x.x_st[1]);
        decode_half_sw<GPTQ, 1>(bits, raw, x.w_st[1], x.wv, x.lane);
        if (more) load_raw_sw<GPTQ>(bits, raw, *x.m, base, s + 1, tile_stride, (k0 >> 5) + 4, nvalid_next, x.tile, x.cg_lds, x.lane);
        multiply_stage<MT>(x.x_st[0], x.w_st[0], nv0, x.wm, x.wn, x.lane, acc);

        // ---- K step 1: multiply stage 1; stage 0 <- first half of the next super-chunk; the two waves of a SIMD (w, w + 4)
        // decode on opposite sides of the multiply -------------------------------------------------------------------------
        wait_vmcnt_le<0>();
        block_sync();
        if (more) issue_x<MT>(x, k0 + SUPER_ROWS, min(2, nvalid_next), x.x_st[0]);
        if (more && x.wm != 0) decode_half_sw<GPTQ, 0>(bits, raw, x.w_st[0], x.wv, x.lane);
        multiply_stage<MT>(x.x_st[1], x.w_st[1], nv1, x.wm, x.wn, x.lane, acc);
        if (more && x.wm == 0) decode_half_sw<GPTQ, 0>(bits, raw, x.w_st[0], x.wv, x.lane);
    }
}

// ---- many rows: weights decoded ONCE per call (wfrag_kernel below) instead of once per workgroup ------------------------------
// With 16384 rows a column block's weights are decoded by 64 workgroups, and the decode is what a K step waits for: the kernel
// with the decode compiled out runs 31 % faster, with the multiply compiled out it still takes half the time
// (profiles/history/r03_prefill_kill.txt).  The pre-pass writes every K step's W stage image (32 KB: [16 tiles][2 chunks][64 lanes][16 B])
// to a scratch buffer; the GEMM then fills its W stages by LDS-DMA like its X stages -- no decode, no packed words, no
// scale tables in the K loop.

// this wave's eighth of one K step's decoded weights -> stage
DEV void issue_w(const u8* wfrag, u32 slot_off, u8* stage, int wv, int lane)
{
    #pragma unroll
    for (int i = 0; i < 4; i++) dma_buf_to_lds16_so(wfrag, (u32)lane * 16u, slot_off + (u32)((wv * 4 + i) * 1024), stage + (size_t)(wv * 4 + i) * 1024);
}

template <int MT>
DEV void run_section_pre(const MfCtx& x, const u8* wfrag, u32 slots, int F, int k_base, int nvl, f32x4 (&acc)[MT][4])
{
    // (slots = byte offset of the section's first W stage image inside the call's scratch: base + scalar offset + lane * 16)
    issue_x<MT>(x, k_base, F == 1 ? min(2, nvl) : 2, x.x_st[0]);
    issue_w(wfrag, slots, x.w_st[0], x.wv, x.lane);
    for (int s = 0; s < F; s++)
    {
        const int k0 = k_base + s * SUPER_ROWS;
        const int nvalid = (s == F - 1) ? nvl : 4;
        const int nv0 = min(2, nvalid), nv1 = nvalid - nv0;
        const bool more = s + 1 < F;
        const int nvalid_next = (s + 1 == F - 1) ? nvl : 4;
        const u32 sl = slots + (u32)(2 * s) * (u32)MF_W_STAGE;
        // K step 0: multiply stage 0; stage 1 <- the second half of this super-chunk
        wait_vmcnt_le<0>();
        block_sync();
        issue_x<MT>(x, k0 + 64, nv1, x.x_st[1]);
        issue_w(wfrag, sl + (u32)MF_W_STAGE, x.w_st[1], x.wv, x.lane);
        multiply_stage<MT, true>(x.x_st[0], x.w_st[0], nv0, x.wm, x.wn, x.lane, acc);
        // K step 1: multiply stage 1; stage 0 <- the first half of the next super-chunk
        wait_vmcnt_le<0>();
        block_sync();
        if (more)
        {
            issue_x<MT>(x, k0 + SUPER_ROWS, min(2, nvalid_next), x.x_st[0]);
            issue_w(wfrag, sl + 2u * (u32)MF_W_STAGE, x.w_st[0], x.wv, x.lane);
        }
        multiply_stage<MT, true>(x.x_st[1], x.w_st[1], nv1, x.wm, x.wn, x.lane, acc);
    }
}

// the pre-pass: workgroup (column block, super-chunk) decodes the 256 columns x 128 K rows into the two W stage images of its
// K steps.  Same loads, same decoders, same rounding as the in-kernel decode: the GEMM's results do not change by a bit.
template <bool GPTQ>
KERNEL void __launch_bounds__(MF_THREADS) wfrag_kernel(const QMatDev m, u8* out, int steps_per_block)
{
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int n_tiles = m.N / TILE_N;
    const int vn = bid_x();
    int sup = bid_y();                                        // super-chunk index over all sections, in the GEMM's K order
    const int n_items = m.n_runs > 0 ? m.n_runs : m.n_desc;
    int step0 = 0;
    for (int r = 0; r < n_items; r++)
    {
        const u32* base; u32 tile_stride; int F, k_base, bits, nvl;
        if (m.n_runs > 0)
        {
            const QRun& R = m.runs[r];
            base = (uniform((int)R.in_tail) ? m.tail : m.qw) + uniform(R.base_word);
            tile_stride = uniform(R.tile_stride); F = uniform((int)R.n_super); k_base = uniform((int)R.k_base);
            bits = uniform((int)R.bits); nvl = uniform((int)R.nvalid_last);
        }
        else
        {
            const QDesc* D = m.desc + r;
            base = (uniform((int)D->in_tail) ? m.tail : m.qw) + uniform(D->base_word);
            tile_stride = uniform(D->tile_stride); F = uniform((int)D->n_super); k_base = uniform((int)D->k_base);
            bits = uniform((int)D->bits); nvl = uniform((int)D->nvalid_last);
        }
        if (F <= 0) continue;
        if (sup >= F) { sup -= F; step0 += 2 * F; continue; }
        int tile[2];
        #pragma unroll
        for (int i = 0; i < 2; i++) tile[i] = min(vn * 16 + 2 * wv + i, n_tiles - 1);
        const int k0 = k_base + sup * SUPER_ROWS;
        const int nvalid = (sup == F - 1) ? nvl : 4;
        TileRaw raw;
        load_raw_sw<GPTQ>(bits, raw, m, base, sup, tile_stride, k0 >> 5, nvalid, tile, m.chunk_group, lane);
        u8* const slot = out + ((size_t)vn * steps_per_block + step0 + 2 * sup) * MF_W_STAGE;
        decode_half_sw<GPTQ, 0>(bits, raw, slot, wv, lane);
        decode_half_sw<GPTQ, 1>(bits, raw, slot + MF_W_STAGE, wv, lane);
        return;
    }
}

template <bool GPTQ, int MT, bool WPRE>
KERNEL void __launch_bounds__(MF_THREADS, 2) qgemm_mfma_kernel(const PrefillArgs args)
{
    // five separate LDS objects, not one dynamic block: hipcc then knows that the LDS-DMA into one X stage does not alias the
    // reads of the other stage / the weight stages / the group map, and does not put an s_waitcnt vmcnt(0) in front of them
    SHARED __attribute__((aligned(16))) u8 lds_x0[MF_X_STAGE(MT)];
    SHARED __attribute__((aligned(16))) u8 lds_x1[MF_X_STAGE(MT)];
    SHARED __attribute__((aligned(16))) u8 lds_w0[MF_W_STAGE];
    SHARED __attribute__((aligned(16))) u8 lds_w1[MF_W_STAGE];
    SHARED __attribute__((aligned(16))) u16 cg_lds[MF_MAX_CHUNKS];
    constexpr int BM = 32 * MT;
    constexpr int PIECES = MT / 2;                          // 1 KB LDS-DMA pieces of the X stage per wave
    const QMatDev& m = args.m;
    const int t = tid(), lane = lane_id();
    const int wv = uniform(wave_id());
    const int wm = wv >> 2, wn = wv & 3;
    const int K = m.K, N = m.N, M = args.M;
    const int n_tiles = N / TILE_N;

    // ---- workgroup -> tile: consecutive ids of one XCD (id % 8) walk a band of 4 row blocks column by column ------------------
    const int nb_n = (N + MF_BN - 1) / MF_BN, nb_m = (M + BM - 1) / BM;
    int vm, vn;
    {
        const int nblk = nb_n * nb_m;
        const int b = bid_x(), xcd = b & 7, i = b >> 3, q = nblk >> 3, r = nblk & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        const int band = v / (4 * nb_n), idx = v - band * 4 * nb_n;
        const int bh = min(4, nb_m - band * 4);
        vn = idx / bh; vm = band * 4 + (idx - vn * bh);
    }
    const int m0 = uniform(vm * BM), n0 = uniform(vn * MF_BN);

    MfCtx x;
    x.m = &m; x.K = K; x.lane = lane; x.wv = wv; x.wm = wm; x.wn = wn;
    x.x_st[0] = lds_x0; x.x_st[1] = lds_x1;
    x.w_st[0] = lds_w0; x.w_st[1] = lds_w1;
    x.cg_lds = cg_lds;
    for (int i = t; i < (K >> 5); i += MF_THREADS) cg_lds[i] = m.chunk_group[i];

    // X stage: piece p (1 KB) holds rows 8 p .. 8 p + 7, the lane's slot is (row = lane >> 3, position = lane & 7) and holds
    // unit u = position ^ ((row >> 1) & 7) of that row.  The staged activations are padded to whole row blocks by the host
    // driver (no row clamp), a wave owns PIECES (even) consecutive pieces, so (row >> 1) & 7 = (4 (piece & 1) + (lane >> 4)) & 7
    // and a piece's source is a wave-uniform base + one of two per-lane offsets.
    {
        const int x_pos = lane & 7, x_r8 = lane >> 3;
        x.x_u0 = x_pos ^ (lane >> 4);
        x.x_off0 = (u32)(x_r8 * K + x.x_u0 * 8) * 2u;
        x.x_off1 = (u32)(x_r8 * K + (x.x_u0 ^ 4) * 8) * 2u;
        x.x_base = args.a;
        x.x_wave = (u32)((size_t)(m0 + wv * PIECES * 8) * K * 2);              // (the host keeps rows x K x 2 below 2 GB: run_prefill)
    }
    #pragma unroll
    for (int i = 0; i < 2; i++) x.tile[i] = min((n0 >> 4) + 2 * wv + i, n_tiles - 1);   // partial last block column: computed, never stored

    f32x4 acc[MT][4];
    #pragma unroll
    for (int mt = 0; mt < MT; mt++)
        #pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    block_sync();                                            // chunk -> group map is in LDS
    const int n_items = m.n_runs > 0 ? m.n_runs : m.n_desc;
    int step0 = 0;                                           // (WPRE) first K step of the section inside this column block's slots
    for (int r = 0; r < n_items; r++)
    {
        const u32* base; u32 tile_stride; int F, k_base, bits, nvl;
        if (m.n_runs > 0)
        {
            const QRun& R = m.runs[r];
            base = (uniform((int)R.in_tail) ? m.tail : m.qw) + uniform(R.base_word);
            tile_stride = uniform(R.tile_stride); F = uniform((int)R.n_super); k_base = uniform((int)R.k_base);
            bits = uniform((int)R.bits); nvl = uniform((int)R.nvalid_last);
        }
        else
        {
            const QDesc* D = m.desc + r;
            base = (uniform((int)D->in_tail) ? m.tail : m.qw) + uniform(D->base_word);
            tile_stride = uniform(D->tile_stride); F = uniform((int)D->n_super); k_base = uniform((int)D->k_base);
            bits = uniform((int)D->bits); nvl = uniform((int)D->nvalid_last);
        }
        if (F <= 0) continue;
        if constexpr (WPRE)
        {
            run_section_pre<MT>(x, args.wfrag, (u32)(((size_t)vn * args.wfrag_steps + step0) * MF_W_STAGE), F, k_base, nvl, acc);
            step0 += 2 * F;
        }
        else run_section<GPTQ, MT>(x, bits, base, tile_stride, F, k_base, nvl, acc);
    }

    // ---- epilogue: lane (i = lane & 15, j = lane >> 4) holds C[row i of the row tile][columns 4 j .. 4 j + 3 of the column tile]
    const int i16 = lane & 15, j4 = lane >> 4;
    const bool vec = !args.c_invperm && !(args.ldc & 3) && !(((size_t)args.c) & 7);
    #pragma unroll
    for (int nt = 0; nt < 4; nt++)
    {
        const int n = n0 + wn * 64 + nt * 16 + 4 * j4;
        if (n >= N) continue;
        float bias[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m.bias)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++) bias[e] = (float)m.bias[n + e];
        }
        #pragma unroll
        for (int mt = 0; mt < MT; mt++)
        {
            const int row = m0 + wm * MT * 16 + mt * 16 + i16;
            if (row >= M) continue;
            f16* cp = args.c + (size_t)row * args.ldc;
            if (vec)
            {
                f16x4 v;
                if (args.c_mode == C_ACCUM)
                {
                    const f16x4 old = *(const f16x4*)(cp + n);
                    #pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (f16)(acc[mt][nt][e] + bias[e] + (float)old[e]);
                }
                else
                {
                    #pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (f16)(acc[mt][nt][e] + bias[e]);
                }
                *(f16x4*)(cp + n) = v;
            }
            else
            {
                #pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const int nn = args.c_invperm ? (int)args.c_invperm[n + e] : n + e;
                    float v = acc[mt][nt][e] + bias[e];
                    if (args.c_mode == C_ACCUM) v += (float)cp[nn];
                    cp[nn] = (f16)v;
                }
            }
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------

// the variant the last call of this process took (exl2_prefill_route_info: tests assert the route they forced)
static int g_last_route[4] = {0, 0, 0, 0};                 // rows, tile rows / 32 (MT), weights pre-decoded (wfrag_kernel) 0 / 1, calls

template <bool GPTQ, int MT, bool WPRE = false>
static int launch_one(const PrefillArgs& p, void* stream)
{
    g_last_route[0] = p.M; g_last_route[1] = MT; g_last_route[2] = WPRE ? 1 : 0; g_last_route[3]++;
    const int nb_n = (p.m.N + MF_BN - 1) / MF_BN, nb_m = (p.M + 32 * MT - 1) / (32 * MT);
    EXL2_REQUIRE((p.m.K >> 5) <= MF_MAX_CHUNKS, "q_gemm (prefill): K = %d too large for the group map in LDS", p.m.K);
    const size_t lds = 0;                               // all LDS is static
    if (getenv("EXL2_PREFILL_TRACE")) fprintf(stderr, "[qgemm_mfma] M=%d K=%d N=%d MT=%d gptq=%d\n", p.M, p.m.K, p.m.N, MT, (int)GPTQ);
    LAUNCH((qgemm_mfma_kernel<GPTQ, MT, WPRE>), dim3((unsigned)(nb_n * nb_m)), dim3(MF_THREADS), lds, stream, p);
    return EXL2_OK;
}

int qgemm_mfma_launch(const PrefillArgs& p, bool gptq, void* stream)
{
    // tile height: 256 rows unless that leaves the 256 CUs with a badly quantized number of rounds
    // (rounds x rows per tile is the cost; the 128-row tile pays ~10 % for decoding the same weights for half the rows)
    int cus = 256;
    {
        hipDeviceProp_t prop;
        static int cached[EXL2_MAX_DEVICES] = {0};
        const int dev = exl2_current_device();
        if (!cached[dev]) cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus = cached[dev];
    }
    const long nb_n = (p.m.N + MF_BN - 1) / MF_BN;
    const long r8 = (nb_n * ((p.M + 255) / 256) + cus - 1) / cus, r4 = (nb_n * ((p.M + 127) / 128) + cus - 1) / cus;
    int mt = (r8 * 20 <= r4 * 11) ? 8 : 4;
    if (const char* e = getenv("EXL2_PREFILL_MT")) { const int v = atoi(e); if (v == 4 || v == 8) mt = v; }
    // many rows: decode the weights once per call into fragment images (wfrag_kernel), multiply from those
    int wpre_min = MF_WPRE_MIN_ROWS;                          // (read per call: tests force both routes)
    if (const char* e = getenv("EXL2_PREFILL_WPRE_MIN_ROWS")) wpre_min = atoi(e);
    if (wpre_min > 0 && p.M >= wpre_min)
    {
        int supers = 0;
        const int n_items = p.m.n_runs > 0 ? p.m.n_runs : 0;
        for (int r = 0; r < n_items; r++) supers += p.m.runs[r].n_super > 0 ? p.m.runs[r].n_super : 0;
        if (supers > 0 && supers <= 65535)
        {
            // the fragment images are K * N * 2 bytes of per-(device, stream) scratch kept until exl2_release_scratch (470 MB for
            // a 70B gate / up matrix, 2 GB for an 8192 x 128256 head): above a cap, or when the allocation fails, the call
            // decodes inside the GEMM instead (no scratch) -- slower, never an error a caller did not have before this route
            f16* buf = nullptr;
            const size_t bytes = (size_t)nb_n * (size_t)(2 * supers) * MF_W_STAGE;
            size_t cap = (size_t)1 << 30;
            if (const char* e = getenv("EXL2_PREFILL_WPRE_MAX_BYTES")) cap = (size_t)strtoull(e, nullptr, 10);
            if (cap >= ((size_t)1 << 31)) cap = ((size_t)1 << 31) - 1;      // (the W stages are filled by buffer loads: a 2 GB window)
            const bool fits = bytes <= cap && prefill_scratch(bytes, stream, 1, &buf) == EXL2_OK;
            if (!fits) { exl2_set_error("%s", ""); buf = nullptr; }
            if (buf)
            {
            if (gptq) LAUNCH((wfrag_kernel<true>), dim3((unsigned)nb_n, (unsigned)supers), dim3(MF_THREADS), 0, stream, p.m, (u8*)buf, 2 * supers);
            else      LAUNCH((wfrag_kernel<false>), dim3((unsigned)nb_n, (unsigned)supers), dim3(MF_THREADS), 0, stream, p.m, (u8*)buf, 2 * supers);
            PrefillArgs q = p;
            q.wfrag = (const u8*)buf; q.wfrag_steps = 2 * supers;
            return mt == 8 ? launch_one<false, 8, true>(q, stream) : launch_one<false, 4, true>(q, stream);
            }
        }
    }
    if (gptq) return mt == 8 ? launch_one<true, 8>(p, stream) : launch_one<true, 4>(p, stream);
    return mt == 8 ? launch_one<false, 8>(p, stream) : launch_one<false, 4>(p, stream);
}

extern "C" int exl2_prefill_route_info(int* out4)
{
    if (out4) for (int i = 0; i < 4; i++) out4[i] = g_last_route[i];
    return EXL2_OK;
}
