// qgemm_prefill.hip -- prefill-shaped q_matrix x fp16 GEMM on the matrix cores, dequantizing INTO the GEMM.
//
// Replaces the reference's M > 32 path: reconstruct_kernel (cuda/q_matrix.cu:328-553) writes the whole fp16 [K, N] matrix
// to HBM, then cuBLAS / hipBLAS Hgemm reads it back (cuda/q_gemm.cu:243-263) -- 2 x 2 K N bytes of traffic per call
// and a library GEMM that knows nothing about the packing.  Here the packed weights are the B operand:
//   * the tile16 layout (qlayout.h) stores, for every 128-row super-chunk of a 16-column tile, exactly the four
//     v_mfma_f32_16x16x32_f16 B fragments of its four 32-row chunks, lane by lane.  A wave loads 64 x b dwords, decodes
//     them in registers (same decoders and the same half(q - zero) * half(scale) rounding as reconstruct -> the weights
//     that enter the MFMA are bit-identical to reconstruct()'s) and feeds the matrix core; B never touches LDS or HBM
//     as fp16;
//   * the activations are made "packed-K ordered" once per call by a row pre-pass (stage_rows_kernel: act-order gather
//     through q_perm, fused with RMSNorm or act(gate) * up where the caller has them), so the GEMM reads A tiles as
//     contiguous 16-byte units -- asynchronous LDS-DMA with an XOR swizzle applied on the GLOBAL side (lane addresses are
//     free), conflict-free ds_read_b128 on the other side;
//   * workgroup = 8 waves (4 x 2), block tile 256 x 256, wave tile 64 x 128 (32 accumulator tiles = 128 VGPRs of the 256
//     a wave may use at 2 waves per SIMD), K step = one super-chunk (<= 128 rows); fp32 accumulation, bias / residual in
//     the epilogue.
// MFMA-bound by design: per K step a wave issues 128 MFMAs against ~500 VALU ops of decode.
#include "qgemm_prefill.h"
#include "errors.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <mutex>

#ifndef PF_BN
#define PF_BN 128
#endif
#define PF_QPT (4 * PF_BN / (PF_WAVES_M * PF_WAVES_N * 64))   // scale chunks staged per thread
#ifndef PF_WAVES_M
#define PF_WAVES_M 2
#endif
#define PF_BM (PF_WAVES_M * 64)
#define PF_WAVES_N 2
#define PF_CT (PF_BN / PF_WAVES_N / 16)     // 16-column tiles per wave (8)
#define PF_THREADS (PF_WAVES_M * PF_WAVES_N * 64)
#define PF_A_BYTES (PF_BM * 256)                 // PF_BM rows x 128 halves
#define PF_SC_BYTES (4 * PF_BN * 2)              // [chunk][column] scales of the current step
#define PF_BUF_BYTES (PF_A_BYTES + 2 * PF_SC_BYTES)   // one stage: A tile + scales + zero points
#define PF_TAB_OFF (2 * PF_BUF_BYTES)                 // chunk -> group map and (EXL2) per-group scale maxima, loaded once
#define PF_LDS_BYTES(K, G) (PF_TAB_OFF + ((((K) >> 5) * 2 + 15) & ~15) + (((G) * 2 + 15) & ~15))

// scale (and GPTQ zero point) of column n for the (up to 4) chunks of one step, as the fp16 values reconstruct() uses
// cg_lds / smax_lds: LDS copies of chunk_group and (EXL2) scale_src made at kernel start -- one global round trip per
// step (the scale code word) instead of a chain of three
template <bool GPTQ>
DEV void step_scales(const QMatDev& m, const u16* cg_lds, const f16* smax_lds, int chunk0, int nvalid, int n, int q0, f16* sc, f16* zp)
{
    #pragma unroll
    for (int i = 0; i < PF_QPT; i++)
    {
        const int q = q0 + i;
        const int g = cg_lds[q < nvalid ? chunk0 + q : chunk0];
        const u32 word = m.q_scale[(size_t)g * (m.N >> 3) + (n >> 3)];
        const int nib = (word >> (4 * (n & 7))) & 15;
        if constexpr (GPTQ) { sc[i] = m.scale_src[(size_t)g * m.N + n]; zp[i] = (f16)(float)(nib + 1); }
        else                { sc[i] = (f16)(float)((nib + 1) * (nib + 1)) * smax_lds[g]; zp[i] = (f16)0.0f; }
    }
}

template <int BITS, bool GPTQ>
DEV void decode_tile(const LaneWords<BITS>& lw, const f16* sc_lds, const f16* zp_lds, int col, f16x8 (&b)[4])
{
    ZC zc[4];
    if constexpr (GPTQ)
    {
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = make_zc(zp_lds[q * PF_BN + col]);
    }
    else
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
        const f16x2 s2 = h2_dup(sc_lds[q * PF_BN + col]);
        const f16x2 b0 = p[4 * q + 0] * s2, b1 = p[4 * q + 1] * s2, b2 = p[4 * q + 2] * s2, b3 = p[4 * q + 3] * s2;
        b[q] = (f16x8){b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
    }
}

// A wave's PF_CT column tiles are processed in batches of CB tiles through two alternating register sets: the packed
// words of batch i + 1 are in flight while batch i decodes and multiplies (CB = 4 up to 4 bits, 2 above: 16 VGPRs a set).
template <int BITS> struct Batch { static constexpr int CB = 2; static constexpr int NB = PF_CT / CB; };
static_assert(PF_CT == 4 || PF_CT == 8, "wave tile: 4 or 8 column tiles");

template <int BITS, int CB>
DEV void load_tiles(LaneWords<BITS> (&w)[CB], const u32* base, u32 tile_stride, int tile0, int n_tiles, int s, int lane)
{
    #pragma unroll
    for (int ct = 0; ct < CB; ct++)
    {
        int tile = tile0 + ct;
        if (tile >= n_tiles) tile = n_tiles - 1;                    // partial last block column: computed, never stored
        load_lane_words<BITS>(base + (size_t)tile * tile_stride + (size_t)s * (64 * BITS), lane, w[ct]);
    }
}

template <int BITS, bool GPTQ, int CB, int CT0>
DEV void multiply_tiles(const LaneWords<BITS> (&w)[CB], const u8* a_lds, const f16* sc_lds, const f16* zp_lds,
                        int wm, int wn, int nvalid, int lane, f32x4 (&acc)[4][PF_CT])
{
    const int i16 = lane & 15, j4 = lane >> 4;
    #pragma unroll
    for (int c = 0; c < CB; c++)
    {
        const int ct = CT0 + c;
        const int col_local = (wn * PF_CT + ct) * 16 + i16;         // this lane's column inside the block tile
        f16x8 b[4];
        decode_tile<BITS, GPTQ>(w[c], sc_lds, zp_lds, col_local, b);
        #pragma unroll
        for (int rt = 0; rt < 4; rt++)
        {
            const int row = wm * 64 + rt * 16 + i16;                // A fragment: lane (i, j) holds row i, k-slot j
            #pragma unroll
            for (int q = 0; q < 4; q++)
            {
                if (q < nvalid)
                {
                    const int p = (4 * q + j4) ^ (row & 15);
                    const f16x8 a = *(const f16x8*)(a_lds + (size_t)row * 256 + p * 16);
                    acc[rt][ct] = mfma_16x16x32_f16(a, b[q], acc[rt][ct]);
                }
            }
        }
        sched_fence();          // keep the decode temporaries of different tiles from overlapping (register budget)
    }
}

struct StepCtx
{
    const QMatDev* m; const f16* a; int M, K, m0, n0, n_tiles;
    u8* a_lds; f16* sc_lds; f16* zp_lds;
    int t, lane, wv, wm, wn, sc_col, sc_q0, sc_n;
    const u16* cg_lds; const f16* smax_lds;
};

// one K step (one super-chunk of `nvalid` 32-row chunks starting at packed row k0) for the whole block tile
template <int BITS, bool GPTQ>
DEV void k_step(const StepCtx& x, const u32* base, u32 tile_stride, int s, int k0, int nvalid, f32x4 (&acc)[4][PF_CT])
{
    constexpr int CB = Batch<BITS>::CB, NB = Batch<BITS>::NB;
    const int tile0 = (x.n0 >> 4) + x.wn * PF_CT;
    LaneWords<BITS> w0[CB], w1[CB];
    load_tiles<BITS, CB>(w0, base, tile_stride, tile0, x.n_tiles, s, x.lane);

    // stage A [256 rows x 32 nvalid K] (swizzled) and this step's scales
    const int chunk0 = k0 >> 5;
    const int units_row = nvalid * 4;                               // 16-byte units of A per row in this step
    f16 sc2[PF_QPT], zp2[PF_QPT];
    step_scales<GPTQ>(*x.m, x.cg_lds, x.smax_lds, chunk0, nvalid, x.sc_n, x.sc_q0, sc2, zp2);
    for (int base_u = x.wv * 64; base_u < PF_BM * 16; base_u += (PF_THREADS / 64) * 64)
    {
        const int slot = base_u + x.lane;                           // LDS position: row = slot >> 4, p = slot & 15
        const int row = slot >> 4, p = slot & 15;
        const int u = p ^ (row & 15);                               // which 16-byte unit of the row lives there
        const int grow = min(x.m0 + row, x.M - 1);
        if (u < units_row)
            dma_to_lds16(x.a + (size_t)grow * x.K + k0 + u * 8, x.a_lds + (size_t)base_u * 16);
    }
    #pragma unroll
    for (int i = 0; i < PF_QPT; i++)
    {
        x.sc_lds[(x.sc_q0 + i) * PF_BN + x.sc_col] = sc2[i];
        if constexpr (GPTQ) x.zp_lds[(x.sc_q0 + i) * PF_BN + x.sc_col] = zp2[i];
    }
    wait_vmcnt_le<0>();
    block_sync();

    load_tiles<BITS, CB>(w1, base, tile_stride, tile0 + CB, x.n_tiles, s, x.lane);
    multiply_tiles<BITS, GPTQ, CB, 0>(w0, x.a_lds, x.sc_lds, x.zp_lds, x.wm, x.wn, nvalid, x.lane, acc);
    if constexpr (NB > 2) load_tiles<BITS, CB>(w0, base, tile_stride, tile0 + 2 * CB, x.n_tiles, s, x.lane);
    multiply_tiles<BITS, GPTQ, CB, CB>(w1, x.a_lds, x.sc_lds, x.zp_lds, x.wm, x.wn, nvalid, x.lane, acc);
    if constexpr (NB > 2)
    {
        load_tiles<BITS, CB>(w1, base, tile_stride, tile0 + 3 * CB, x.n_tiles, s, x.lane);
        multiply_tiles<BITS, GPTQ, CB, 2 * CB>(w0, x.a_lds, x.sc_lds, x.zp_lds, x.wm, x.wn, nvalid, x.lane, acc);
        multiply_tiles<BITS, GPTQ, CB, 3 * CB>(w1, x.a_lds, x.sc_lds, x.zp_lds, x.wm, x.wn, nvalid, x.lane, acc);
    }
    block_sync();
}

// CB column tiles decoded to B fragments in registers, then every A fragment of the wave's 64 rows is read ONCE from LDS
// and used for all CB tiles (multiply_tiles re-reads it per tile: 4x the ds_read_b128 traffic, LDS-bound).
template <int BITS, bool GPTQ, int CB, int CT0>
DEV void multiply_batch(const LaneWords<BITS> (&w)[CB], const u8* a_lds, const f16* sc_lds, const f16* zp_lds,
                        int wm, int wn, int lane, f32x4 (&acc)[4][PF_CT])
{
    const int i16 = lane & 15, j4 = lane >> 4;
    f16x8 b[CB][4];
    #pragma unroll
    for (int c = 0; c < CB; c++)
        decode_tile<BITS, GPTQ>(w[c], sc_lds, zp_lds, (wn * PF_CT + CT0 + c) * 16 + i16, b[c]);
    #pragma unroll
    for (int rt = 0; rt < 4; rt++)
    {
        const int row = wm * 64 + rt * 16 + i16;
        f16x8 a[4];
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int p = (4 * q + j4) ^ (row & 15);
            a[q] = *(const f16x8*)(a_lds + (size_t)row * 256 + p * 16);
        }
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            #pragma unroll
            for (int c = 0; c < CB; c++) acc[rt][CT0 + c] = mfma_16x16x32_f16(a[q], b[c][q], acc[rt][CT0 + c]);
        }
        sched_fence();          // one row tile's A fragments live at a time (register budget: 128 accumulators)
    }
}

// All F full super-chunks of one run, software-pipelined: while step s multiplies, the A tile and the scales of step
// s + 1 are in flight into the other LDS stage and the first batch of its packed weights into registers; one barrier
// per step.
template <int BITS, bool GPTQ>
DEV void run_pipelined(const StepCtx& x, u8* smem, const u32* base, u32 tile_stride, int F, int k_base,
                       f32x4 (&acc)[4][PF_CT])
{
    constexpr int CB = Batch<BITS>::CB, NB = Batch<BITS>::NB;
    const int tile0 = (x.n0 >> 4) + x.wn * PF_CT;
    u8* a_buf[2] = {smem, smem + PF_BUF_BYTES};
    auto sc_of = [&](int b) { return (f16*)(a_buf[b] + PF_A_BYTES); };
    auto zp_of = [&](int b) { return (f16*)(a_buf[b] + PF_A_BYTES + PF_SC_BYTES); };
    auto issue_a = [&](int s, int b)
    {
        const int k0 = k_base + s * SUPER_ROWS;
        for (int base_u = x.wv * 64; base_u < PF_BM * 16; base_u += (PF_THREADS / 64) * 64)
        {
            const int slot = base_u + x.lane;
            const int row = slot >> 4, p = slot & 15;
            const int u = p ^ (row & 15);
            const int grow = min(x.m0 + row, x.M - 1);
            dma_to_lds16(x.a + (size_t)grow * x.K + k0 + u * 8, a_buf[b] + (size_t)base_u * 16);
        }
    };
    LaneWords<BITS> w0[CB], w1[CB];
    f16 sc2[PF_QPT], zp2[PF_QPT];
    issue_a(0, 0);
    step_scales<GPTQ>(*x.m, x.cg_lds, x.smax_lds, k_base >> 5, 4, x.sc_n, x.sc_q0, sc2, zp2);
    load_tiles<BITS, CB>(w0, base, tile_stride, tile0, x.n_tiles, 0, x.lane);
    for (int s = 0; s < F; s++)
    {
        const int b = s & 1;
        f16* scl = sc_of(b); f16* zpl = zp_of(b);
        #pragma unroll
        for (int i = 0; i < PF_QPT; i++)
        {
            scl[(x.sc_q0 + i) * PF_BN + x.sc_col] = sc2[i];
            if constexpr (GPTQ) zpl[(x.sc_q0 + i) * PF_BN + x.sc_col] = zp2[i];
        }
        wait_vmcnt_le<0>();
        block_sync();
        if (s + 1 < F)
        {
            issue_a(s + 1, b ^ 1);
            step_scales<GPTQ>(*x.m, x.cg_lds, x.smax_lds, (k_base >> 5) + 4 * (s + 1), 4, x.sc_n, x.sc_q0, sc2, zp2);
        }
        load_tiles<BITS, CB>(w1, base, tile_stride, tile0 + CB, x.n_tiles, s, x.lane);
        multiply_batch<BITS, GPTQ, CB, 0>(w0, a_buf[b], scl, zpl, x.wm, x.wn, x.lane, acc);
        if constexpr (NB == 2)
        {
            if (s + 1 < F) load_tiles<BITS, CB>(w0, base, tile_stride, tile0, x.n_tiles, s + 1, x.lane);
            multiply_batch<BITS, GPTQ, CB, CB>(w1, a_buf[b], scl, zpl, x.wm, x.wn, x.lane, acc);
        }
        else
        {
            load_tiles<BITS, CB>(w0, base, tile_stride, tile0 + 2 * CB, x.n_tiles, s, x.lane);
            multiply_batch<BITS, GPTQ, CB, CB>(w1, a_buf[b], scl, zpl, x.wm, x.wn, x.lane, acc);
            load_tiles<BITS, CB>(w1, base, tile_stride, tile0 + 3 * CB, x.n_tiles, s, x.lane);
            multiply_batch<BITS, GPTQ, CB, 2 * CB>(w0, a_buf[b], scl, zpl, x.wm, x.wn, x.lane, acc);
            if (s + 1 < F) load_tiles<BITS, CB>(w0, base, tile_stride, tile0, x.n_tiles, s + 1, x.lane);
            multiply_batch<BITS, GPTQ, CB, 3 * CB>(w1, a_buf[b], scl, zpl, x.wm, x.wn, x.lane, acc);
        }
    }
    block_sync();                                            // stage buffers are free again for whoever comes next
}

template <bool GPTQ>
KERNEL void __launch_bounds__(PF_THREADS, 2) qgemm_prefill_kernel(const PrefillArgs args)
{
    DYN_SMEM(smem);
    const QMatDev& m = args.m;
    const int t = tid();
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int wm = wv / PF_WAVES_N, wn = wv - wm * PF_WAVES_N;
    const int n0 = bid_x() * PF_BN;
    const int m0 = bid_y() * PF_BM;
    const int K = m.K;
    const int n_tiles = m.N / TILE_N;

    u8* a_lds = (u8*)smem;
    f16* sc_lds = (f16*)(smem + PF_A_BYTES);
    f16* zp_lds = (f16*)(smem + PF_A_BYTES + PF_SC_BYTES);

    f32x4 acc[4][PF_CT];
    #pragma unroll
    for (int rt = 0; rt < 4; rt++)
        #pragma unroll
        for (int ct = 0; ct < PF_CT; ct++) acc[rt][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // scale staging: thread t (of 512) owns column (t & 255) and chunks {2 (t >> 8), +1} of the step
    const int sc_col = t % PF_BN;
    const int sc_q0 = (t / PF_BN) * PF_QPT;
    const int sc_n = min(n0 + sc_col, m.N - 1);

    const int i16 = lane & 15, j4 = lane >> 4;
    StepCtx x;
    x.m = &m; x.a = args.a; x.M = args.M; x.K = K; x.m0 = m0; x.n0 = n0; x.n_tiles = n_tiles;
    x.a_lds = a_lds; x.sc_lds = sc_lds; x.zp_lds = zp_lds;
    x.t = t; x.lane = lane; x.wv = wv; x.wm = wm; x.wn = wn; x.sc_col = sc_col; x.sc_q0 = sc_q0; x.sc_n = sc_n;
    {
        u16* cg = (u16*)(smem + PF_TAB_OFF);
        f16* sm = (f16*)(smem + PF_TAB_OFF + (((K >> 5) * 2 + 15) & ~15));
        for (int i = t; i < (K >> 5); i += PF_THREADS) cg[i] = m.chunk_group[i];
        if constexpr (!GPTQ) for (int i = t; i < m.G; i += PF_THREADS) sm[i] = m.scale_src[i];
        x.cg_lds = cg; x.smax_lds = sm;
        block_sync();
    }
    if (m.n_runs > 0)
    {
        // one contiguous stream per bit-width section (QRun): full runs are software-pipelined, a partial last
        // super-chunk (its own run, padded side buffer) takes the single-step path
        for (int ri = 0; ri < m.n_runs; ri++)
        {
            const QRun& run = m.runs[ri];
            const int bits = uniform((int)run.bits);
            const u32* base = (uniform((int)run.in_tail) ? m.tail : m.qw) + uniform(run.base_word);
            const u32 tile_stride = uniform(run.tile_stride);
            const int F = uniform((int)run.n_super), k_base = uniform((int)run.k_base);
            const int nvl = uniform((int)run.nvalid_last);
            if (nvl == 4)
            {
                switch (GPTQ ? 4 : bits)
                {
                    case 4: run_pipelined<4, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                    case 8: run_pipelined<8, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                    case 6: run_pipelined<6, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                    case 5: run_pipelined<5, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                    case 3: run_pipelined<3, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                    default: run_pipelined<2, GPTQ>(x, smem, base, tile_stride, F, k_base, acc); break;
                }
            }
            else
            {
                switch (GPTQ ? 4 : bits)
                {
                    case 4: k_step<4, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                    case 8: k_step<8, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                    case 6: k_step<6, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                    case 5: k_step<5, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                    case 3: k_step<3, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                    default: k_step<2, GPTQ>(x, base, tile_stride, 0, k_base, nvl, acc); break;
                }
            }
        }
    }
    else
    for (int d = 0; d < m.n_desc; d++)
    {
        const QDesc* dp = m.desc + d;
        const int n_super = uniform((int)dp->n_super);
        const int bits = uniform((int)dp->bits);
        const u32* base = (uniform((int)dp->in_tail) ? m.tail : m.qw) + uniform(dp->base_word);
        const u32 tile_stride = uniform(dp->tile_stride);
        const int k_base = uniform((int)dp->k_base);
        const int nvalid_last = uniform((int)dp->nvalid_last);
        for (int s = 0; s < n_super; s++)
        {
            const int nvalid = (s == n_super - 1) ? nvalid_last : 4;
            const int k0 = k_base + s * SUPER_ROWS;
            switch (GPTQ ? 4 : bits)
            {
                case 4: k_step<4, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
                case 8: k_step<8, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
                case 6: k_step<6, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
                case 5: k_step<5, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
                case 3: k_step<3, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
                default: k_step<2, GPTQ>(x, base, tile_stride, s, k0, nvalid, acc); break;
            }
        }
    }

    // ---- epilogue: D fragment lane (c = l & 15, j) holds rows 4 j .. 4 j + 3 of column c ---------------------------------
    #pragma unroll
    for (int ct = 0; ct < PF_CT; ct++)
    {
        const int n = n0 + (wn * PF_CT + ct) * 16 + i16;
        if (n >= m.N) continue;
        const float bias = m.bias ? (float)m.bias[n] : 0.0f;
        const int nn = args.c_invperm ? (int)args.c_invperm[n] : n;
        #pragma unroll
        for (int rt = 0; rt < 4; rt++)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                const int row = m0 + wm * 64 + rt * 16 + j4 * 4 + e;
                if (row < args.M)
                {
                    f16* cp = args.c + (size_t)row * args.ldc + nn;
                    float v = acc[rt][ct][e] + bias;
                    if (args.c_mode == C_ACCUM) v += (float)*cp;
                    *cp = (f16)v;
                }
            }
        }
    }
}

// ---- row pre-pass: out[r, k'] = f(a[r, perm[k']]) (f: identity | RMSNorm | act | act(gate) * up) -----------------------
// One workgroup per row; the row is read once (contiguous), normalised / activated, and written in the matrix' packed
// K order.  Numerics = stage_rows (qgemv_common.h) = rms_norm.cu / q_mlp_activation.cuh.

struct RowStageArgs
{
    const f16* a; const f16* a2; const f16* norm_w; const u16* perm; f16* out;
    int K, lda, mode; float eps;
    int parts;               // stage_rows_kernel, few rows: blockIdx.y = which slice of the OUTPUT row this workgroup writes (every
                             // workgroup of a row transforms the whole row -- a few KB from the L2 -- so that rows x parts fill the chip)
};

DEV void stage_rows_body(const RowStageArgs& s, char* smem, int slice = 0, int slices = 1)
{
    const int r = bid_x();
    const int t = tid(), nt = nthreads(), lane = lane_id(), wv = wave_id(), nw = nt >> 6;
    f16* row = (f16*)smem;                                   // [K] transformed values in ORIGINAL order
    float* part = (float*)(smem + (size_t)s.K * 2);
    const f16* ar = s.a + (size_t)r * s.lda;
    const f16* br = s.a2 ? s.a2 + (size_t)r * s.lda : nullptr;
    const int oct = s.K >> 3;
    float rms = 1.0f;
    if (s.mode == A_RMSNORM)
    {
        float ss = 0.0f;
        for (int i = t; i < oct; i += nt)
        {
            const f16x8 v = ((const f16x8*)ar)[i];
            #pragma unroll
            for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
        }
        ss = wave_allreduce_add(ss);
        part[wv] = ss;
        block_sync();
        float tot = 0.0f;
        for (int w = 0; w < nw; w++) tot += part[w];
        rms = fast_rsqrt(tot * (1.0f / (float)s.K) + s.eps);
    }
    for (int i = t; i < oct; i += nt)
    {
        const f16x8 x = ((const f16x8*)ar)[i];
        f16x8 y = {0, 0, 0, 0, 0, 0, 0, 0};
        if (s.mode == A_RMSNORM) y = ((const f16x8*)s.norm_w)[i];
        else if (s.mode == A_SILU_MUL || s.mode == A_GELU_MUL) y = ((const f16x8*)br)[i];
        f16x8 v;
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            f16 xv = x[e];
            if (s.mode == A_RMSNORM)
            {
                const float f = fmaxf(-65504.0f, fminf((float)xv, 65504.0f));
                xv = (f16)((f * (float)y[e]) * rms);
            }
            else if (s.mode == A_SILU_MUL) xv = clamp_h(act_h(xv, false) * y[e]);
            else if (s.mode == A_GELU_MUL) xv = clamp_h(act_h(xv, true) * y[e]);
            else if (s.mode == A_SILU) xv = act_h(xv, false);
            else if (s.mode == A_GELU) xv = act_h(xv, true);
            v[e] = xv;
        }
        ((f16x8*)row)[i] = v;
    }
    block_sync();
    f16* out = s.out + (size_t)r * s.K;
    const int o_lo = (int)((long long)oct * slice / slices), o_hi = (int)((long long)oct * (slice + 1) / slices);
    for (int i = o_lo + t; i < o_hi; i += nt)
    {
        f16x8 v;
        if (s.perm)
        {
            const u32x4 pv = ((const u32x4*)s.perm)[i];
            v[0] = row[pv.x & 0xFFFF]; v[1] = row[pv.x >> 16]; v[2] = row[pv.y & 0xFFFF]; v[3] = row[pv.y >> 16];
            v[4] = row[pv.z & 0xFFFF]; v[5] = row[pv.z >> 16]; v[6] = row[pv.w & 0xFFFF]; v[7] = row[pv.w >> 16];
        }
        else v = ((const f16x8*)row)[i];
        ((f16x8*)out)[i] = v;
    }
}

KERNEL void __launch_bounds__(256) stage_rows_kernel(const RowStageArgs s)
{
    DYN_SMEM(smem);
    if (s.parts > 1) stage_rows_body(s, (char*)smem, bid_y(), s.parts);
    else stage_rows_body(s, (char*)smem);
}

// the same for the matrices of a fused launch (blockIdx.y = matrix): one launch ahead of the phased decode kernel
struct RowStageMulti { RowStageArgs s[MAX_FUSED_MATS]; };
KERNEL void __launch_bounds__(256) stage_rows_multi_kernel(const RowStageMulti a)
{
    DYN_SMEM(smem);
    stage_rows_body(a.s[bid_y()], (char*)smem);
}

// ---- host ---------------------------------------------------------------------------------------------------------------

// Scratch for the packed-order activations, one per (device, stream): launches on one stream are ordered, two streams
// never share a buffer.  A buffer that has been handed out is NEVER freed -- a captured graph (GreedyGraphDecoder,
// PipelineStage) may have its address baked into kernel arguments -- so a larger request allocates a new one (at least
// twice the old size, which bounds the retired total by the live size) and the old one stays allocated.  Growing inside
// a capture is refused.  exl2_release_scratch (model unload, after the graphs that used the stream have been destroyed) frees
// the live and the retired buffers of a stream, so processes that cycle through models or streams do not accumulate them.
struct StageSlot { int dev; void* stream; int kind; f16* buf; size_t bytes; std::vector<f16*> retired; };
static std::vector<StageSlot> g_stage_slots;
static std::mutex g_stage_mutex;

int prefill_scratch(size_t bytes, void* stream, int kind, f16** out)
{
    const int dev = exl2_current_device();
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    StageSlot* slot = nullptr;
    for (StageSlot& s : g_stage_slots) if (s.dev == dev && s.stream == stream && s.kind == kind) { slot = &s; break; }
    if (!slot) { g_stage_slots.push_back(StageSlot{dev, stream, kind, nullptr, 0, {}}); slot = &g_stage_slots.back(); }
    if (slot->bytes < bytes)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cs);
        EXL2_REQUIRE(cs == hipStreamCaptureStatusNone, "prefill: staging scratch cannot grow inside a graph capture");
        size_t want = bytes < (size_t)(1 << 20) ? (size_t)(1 << 20) : bytes;
        if (want < slot->bytes * 2) want = slot->bytes * 2;
        f16* fresh = nullptr;
        if (hipMalloc((void**)&fresh, want) != hipSuccess)
        {
            (void)hipGetLastError();
            want = bytes;
            if (hipMalloc((void**)&fresh, want) != hipSuccess)
            {
                (void)hipGetLastError();
                EXL2_FAIL(EXL2_E_OOM, "HIP out of memory (prefill staging scratch: %zu bytes)", bytes);
            }
        }
        // kind 2 (qgemm_skinny.hip's split-K scratch) starts with 64 KB of arrival tickets that must read zero (the kernel leaves them so)
        if (kind == 2) HIP_TRY(hipMemsetAsync(fresh, 0, (size_t)1 << 16, (hipStream_t)stream));
        if (slot->buf) slot->retired.push_back(slot->buf);      // stays allocated until exl2_release_scratch (see above)
        slot->buf = fresh;
        slot->bytes = want;
    }
    *out = slot->buf;
    return EXL2_OK;
}
static int stage_scratch(size_t bytes, void* stream, f16** out) { return prefill_scratch(bytes, stream, 0, out); }

// Frees the staging buffers (live and retired) of the current device: those of `stream`, or of every stream when all_streams
// != 0.  The caller guarantees that no captured graph that ran a prefill / batched-decode launch on those streams will be
// replayed again (model.unload() calls it after its decoders' graphs are gone).  Returns the number of bytes released.
extern "C" long long exl2_release_scratch(void* stream, int all_streams)
{
    const int dev = exl2_current_device();
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    long long freed = 0;
    for (size_t i = 0; i < g_stage_slots.size();)
    {
        StageSlot& s = g_stage_slots[i];
        if (s.dev != dev || !(all_streams || s.stream == stream)) { i++; continue; }
        (void)hipDeviceSynchronize();
        if (s.buf) { (void)hipFree(s.buf); freed += (long long)s.bytes; }
        for (f16* r : s.retired) (void)hipFree(r);
        g_stage_slots.erase(g_stage_slots.begin() + i);
    }
    return freed;
}

// Pre-pass for the phased decode kernel (qgemv_stream.hip): rows of every job in its packed K order, into the per-device
// scratch; out[i] = where job i's [M, K_i] rows went.  A plain, unpermuted input is used in place (out[i] = a, ld = lda).
int stage_rows_for_decode(const GemvJob* jobs, int n_jobs, int M, void* stream, const f16** out, int* out_ld)
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
    {
        (void)hipFuncSetAttribute((const void*)stage_rows_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    size_t total = 0;
    int k_max = 0, n_staged = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        const GemvJob& j = jobs[i];
        if (j.a_mode == A_PLAIN && !j.m.perm) continue;
        EXL2_REQUIRE((size_t)j.m.K * 2 + 64 <= 150 * 1024, "q_gemm: K = %d too large for the row pre-pass", j.m.K);
        total += (size_t)M * j.m.K * 2;
        if (j.m.K > k_max) k_max = j.m.K;
        n_staged++;
    }
    f16* stage = nullptr;
    if (n_staged) { const int rc = stage_scratch(total, stream, &stage); if (rc) return rc; }
    RowStageMulti ra;
    memset(&ra, 0, sizeof(ra));
    int n = 0;
    size_t off = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        const GemvJob& j = jobs[i];
        if (j.a_mode == A_PLAIN && !j.m.perm) { out[i] = j.a; out_ld[i] = j.lda; continue; }
        RowStageArgs& s = ra.s[n++];
        s.a = j.a; s.a2 = j.a2; s.norm_w = j.norm_w; s.perm = j.m.perm; s.out = stage + off; s.K = j.m.K; s.lda = j.lda;
        s.mode = j.a_mode; s.eps = j.norm_eps;
        out[i] = stage + off; out_ld[i] = j.m.K;
        off += (size_t)M * j.m.K;
    }
    if (n) LAUNCH(stage_rows_multi_kernel, dim3((unsigned)M, (unsigned)n, 1), dim3(256), (size_t)k_max * 2 + 64, stream, ra);
    return EXL2_OK;
}

#define PF_ROW_CHUNK 16384             // rows staged (and multiplied) per pass: 16384 x 11008 halves = 360 MB of scratch at most
#define PF_MFMA_MIN_ROWS 129           // from here up the 256-column LDS-decode kernel (qgemm_mfma.hip) takes the pass

// returns 0 when done, 1 when the prefill kernel does not apply (caller falls back), < 0 on error
int qgemm_prefill_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream)
{
    if (n_jobs < 1 || M < 1) return -1;
    const char* off = getenv("EXL2_PREFILL_GENERIC");
    if (off && atoi(off)) return 1;
    for (int i = 0; i < n_jobs; i++)
    {
        const GemvJob& j = jobs[i];
        if (j.r_weights) return 1;                                              // MoE routing: decode-shaped path only
        if ((((size_t)j.a) & 15) || (j.lda & 7) || (j.m.perm && (((size_t)j.m.perm) & 15))) return 1;
        if ((j.a_mode == A_SILU_MUL || j.a_mode == A_GELU_MUL) && (!j.a2 || (((size_t)j.a2) & 15))) return 1;
        if (j.a_mode == A_RMSNORM && (!j.norm_w || (((size_t)j.norm_w) & 15))) return 1;
        if ((size_t)j.m.K * 2 + 64 > 150 * 1024) return 1;
    }
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
    {
        (void)hipFuncSetAttribute((const void*)qgemm_prefill_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qgemm_prefill_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)stage_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    int k_max = 0;
    for (int i = 0; i < n_jobs; i++) if (jobs[i].m.K > k_max) k_max = jobs[i].m.K;
    int mfma_min_rows = PF_MFMA_MIN_ROWS;
    if (const char* e = getenv("EXL2_PREFILL_MFMA_MIN_ROWS")) mfma_min_rows = atoi(e);      // tests / A-B runs (0 = never)
    int chunk = M < PF_ROW_CHUNK ? M : PF_ROW_CHUNK;
    // qgemm_mfma.hip fills its stages with BUFFER loads (u32 byte offsets into a 2 GB window over the staged rows): a chunk's
    // (rows + 256) x K x 2 bytes must stay below 2^31, or out-of-range pieces would read zeros silently (round-5 advisor finding;
    // K > ~64.5 k at the default chunk).  The chunk shrinks instead of the route changing.
    while ((size_t)(chunk + 256) * (size_t)k_max * 2 >= ((size_t)1 << 31) && chunk > 256) chunk = (chunk / 2 + 255) & ~255;
    EXL2_REQUIRE((size_t)(chunk + 256) * (size_t)k_max * 2 < ((size_t)1 << 31), "prefill: K = %d is too wide for the staged-row window", k_max);
    f16* stage = nullptr;
    // + 256 rows: qgemm_mfma.hip reads whole row blocks (rows >= M hold stale values, feed only rows that are never stored)
    { const int rc = stage_scratch((size_t)(chunk + 256) * k_max * 2, stream, &stage); if (rc) return rc; }

    for (int r0 = 0; r0 < M; r0 += chunk)
    {
        const int rows = M - r0 < chunk ? M - r0 : chunk;
        for (int i = 0; i < n_jobs; i++)
        {
            const GemvJob& j = jobs[i];
            // q | k | v and gate | up multiply the same rows in the same packed order (one act-order permutation, checked at
            // make time: GemvJob::rows_as_prev): the staged copy of the previous job of this pass is still in the scratch
            const bool same_rows = i > 0 && j.rows_as_prev && jobs[i - 1].a == j.a && jobs[i - 1].a2 == j.a2 &&
                                   jobs[i - 1].lda == j.lda && jobs[i - 1].m.K == j.m.K && jobs[i - 1].a_mode == j.a_mode &&
                                   jobs[i - 1].norm_w == j.norm_w && jobs[i - 1].norm_eps == j.norm_eps;
            if (!same_rows)
            {
                RowStageArgs s;
                memset(&s, 0, sizeof(s));
                s.a = j.a + (size_t)r0 * j.lda; s.a2 = j.a2 ? j.a2 + (size_t)r0 * j.lda : nullptr;
                s.norm_w = j.norm_w; s.perm = j.m.perm; s.out = stage; s.K = j.m.K; s.lda = j.lda; s.mode = j.a_mode; s.eps = j.norm_eps;
                // few rows: rows x parts workgroups ~ one per CU (a 32-row pre-pass on 32 CUs took 6.8 us, 20 % of a 32-sequence decode layer)
                s.parts = rows >= 128 ? 1 : (256 / rows > 16 ? 16 : 256 / rows);
                if (getenv("EXL2_STAGE_PARTS")) s.parts = atoi(getenv("EXL2_STAGE_PARTS"));
                if (s.parts < 1) s.parts = 1;
                LAUNCH(stage_rows_kernel, dim3((unsigned)rows, (unsigned)s.parts, 1), dim3(256), (size_t)j.m.K * 2 + 64, stream, s);
            }

            PrefillArgs p;
            memset(&p, 0, sizeof(p));
            p.m = j.m; p.a = stage; p.c = j.c + (size_t)r0 * j.ldc; p.ldc = j.ldc; p.c_invperm = j.c_invperm;
            p.M = rows; p.c_mode = j.c_mode;
            if (mfma_min_rows > 0 && rows >= mfma_min_rows)
            {
                const int rc = qgemm_mfma_launch(p, gptq, stream);
                if (rc) return rc;
                continue;
            }
            {
                // 17 .. 128 rows: the weight-stream-bound kernel (qgemm_skinny.hip), the following jobs over the SAME staged rows
                // (q | k | v, gate | up) in the same launch; EXL2_PREFILL_SKINNY=0: the generic kernel (A/B runs, tests)
                const char* e = getenv("EXL2_PREFILL_SKINNY");
                if (rows <= 128 && !(e && !atoi(e)))
                {
                    PrefillArgs pas[3];
                    int n = 1;
                    pas[0] = p;
                    while (n < 3 && i + n < n_jobs && !getenv("EXL2_SKINNY_UNFUSED"))
                    {
                        const GemvJob& q = jobs[i + n];
                        const GemvJob& pr = jobs[i + n - 1];
                        const bool same = q.rows_as_prev && pr.a == q.a && pr.a2 == q.a2 && pr.lda == q.lda && pr.m.K == q.m.K &&
                                          pr.a_mode == q.a_mode && pr.norm_w == q.norm_w && pr.norm_eps == q.norm_eps;
                        if (!same) break;
                        PrefillArgs& pq = pas[n];
                        memset(&pq, 0, sizeof(pq));
                        pq.m = q.m; pq.a = stage; pq.c = q.c + (size_t)r0 * q.ldc; pq.ldc = q.ldc; pq.c_invperm = q.c_invperm;
                        pq.M = rows; pq.c_mode = q.c_mode;
                        n++;
                    }
                    int rc = qgemm_skinny_launch(pas, n, gptq, stream);
                    if (rc == 1 && n > 1) { n = 1; rc = qgemm_skinny_launch(pas, 1, gptq, stream); }
                    if (rc < 0) return rc;
                    if (rc == 0) { i += n - 1; continue; }
                }
            }
            dim3 grid((unsigned)((j.m.N + PF_BN - 1) / PF_BN), (unsigned)((rows + PF_BM - 1) / PF_BM), 1);
            const size_t lds = PF_LDS_BYTES(j.m.K, j.m.G);
            if (gptq) LAUNCH((qgemm_prefill_kernel<true>), grid, dim3(PF_THREADS), lds, stream, p);
            else      LAUNCH((qgemm_prefill_kernel<false>), grid, dim3(PF_THREADS), lds, stream, p);
        }
    }
    return 0;
}
