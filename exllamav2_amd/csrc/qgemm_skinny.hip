// qgemm_skinny.hip -- q_matrix x fp16 rows for 17 .. 128 rows (round 6): short prompts, prompt chunks, batches of 17 .. 128 sequences.
//
// Replaces, for these row counts, the reference's reconstruct_kernel + library Hgemm (cuda/q_gemm.cu:243-263: the whole fp16 [K, N]
// matrix written to HBM and read back for a product that needs K N / 2 bytes of weights) and this library's own round-1 kernel
// (128 x 128 tiles, ONE K step in flight, a barrier-synchronised global round trip per step = 3-4 us x K / 128 steps: 137 us for a 7B
// gate_proj at 64 rows, on 86 of 256 CUs; removed in round 6, git history).
//
// The product is bound by the weight stream (64 rows: 256 FLOP per weight byte, the chip's ridge is ~310), so the kernel is built like
// the decode kernels -- many small workgroups, every request far ahead of its use -- and not like a GEMM:
//   * workgroup = 4 waves = 128 columns x ALL rows x a slice of K (blockIdx.y = K split, sized on the host so that ~2 workgroups per
//     CU are out); wave w owns two 16-column tiles and streams THEIR items (128 K-rows = one contiguous 64 b-dword run per tile) through
//     a register ring NST deep, requested NST - 1 steps ahead with the non-temporal policy;
//   * the activation rows of a K step (16 RB rows x 128 halfs, RB = ceil(M / 16)) arrive by LDS-DMA (buffer form: counted by vmcnt like
//     any load) into a ring of NST stages, NST - 1 steps ahead; the step's 4 x 128 group scales (the fp16 values reconstruct() uses)
//     travel with them, one q_scale word per thread, and double as the fence that tells a wave its share of the rows has landed;
//     ONE workgroup barrier per step;
//   * a wave decodes its two tiles to B fragments in registers (same decoders, same half(q - zero) * half(scale) rounding as
//     reconstruct(): what enters the matrix core is bit-identical to reconstruct()'s weight), reads each A fragment ONCE from LDS
//     (swizzled, conflict-free ds_read_b128) and uses it for both tiles: 8 RB v_mfma_f32_16x16x32_f16 per step and wave;
//   * split K: partial tiles leave as agent-scope 8-byte stores in FRAGMENT order (a lane's four accumulators are 16 contiguous
//     bytes), the LAST split to arrive (ticket) adds them in split order -- the result does not depend on arrival order -- and runs
//     the epilogue (bias, accumulate-into-c, the consumer's column order).
// Algorithmic bytes: packed weights + scales once, M K 2 of rows per 128-column group (L2), M N 2 out.
#include "qgemm_prefill.h"
#include "errors.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <mutex>

#ifdef SK_DBG
#define SK_SKIP(x, bit) ((x).dbg & (bit))       // timing experiments of tools/debug/skinny_dbg.py: parts left out, results wrong
#define SK_STAMP(i) do { if (args.trace && t == 0) args.trace[((size_t)bid_y() * gdim_x() + bid_x()) * 8 + (i)] = realtime_stamp(); } while (0)
#else
#define SK_SKIP(x, bit) false
#define SK_STAMP(i) do { } while (0)
#endif
#define SK_BN 128
#define SK_CT 2                                   // 16-column tiles per wave
#define SK_WAVES 4
#define SK_THREADS (SK_WAVES * 64)
#define SK_QPT 2                                  // scale chunks staged per thread (4 chunks x 128 columns / 256 threads)
#define SK_A_BYTES(RB) ((RB) * 16 * 256)          // one stage of rows: 16 RB rows x 128 halfs
#define SK_SC_BYTES (4 * SK_BN * 2)               // [chunk][column] scales (zero points) of one step
#define SK_STAGE_BYTES(RB) (SK_A_BYTES(RB) + 2 * SK_SC_BYTES)
#define SK_NST(RB) ((RB) <= 4 ? 4 : 3)            // stages = register-ring depth: requests run NST - 1 steps ahead
#define SK_LDS_BYTES(RB, chunks) (SK_NST(RB) * SK_STAGE_BYTES(RB) + (((chunks) * 2 + 15) & ~15))

struct SkCtx
{
    const QMatDev* m; const f16* a;
    int M, K, n0, n_tiles, tile0;                 // tile0: this wave's first 16-column tile
    int t, lane, wv;
    int sc_col, sc_q0, sc_n;                      // scale staging: this thread's column and first chunk of a step
    const u16* cg_lds;                            // chunk -> group map [K / 32] (LDS)
    int dbg;
    u32 a_voff[8];                                // per-lane byte offsets of this wave's RB row-copy instructions (fixed for the launch)
};

// scale (and GPTQ zero point) of this thread's column for its SK_QPT chunks of the step that starts at chunk `chunk0`, as the fp16
// values reconstruct() uses (q_matrix.cu:328-553: EXL2 (code + 1)^2 * max, GPTQ scale and code + 1).  Two halves: the REQUEST (the
// loaded words stay untouched in registers while other requests are issued behind them -- arithmetic on them here would make the
// compiler wait for them, and with them for everything requested before) and, steps later, the arithmetic.
struct SkScaleRaw { u32 word[SK_QPT]; f16 s[SK_QPT]; };
template <bool GPTQ>
DEV void sk_scales_request(const SkCtx& x, int chunk0, int nvalid, SkScaleRaw& r)
{
    const QMatDev& m = *x.m;
    #pragma unroll
    for (int i = 0; i < SK_QPT; i++)
    {
        const int q = x.sc_q0 + i;
        const int g = x.cg_lds[q < nvalid ? chunk0 + q : chunk0];
        r.word[i] = m.q_scale[(size_t)g * (m.N >> 3) + (x.sc_n >> 3)];
        if constexpr (GPTQ) r.s[i] = m.scale_src[(size_t)g * m.N + x.sc_n];
        else                r.s[i] = m.scale_src[g];
    }
}
template <bool GPTQ>
DEV void sk_scales_finish(const SkCtx& x, const SkScaleRaw& r, f16* sc, f16* zp)
{
    #pragma unroll
    for (int i = 0; i < SK_QPT; i++)
    {
        const int nib = (r.word[i] >> (4 * (x.sc_n & 7))) & 15;
        if constexpr (GPTQ) { sc[i] = r.s[i]; zp[i] = (f16)(float)(nib + 1); }
        else                { sc[i] = (f16)(float)((nib + 1) * (nib + 1)) * r.s[i]; zp[i] = (f16)0.0f; }
    }
}

template <int BITS, bool GPTQ>
DEV void sk_decode_tile(const LaneWords<BITS>& lw, const f16* sc_lds, const f16* zp_lds, int col, f16x8 (&b)[4])
{
    ZC zc[4];
    if constexpr (GPTQ)
    {
        #pragma unroll
        for (int q = 0; q < 4; q++) zc[q] = make_zc(zp_lds[q * SK_BN + col]);
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
        const f16x2 s2 = h2_dup(sc_lds[q * SK_BN + col]);
        const f16x2 b0 = p[4 * q + 0] * s2, b1 = p[4 * q + 1] * s2, b2 = p[4 * q + 2] * s2, b3 = p[4 * q + 3] * s2;
        b[q] = (f16x8){b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
    }
}

template <int BITS>
DEV void sk_load_tiles(LaneWords<BITS> (&w)[SK_CT], const SkCtx& x, const u32* base, u32 tile_stride, int s)
{
    #pragma unroll
    for (int ct = 0; ct < SK_CT; ct++)
    {
        int tile = x.tile0 + ct;
        if (tile >= x.n_tiles) tile = x.n_tiles - 1;                 // partial last column group: computed, never stored
        load_lane_words<BITS>(base + (size_t)tile * tile_stride + (size_t)s * (64 * BITS), x.lane, w[ct]);
    }
}

// the rows of the K step at packed row k0 -> stage `a_stage` (swizzled: unit u of row r sits at slot u ^ (r & 15)); `units_row`
// 16-byte units per row are valid (16 for a full step)
template <int RB>
DEV void sk_issue_rows(const SkCtx& x, int k0, u8* a_stage, int units_row)
{
    if (SK_SKIP(x, 4)) return;
    #pragma unroll
    for (int i = 0; i < RB; i++)
    {
        const int base_u = (i * SK_WAVES + x.wv) * 64;               // first LDS slot of this copy instruction
        if (units_row >= 16) dma_buf_to_lds16(x.a + k0, x.a_voff[i], a_stage + (size_t)base_u * 16);
        else
        {
            const int slot = base_u + x.lane, row = slot >> 4, u = (slot & 15) ^ (row & 15);
            if (u < units_row) dma_buf_to_lds16(x.a + k0, x.a_voff[i], a_stage + (size_t)base_u * 16);
        }
    }
}

// both tiles decoded to B fragments, then every A fragment of the RB row blocks is read once and used for both
template <int BITS, bool GPTQ, int RB>
DEV void sk_multiply(const SkCtx& x, const LaneWords<BITS> (&w)[SK_CT], const u8* a_lds, const f16* sc_lds, const f16* zp_lds, int wv, int lane,
                     int nvalid, f32x4 (&acc)[RB][SK_CT])
{
    const int i16 = lane & 15, j4 = lane >> 4;
    f16x8 b[SK_CT][4];
    #pragma unroll
    for (int c = 0; c < SK_CT; c++)
    {
        if (SK_SKIP(x, 2)) { for (int q = 0; q < 4; q++) b[c][q] = __builtin_bit_cast(f16x8, (u32x4){w[c].w[0], w[c].w[1], w[c].w[0], w[c].w[1]}); }
        else sk_decode_tile<BITS, GPTQ>(w[c], sc_lds, zp_lds, (wv * SK_CT + c) * 16 + i16, b[c]);
        sched_fence();                                               // one tile's decode temporaries at a time
    }
    if (SK_SKIP(x, 1)) { acc[0][0][0] += (float)b[0][0][0] + (float)b[1][3][7]; return; }
    // the A fragments of row block rb + 1 are requested before the matrix-core work of row block rb
    auto read_a = [&](int rb, f16x8 (&a)[4])
    {
        const int row = rb * 16 + i16;
        #pragma unroll
        for (int q = 0; q < 4; q++) a[q] = *(const f16x8*)(a_lds + (size_t)row * 256 + ((4 * q + j4) ^ (row & 15)) * 16);
    };
    // two row blocks at a time: four independent accumulator chains (2 row blocks x 2 tiles) rotate through the matrix core, the
    // next pair's fragments are requested before the current pair's work
    constexpr int NP = (RB + 1) / 2;
    f16x8 a[2][2][4];
    read_a(0, a[0][0]);
    if (RB > 1) read_a(1, a[0][1]);
    #pragma unroll
    for (int pr = 0; pr < NP; pr++)
    {
        const int rb0 = 2 * pr, cur = pr & 1;
        if (rb0 + 2 < RB) read_a(rb0 + 2, a[cur ^ 1][0]);
        if (rb0 + 3 < RB) read_a(rb0 + 3, a[cur ^ 1][1]);
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            if (q < nvalid)
            {
                #pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    if (rb0 + h < RB)
                    {
                        #pragma unroll
                        for (int c = 0; c < SK_CT; c++) acc[rb0 + h][c] = mfma_16x16x32_f16(a[cur][h][q], b[c][q], acc[rb0 + h][c]);
                    }
                }
            }
        }
    }
}

// F full items of one bit-width section, starting at item 0 of `base` (already advanced to the split's first item), packed row k_base
template <int BITS, bool GPTQ, int RB>
DEV void sk_run_stream(const SkCtx& x, u8* smem, const u32* base, u32 tile_stride, int F, int k_base, f32x4 (&acc)[RB][SK_CT])
{
    constexpr int NST = SK_NST(RB), L = NST - 1;
    auto a_of  = [&](int st) { return smem + (size_t)st * SK_STAGE_BYTES(RB); };
    auto sc_of = [&](int st) { return (f16*)(smem + (size_t)st * SK_STAGE_BYTES(RB) + SK_A_BYTES(RB)); };
    auto zp_of = [&](int st) { return (f16*)(smem + (size_t)st * SK_STAGE_BYTES(RB) + SK_A_BYTES(RB) + SK_SC_BYTES); };
    LaneWords<BITS> ring[NST][SK_CT];
    SkScaleRaw scr[NST];
    // Every request below is UNCONDITIONAL (past the end of the run the last item is requested again and never multiplied): a request
    // under a branch makes the compiler's counted waits at the join as strict as its shortest path needs -- vmcnt(0) in practice,
    // one exposed round trip per step (measured: 1.6-2.0 us per step instead of ~0.3).
    // prologue: steps 0 .. L - 1 requested (rows, then the scales that fence them, then the weights)
    #pragma unroll
    for (int u = 0; u < L; u++)
    {
        const int su = min(u, F - 1);
        sk_issue_rows<RB>(x, k_base + su * SUPER_ROWS, a_of(u), 16);
        asm volatile("" ::: "memory");                               // the copies stay in front of the scale loads (their fence)
        sk_scales_request<GPTQ>(x, (k_base >> 5) + 4 * su, 4, scr[u]);
        sk_load_tiles<BITS>(ring[u], x, base, tile_stride, su);
    }
    for (int s0 = 0; s0 < F; s0 += NST)
    {
        #pragma unroll
        for (int u = 0; u < NST; u++)
        {
            const int s = s0 + u;
            // this step's scales (requested L steps ago, behind the row copies: when they are here the copies have landed)
            f16* scl = sc_of(u); f16* zpl = zp_of(u);
            f16 sc[SK_QPT], zp[SK_QPT];
            sk_scales_finish<GPTQ>(x, scr[u], sc, zp);
            #pragma unroll
            for (int i = 0; i < SK_QPT; i++)
            {
                scl[(x.sc_q0 + i) * SK_BN + x.sc_col] = sc[i];
                if constexpr (GPTQ) zpl[(x.sc_q0 + i) * SK_BN + x.sc_col] = zp[i];
            }
            block_sync_lds();                                        // every wave's rows + scales of step s are in LDS; step s - 1 is read out
            const int un = (u + L) % NST;                            // the stage / ring slot step s - 1 used = where step s + L goes
            const int sn = min(s + L, F - 1);
            sk_load_tiles<BITS>(ring[un], x, base, tile_stride, sn);
            sk_issue_rows<RB>(x, k_base + sn * SUPER_ROWS, a_of(un), 16);
            asm volatile("" ::: "memory");
            sk_scales_request<GPTQ>(x, (k_base >> 5) + 4 * sn, 4, scr[un]);
            if (s < F) sk_multiply<BITS, GPTQ, RB>(x, ring[u], a_of(u), scl, zpl, x.wv, x.lane, 4, acc);
        }
    }
    wait_vmcnt0();                                                   // (requests past the end of the run)
    block_sync_lds();                                                // the stages are free for whoever comes next
}

// the partial last item of a section (its own run, padded side buffer): one synchronous step
template <int BITS, bool GPTQ, int RB>
DEV void sk_tail_step(const SkCtx& x, u8* smem, const u32* base, u32 tile_stride, int k0, int nvalid, f32x4 (&acc)[RB][SK_CT])
{
    u8* a_lds = smem;
    f16* scl = (f16*)(smem + SK_A_BYTES(RB)); f16* zpl = (f16*)(smem + SK_A_BYTES(RB) + SK_SC_BYTES);
    LaneWords<BITS> w[SK_CT];
    f16 sc[SK_QPT], zp[SK_QPT];
    SkScaleRaw raw;
    sk_issue_rows<RB>(x, k0, a_lds, nvalid * 4);
    sk_scales_request<GPTQ>(x, k0 >> 5, nvalid, raw);
    sk_load_tiles<BITS>(w, x, base, tile_stride, 0);
    sk_scales_finish<GPTQ>(x, raw, sc, zp);
    #pragma unroll
    for (int i = 0; i < SK_QPT; i++)
    {
        scl[(x.sc_q0 + i) * SK_BN + x.sc_col] = sc[i];
        if constexpr (GPTQ) zpl[(x.sc_q0 + i) * SK_BN + x.sc_col] = zp[i];
    }
    wait_vmcnt_le<0>();
    block_sync();
    sk_multiply<BITS, GPTQ, RB>(x, w, a_lds, scl, zpl, x.wv, x.lane, nvalid, acc);
    block_sync();
}

// one contiguous stream of items per 16-column tile: a bit-width section (QRun, kernel arguments) or, for matrices with more sections
// than the argument block holds (n_runs = 0), a descriptor (QDesc, device memory); its last item may be partial (`tail_nv` chunks)
struct SkSegment { const u32* base; u32 tile_stride; int n_full, k_base, bits, tail_nv; };
DEV SkSegment sk_segment(const QMatDev& m, int i)
{
    SkSegment g;
    int n_super, nvl, in_tail; u32 base_word;
    if (m.n_runs > 0)
    {
        const QRun& r = m.runs[i];
        n_super = uniform((int)r.n_super); nvl = uniform((int)r.nvalid_last); in_tail = uniform((int)r.in_tail); base_word = uniform(r.base_word);
        g.tile_stride = uniform(r.tile_stride); g.k_base = uniform((int)r.k_base); g.bits = uniform((int)r.bits);
    }
    else
    {
        const QDesc* d = m.desc + i;
        n_super = uniform((int)d->n_super); nvl = uniform((int)d->nvalid_last); in_tail = uniform((int)d->in_tail); base_word = uniform(d->base_word);
        g.tile_stride = uniform(d->tile_stride); g.k_base = uniform((int)d->k_base); g.bits = uniform((int)d->bits);
    }
    g.base = (in_tail ? m.tail : m.qw) + base_word;
    g.n_full = nvl == 4 ? n_super : n_super - 1;
    g.tail_nv = nvl == 4 ? 0 : nvl;
    return g;
}

#define SK_MAX_JOBS 3        // matrices of one launch: q | k | v, gate | up multiply the same staged rows (GemvJob::rows_as_prev)
struct SkinnyJob { QMatDev m; f16* c; const u16* c_invperm; int ldc, c_mode; };
struct SkinnyArgs
{
    SkinnyJob job[SK_MAX_JOBS];
    int group0[SK_MAX_JOBS + 1];   // first column group (blockIdx.x) of every matrix
    const f16* a;           // [M, K] in packed K order, row stride K (stage_rows_kernel's output)
    int M;
    int ks;                 // K splits (gridDim.y)
    f32x4* part;            // [column group][split][wave][row block][tile][lane] partial accumulators (ks > 1)
    u32* tickets;           // [column group], zero between launches
    int dbg;                // SK_DBG build only: timing experiments (wrong results)
    u64* trace;             // SK_DBG build only: [workgroup][8] 100 MHz stamps
};

template <bool GPTQ, int RB>
KERNEL void __launch_bounds__(SK_THREADS, RB == 8 ? 1 : 2) qgemm_skinny_kernel(const SkinnyArgs args)
{
    DYN_SMEM(smem);
    const int jb = (bid_x() >= args.group0[1] ? 1 : 0) + (bid_x() >= args.group0[2] ? 1 : 0);
    const SkinnyJob& job = args.job[jb];
    const QMatDev& m = job.m;
    const int t = tid(), lane = lane_id(), wv = uniform(wave_id());
    const int KS = args.ks, z = bid_y();
    const int n0 = (bid_x() - args.group0[jb]) * SK_BN;
    const int i16 = lane & 15, j4 = lane >> 4;

    SK_STAMP(0);
    SkCtx x;
    x.m = &m; x.a = args.a; x.M = args.M; x.K = m.K; x.n0 = n0; x.n_tiles = m.N / TILE_N; x.tile0 = (n0 >> 4) + wv * SK_CT;
    x.t = t; x.lane = lane; x.wv = wv; x.dbg = args.dbg;
    x.sc_col = t % SK_BN; x.sc_q0 = (t / SK_BN) * SK_QPT; x.sc_n = min(n0 + x.sc_col, m.N - 1);
    #pragma unroll
    for (int i = 0; i < RB; i++)
    {
        const int slot = (i * SK_WAVES + wv) * 64 + lane, row = slot >> 4, u = (slot & 15) ^ (row & 15);
        x.a_voff[i] = (u32)(((size_t)min(row, args.M - 1) * m.K + u * 8) * 2);
    }

    // this split's share: full items [it_lo, it_hi) counted over the sections in K order; partial last items go to the last split
    const int n_seg = m.n_runs > 0 ? m.n_runs : m.n_desc;
    int total = 0;
    for (int si = 0; si < n_seg; si++) total += sk_segment(m, si).n_full;
    const int it_lo = (int)((long long)total * z / KS), it_hi = (int)((long long)total * (z + 1) / KS);
    {
        u16* cg = (u16*)(smem + SK_NST(RB) * SK_STAGE_BYTES(RB));
        for (int i = t; i < (m.K >> 5); i += SK_THREADS) cg[i] = m.chunk_group[i];
        x.cg_lds = cg;
        block_sync();
    }
    SK_STAMP(1);

    f32x4 acc[RB][SK_CT];
    #pragma unroll
    for (int rb = 0; rb < RB; rb++)
        #pragma unroll
        for (int ct = 0; ct < SK_CT; ct++) acc[rb][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    int pos = 0;
    for (int si = 0; si < n_seg; si++)
    {
        const SkSegment sg = sk_segment(m, si);
        const int bits = GPTQ ? 4 : sg.bits;
        if (sg.n_full > 0)
        {
            const int f0 = max(it_lo - pos, 0), f1 = min(it_hi - pos, sg.n_full);
            pos += sg.n_full;
            if (f1 > f0)
            {
                const u32* base = sg.base + (size_t)f0 * (size_t)(64 * bits);
                const int k_base = sg.k_base + f0 * SUPER_ROWS, F = f1 - f0;
                switch (bits)
                {
                    case 4: sk_run_stream<4, GPTQ, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                    case 8: if constexpr (!GPTQ) sk_run_stream<8, false, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                    case 6: if constexpr (!GPTQ) sk_run_stream<6, false, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                    case 5: if constexpr (!GPTQ) sk_run_stream<5, false, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                    case 3: if constexpr (!GPTQ) sk_run_stream<3, false, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                    default: if constexpr (!GPTQ) sk_run_stream<2, false, RB>(x, smem, base, sg.tile_stride, F, k_base, acc); break;
                }
            }
        }
        if (sg.tail_nv > 0 && z == KS - 1)
        {
            const u32* base = sg.base + (size_t)sg.n_full * (size_t)(64 * bits);
            const int k0 = sg.k_base + sg.n_full * SUPER_ROWS, nvl = sg.tail_nv;
            switch (bits)
            {
                case 4: sk_tail_step<4, GPTQ, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
                case 8: if constexpr (!GPTQ) sk_tail_step<8, false, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
                case 6: if constexpr (!GPTQ) sk_tail_step<6, false, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
                case 5: if constexpr (!GPTQ) sk_tail_step<5, false, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
                case 3: if constexpr (!GPTQ) sk_tail_step<3, false, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
                default: if constexpr (!GPTQ) sk_tail_step<2, false, RB>(x, smem, base, sg.tile_stride, k0, nvl, acc); break;
            }
        }
    }

    SK_STAMP(2);
    // ---- split K: the partial tile leaves at agent scope; the last split to arrive adds the KS partials in split order -------------
    if (KS > 1 && !SK_SKIP(x, 8))
    {
        const size_t per_wg = (size_t)SK_WAVES * RB * SK_CT * 64;                  // f32x4 units of one workgroup's partial tile
        f32x4* const mine = args.part + ((size_t)bid_x() * KS + z) * per_wg + (size_t)wv * RB * SK_CT * 64;
        #pragma unroll
        for (int rb = 0; rb < RB; rb++)
            #pragma unroll
            for (int ct = 0; ct < SK_CT; ct++) store_agent_f32x4(mine + (rb * SK_CT + ct) * 64 + lane, acc[rb][ct]);
        wait_vmcnt0();
        block_sync();
        SK_STAMP(3);
        u32* const tk = (u32*)smem;
        if (t == 0) *tk = ticket_add_agent(args.tickets + bid_x(), 1u);
        block_sync();
        SK_STAMP(4);
        if (*tk != (u32)(KS - 1)) return;
        if (t == 0) store_relaxed_agent(args.tickets + bid_x(), 0u);              // every split of this group has arrived: zero for the next call
        const f32x4* const all = args.part + (size_t)bid_x() * KS * per_wg + (size_t)wv * RB * SK_CT * 64;
        #pragma unroll
        for (int rb = 0; rb < RB; rb++)
            #pragma unroll
            for (int ct = 0; ct < SK_CT; ct++) acc[rb][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        #pragma unroll 2
        for (int zz = 0; zz < KS; zz++)
        {
            #pragma unroll
            for (int rb = 0; rb < RB; rb++)
                #pragma unroll
                for (int ct = 0; ct < SK_CT; ct++) acc[rb][ct] += load_agent_f32x4(all + (size_t)zz * per_wg + (rb * SK_CT + ct) * 64 + lane);
        }
    }
    SK_STAMP(5);
    // ---- epilogue: D fragment lane (c = l & 15, j) holds rows 4 j .. 4 j + 3 of column c ---------------------------------------------
    if (SK_SKIP(x, 16)) { if (acc[0][0][0] == 123.456f) job.c[0] = (f16)1.0f; return; }
    #pragma unroll
    for (int ct = 0; ct < SK_CT; ct++)
    {
        const int n = n0 + (wv * SK_CT + ct) * 16 + i16;
        if (n >= m.N) continue;
        const float bias = m.bias ? (float)m.bias[n] : 0.0f;
        const int nn = job.c_invperm ? (int)job.c_invperm[n] : n;
        #pragma unroll
        for (int rb = 0; rb < RB; rb++)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                const int row = rb * 16 + j4 * 4 + e;
                if (row < args.M)
                {
                    f16* cp = job.c + (size_t)row * job.ldc + nn;
                    float v = acc[rb][ct][e] + bias;
                    if (job.c_mode == C_ACCUM) v += (float)*cp;
                    *cp = (f16)v;
                }
            }
        }
    }
    wait_vmcnt0();
    SK_STAMP(6);
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------

template <bool GPTQ, int RB>
static int sk_launch_rb(const SkinnyArgs& p, dim3 grid, size_t lds, void* stream)
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
        (void)hipFuncSetAttribute((const void*)qgemm_skinny_kernel<GPTQ, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LAUNCH((qgemm_skinny_kernel<GPTQ, RB>), grid, dim3(SK_THREADS), lds, stream, p);
    return EXL2_OK;
}

// n (1 .. SK_MAX_JOBS) matrices over the SAME staged rows in one launch.  0 = launched, 1 = does not apply (the caller takes the generic
// kernel, one matrix at a time), < 0 = error
int qgemm_skinny_launch(const PrefillArgs* pas, int n, bool gptq, void* stream)
{
    if (n < 1 || n > SK_MAX_JOBS) return 1;
    const int M = pas[0].M, K = pas[0].m.K;
    if (M < 1 || M > 128) return 1;
    if ((size_t)128 * (size_t)K * 2 >= ((size_t)1 << 31)) return 1;               // the row copies address a 2 GB buffer window
    const int rb = (M + 15) / 16, RBT = rb <= 2 ? 2 : rb <= 4 ? 4 : rb <= 6 ? 6 : 8;
    SkinnyArgs p;
    memset(&p, 0, sizeof(p));
    int groups = 0, min_items = 0x7fffffff;
    for (int i = 0; i < n; i++)
    {
        const QMatDev& m = pas[i].m;
        if (m.K != K || pas[i].M != M || pas[i].a != pas[0].a) return 1;
        int full_items = 0;
        for (int ri = 0; ri < m.n_runs; ri++) if (m.runs[ri].nvalid_last == 4) full_items += (int)m.runs[ri].n_super;
        if (m.n_runs <= 0) full_items = (K >> 7) - m.n_desc > 0 ? (K >> 7) - m.n_desc : 0;       // (descriptors live on the device: a lower bound)
        if (full_items < min_items) min_items = full_items;
        p.job[i].m = m;
        if (getenv("EXL2_SKINNY_FORCE_DESC")) p.job[i].m.n_runs = 0;      // tests: walk the descriptors (what a matrix with > MAX_RUNS sections takes)
        p.job[i].c = pas[i].c; p.job[i].ldc = pas[i].ldc; p.job[i].c_invperm = pas[i].c_invperm; p.job[i].c_mode = pas[i].c_mode;
        p.group0[i] = groups;
        groups += (m.N + SK_BN - 1) / SK_BN;
    }
    for (int i = n; i <= SK_MAX_JOBS; i++) p.group0[i] = i == n ? groups : 0x7fffffff;
    for (int i = n + 1; i <= SK_MAX_JOBS; i++) p.group0[i] = 0x7fffffff;
    // K splits: as many as fit the chip in ONE round (2 workgroups per CU; RB = 8: one, its stages fill the LDS -- a workgroup
    // that has to wait for a slot doubles the launch), >= 3 items per split
    int ks = (RBT == 8 ? 256 : 512) / groups;
    if (ks > 16) ks = 16;
    if (ks > min_items / 3) ks = min_items / 3;
    if (const char* e = getenv("EXL2_SKINNY_SPLITK")) { const int v = atoi(e); if (v >= 1) ks = v > 32 ? 32 : v; if (ks > min_items) ks = min_items; }
    if (ks < 1) ks = 1;
    const size_t lds = SK_LDS_BYTES(RBT, K >> 5);
    if (lds > 160 * 1024) return 1;
#ifdef SK_DBG
    if (const char* e = getenv("EXL2_SKINNY_DBG")) p.dbg = atoi(e);
    if (const char* e = getenv("EXL2_SKINNY_TRACE_PTR")) p.trace = (u64*)strtoull(e, nullptr, 0);
#endif
    p.a = pas[0].a; p.M = M; p.ks = ks;
    if (ks > 1)
    {
        const size_t part_bytes = (size_t)groups * ks * SK_WAVES * RBT * SK_CT * 64 * sizeof(f32x4);
        const size_t tick_bytes = (size_t)1 << 16;
        EXL2_REQUIRE((size_t)groups * 4 <= tick_bytes, "q_gemm: too many column groups for the split-K tickets");
        f16* buf = nullptr;
        const int rc = prefill_scratch(tick_bytes + part_bytes, stream, 2, &buf); if (rc) return rc;      // (tickets zeroed at allocation)
        p.tickets = (u32*)buf; p.part = (f32x4*)((u8*)buf + tick_bytes);
    }
    const dim3 grid((unsigned)groups, (unsigned)ks, 1);
    if (getenv("EXL2_SKINNY_DEBUG")) fprintf(stderr, "[skinny] M %d K %d matrices %d RB %d grid %d x %d lds %zu items %d\n", M, K, n, RBT, groups, ks, lds, min_items);
    if (gptq)
        switch (RBT)
        {
            case 2: return sk_launch_rb<true, 2>(p, grid, lds, stream);
            case 4: return sk_launch_rb<true, 4>(p, grid, lds, stream);
            case 6: return sk_launch_rb<true, 6>(p, grid, lds, stream);
            default: return sk_launch_rb<true, 8>(p, grid, lds, stream);
        }
    switch (RBT)
    {
        case 2: return sk_launch_rb<false, 2>(p, grid, lds, stream);
        case 4: return sk_launch_rb<false, 4>(p, grid, lds, stream);
        case 6: return sk_launch_rb<false, 6>(p, grid, lds, stream);
        default: return sk_launch_rb<false, 8>(p, grid, lds, stream);
    }
}
