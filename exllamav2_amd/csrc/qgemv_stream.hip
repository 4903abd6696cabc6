// qgemv_stream.hip -- the decode-shaped q_gemm kernel (M <= 16 rows, activations fit in LDS in one piece).
//
// Replaces gemm_half_q_half_kernel (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565) on the hot decode path; the generic
// kernel in qgemv.hip stays for shapes that need phased activation staging.
//
// A 10-90 MB GEMV on MI355X lasts 5-20 us, so what bounds it is the number of DEPENDENT memory round trips between
// launch and the last store (1-2 us each when the data is HBM-cold) and how early the weight stream starts:
//   * everything the prologue needs lives in ONE by-value block of kernel arguments (JobHot): no pointer chasing through
//     the argument segment before the first load can be issued;
//   * the prologue's inputs (x rows, norm weight / up rows, the make-time pack = q_perm + chunk->group map, this
//     workgroup's slice of the make-time [tile][G][16] scale table) are copied global -> LDS by asynchronous LDS-DMA
//     issued in the kernel's first cycles; they travel alone (behind the weight flood they would queue behind ~all of
//     it), then a first sip of the weight slice goes into the register ring, the rest after the LDS work; barriers in
//     the prologue order LDS only (block_sync_lds) -- a __syncthreads() would drain vmcnt, i.e. wait for every
//     prefetched weight load.  The prologue runs at 3 waves per SIMD and is bound by its instruction count, so anything
//     computable at load time (scale tables) is;
//   * the activation permutation (act-order), RMSNorm and SiLU(gate)*up happen LDS -> LDS;
//   * a wavefront streams a CONTIGUOUS slice of one 16-column tile (tile16 layout: a tile's K range is one linear
//     stream); every ring load is unconditional (clamped), so the compiler's counted vmcnt lets item i decode while
//     items i+1.. are in flight;
//   * a workgroup of W waves owns W/S tiles, S waves splitting each tile's K range; the S partial sums of a tile are
//     combined through LDS in a fixed order (deterministic; no atomics, no cross-workgroup traffic);
//   * decode = magic-number half2 unpack (qlayout.h) -> exact (q - zero) in fp16 -> B fragment of
//     v_mfma_f32_16x16x32_f16, group scale applied to the fp32 partial sums (qgemv_common.h); bias / residual / MoE
//     routing weight in the epilogue.
#include "qgemv_common.h"
#include "errors.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// EXL2_TRACE build (tools/trace_gemv.py): per-wave timestamps at the phase boundaries of the kernel
#ifdef EXL2_TRACE
#define TRACE_POINT(i) do { if (args.trace && lane_id() == 0) args.trace[((size_t)(bid_y() * gdim_x() + bid_x()) * 16 + wave_id()) * 16 + (i)] = realtime_stamp(); } while (0)
#else
#define TRACE_POINT(i) do { } while (0)
#endif

// What the prologue and the main weight run need, flat and by value (selected with scalar selects, never indexed).
struct alignas(64) JobHot          // 64-byte aligned entries: whole-line scalar loads at a dynamic index
{
    const u32* main_ptr;          // (tile 0, super-chunk 0) of the main run
    const u8*  pack;              // make-time prologue pack: [q_perm][chunk -> group map]
    const f16* sc_tab;            // [tile][G][16] scales ; zp_tab: GPTQ zero points
    const f16* zp_tab;
    const u16* perm;
    const f16* a; const f16* a2; const f16* norm_w;
    u32 main_tile_stride; int main_F; int main_chunk0;
    int tile0, n_tiles, K, G, N, lda, a_mode, a_stride;
    float norm_eps;
    u32 lds_scale_off, lds_zp_off, lds_cg_off, lds_rmf_off, lds_rawx_off, lds_raw2_off, lds_perm_off;
    u32 pack_units, pack_cg_off;
    int main_bits;                // bit width of this matrix' main run (the early ring fill applies when it equals MB)
    int n_runs;
    const f16* r_weights; int r_stride;  // MoE routing weights (nullable): a launch whose rows all weigh zero exits at once
};

struct StreamArgs
{
    JobHot hot[MAX_FUSED_MATS];
    GemvJob job[MAX_FUSED_MATS];  // the rest (minor runs, epilogue): read while the weights stream
    int n_jobs;
    int M;          // rows (<= MAX_GEMV_ROWS)
    int S;          // waves per tile (power of two)
    int TPW;        // tiles per workgroup = waves / S
    u64* trace;     // EXL2_TRACE build only: [block][wave][16] timestamps
};

struct RunSlice { const u32* ptr0; int n; int chunk0; };

// slice r of S of a full run, for one tile
DEV RunSlice slice_of(const QRun& run, const QMatDev& m, int tile, int r, int S)
{
    const int F = (int)run.n_super;
    const int i0 = (int)(((long long)r * F) / S), i1 = (int)(((long long)(r + 1) * F) / S);
    RunSlice s;
    s.n = i1 - i0;
    s.ptr0 = (run.in_tail ? m.tail : m.qw) + run.base_word + (size_t)tile * run.tile_stride + (size_t)i0 * (64u * run.bits);
    s.chunk0 = ((int)run.k_base >> 5) + 4 * i0;
    return s;
}

template <int BITS, bool GPTQ>
DEV void do_run(const QRun& run, const QMatDev& m, int tile, int r, int S, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if (run.nvalid_last != 4)
    {
        // partial super-chunk (one per section at most): the split's first wave takes it
        if (r != 0) return;
        LaneWords<BITS> w;
        const u32* p = m.tail + run.base_word + (size_t)tile * run.tile_stride;
        load_lane_words<BITS>(p, lane, w);
        gemv_super<BITS, GPTQ, false>(w, ph, (int)run.k_base >> 5, (int)run.nvalid_last, lane, acc);
        return;
    }
    const RunSlice s = slice_of(run, m, tile, r, S);
    LaneWords<BITS> b[MINOR_DEPTH];
    stream_items<BITS, GPTQ, MINOR_DEPTH>(s.ptr0, s.n, s.chunk0, ph, lane, acc, b, false);
}

template <bool GPTQ>
DEV void do_run_any(const QRun& run, const QMatDev& m, int tile, int r, int S, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if constexpr (GPTQ) do_run<4, true>(run, m, tile, r, S, ph, lane, acc);
    else
    {
        switch (run.bits)
        {
            case 4: do_run<4, false>(run, m, tile, r, S, ph, lane, acc); break;
            case 8: do_run<8, false>(run, m, tile, r, S, ph, lane, acc); break;
            case 6: do_run<6, false>(run, m, tile, r, S, ph, lane, acc); break;
            case 5: do_run<5, false>(run, m, tile, r, S, ph, lane, acc); break;
            case 3: do_run<3, false>(run, m, tile, r, S, ph, lane, acc); break;
            default: do_run<2, false>(run, m, tile, r, S, ph, lane, acc); break;
        }
    }
}

// MB = bit width of the main (largest) run, whose ring fill is issued with the prologue; 0 = no early fill
// MIXED: the fused matrices do not all have MB as their main width (then the early fill is decided per workgroup)
template <bool GPTQ, int MB, bool MIXED = false>
KERNEL void __launch_bounds__(1024) qgemv_stream_kernel(const StreamArgs args)
{
    DYN_SMEM(smem);
    TRACE_POINT(0);

    // blockIdx.y = matrix of a fused launch: the whole parameter block arrives with ONE batch of scalar loads
    const int ji = bid_y();
    const JobHot h = args.hot[ji];                           // by value: one batch of scalar loads, not one per use
    if (bid_x() * args.TPW >= h.n_tiles) return;              // fused matrices of different widths share grid.x
    const int M = args.M;
    if (h.r_weights)
    {
        // q_gemm_kernel.cuh:189-200: nothing to do for an expert no row is routed to (wave-uniform scalar loads)
        u32 any = 0;
        for (int rr = 0; rr < M; rr++) any |= (u32)as_u16(h.r_weights[(size_t)rr * h.r_stride]);
        if (uniform(any) == 0) return;
    }
    const int S = args.S;
    const int TPW = args.TPW;

    const int t = tid();
    const int nt = nthreads();
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int nw = nt >> 6;
    const int gidx = wv / S;                                  // tile slot inside the workgroup
    const int r = wv - gidx * S;                              // K-slice of that tile
    const int n_tiles = h.n_tiles;
    const int tile_base = bid_x() * TPW;
    int tile = tile_base + gidx;
    const bool tile_ok = tile < n_tiles;
    if (!tile_ok) tile = n_tiles - 1;                         // idle slot: compute on a valid tile, never store
    TRACE_POINT(12);
#ifdef EXL2_TRACE
    if (args.trace && lane_id() == 0)      // where the wave runs: HW_ID (cu / sh / se) and XCC_ID
        args.trace[((size_t)(bid_y() * gdim_x() + bid_x()) * 16 + wave_id()) * 16 + 13] =
            ((u64)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    const int K = h.K, G = h.G, oct = K >> 3;

    f16* a_lds  = (f16*)smem;
    f16* sc_all = (f16*)(smem + h.lds_scale_off);
    f16* zp_all = (f16*)(smem + h.lds_zp_off);
    u16* cg_lds = (u16*)(smem + h.lds_cg_off);
    StageLds L;
    L.rawx = (f16*)(smem + h.lds_rawx_off); L.raw2 = (f16*)(smem + h.lds_raw2_off);
    L.perm = (u16*)(smem + h.lds_perm_off); L.rms = (float*)(smem + h.lds_rmf_off);
    float* red  = (float*)smem;                               // aliases a_lds after the streaming

    // ---- issue: prologue inputs (LDS-DMA) --------------------------------------------------------------------------------
    const bool lds_stage = h.lds_rawx_off != 0;             // rows staged through LDS; else gathered from global memory
    {
        const f16* a = h.a; const int lda = h.lda;
        if (lds_stage)
        for (int rr = 0; rr < M; rr++)
            dma_units16([&](int u) { return (const void*)(a + (size_t)rr * lda + (size_t)u * 8); }, L.rawx + (size_t)rr * K, oct, wv, nw, lane);
        if (!lds_stage) { }
        else if (h.a_mode == A_RMSNORM)
        {
            const f16* w = h.norm_w;
            dma_units16([&](int u) { return (const void*)(w + (size_t)u * 8); }, L.raw2, oct, wv, nw, lane);
        }
        else if (h.a_mode == A_SILU_MUL || h.a_mode == A_GELU_MUL)
        {
            const f16* a2 = h.a2;
            for (int rr = 0; rr < M; rr++)
                dma_units16([&](int u) { return (const void*)(a2 + (size_t)rr * lda + (size_t)u * 8); }, L.raw2 + (size_t)rr * K, oct, wv, nw, lane);
        }
        // make-time pack [q_perm][chunk -> group map]: one contiguous copy (the permutation only when rows go through LDS)
        {
            const u8* pk = h.pack;
            const int skip = lds_stage ? 0 : (int)(h.pack_cg_off >> 4);
            dma_units16([&](int u) { return (const void*)(pk + ((size_t)(skip + u) << 4)); }, lds_stage ? smem + h.lds_perm_off : smem + h.lds_cg_off,
                        (int)h.pack_units - skip, wv, nw, lane, 1 % nw);
        }
        // scale (and GPTQ zero-point) tables of this workgroup's tiles: contiguous in the make-time [tile][G][16] layout
        {
            const int nt_here = min(TPW, n_tiles - tile_base);
            const f16* st = h.sc_tab + (size_t)tile_base * G * 16;
            dma_units16([&](int u) { return (const void*)(st + (size_t)u * 8); }, sc_all, nt_here * G * 2, wv, nw, lane, 2 % nw);
            if constexpr (GPTQ)
            {
                const f16* zt = h.zp_tab + (size_t)tile_base * G * 16;
                dma_units16([&](int u) { return (const void*)(zt + (size_t)u * 8); }, zp_all, nt_here * G * 2, wv, nw, lane, 3 % nw);
            }
        }
    }
    TRACE_POINT(1);
    // The prologue inputs travel alone: issued behind the weight flood they would queue behind ~all of it (the memory
    // system serves the chip's requests roughly in arrival order) and the first barrier would open only when the whole
    // matrix has been read.  One unloaded round trip (~1 us) later the ring fill goes out and the LDS work below overlaps
    // with the weights' flight.
    wait_vmcnt_le<0>();
    block_sync_lds();
    TRACE_POINT(2);
    constexpr int DM = MainDepth<(MB ? MB : 4)>::v;
#ifndef FIRST_SIP_ITEMS
#define FIRST_SIP_ITEMS 2
#endif
    constexpr int FIRST_SIP = DM < FIRST_SIP_ITEMS ? DM : FIRST_SIP_ITEMS;
    LaneWords<(MB ? MB : 4)> pre[DM];
    RunSlice ms; ms.n = 0; ms.ptr0 = nullptr; ms.chunk0 = 0;
    // fused matrices may have different main widths (q / k vs v in low-bpw models): the early fill serves the ones that
    // match this instantiation, the others stream all their runs through the run loop below
    const bool fast_main = MB != 0 && (!MIXED || h.main_bits == MB);
    if (fast_main)
    {
        const int F = h.main_F;
        const int i0 = (int)(((long long)r * F) / S), i1 = (int)(((long long)(r + 1) * F) / S);
        ms.n = i1 - i0;
        ms.ptr0 = h.main_ptr + (size_t)tile * h.main_tile_stride + (size_t)i0 * (64u * MB);
        ms.chunk0 = h.main_chunk0 + 4 * i0;
        // a first sip only: more than ~2 KB per wave overflows the CU's request queue and the wave would sit in the issue
        // stage instead of doing the LDS work below; the rest of the ring goes out right after that work
        ring_fill<(MB ? MB : 4), DM, 0, FIRST_SIP>(pre, ms.ptr0, ms.n, lane);
    }
    TRACE_POINT(9);

    // ---- prologue: tables and activations, LDS -> LDS -------------------------------------------------------------------
    TRACE_POINT(10);
    if (lds_stage)
    {
        if (h.a_mode == A_RMSNORM)
        {
            stage_rms_lds(L, K, h.norm_eps, M, lane, wv, nw);
            block_sync_lds();
        }
        TRACE_POINT(11);
        const bool hp = h.perm != nullptr;
        switch (h.a_mode)
        {
            case A_PLAIN:    stage_shuffle_lds<A_PLAIN>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
            case A_RMSNORM:  stage_shuffle_lds<A_RMSNORM>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
            case A_SILU_MUL: stage_shuffle_lds<A_SILU_MUL>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
            case A_GELU_MUL: stage_shuffle_lds<A_GELU_MUL>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
            case A_SILU:     stage_shuffle_lds<A_SILU>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
            default:         stage_shuffle_lds<A_GELU>(L, hp, a_lds, h.a_stride, K, M, t, nt); break;
        }
    }
    else
    {
        // many rows (M x K does not fit twice in LDS): gather straight from global memory -- three dependent round trips,
        // amortised over M rows (stage_rows, qgemv_common.h)
        const GemvJob& jc = args.job[ji];
        float* rmf_w = L.rms + 16 + wv * 16;                 // per-wave copy: no block barrier needed
        if (h.a_mode == A_RMSNORM)
        {
            for (int rr = 0; rr < M; rr++)
            {
                const f16x8* xr = (const f16x8*)(h.a + (size_t)rr * h.lda);
                float ss = 0.0f;
                for (int i = lane; i < oct; i += 64)
                {
                    const f16x8 v = xr[i];
                    #pragma unroll
                    for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
                }
                ss = wave_allreduce_add(ss);
                rmf_w[rr] = fast_rsqrt(ss * (1.0f / (float)K) + h.norm_eps);
            }
        }
        switch (h.a_mode)
        {
            case A_PLAIN:    stage_rows<A_PLAIN>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
            case A_RMSNORM:  stage_rows<A_RMSNORM>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
            case A_SILU_MUL: stage_rows<A_SILU_MUL>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
            case A_GELU_MUL: stage_rows<A_GELU_MUL>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
            case A_SILU:     stage_rows<A_SILU>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
            default:         stage_rows<A_GELU>(jc, jc.m, h.a, h.a2, a_lds, rmf_w, 0, oct, M, t, nt); break;
        }
    }
    if (fast_main) ring_fill<(MB ? MB : 4), DM, FIRST_SIP, DM>(pre, ms.ptr0, ms.n, lane);
    TRACE_POINT(3);
    block_sync_lds();
    TRACE_POINT(4);

    // ---- stream -------------------------------------------------------------------------------------------------------
    const GemvJob& job = args.job[ji];
    const QMatDev& m = job.m;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    PhaseCtx ph;
    ph.a_lds = a_lds; ph.sc_lds = sc_all + (size_t)gidx * G * 16; ph.zp_lds = zp_all + (size_t)gidx * G * 16;
    ph.cg_lds = cg_lds; ph.a_stride = h.a_stride;
    ph.M = M; ph.phase_k0 = 0;

    if (fast_main) stream_items<(MB ? MB : 4), GPTQ, DM>(ms.ptr0, ms.n, ms.chunk0, ph, lane, acc, pre, true);
    TRACE_POINT(5);
    if (!fast_main || h.n_runs > 1)
    {
        for (int i = 0; i < m.n_runs; i++)
        {
            if (fast_main && i == m.main_run) continue;
            do_run_any<GPTQ>(m.runs[i], m, tile, r, S, ph, lane, acc);
        }
    }

    // ---- combine the S slices of every tile (fixed order) + epilogue ----------------------------------------------------
    TRACE_POINT(6);
    block_sync();
    TRACE_POINT(7);
    {
        const int c = lane & 15, j = lane >> 4;
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int row = j * 4 + q;
            if (row < M) red[(wv * 16 + row) * 16 + c] = acc[q];
        }
    }
    block_sync();
    for (int idx = t; idx < TPW * M * 16; idx += nt)
    {
        const int slot = idx / (M * 16);
        const int rem = idx - slot * (M * 16);
        const int row = rem >> 4, c = rem & 15;
        const int tl = tile_base + slot;
        if (tl >= n_tiles) continue;
        float v = 0.0f;
        for (int w = 0; w < S; w++) v += red[((slot * S + w) * 16 + row) * 16 + c];
        const int n = tl * 16 + c;
        bool skip = false;
        if (job.r_weights)
        {
            const f16 rw = job.r_weights[(size_t)row * job.r_stride];
            if (as_u16(rw) == 0) skip = true;                       // q_gemm_kernel.cuh:189-200
            if (job.mul_r_weights) v *= (float)rw;
        }
        if (!skip)
        {
            if (m.bias) v += (float)m.bias[n];
            f16* cp = job.c + (size_t)row * job.ldc + (job.c_invperm ? (int)job.c_invperm[n] : n);
            if (job.c_mode == C_ACCUM) v += (float)*cp;
            *cp = (f16)v;
        }
    }
    TRACE_POINT(8);
}

// ---- many rows x large K: phased kernel ---------------------------------------------------------------------------------
//
// When M rows of K activations do not fit in LDS next to their raw copy (M = 5..16 at K >= 4096, e.g. a batch of 16
// sequences decoding), the activations are brought into each matrix' packed K order ONCE by a row pre-pass
// (stage_rows_for_decode, qgemm_prefill.hip: norm / activation / act-order permutation, M small workgroups) instead of
// once per workgroup, and this kernel walks K in phases: LDS-DMA the [M, rows] slab of the packed rows (contiguous
// row segments, one round trip out of L2), stream the matching super-chunks of every tile, next slab.  The partial sums
// stay in registers across phases; everything else (tile16 stream, S-way K split, fixed-order combine, epilogue) is the
// streaming kernel's.
struct PhasedArgs
{
    GemvJob job[MAX_FUSED_MATS];
    const f16* ap[MAX_FUSED_MATS];      // packed-order rows [M, ap_ld]
    int ap_ld[MAX_FUSED_MATS];
    u32 lds_scale_off[MAX_FUSED_MATS], lds_zp_off[MAX_FUSED_MATS], lds_cg_off[MAX_FUSED_MATS];
    int n_jobs, M, S, TPW;
    int items_max;                      // super-chunks of K per phase that fit the slab
};

template <int BITS, bool GPTQ>
DEV void phase_items(const QRun& run, const QMatDev& m, int tile, int it0, int n, int r, int S, const PhaseCtx& ph, int lane, f32x4& acc)
{
    const int i0 = it0 + (int)(((long long)r * n) / S), i1 = it0 + (int)(((long long)(r + 1) * n) / S);
    const u32* ptr0 = (run.in_tail ? m.tail : m.qw) + run.base_word + (size_t)tile * run.tile_stride + (size_t)i0 * (64u * BITS);
    LaneWords<BITS> b[MINOR_DEPTH];
    stream_items<BITS, GPTQ, MINOR_DEPTH>(ptr0, i1 - i0, ((int)run.k_base >> 5) + 4 * i0, ph, lane, acc, b, false);
}

template <int BITS, bool GPTQ>
DEV void phase_tail(const QRun& run, const QMatDev& m, int tile, const PhaseCtx& ph, int lane, f32x4& acc)
{
    LaneWords<BITS> w;
    load_lane_words<BITS>(m.tail + run.base_word + (size_t)tile * run.tile_stride, lane, w);
    gemv_super<BITS, GPTQ, false>(w, ph, (int)run.k_base >> 5, (int)run.nvalid_last, lane, acc);
}

template <bool GPTQ>
KERNEL void __launch_bounds__(1024) qgemv_phased_kernel(const PhasedArgs args)
{
    DYN_SMEM(smem);
    const int ji = bid_y();
    const GemvJob& job = args.job[ji];
    const QMatDev& m = job.m;
    const int TPW = args.TPW, S = args.S, M = args.M;
    const int n_tiles = m.N / TILE_N;
    if (bid_x() * TPW >= n_tiles) return;
    if (job.r_weights)
    {
        u32 any = 0;
        for (int rr = 0; rr < M; rr++) any |= (u32)as_u16(job.r_weights[(size_t)rr * job.r_stride]);
        if (uniform(any) == 0) return;                      // q_gemm_kernel.cuh:189-200
    }
    const int t = tid(), nt = nthreads(), lane = lane_id();
    const int wv = uniform(wave_id()), nw = nt >> 6;
    const int gidx = wv / S, r = wv - gidx * S;
    const int tile_base = bid_x() * TPW;
    int tile = tile_base + gidx;
    if (tile >= n_tiles) tile = n_tiles - 1;                  // idle slot: compute on a valid tile, never store
    const int G = m.G;

    f16* a_lds  = (f16*)smem;
    f16* sc_all = (f16*)(smem + args.lds_scale_off[ji]);
    f16* zp_all = (f16*)(smem + args.lds_zp_off[ji]);
    u16* cg_lds = (u16*)(smem + args.lds_cg_off[ji]);
    float* red  = (float*)smem;

    // tables: chunk -> group map (tail of the make-time pack) and this workgroup's [tile][G][16] scale slices
    {
        const u8* pk = m.pack;
        const int skip = (int)(m.pack_cg_off >> 4);
        dma_units16([&](int u) { return (const void*)(pk + ((size_t)(skip + u) << 4)); }, cg_lds, (int)m.pack_units - skip, wv, nw, lane);
        const int nt_here = min(TPW, n_tiles - tile_base);
        const f16* st = m.sc_tab + (size_t)tile_base * G * 16;
        dma_units16([&](int u) { return (const void*)(st + (size_t)u * 8); }, sc_all, nt_here * G * 2, wv, nw, lane, 1 % nw);
        if constexpr (GPTQ)
        {
            const f16* zt = m.zp_tab + (size_t)tile_base * G * 16;
            dma_units16([&](int u) { return (const void*)(zt + (size_t)u * 8); }, zp_all, nt_here * G * 2, wv, nw, lane, 2 % nw);
        }
    }

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    PhaseCtx ph;
    ph.sc_lds = sc_all + (size_t)gidx * G * 16; ph.zp_lds = zp_all + (size_t)gidx * G * 16;
    ph.cg_lds = cg_lds; ph.a_lds = a_lds; ph.M = M;
    const f16* ap = args.ap[ji];
    const int ap_ld = args.ap_ld[ji];
    bool first = true;
    for (int i = 0; i < m.n_runs; i++)
    {
        const QRun& run = m.runs[i];
        const bool full = run.nvalid_last == 4;
        const int F = full ? (int)run.n_super : 1;
        const int n_ph = (F + args.items_max - 1) / args.items_max;
        const int ipp = (F + n_ph - 1) / n_ph;
        for (int it0 = 0; it0 < F; it0 += ipp)
        {
            const int n = min(ipp, F - it0);
            const int k0 = (int)run.k_base + it0 * SUPER_ROWS;
            const int rows = full ? n * SUPER_ROWS : (int)run.nvalid_last * 32;
            const int a_stride = rows + 8;
            if (!first) block_sync_lds();                    // the previous slab's readers are done
            first = false;
            const int upr = rows >> 3;
            for (int rr = 0; rr < M; rr++)
                dma_units16([&](int u) { return (const void*)(ap + (size_t)rr * ap_ld + k0 + (size_t)u * 8); },
                            a_lds + (size_t)rr * a_stride, upr, wv, nw, lane, rr % nw);
            wait_vmcnt_le<0>();
            block_sync_lds();
            ph.a_stride = a_stride; ph.phase_k0 = k0;
            if (!full)
            {
                if (r != 0) continue;                         // partial super-chunk: the split's first wave takes it
                if constexpr (GPTQ) phase_tail<4, true>(run, m, tile, ph, lane, acc);
                else switch (run.bits)
                {
                    case 4: phase_tail<4, false>(run, m, tile, ph, lane, acc); break;
                    case 8: phase_tail<8, false>(run, m, tile, ph, lane, acc); break;
                    case 6: phase_tail<6, false>(run, m, tile, ph, lane, acc); break;
                    case 5: phase_tail<5, false>(run, m, tile, ph, lane, acc); break;
                    case 3: phase_tail<3, false>(run, m, tile, ph, lane, acc); break;
                    default: phase_tail<2, false>(run, m, tile, ph, lane, acc); break;
                }
                continue;
            }
            if constexpr (GPTQ) phase_items<4, true>(run, m, tile, it0, n, r, S, ph, lane, acc);
            else switch (run.bits)
            {
                case 4: phase_items<4, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
                case 8: phase_items<8, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
                case 6: phase_items<6, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
                case 5: phase_items<5, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
                case 3: phase_items<3, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
                default: phase_items<2, false>(run, m, tile, it0, n, r, S, ph, lane, acc); break;
            }
        }
    }

    // combine the S slices of every tile (fixed order) + epilogue: as in qgemv_stream_kernel
    block_sync();
    {
        const int c = lane & 15, j = lane >> 4;
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int row = j * 4 + q;
            if (row < M) red[(wv * 16 + row) * 16 + c] = acc[q];
        }
    }
    block_sync();
    for (int idx = t; idx < TPW * M * 16; idx += nt)
    {
        const int slot = idx / (M * 16);
        const int rem = idx - slot * (M * 16);
        const int row = rem >> 4, c = rem & 15;
        const int tl = tile_base + slot;
        if (tl >= n_tiles) continue;
        float v = 0.0f;
        for (int w = 0; w < S; w++) v += red[((slot * S + w) * 16 + row) * 16 + c];
        const int n = tl * 16 + c;
        bool skip = false;
        if (job.r_weights)
        {
            const f16 rw = job.r_weights[(size_t)row * job.r_stride];
            if (as_u16(rw) == 0) skip = true;
            if (job.mul_r_weights) v *= (float)rw;
        }
        if (!skip)
        {
            if (m.bias) v += (float)m.bias[n];
            f16* cp = job.c + (size_t)row * job.ldc + (job.c_invperm ? (int)job.c_invperm[n] : n);
            if (job.c_mode == C_ACCUM) v += (float)*cp;
            *cp = (f16)v;
        }
    }
}

// ---- host --------------------------------------------------------------------------------------------------------------

#ifdef EXL2_TRACE
static u64* g_trace_buf = nullptr;
static int g_trace_which = 0, g_trace_count = 0;
// trace the `which`-th streaming launch after this call (0 = the next one)
extern "C" void exl2_debug_set_trace(void* p, int which) { g_trace_buf = (u64*)p; g_trace_which = which; g_trace_count = 0; }
#endif

static inline u32 align16s(u32 x) { return (x + 15u) & ~15u; }

static int num_cus()
{
    static int n[EXL2_MAX_DEVICES] = {0};
    const int dev = exl2_current_device();
    if (n[dev] <= 0)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
        if (n[dev] <= 0) n[dev] = 256;
    }
    return n[dev];
}

template <bool GPTQ, int MB>
static void launch_variant(const StreamArgs& args, dim3 grid, dim3 block, u32 lds, void* stream, bool mixed = false)
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
    {
        (void)hipFuncSetAttribute((const void*)qgemv_stream_kernel<GPTQ, MB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qgemv_stream_kernel<GPTQ, MB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (mixed) LAUNCH((qgemv_stream_kernel<GPTQ, MB, true>), grid, block, lds, stream, args);
    else       LAUNCH((qgemv_stream_kernel<GPTQ, MB, false>), grid, block, lds, stream, args);
}

static inline bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

int stage_rows_for_decode(const GemvJob* jobs, int n_jobs, int M, void* stream, const f16** out, int* out_ld);   // qgemm_prefill.hip

// slab budget of the phased kernel (bytes of LDS for the [M, rows] activations of one phase)
#define PHASED_SLAB_BYTES (136u * 1024u)

static int qgemv_phased_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, int TPW, int S, void* stream)
{
    PhasedArgs args;
    memset(&args, 0, sizeof(args));
    int items_max = (int)(PHASED_SLAB_BYTES / ((u32)M * 2) - 8) / SUPER_ROWS;
    const char* fi = getenv("EXL2_GEMV_PHASE_ITEMS");       // tests: small slabs = many phases at small K
    if (fi && atoi(fi) > 0 && atoi(fi) < items_max) items_max = atoi(fi);
    if (items_max < 1) return 1;
    for (int i = 0; i < n_jobs; i++)
        if ((jobs[i].m.K & 7) || (jobs[i].a_mode == A_PLAIN && !jobs[i].m.perm && (jobs[i].lda & 7))) return 1;
    // tables must fit beside the slab: fewer tiles per workgroup for matrices with many groups
    u32 lds = 0;
    for (;;)
    {
        lds = 0;
        for (int i = 0; i < n_jobs; i++)
        {
            const QMatDev& m = jobs[i].m;
            int f_max = 1;
            for (int q = 0; q < m.n_runs; q++) if (m.runs[q].nvalid_last == 4 && (int)m.runs[q].n_super > f_max) f_max = (int)m.runs[q].n_super;
            const int n_ph = (f_max + items_max - 1) / items_max;
            const int ipp = (f_max + n_ph - 1) / n_ph;
            u32 total = align16s((u32)M * (u32)(ipp * SUPER_ROWS + 8) * 2);
            const u32 red_bytes = (u32)(TPW * S) * 16 * 16 * 4;
            if (total < red_bytes) total = red_bytes;
            args.lds_scale_off[i] = total; total += align16s((u32)TPW * m.G * 32);
            args.lds_zp_off[i] = total;    total += gptq ? align16s((u32)TPW * m.G * 32) : 0;
            args.lds_cg_off[i] = total;    total += m.pack_units * 16 - m.pack_cg_off;
            if (total > lds) lds = total;
        }
        if (lds <= 160 * 1024) break;
        if (TPW > 1) { TPW = (TPW + 1) / 2; S = 16 / TPW; continue; }
        if (items_max > 1) { items_max = (items_max * 3) / 4 > 0 ? (items_max * 3) / 4 : 1; continue; }
        return 1;
    }
    {
        const int rc = stage_rows_for_decode(jobs, n_jobs, M, stream, args.ap, args.ap_ld);
        if (rc) return rc;
    }
    int blk_max = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        args.job[i] = jobs[i];
        const int blks = (jobs[i].m.N / TILE_N + TPW - 1) / TPW;
        if (blks > blk_max) blk_max = blks;
    }
    args.n_jobs = n_jobs; args.M = M; args.S = S; args.TPW = TPW; args.items_max = items_max;
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr))
    {
        (void)hipFuncSetAttribute((const void*)qgemv_phased_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qgemv_phased_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    dim3 grid((unsigned)blk_max, (unsigned)n_jobs, 1), block((unsigned)(TPW * S * 64), 1, 1);
    if (getenv("EXL2_DEBUG_ROUTE")) fprintf(stderr, "q_gemm route: phased M=%d jobs=%d TPW=%d S=%d items=%d lds=%u\n", M, n_jobs, TPW, S, items_max, lds);
    if (gptq) LAUNCH((qgemv_phased_kernel<true>), grid, block, lds, stream, args);
    else      LAUNCH((qgemv_phased_kernel<false>), grid, block, lds, stream, args);
    return 0;
}

// returns 0 when launched, 1 when this kernel does not apply (caller falls back to the generic kernel), < 0 on error
int qgemv_stream_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream)
{
    if (n_jobs < 1 || n_jobs > MAX_FUSED_MATS || M < 1) EXL2_FAIL(EXL2_E_INVALID, "q_gemm: %d fused matrices / %d rows not launchable", n_jobs, M);
    if (M > MAX_GEMV_ROWS) return 1;
    const char* off = getenv("EXL2_GEMV_GENERIC");
    if (off && atoi(off)) return 1;
    long long tiles = 0;
    int min_items = 1 << 30, mb = -1;
    long long bits_weight[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_jobs; i++)
    {
        const GemvJob& j = jobs[i];
        const QMatDev& m = j.m;
        if (m.n_runs <= 0) return 1;
        // LDS-DMA sources: 16-byte units
        if (!aligned16(j.a) || (j.lda & 7) || (m.perm && !aligned16(m.perm))) return 1;
        if ((j.a_mode == A_SILU_MUL || j.a_mode == A_GELU_MUL) && !aligned16(j.a2)) return 1;
        if (j.a_mode == A_RMSNORM && !aligned16(j.norm_w)) return 1;
        if (!m.pack || !m.sc_tab || (gptq && !m.zp_tab)) return 1;
        tiles += m.N / TILE_N;
        const QRun& mr = m.runs[m.main_run];
        const int items = mr.nvalid_last == 4 ? (int)mr.n_super : 0;
        if (items < min_items) min_items = items;
        const int b = items > 0 ? (int)mr.bits : 0;
        if (b >= 2 && b <= 8) bits_weight[b] += (long long)items * b * (m.N / TILE_N);
    }
    for (int b = 2; b <= 8; b++) if (bits_weight[b] > 0 && (mb < 0 || bits_weight[b] > bits_weight[mb])) mb = b;
    if (mb < 0) mb = 0;

    // Shape of the launch.  Measured on MI355X (tools/trace_gemv.py): the dispatcher spreads workgroups breadth-first over
    // the CUs, and a CU needs ~6.5 us + 29 ns per super-chunk it decodes -- the kernel lasts as long as its most loaded
    // CU.  So: at most one workgroup per CU, every workgroup the same number of 16-column tiles (TPW), and the 16 wave
    // slots of a CU split each tile's K range S ways (S need not be a power of two).
    const int ncu = num_cus();
    int t_max = 0;
    for (int i = 0; i < n_jobs; i++) if (jobs[i].m.N / TILE_N > t_max) t_max = jobs[i].m.N / TILE_N;
    const int wgs_per_job = ncu / n_jobs > 0 ? ncu / n_jobs : 1;
    int TPW = (t_max + wgs_per_job - 1) / wgs_per_job;
    if (TPW > 16) TPW = 16;
    int S = 16 / TPW;
    if (S > min_items / 2) S = min_items / 2;
    if (S < 1) S = 1;
    const char* fs = getenv("EXL2_GEMV_SPLIT");
    if (fs && atoi(fs) > 0) S = atoi(fs);
    const char* ft = getenv("EXL2_GEMV_TPW");
    if (ft && atoi(ft) > 0) TPW = atoi(ft);
    if (TPW * S > 16) TPW = 16 / S > 0 ? 16 / S : 1;
    {
        // more than a few rows whose raw + staged copies do not fit in LDS: pre-pass + phased kernel
        bool fits = true;
        for (int i = 0; i < n_jobs; i++)
        {
            const GemvJob& j = jobs[i];
            const QMatDev& m = j.m;
            const u32 row_bytes = align16s((u32)M * m.K * 2);
            const bool two = j.a_mode == A_SILU_MUL || j.a_mode == A_GELU_MUL;
            const unsigned long long need = (unsigned long long)align16s((u32)M * (m.K + 8) * 2) + row_bytes
                + (two ? row_bytes : (j.a_mode == A_RMSNORM ? align16s((u32)m.K * 2) : 0)) + (unsigned long long)m.pack_units * 16
                + (unsigned long long)align16s((u32)m.G * 32) * (gptq ? 2 : 1) + 64 + 16 * 16 * 4;
            if (need > 160 * 1024) fits = false;
        }
        const char* fp = getenv("EXL2_GEMV_PHASED");        // 0: never, 1: whenever M > 1 (tests), default: M > 4 and no fit
        const int force = fp ? atoi(fp) : -1;
        if (force != 0 && ((force == 1 && M > 1) || (M > 4 && !fits)))
        {
            const int rc = qgemv_phased_launch(jobs, n_jobs, M, gptq, TPW, S, stream);
            if (rc <= 0) return rc;
        }
    }
    // many quantisation groups (small group size x large K): the per-tile scale tables may not fit for TPW tiles --
    // fewer tiles per workgroup (more workgroups, possibly a second round) still beats the generic kernel by far
    for (;;)
    {
        u32 worst = 0;
        for (int i = 0; i < n_jobs; i++)
        {
            const QMatDev& m = jobs[i].m;
            u32 a_bytes = align16s((u32)M * (m.K + 8) * 2);
            const u32 red_bytes = (u32)(TPW * S) * 16 * 16 * 4;
            if (a_bytes < red_bytes) a_bytes = red_bytes;
            const u32 t = a_bytes + align16s((u32)TPW * m.G * 32) * (gptq ? 2 : 1) + 64 + 16 * 16 * 4 + (m.pack_units * 16 - m.pack_cg_off);
            if (t > worst) worst = t;
        }
        if (worst <= 160 * 1024 || TPW == 1) break;
        TPW = (TPW + 1) / 2;
        S = 16 / TPW;
        if (S > min_items / 2) S = min_items / 2;
        if (S < 1) S = 1;
    }
    const int W = TPW * S;

    const char* nls = getenv("EXL2_GEMV_NO_LDS_STAGE");       // tests: force the many-rows staging route
    const bool no_lds_stage = nls && atoi(nls);
    StreamArgs args;
    memset(&args, 0, sizeof(args));
    args.n_jobs = n_jobs; args.M = M; args.S = S; args.TPW = TPW;
#ifdef EXL2_TRACE
    args.trace = (g_trace_buf && g_trace_count++ == g_trace_which) ? g_trace_buf : nullptr;
#endif
    u32 lds = 0;
    int blk_max = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        GemvJob& j = args.job[i];
        JobHot& h = args.hot[i];
        j = jobs[i];
        const QMatDev& m = j.m;
        j.tile0 = 0;
        const int blks = (m.N / TILE_N + TPW - 1) / TPW;
        if (blks > blk_max) blk_max = blks;
        j.rows_per_phase = m.K;
        j.a_stride = m.K + 8;
        u32 a_bytes = align16s((u32)M * j.a_stride * 2);
        const u32 red_bytes = (u32)W * 16 * 16 * 4;
        if (a_bytes < red_bytes) a_bytes = red_bytes;
        const u32 row_bytes = align16s((u32)M * m.K * 2);
        const bool two = j.a_mode == A_SILU_MUL || j.a_mode == A_GELU_MUL;
        u32 total = a_bytes;
        h.lds_scale_off = total;  total += align16s((u32)TPW * m.G * 32);
        h.lds_zp_off = total;     total += gptq ? align16s((u32)TPW * m.G * 32) : 0;
        h.lds_rmf_off = total;    total += 64 + 16 * 16 * 4;
        {
            // rows staged through LDS: the whole pack image [q_perm][chunk -> group map] + raw rows; many-rows route (gather
            // from global memory): only the chunk -> group map part of the pack lives in LDS
            const u32 raw = row_bytes + (two ? row_bytes : (j.a_mode == A_RMSNORM ? align16s((u32)m.K * 2) : 0));
            if (total + m.pack_units * 16 + raw <= 160 * 1024 && !no_lds_stage)
            {
                h.lds_perm_off = total;   total += m.pack_units * 16;
                h.lds_cg_off = h.lds_perm_off + m.pack_cg_off;
                h.lds_rawx_off = total;   total += row_bytes;
                h.lds_raw2_off = total;   total += two ? row_bytes : (j.a_mode == A_RMSNORM ? align16s((u32)m.K * 2) : 0);
            }
            else
            {
                h.lds_perm_off = 0;
                h.lds_cg_off = total;     total += m.pack_units * 16 - m.pack_cg_off;
                h.lds_rawx_off = 0; h.lds_raw2_off = 0;
            }
        }
        if (total > lds) lds = total;

        const QRun& mr = m.runs[m.main_run];
        h.main_ptr = (mr.in_tail ? m.tail : m.qw) + mr.base_word;
        h.main_tile_stride = mr.tile_stride; h.main_F = (int)mr.n_super; h.main_chunk0 = (int)mr.k_base >> 5;
        h.main_bits = mr.nvalid_last == 4 ? (int)mr.bits : 0;
        h.pack = m.pack; h.pack_units = m.pack_units; h.pack_cg_off = m.pack_cg_off; h.sc_tab = m.sc_tab; h.zp_tab = m.zp_tab;
        h.perm = m.perm;
        h.a = j.a; h.a2 = j.a2; h.norm_w = j.norm_w;
        h.tile0 = j.tile0; h.n_tiles = m.N / TILE_N; h.K = m.K; h.G = m.G; h.N = m.N; h.lda = j.lda; h.a_mode = j.a_mode;
        h.a_stride = j.a_stride; h.norm_eps = j.norm_eps; h.n_runs = m.n_runs;
        h.r_weights = j.r_weights; h.r_stride = j.r_stride;
    }
    if (lds > 160 * 1024) return 1;
    bool mixed = false;
    for (int i = 0; i < n_jobs; i++) if (args.hot[i].main_bits != mb) mixed = true;
    dim3 grid((unsigned)blk_max, (unsigned)n_jobs, 1), block((unsigned)(W * 64), 1, 1);
    if (gptq)
    {
        if (mb == 4) launch_variant<true, 4>(args, grid, block, lds, stream);
        else         launch_variant<true, 0>(args, grid, block, lds, stream);
    }
    else
    {
        switch (mb)
        {
            case 4: launch_variant<false, 4>(args, grid, block, lds, stream, mixed); break;
            case 8: launch_variant<false, 8>(args, grid, block, lds, stream, mixed); break;
            case 6: launch_variant<false, 6>(args, grid, block, lds, stream, mixed); break;
            case 5: launch_variant<false, 5>(args, grid, block, lds, stream, mixed); break;
            case 3: launch_variant<false, 3>(args, grid, block, lds, stream, mixed); break;
            case 2: launch_variant<false, 2>(args, grid, block, lds, stream, mixed); break;
            default: launch_variant<false, 0>(args, grid, block, lds, stream); break;
        }
    }
    return 0;
}
