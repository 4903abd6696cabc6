// qgemv_stream.hip -- the decode-shaped q_gemm kernel (M <= 16 rows, activations fit in LDS in one piece).
//
// Replaces gemm_half_q_half_kernel (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565) on the hot decode path; the generic
// kernel in qgemv.hip stays for shapes that need phased activation staging.
//
// What bounds a 10-90 MB GEMV on MI355X is not ALU but (1) bytes in flight per CU and (2) the number of DEPENDENT memory
// round trips between launch and the last store (~1 us each when the data is HBM-cold).  So:
//   * a wavefront streams a CONTIGUOUS slice of one 16-column tile (the tile16 layout makes a tile's K range one linear
//     stream) with a 4-deep register ring: loads are issued unconditionally in the steady state, so the compiler's
//     counted vmcnt keeps three 1-KB loads per wave in flight while the fourth decodes;
//   * the first ring fill of the largest bit-width section is issued BEFORE the prologue (run parameters travel in the
//     kernel arguments = SGPRs, nothing has to be fetched to compute the addresses), so weights, q_perm, scales and
//     descriptors are all in flight together: launch -> {everything} -> gather a[perm] -> decode -> reduce -> store;
//   * a workgroup of W waves owns W/S tiles, S waves splitting each tile's K range; the activation vector is gathered
//     through q_perm into LDS once per workgroup and shared by all its tiles; the S partial sums of a tile are combined
//     through LDS in a fixed order (deterministic; no atomics, no cross-workgroup traffic);
//   * decode = magic-number half2 unpack (qlayout.h) -> exact (q - zero) * scale in fp16 like reconstruct() -> B fragment
//     of v_mfma_f32_16x16x32_f16, fp32 accumulate; RMSNorm / SiLU*up are folded into the activation staging, bias /
//     residual / MoE weight into the epilogue (qgemv_common.h).
#include "qgemv_common.h"
#include <stdlib.h>
#include <string.h>

struct StreamArgs
{
    GemvJob job[MAX_FUSED_MATS];
    int n_jobs;
    int M;          // rows (<= MAX_GEMV_ROWS)
    int S;          // waves per tile (power of two)
    int TPW;        // tiles per workgroup = waves / S
};

// ---- streaming one wave's slice of one run ---------------------------------------------------------------------------

template <int BITS> DEV void ring_load(LaneWords<BITS>& b, const u32* p, int lane) { load_lane_words<BITS>(p, lane, b); }

// items [0, n) at ptr0 + i * 64 * BITS words, chunk index chunk0 + 4 i.  `b` may already hold items 0..min(n,4)-1.
template <int BITS, bool GPTQ>
DEV void stream_items(const u32* ptr0, int n, int chunk0, const PhaseCtx& ph, int lane, f32x4& acc,
                      LaneWords<BITS> (&b)[4], bool preloaded)
{
    constexpr size_t STEP = 64 * BITS;
    if (n <= 0) return;
    if (!preloaded)
    {
        #pragma unroll
        for (int u = 0; u < 4; u++) if (u < n) ring_load<BITS>(b[u], ptr0 + (size_t)u * STEP, lane);
    }
    int i = 0;
    // steady state: every load is unconditional -> exact vmcnt(3 * loads per item) before each decode
    while (i + 8 <= n)
    {
        #pragma unroll
        for (int u = 0; u < 4; u++)
        {
            gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
            ring_load<BITS>(b[u], ptr0 + (size_t)(i + u + 4) * STEP, lane);
        }
        i += 4;
    }
    // drain: at most 7 items left, the ring holds items i .. min(i + 4, n) - 1
    #pragma unroll
    for (int u = 0; u < 4; u++)
    {
        if (i + u < n) gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
        if (i + u + 4 < n) ring_load<BITS>(b[u], ptr0 + (size_t)(i + u + 4) * STEP, lane);
    }
    i += 4;
    #pragma unroll
    for (int u = 0; u < 4; u++)
        if (i + u < n) gemv_super<BITS, GPTQ, true>(b[u], ph, chunk0 + 4 * (i + u), 4, lane, acc);
}

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
    LaneWords<BITS> b[4];
    stream_items<BITS, GPTQ>(s.ptr0, s.n, s.chunk0, ph, lane, acc, b, false);
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

// MB = bit width of the main (largest) run, whose first ring fill is issued ahead of the prologue; 0 = no early fill
template <bool GPTQ, int MB>
KERNEL void __launch_bounds__(1024) qgemv_stream_kernel(const StreamArgs args)
{
    DYN_SMEM(smem);

    int ji = 0;
    #pragma unroll
    for (int i = 1; i < MAX_FUSED_MATS; i++)
        if (i < args.n_jobs && bid_x() >= args.job[i].tile0) ji = i;
    const GemvJob& job = args.job[ji];
    const QMatDev& m = job.m;
    const int M = args.M;
    const int S = args.S;

    const int t = tid();
    const int nt = nthreads();
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int nw = nt >> 6;
    const int gidx = wv / S;                                  // tile slot inside the workgroup
    const int r = wv - gidx * S;                              // K-slice of that tile
    const int n_tiles = m.N / TILE_N;
    int tile = (bid_x() - job.tile0) * args.TPW + gidx;
    const bool tile_ok = tile < n_tiles;
    if (!tile_ok) tile = n_tiles - 1;                         // idle slot: compute on a valid tile, never store

    f16* a_lds  = (f16*)smem;
    f16* sc_lds = (f16*)(smem + job.lds_scale_off) + (size_t)gidx * m.G * 16;
    f16* zp_lds = (f16*)(smem + job.lds_zp_off) + (size_t)gidx * m.G * 16;
    u16* cg_lds = (u16*)(smem + job.lds_cg_off);
    float* rmf_lds = (float*)(smem + job.lds_rmf_off) + wv * 16;
    float* red  = (float*)smem;                               // aliases a_lds after the streaming

    // ---- early ring fill of the main run (addresses come from kernel arguments only) ---------------------------------
    LaneWords<(MB ? MB : 4)> pre[4];
    RunSlice ms; ms.n = 0; ms.ptr0 = nullptr; ms.chunk0 = 0;
    if constexpr (MB != 0)
    {
        ms = slice_of(m.runs[m.main_run], m, tile, r, S);
        #pragma unroll
        for (int u = 0; u < 4; u++) if (u < ms.n) ring_load<MB>(pre[u], ms.ptr0 + (size_t)u * (64 * MB), lane);
    }

    // ---- prologue: independent loads first (chunk map, this tile's group scales), then the activation gather ----------
    for (int i = t; i < (m.K >> 5); i += nt) cg_lds[i] = m.chunk_group[i];
    {
        const int n8 = m.N >> 3;
        const int lt = r * 64 + lane;                         // thread index inside the tile's S-wave group
        for (int idx = lt; idx < m.G * 16; idx += S * 64)
        {
            const int g = idx >> 4, c = idx & 15;
            const int n = tile * 16 + c;
            const u32 word = m.q_scale[(size_t)g * n8 + (n >> 3)];
            const int nib = (word >> (4 * (n & 7))) & 15;
            if constexpr (GPTQ)
            {
                sc_lds[idx] = m.scale_src[(size_t)g * m.N + n];
                zp_lds[idx] = (f16)(float)(nib + 1);
            }
            else
            {
                sc_lds[idx] = (f16)(float)((nib + 1) * (nib + 1)) * m.scale_src[g];
            }
        }
    }
    if (job.a_mode == A_RMSNORM)
    {
        for (int rr = 0; rr < M; rr++)
        {
            const f16x8* xr = (const f16x8*)(job.a + (size_t)rr * job.lda);
            float ss = 0.0f;
            for (int i = lane; i < (m.K >> 3); i += 64)
            {
                const f16x8 v = xr[i];
                #pragma unroll
                for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
            }
            ss = wave_allreduce_add(ss);
            rmf_lds[rr] = fast_rsqrt(ss * (1.0f / (float)m.K) + job.norm_eps);
        }
    }
    {
        const int oct = m.K >> 3;
        switch (job.a_mode)
        {
            case A_PLAIN:    stage_rows<A_PLAIN>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
            case A_RMSNORM:  stage_rows<A_RMSNORM>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
            case A_SILU_MUL: stage_rows<A_SILU_MUL>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
            case A_GELU_MUL: stage_rows<A_GELU_MUL>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
            case A_SILU:     stage_rows<A_SILU>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
            default:         stage_rows<A_GELU>(job, m, job.a, job.a2, a_lds, rmf_lds, 0, oct, M, t, nt); break;
        }
    }
    block_sync();

    // ---- stream -------------------------------------------------------------------------------------------------------
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    PhaseCtx ph;
    ph.a_lds = a_lds; ph.sc_lds = sc_lds; ph.zp_lds = zp_lds; ph.cg_lds = cg_lds; ph.a_stride = job.a_stride;
    ph.M = M; ph.phase_k0 = 0;

    if constexpr (MB != 0) stream_items<MB, GPTQ>(ms.ptr0, ms.n, ms.chunk0, ph, lane, acc, pre, true);
    for (int i = 0; i < m.n_runs; i++)
    {
        if (MB != 0 && i == m.main_run) continue;
        do_run_any<GPTQ>(m.runs[i], m, tile, r, S, ph, lane, acc);
    }

    // ---- combine the S slices of every tile (fixed order) + epilogue ----------------------------------------------------
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
    for (int idx = t; idx < args.TPW * M * 16; idx += nt)
    {
        const int slot = idx / (M * 16);
        const int rem = idx - slot * (M * 16);
        const int row = rem >> 4, c = rem & 15;
        const int tl = (bid_x() - job.tile0) * args.TPW + slot;
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
            f16* cp = job.c + (size_t)row * job.ldc + n;
            if (job.c_mode == C_ACCUM) v += (float)*cp;
            *cp = (f16)v;
        }
    }
}

// ---- host --------------------------------------------------------------------------------------------------------------

static inline u32 align16s(u32 x) { return (x + 15u) & ~15u; }

static int num_cus()
{
    static int n = 0;
    if (n <= 0)
    {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <bool GPTQ, int MB>
static void launch_variant(const StreamArgs& args, dim3 grid, dim3 block, u32 lds, void* stream)
{
    static bool attr = false;
    if (!attr)
    {
        (void)hipFuncSetAttribute((const void*)qgemv_stream_kernel<GPTQ, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    LAUNCH((qgemv_stream_kernel<GPTQ, MB>), grid, block, lds, stream, args);
}

// returns 0 when launched, 1 when this kernel does not apply (caller falls back to the generic kernel), < 0 on error
int qgemv_stream_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream)
{
    if (n_jobs < 1 || n_jobs > MAX_FUSED_MATS || M < 1) return -1;
    if (M > MAX_GEMV_ROWS) return 1;
    const char* off = getenv("EXL2_GEMV_GENERIC");
    if (off && atoi(off)) return 1;
    long long tiles = 0;
    int min_items = 1 << 30, mb = -1;
    for (int i = 0; i < n_jobs; i++)
    {
        const QMatDev& m = jobs[i].m;
        if (m.n_runs <= 0) return 1;
        if ((long long)M * (m.K + 8) * 2 > 96 * 1024) return 1;                 // activations must fit in LDS in one piece
        tiles += m.N / TILE_N;
        const QRun& mr = m.runs[m.main_run];
        const int items = mr.nvalid_last == 4 ? (int)mr.n_super : 0;
        if (items < min_items) min_items = items;
        const int b = items > 0 ? (int)mr.bits : 0;
        if (mb < 0) mb = b; else if (mb != b) mb = 0;
    }
    if (mb < 0) mb = 0;

    // S waves per tile: enough wavefronts to keep ~12 per CU streaming, but at least ~4 super-chunks per wave
    const long long want = (long long)num_cus() * 12;
    int S = 1;
    while (S < 16 && tiles * S < want && min_items / (S * 2) >= 4) S *= 2;
    const char* fs = getenv("EXL2_GEMV_SPLIT");
    if (fs && atoi(fs) > 0) S = atoi(fs);
    int W = S > 8 ? S : 8;
    const char* fw = getenv("EXL2_GEMV_WAVES");
    if (fw && atoi(fw) >= S) W = atoi(fw);
    const int TPW = W / S;

    StreamArgs args;
    memset(&args, 0, sizeof(args));
    args.n_jobs = n_jobs; args.M = M; args.S = S; args.TPW = TPW;
    u32 lds = 0;
    int blk0 = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        GemvJob& j = args.job[i];
        j = jobs[i];
        j.tile0 = blk0;
        blk0 += (j.m.N / TILE_N + TPW - 1) / TPW;
        j.rows_per_phase = j.m.K;
        j.a_stride = j.m.K + 8;
        u32 a_bytes = align16s((u32)M * j.a_stride * 2);
        const u32 red_bytes = (u32)W * 16 * 16 * 4;
        if (a_bytes < red_bytes) a_bytes = red_bytes;
        j.lds_scale_off = a_bytes;
        j.lds_zp_off = j.lds_scale_off + align16s((u32)TPW * j.m.G * 32);
        u32 total = j.lds_zp_off + (gptq ? align16s((u32)TPW * j.m.G * 32) : 0);
        j.lds_cg_off = total;   total += align16s((u32)(j.m.K >> 5) * 2);
        j.lds_rmf_off = total;  total += 16 * 16 * 4;
        j.lds_desc_off = total;
        if (total > lds) lds = total;
    }
    if (lds > 160 * 1024) return 1;
    dim3 grid((unsigned)blk0, 1, 1), block((unsigned)(W * 64), 1, 1);
    if (gptq)
    {
        if (mb == 4) launch_variant<true, 4>(args, grid, block, lds, stream);
        else         launch_variant<true, 0>(args, grid, block, lds, stream);
    }
    else
    {
        switch (mb)
        {
            case 4: launch_variant<false, 4>(args, grid, block, lds, stream); break;
            case 8: launch_variant<false, 8>(args, grid, block, lds, stream); break;
            case 6: launch_variant<false, 6>(args, grid, block, lds, stream); break;
            case 5: launch_variant<false, 5>(args, grid, block, lds, stream); break;
            case 3: launch_variant<false, 3>(args, grid, block, lds, stream); break;
            case 2: launch_variant<false, 2>(args, grid, block, lds, stream); break;
            default: launch_variant<false, 0>(args, grid, block, lds, stream); break;
        }
    }
    return 0;
}
