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
//   * about one workgroup per CU: a workgroup owns a CONTIGUOUS range of 16-column tiles and its waves split that
//     range's super-chunks evenly (a wave's share may straddle two tiles), so every wave streams the same number of
//     bytes whatever the shape; the activation vector is gathered through q_perm into LDS once per workgroup and all
//     group-scale tables of the range are built in the same round trip; per-tile partial sums are combined through
//     LDS in a fixed wave order (deterministic; no atomics, no cross-workgroup traffic);
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
    int M;          // rows (<= MAX_STREAM_ROWS)
    int probe;      // profiling aid (EXL2_GEMV_PROBE bit mask): 1 skip weight streaming, 2 skip activation staging,
                    // 8 skip the early ring fill; results are wrong when 1 or 2 is set
};
#define MAX_STREAM_ROWS 8
#define MAX_WAVE_TILES 4          // tiles one wave's share may touch (host guarantees it)

// ---- streaming a contiguous stretch of one run ------------------------------------------------------------------------

template <int BITS> DEV void ring_load(LaneWords<BITS>& b, const u32* p, int lane) { load_lane_words<BITS>(p, lane, b); }

// items [0, n) at ptr0 + i * 64 * BITS words, chunk index chunk0 + 4 i.  With `preloaded`, b[0 .. min(n,4)) are in flight.
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

// a stretch of `cnt` consecutive super-chunks of run `run`, starting at super-chunk s, of tile `tile`
struct Stretch { QRun run; int s, cnt; };        // run by VALUE: pointers into the kernel-argument block would force it to scratch

DEV const u32* stretch_ptr(const Stretch& st, const QMatDev& m, int tile)
{
    const QRun& r = st.run;
    return (r.in_tail ? m.tail : m.qw) + r.base_word + (size_t)tile * r.tile_stride + (size_t)st.s * (64u * r.bits);
}

template <int BITS, bool GPTQ>
DEV void do_stretch(const Stretch& st, const QMatDev& m, int tile, const PhaseCtx& ph, int lane, f32x4& acc)
{
    const QRun& r = st.run;
    const u32* p = stretch_ptr(st, m, tile);
    const int chunk0 = ((int)r.k_base >> 5) + 4 * st.s;
    if (r.nvalid_last != 4)
    {
        LaneWords<BITS> w;                                   // partial super-chunk: a run of its own, one item
        load_lane_words<BITS>(p, lane, w);
        gemv_super<BITS, GPTQ, false>(w, ph, chunk0, (int)r.nvalid_last, lane, acc);
        return;
    }
    LaneWords<BITS> b[4];
    stream_items<BITS, GPTQ>(p, st.cnt, chunk0, ph, lane, acc, b, false);
}

template <bool GPTQ>
DEV void do_stretch_any(const Stretch& st, const QMatDev& m, int tile, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if constexpr (GPTQ) do_stretch<4, true>(st, m, tile, ph, lane, acc);
    else
    {
        switch (st.run.bits)
        {
            case 4: do_stretch<4, false>(st, m, tile, ph, lane, acc); break;
            case 8: do_stretch<8, false>(st, m, tile, ph, lane, acc); break;
            case 6: do_stretch<6, false>(st, m, tile, ph, lane, acc); break;
            case 5: do_stretch<5, false>(st, m, tile, ph, lane, acc); break;
            case 3: do_stretch<3, false>(st, m, tile, ph, lane, acc); break;
            default: do_stretch<2, false>(st, m, tile, ph, lane, acc); break;
        }
    }
}

// position `off` (0 <= off < items per tile) -> run and super-chunk inside it; limit = items left in that run
DEV Stretch locate(const QMatDev& m, int off)
{
    Stretch st; st.run = m.runs[0]; st.s = 0; st.cnt = 0;
    int acc_items = 0;
    for (int i = 0; i < m.n_runs; i++)
    {
        const int n = (int)m.runs[i].n_super;
        if (off < acc_items + n) { st.run = m.runs[i]; st.s = off - acc_items; st.cnt = n - st.s; return st; }
        acc_items += n;
    }
    return st;
}

// MB = bit width whose first ring fill is issued ahead of the prologue when a wave's share starts in such a run; 0 = off
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

    const int t = tid();
    const int nt = nthreads();
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int nw = nt >> 6;
    const int n_tiles = m.N / TILE_N;
    const int tpw = job.tiles_per_wg;
    const int t0 = (bid_x() - job.tile0) * tpw;               // first tile of this workgroup
    const int ntl = min(tpw, n_tiles - t0);                   // tiles of this workgroup (>= 1)
    const int ipt = job.items_per_tile;
    const int tot = ntl * ipt;
    const int i0 = (int)(((long long)wv * tot) / nw), i1 = (int)(((long long)(wv + 1) * tot) / nw);   // this wave's share
    const int first_tile = i0 / ipt;                          // local index of the first tile this wave touches

    f16* a_lds  = (f16*)smem;
    f16* sc_all = (f16*)(smem + job.lds_scale_off);           // [tile_local][G][16]
    f16* zp_all = (f16*)(smem + job.lds_zp_off);
    u16* cg_lds = (u16*)(smem + job.lds_cg_off);
    float* rmf_lds = (float*)(smem + job.lds_rmf_off) + wv * 16;
    float* red  = (float*)(smem + job.lds_red_off);           // [wave][MAX_WAVE_TILES][M][16]

    // ---- early ring fill: this wave's first stretch, when it lies in an MB-bit full run (addresses from SGPRs only) -----
    LaneWords<(MB ? MB : 4)> pre[4];
    Stretch es = locate(m, i0 - first_tile * ipt);
    es.cnt = min(es.cnt, i1 - i0);
    bool early = false;
    if constexpr (MB != 0)
    {
        early = i1 > i0 && es.run.bits == MB && es.run.nvalid_last == 4 && !(args.probe & 9);
        if (early)
        {
            const u32* p = stretch_ptr(es, m, t0 + first_tile);
            #pragma unroll
            for (int u = 0; u < 4; u++) if (u < es.cnt) ring_load<MB>(pre[u], p + (size_t)u * (64 * MB), lane);
        }
    }

    // ---- prologue, step 1: ISSUE every independent load (chunk map, the group scales of all tiles of this workgroup,
    //      q_perm vectors of this thread's staging slots) before consuming any of them: together with the early ring fill
    //      they cost ONE memory round trip.  Fixed slot counts keep it straight-line; leftovers take the loops below. ------
    constexpr int NSC = 4;
    const int n8 = m.N >> 3;
    const int oct = m.K >> 3;
    const int n_sc = ntl * m.G * 16;
    u32 scw[NSC]; f16 scm[NSC];
    #pragma unroll
    for (int u = 0; u < NSC; u++)
    {
        const int idx = t + u * nt;
        scw[u] = 0; scm[u] = (f16)0.0f;
        if (idx < n_sc)
        {
            const int tl = idx / (m.G * 16), rem = idx - tl * (m.G * 16);
            const int g = rem >> 4, n = (t0 + tl) * 16 + (rem & 15);
            scw[u] = m.q_scale[(size_t)g * n8 + (n >> 3)];
            scm[u] = GPTQ ? m.scale_src[(size_t)g * m.N + n] : m.scale_src[g];
        }
    }
    u16 cgv[2] = {0, 0};
    #pragma unroll
    for (int u = 0; u < 2; u++) { const int i = t + u * nt; if (i < (m.K >> 5)) cgv[u] = m.chunk_group[i]; }
    u32x4 pv[STAGE_SLOTS];
    #pragma unroll
    for (int u = 0; u < STAGE_SLOTS; u++)
    {
        const int idx = t + u * nt;
        pv[u] = (u32x4){0, 0, 0, 0};
        if (m.perm && idx < M * oct) pv[u] = *(const u32x4*)(m.perm + (idx % oct) * 8);
    }

    // RMSNorm statistics (rms_norm.cu:68-76,118): every wave reduces the whole row itself from L2 (no workgroup barrier)
    if (job.a_mode == A_RMSNORM)
    {
        for (int rr = 0; rr < M; rr++)
        {
            const f16x8* xr = (const f16x8*)(job.a + (size_t)rr * job.lda);
            float ss = 0.0f;
            for (int j0 = 0; j0 < oct; j0 += 8 * 64)
            {
                f16x8 xv[8];
                #pragma unroll
                for (int u = 0; u < 8; u++) { const int i = j0 + u * 64 + lane; xv[u] = i < oct ? xr[i] : (f16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
                #pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    #pragma unroll
                    for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)xv[u][e], 65504.0f)); ss = fmaf(f, f, ss); }
                }
            }
            ss = wave_allreduce_add(ss);
            rmf_lds[rr] = fast_rsqrt(ss * (1.0f / (float)m.K) + job.norm_eps);
        }
    }

    // ---- prologue, step 2: consume.  Tables -> LDS, then the activation gather through the pre-loaded permutation ------
    #pragma unroll
    for (int u = 0; u < 2; u++) { const int i = t + u * nt; if (i < (m.K >> 5)) cg_lds[i] = cgv[u]; }
    for (int i = t + 2 * nt; i < (m.K >> 5); i += nt) cg_lds[i] = m.chunk_group[i];
    #pragma unroll
    for (int u = 0; u < NSC; u++)
    {
        const int idx = t + u * nt;
        if (idx < n_sc)
        {
            const int n = (idx % (m.G * 16)) & 15;            // column inside the tile; tiles start at multiples of 16
            const int nib = (scw[u] >> (4 * (n & 7))) & 15;
            if constexpr (GPTQ) { sc_all[idx] = scm[u]; zp_all[idx] = (f16)(float)(nib + 1); }       // q_matrix.cu:265-270
            else sc_all[idx] = (f16)(float)((nib + 1) * (nib + 1)) * scm[u];                         // qdq_util.cuh:24-30
        }
    }
    for (int idx = t + NSC * nt; idx < n_sc; idx += nt)
    {
        const int tl = idx / (m.G * 16), rem = idx - tl * (m.G * 16);
        const int g = rem >> 4, n = (t0 + tl) * 16 + (rem & 15);
        const int nib = (m.q_scale[(size_t)g * n8 + (n >> 3)] >> (4 * (n & 7))) & 15;
        if constexpr (GPTQ) { sc_all[idx] = m.scale_src[(size_t)g * m.N + n]; zp_all[idx] = (f16)(float)(nib + 1); }
        else sc_all[idx] = (f16)(float)((nib + 1) * (nib + 1)) * m.scale_src[g];
    }
    if (!(args.probe & 2))
    switch (job.a_mode)
    {
        case A_PLAIN:    stage_preloaded<A_PLAIN>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
        case A_RMSNORM:  stage_preloaded<A_RMSNORM>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
        case A_SILU_MUL: stage_preloaded<A_SILU_MUL>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
        case A_GELU_MUL: stage_preloaded<A_GELU_MUL>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
        case A_SILU:     stage_preloaded<A_SILU>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
        default:         stage_preloaded<A_GELU>(job, m, job.a, job.a2, a_lds, rmf_lds, oct, M, t, nt, pv); break;
    }
    block_sync();

    // ---- stream this wave's share: stretches of runs, tile after tile --------------------------------------------------
    PhaseCtx ph;
    ph.a_lds = a_lds; ph.cg_lds = cg_lds; ph.a_stride = job.a_stride; ph.M = M; ph.phase_k0 = 0;
    {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const int c = lane & 15, j = lane >> 4;
        int cur = i0;
        bool first = true;
        while (cur < i1 && !(args.probe & 1))
        {
            const int tl = cur / ipt;
            Stretch st = locate(m, cur - tl * ipt);
            st.cnt = min(st.cnt, i1 - cur);
            ph.sc_lds = sc_all + (size_t)tl * m.G * 16;
            ph.zp_lds = zp_all + (size_t)tl * m.G * 16;
            if (MB != 0 && first && early)
                stream_items<(MB ? MB : 4), GPTQ>(stretch_ptr(st, m, t0 + tl), st.cnt, ((int)st.run.k_base >> 5) + 4 * st.s,
                                                 ph, lane, acc, pre, true);
            else
                do_stretch_any<GPTQ>(st, m, t0 + tl, ph, lane, acc);
            first = false;
            cur += st.cnt;
            if (cur == i1 || cur - tl * ipt == ipt)
            {
                // tile finished (or share exhausted): park the partial sums of this tile
                float* rp = red + ((size_t)(wv * MAX_WAVE_TILES + (tl - first_tile)) * M) * 16;
                #pragma unroll
                for (int q = 0; q < 4; q++) { const int row = j * 4 + q; if (row < M) rp[row * 16 + c] = acc[q]; }
                acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    }

    // ---- combine the partial sums of every tile in wave order + epilogue -------------------------------------------------
    block_sync();
    for (int idx = t; idx < ntl * M * 16; idx += nt)
    {
        const int tl = idx / (M * 16);
        const int rem = idx - tl * (M * 16);
        const int row = rem >> 4, c = rem & 15;
        float v = 0.0f;
        if (!(args.probe & 1))
        for (int w = 0; w < nw; w++)
        {
            const int w0 = (int)(((long long)w * tot) / nw), w1 = (int)(((long long)(w + 1) * tot) / nw);
            if (w1 > w0 && w0 < (tl + 1) * ipt && w1 > tl * ipt)
                v += red[((size_t)(w * MAX_WAVE_TILES + (tl - w0 / ipt)) * M + row) * 16 + c];
        }
        const int n = (t0 + tl) * 16 + c;
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
    if (!attr && lds > 64 * 1024)
    {
        (void)hipFuncSetAttribute((const void*)qgemv_stream_kernel<GPTQ, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    LAUNCH((qgemv_stream_kernel<GPTQ, MB>), grid, block, lds, stream, args);
}

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && atoi(v) > 0) ? atoi(v) : dflt; }

// returns 0 when launched, 1 when this kernel does not apply (caller falls back to the generic kernel), < 0 on error
int qgemv_stream_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream)
{
    if (n_jobs < 1 || n_jobs > MAX_FUSED_MATS || M < 1) return -1;
    if (M > MAX_STREAM_ROWS) return 1;
    if (env_int("EXL2_GEMV_GENERIC", 0)) return 1;

    long long items_total = 0;
    int ipt[MAX_FUSED_MATS];
    int mb = -1;
    for (int i = 0; i < n_jobs; i++)
    {
        const QMatDev& m = jobs[i].m;
        if (m.n_runs <= 0) return 1;
        if ((long long)M * (m.K + 8) * 2 > 96 * 1024) return 1;                 // activations must fit in LDS in one piece
        ipt[i] = 0;
        for (int r = 0; r < m.n_runs; r++) ipt[i] += m.runs[r].n_super;
        items_total += (long long)ipt[i] * (m.N / TILE_N);
        const QRun& mr = m.runs[m.main_run];
        const int b = mr.nvalid_last == 4 ? (int)mr.bits : 0;
        if (mb < 0) mb = b; else if (mb != b) mb = 0;
    }
    if (mb < 0) mb = 0;

    // ~one workgroup per CU, W waves each; every workgroup owns a contiguous tile range of ONE matrix
    const int W = env_int("EXL2_GEMV_WAVES", 16);
    const int target_wgs = env_int("EXL2_GEMV_WGS", num_cus());
    StreamArgs args;
    memset(&args, 0, sizeof(args));
    args.n_jobs = n_jobs; args.M = M;
    args.probe = env_int("EXL2_GEMV_PROBE", 0);
    u32 lds = 0;
    int blk0 = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        GemvJob& j = args.job[i];
        j = jobs[i];
        const int tiles = j.m.N / TILE_N;
        long long share = ((long long)target_wgs * ipt[i] * tiles + items_total - 1) / items_total;     // workgroups of this job
        if (share < 1) share = 1;
        if (share > tiles) share = tiles;
        int tpw = (int)((tiles + share - 1) / share);
        // a wave's share (tpw * ipt / W items) must not touch more than MAX_WAVE_TILES tiles
        while (tpw > 1 && ((long long)tpw * ipt[i] / W) / ipt[i] + 2 > MAX_WAVE_TILES) tpw--;
        j.tiles_per_wg = tpw;
        j.items_per_tile = ipt[i];
        j.tile0 = blk0;
        blk0 += (tiles + tpw - 1) / tpw;
        j.rows_per_phase = j.m.K;
        j.a_stride = j.m.K + 8;
        u32 total = align16s((u32)M * j.a_stride * 2);
        j.lds_scale_off = total;  total += align16s((u32)tpw * j.m.G * 32);
        j.lds_zp_off = total;     total += gptq ? align16s((u32)tpw * j.m.G * 32) : 0;
        j.lds_cg_off = total;     total += align16s((u32)(j.m.K >> 5) * 2);
        j.lds_rmf_off = total;    total += 16 * 16 * 4;
        j.lds_red_off = total;    total += (u32)W * MAX_WAVE_TILES * M * 16 * 4;
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
