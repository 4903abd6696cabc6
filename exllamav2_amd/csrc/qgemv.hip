// qgemv.hip -- skinny (decode) q_matrix x fp16 kernel for gfx950.
//
// Replaces the reference's gemm_half_q_half_kernel / gemm_half_q_half_gptq_kernel
// (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565, q_gemm_kernel_gptq.cuh:61-246) for M <= 16 rows.
//
// Design:
//   * one workgroup owns one 16-column tile for the WHOLE K: no split-K across workgroups, no atomics (the reference
//     combines 64 K-slices with CAS-emulated fp16 atomics), deterministic fp32 reduction order;
//   * the wavefronts of the workgroup take super-chunks (128 K-rows x 16 columns = one 64-lane x b-dword load)
//     round-robin; every vector load instruction of a wave reads one contiguous run of the re-laid weight stream,
//     non-temporal, straight to VGPRs (weights are read exactly once);
//   * a lane's b dwords decode with the magic-number half2 trick into the B fragment of v_mfma_f32_16x16x32_f16; the
//     activation rows (M <= 16, gathered through q_perm once per workgroup into LDS) are the A fragment.  The matrix
//     core does the 2*M*16*32 flops of a chunk in 4 passes whatever M is, so M = 1..16 cost the same VALU work;
//   * the B operand is the exact (q - zero) in fp16; the group scale multiplies the fp32 partial sum (qgemv_common.h):
//     gemm(I) still equals reconstruct() bit for bit, general results equal matmul(x, reconstruct()) within fp16
//     rounding of the individual weights;
//   * optional prologue fusions while the activations are staged: RMSNorm (rms_norm.cu numerics) or SiLU(gate)*up;
//     optional epilogue: bias, accumulate into the residual (c += a*W), MoE routing weight.
#include "qgemv_common.h"
#include "errors.h"
#include <stdlib.h>

// ---- a wave's work list: the super-chunks g = g0 + wv, g0 + wv + nw, ... of the descriptors [d, de) ------------------
// [d, de) is a run of descriptors with the SAME bit width (consecutive descriptors of one section are one contiguous
// stream; they are only split so that M > 1 can stage the activations in phases).

struct Item { const u32* ptr; int chunk0; int nvalid; };

struct Cursor
{
    const QDesc* desc; const u32* qw; const u32* tail; int tile;
    int d, de;          // current / end descriptor
    int g, step;        // next global super-chunk index of this wave, stride
};

// advance to this wave's next item; false when the list is exhausted.  Everything here is wave-uniform (SALU).
template <int BITS>
DEV bool next_item(Cursor& c, Item& it)
{
    while (c.d < c.de)
    {
        const QDesc* dp = c.desc + c.d;
        const int pre = uniform((int)dp->sc_prefix), n = uniform((int)dp->n_super);
        if (c.g >= pre + n) { c.d++; continue; }
        const int s = c.g - pre;
        it.nvalid = (s == n - 1) ? uniform((int)dp->nvalid_last) : 4;
        it.chunk0 = (uniform((int)dp->k_base) >> 5) + 4 * s;
        it.ptr = (uniform((int)dp->in_tail) ? c.tail : c.qw) + uniform(dp->base_word)
                 + (size_t)c.tile * uniform(dp->tile_stride) + (size_t)s * (64 * BITS);
        c.g += c.step;
        return true;
    }
    return false;
}

// Stream the wave's items of one run in batches of four: the four vector loads of a batch are issued back to back,
// unconditionally (missing items of the last batch re-read the batch's first item -- cache hits, results unused), so
// the code is straight-line and the compiler's counted vmcnt lets item i decode while items i+1.. are still in flight.
template <int BITS, bool GPTQ>
DEV void gemv_run(Cursor cur, const PhaseCtx& ph, int lane, f32x4& acc)
{
    for (;;)
    {
        Item it[4];
        int v = 0;
        #pragma unroll
        for (int i = 0; i < 4; i++)
        {
            if (v == i && next_item<BITS>(cur, it[i])) v = i + 1;
            else it[i] = it[0];
        }
        if (v == 0) return;
        LaneWords<BITS> w[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) load_lane_words<BITS>(it[i].ptr, lane, w[i]);
        #pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < v) gemv_super<BITS, GPTQ, true>(w[i], ph, it[i].chunk0, 4, lane, acc);
        if (v < 4) return;
    }
}

// a partial super-chunk (at most one per bit-width section, its own descriptor): rare, no pipelining
template <int BITS, bool GPTQ>
DEV void gemv_tail(Cursor cur, const PhaseCtx& ph, int lane, f32x4& acc)
{
    Item it;
    LaneWords<BITS> w;
    while (next_item<BITS>(cur, it))
    {
        load_lane_words<BITS>(it.ptr, lane, w);
        gemv_super<BITS, GPTQ, false>(w, ph, it.chunk0, it.nvalid, lane, acc);
    }
}

template <int BITS, bool GPTQ>
DEV void gemv_dispatch(const Cursor& cur, bool partial, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if (partial) gemv_tail<BITS, GPTQ>(cur, ph, lane, acc);
    else         gemv_run<BITS, GPTQ>(cur, ph, lane, acc);
}

DEV int desc_rows(const QDesc* d)
{
    return d->in_tail ? (int)d->nvalid_last * 32 : (int)d->n_super * SUPER_ROWS;
}

template <bool GPTQ>
KERNEL void __launch_bounds__(1024) qgemv_kernel(const GemvArgs args)
{
    DYN_SMEM(smem);

    // which job / tile
    int ji = 0;
    #pragma unroll
    for (int i = 1; i < MAX_FUSED_MATS; i++)
        if (i < args.n_jobs && bid_x() >= args.job[i].tile0) ji = i;
    const GemvJob& job = args.job[ji];
    const QMatDev& m = job.m;
    const int tile = bid_x() - job.tile0;
    const int row0 = bid_y() * MAX_GEMV_ROWS;                 // row block (M > 16 handled by grid.y)
    const int M = min(args.M - row0, MAX_GEMV_ROWS);

    const int t = tid();
    const int nt = nthreads();
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int nw = nt >> 6;

    f16* a_lds  = (f16*)smem;
    f16* sc_lds = (f16*)(smem + job.lds_scale_off);
    f16* zp_lds = (f16*)(smem + job.lds_zp_off);
    float* red  = (float*)smem;                               // aliases a_lds after the last phase

    const f16* a  = job.a  + (size_t)row0 * job.lda;
    const f16* a2 = job.a2 ? job.a2 + (size_t)row0 * job.lda : nullptr;

    QDesc* desc_lds = (QDesc*)(smem + job.lds_desc_off);
    u16* cg_lds = (u16*)(smem + job.lds_cg_off);
    float* rmf_lds = (float*)(smem + job.lds_rmf_off) + wv * 16;          // per-wave copy: no block barrier needed

    // ---- prologue, part 1: every load below is independent of the others -> one memory latency for all of them ------
    //   descriptors and the chunk->group map -> LDS (the streaming loop must not issue any vector-memory load besides
    //   the weights); per-group scales / zero points of this tile's 16 columns -> LDS [G][16]
    {
        const u32* src = (const u32*)m.desc;
        u32* dst = (u32*)desc_lds;
        for (int i = t; i < m.n_desc * (int)(sizeof(QDesc) / 4); i += nt) dst[i] = src[i];
        for (int i = t; i < (m.K >> 5); i += nt) cg_lds[i] = m.chunk_group[i];
        const int n8 = m.N >> 3;
        for (int idx = t; idx < m.G * 16; idx += nt)
        {
            const int g = idx >> 4, c = idx & 15;
            const int n = tile * 16 + c;
            const u32 word = m.q_scale[(size_t)g * n8 + (n >> 3)];
            const int nib = (word >> (4 * (n & 7))) & 15;
            if constexpr (GPTQ)
            {
                sc_lds[idx] = m.scale_src[(size_t)g * m.N + n];
                zp_lds[idx] = (f16)(float)(nib + 1);           // (q - (zero + 1)) * scale, q_matrix.cu:265-270
            }
            else
            {
                sc_lds[idx] = (f16)(float)((nib + 1) * (nib + 1)) * m.scale_src[g];      // qdq_util.cuh:24-30
            }
        }
    }

    // ---- RMSNorm statistics (rms_norm.cu:68-76,118): every wave reduces the whole row itself (L1/L2 hits), which
    //      costs K/512 16-byte loads per lane and saves two workgroup barriers ------------------------------------------
    if (job.a_mode == A_RMSNORM)
    {
        for (int r = 0; r < M; r++)
        {
            const f16x8* xr = (const f16x8*)(a + (size_t)r * job.lda);
            float ss = 0.0f;
            for (int i = lane; i < (m.K >> 3); i += 64)
            {
                const f16x8 v = xr[i];
                #pragma unroll
                for (int e = 0; e < 8; e++) { const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f)); ss = fmaf(f, f, ss); }
            }
            ss = wave_allreduce_add(ss);
            rmf_lds[r] = fast_rsqrt(ss * (1.0f / (float)m.K) + job.norm_eps);   // every lane stores the same value
        }
    }

    // ---- phases over K ------------------------------------------------------------------------------------------------
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    PhaseCtx ph;
    ph.a_lds = a_lds; ph.sc_lds = sc_lds; ph.zp_lds = zp_lds; ph.cg_lds = cg_lds; ph.a_stride = job.a_stride; ph.M = M;

    int di = 0;
    bool first = true;
    while (di < m.n_desc)
    {
        int k0, rows, de;
        if (job.rows_per_phase >= m.K) { k0 = 0; rows = m.K; de = m.n_desc; }          // one phase (always for M = 1)
        else
        {
            k0 = uniform((int)m.desc[di].k_base);
            rows = 0; de = di;
            while (de < m.n_desc)
            {
                const int r = uniform(desc_rows(m.desc + de));
                if (de > di && rows + r > job.rows_per_phase) break;
                rows += r; de++;
            }
        }

        if (!first) block_sync();              // previous phase's A reads are done
        first = false;

        // stage a[:, perm[k0 .. k0+rows)] into LDS, 8 K-values per thread per step; the mode is uniform and hoisted so
        // the 8 gathers of a step are issued back to back (one latency, not eight)
        const int oct = rows >> 3;
        switch (job.a_mode)
        {
            case A_PLAIN:    stage_rows<A_PLAIN>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
            case A_RMSNORM:  stage_rows<A_RMSNORM>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
            case A_SILU_MUL: stage_rows<A_SILU_MUL>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
            case A_GELU_MUL: stage_rows<A_GELU_MUL>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
            case A_SILU:     stage_rows<A_SILU>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
            default:         stage_rows<A_GELU>(job, m, a, a2, a_lds, rmf_lds, k0, oct, M, t, nt); break;
        }
        block_sync();

        ph.phase_k0 = k0;
        // runs of descriptors with equal bit width (partial super-chunks are runs of their own)
        for (int d = di; d < de; )
        {
            const int bits = uniform((int)desc_lds[d].bits);
            const bool partial = uniform((int)desc_lds[d].nvalid_last) != 4;
            int e = d + 1;
            if (!partial)
                while (e < de && uniform((int)desc_lds[e].bits) == bits && uniform((int)desc_lds[e].nvalid_last) == 4) e++;
            Cursor cur;
            cur.desc = desc_lds; cur.qw = m.qw; cur.tail = m.tail; cur.tile = tile;
            cur.d = d; cur.de = e; cur.step = nw;
            cur.g = uniform((int)desc_lds[d].sc_prefix) + wv;
            if constexpr (GPTQ) gemv_dispatch<4, true>(cur, partial, ph, lane, acc);
            else
            {
                switch (bits)
                {
                    case 4: gemv_dispatch<4, false>(cur, partial, ph, lane, acc); break;
                    case 8: gemv_dispatch<8, false>(cur, partial, ph, lane, acc); break;
                    case 6: gemv_dispatch<6, false>(cur, partial, ph, lane, acc); break;
                    case 5: gemv_dispatch<5, false>(cur, partial, ph, lane, acc); break;
                    case 3: gemv_dispatch<3, false>(cur, partial, ph, lane, acc); break;
                    default: gemv_dispatch<2, false>(cur, partial, ph, lane, acc); break;
                }
            }
            d = e;
        }
        di = de;
    }

    // ---- deterministic cross-wave reduction + epilogue -----------------------------------------------------------------
    block_sync();
    {
        const int c = lane & 15, j = lane >> 4;
        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int row = j * 4 + r;
            if (row < M) red[(wv * 16 + row) * 16 + c] = acc[r];
        }
    }
    block_sync();
    if (t < M * 16)
    {
        const int row = t >> 4, c = t & 15;
        float v = 0.0f;
        for (int w = 0; w < nw; w++) v += red[(w * 16 + row) * 16 + c];
        const int n = tile * 16 + c;
        const int grow = row0 + row;
        bool skip = false;
        if (job.r_weights)
        {
            const f16 rw = job.r_weights[(size_t)grow * job.r_stride];
            if (as_u16(rw) == 0) skip = true;                       // q_gemm_kernel.cuh:189-200: zero weight -> no-op
            if (job.mul_r_weights) v *= (float)rw;
        }
        if (!skip)
        {
            if (m.bias) v += (float)m.bias[n];
            f16* cp = job.c + (size_t)grow * job.ldc + (job.c_invperm ? (int)job.c_invperm[n] : n);
            if (job.c_mode == C_ACCUM) v += (float)*cp;
            *cp = (f16)v;
        }
    }
}

// ---- host launcher ---------------------------------------------------------------------------------------------------

static int device_cus()
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

static int pick_waves(long long tiles, int max_super_per_tile)
{
    // enough wavefronts in flight to cover HBM latency (~16+ waves/CU over 256 CUs) without starving each wave of work
    const int g_num_cus = device_cus();
    const char* force = getenv("EXL2_GEMV_WAVES");
    if (force && atoi(force) > 0) return atoi(force);
    // keep >= ~16 wavefronts per CU streaming; prefer several small workgroups per CU (their prologues overlap each
    // other's streaming) and only widen the workgroup when there are too few column tiles to go round
    const long long target = (long long)g_num_cus * 16;
    int w = 4;
    while (w < 16 && tiles * w < target && w * 2 <= max_super_per_tile) w *= 2;
    return w;
}

static inline u32 align16(u32 x) { return (x + 15u) & ~15u; }

// Fills the LDS layout of a job for M rows; returns dynamic LDS bytes needed (excluding nothing).
static u32 plan_job_lds(GemvJob& j, int M, bool gptq, int nwaves)
{
    const int budget_halfs = (64 * 1024) / 2;                 // activation staging budget per phase
    int rpp = budget_halfs / (M < 1 ? 1 : M);
    rpp -= 8;
    rpp = (rpp / SUPER_ROWS) * SUPER_ROWS;
    if (rpp < QDESC_MAX_SUPER * SUPER_ROWS) rpp = QDESC_MAX_SUPER * SUPER_ROWS;
    if (rpp > j.m.K) rpp = ((j.m.K + 31) / 32) * 32;
    j.rows_per_phase = rpp;
    j.a_stride = rpp + 8;
    u32 a_bytes = align16((u32)M * j.a_stride * 2);
    const u32 red_bytes = (u32)nwaves * 16 * 16 * 4;
    if (a_bytes < red_bytes) a_bytes = red_bytes;
    j.lds_scale_off = a_bytes;
    j.lds_zp_off = j.lds_scale_off + align16((u32)j.m.G * 32);            // [G][16] halfs
    u32 total = j.lds_zp_off + (gptq ? align16((u32)j.m.G * 32) : 0);
    j.lds_cg_off = total;
    total += align16((u32)(j.m.K >> 5) * 2);
    j.lds_rmf_off = total;
    total += 16 * 16 * 4;                                                 // rmf[wave][16]
    j.lds_desc_off = total;
    total += align16((u32)j.m.n_desc * (u32)sizeof(QDesc));
    return total;
}

// Launch up to MAX_FUSED_MATS jobs (same M, same format family) as one grid.
int qgemv_stream_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream);
int qgemm_prefill_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream);

int qgemv_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream)
{
    if (n_jobs < 1 || n_jobs > MAX_FUSED_MATS || M < 1) EXL2_FAIL(EXL2_E_INVALID, "q_gemm: %d fused matrices / %d rows not launchable", n_jobs, M);
    {
        // decode-shaped calls go to the streaming kernel (qgemv_stream.hip); this generic kernel handles the rest
        const int rc = qgemv_stream_launch(jobs, n_jobs, M, gptq, stream);
        if (rc <= 0) return rc;
    }
    if (M > MAX_GEMV_ROWS)
    {
        // prefill-shaped calls: dequantize-into-MFMA GEMM (qgemm_prefill.hip)
        const int rc = qgemm_prefill_launch(jobs, n_jobs, M, gptq, stream);
        if (rc <= 0) return rc;
    }
    GemvArgs args;
    args.n_jobs = n_jobs;
    args.M = M;
    long long tiles = 0;
    int max_super = 1 << 30;
    for (int i = 0; i < n_jobs; i++)
    {
        const int k_super = (jobs[i].m.K + SUPER_ROWS - 1) / SUPER_ROWS;
        if (k_super < max_super) max_super = k_super;
        tiles += jobs[i].m.N / TILE_N;
    }
    const int Mb = M < MAX_GEMV_ROWS ? M : MAX_GEMV_ROWS;
    const int nwaves = pick_waves(tiles, max_super);
    u32 lds = 0;
    int tile0 = 0;
    for (int i = 0; i < n_jobs; i++)
    {
        args.job[i] = jobs[i];
        args.job[i].tile0 = tile0;
        tile0 += jobs[i].m.N / TILE_N;
        const u32 l = plan_job_lds(args.job[i], Mb, gptq, nwaves);
        if (l > lds) lds = l;
    }
    if (lds > 160 * 1024) EXL2_FAIL(EXL2_E_UNSUPPORTED, "q_gemm: %u bytes of LDS needed, 160 KB available", lds);
    dim3 grid((unsigned)tiles, (unsigned)((M + MAX_GEMV_ROWS - 1) / MAX_GEMV_ROWS), 1);
    dim3 block(nwaves * 64, 1, 1);
    static bool attr_set[2][EXL2_MAX_DEVICES] = {{false}};
    if (exl2_first_on_device(attr_set[gptq ? 1 : 0]))
    {
        if (gptq) hipFuncSetAttribute((const void*)qgemv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        else      hipFuncSetAttribute((const void*)qgemv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (gptq) LAUNCH(qgemv_kernel<true>, grid, block, lds, stream, args);
    else      LAUNCH(qgemv_kernel<false>, grid, block, lds, stream, args);
    return 0;
}
