// qgemm_prefill.hip -- host driver of the many-row q_matrix x fp16 products (more than 16 rows) and their row pre-pass.
//
// Replaces the reference's M > 32 path: reconstruct_kernel (cuda/q_matrix.cu:328-553) writes the whole fp16 [K, N] matrix to HBM, then
// cuBLAS / hipBLAS Hgemm reads it back (cuda/q_gemm.cu:243-263).  Here the packed weights are the B operand of the matrix cores:
//   * the activations are made "packed-K ordered" once per call by a row pre-pass (stage_rows_kernel: act-order gather through q_perm,
//     fused with RMSNorm or act(gate) * up where the caller has them), so the GEMM kernels read A tiles as contiguous 16-byte units;
//   * 17 .. 128 rows: qgemm_skinny.hip (weight-stream bound; q | k | v and gate | up share one launch each);
//   * more rows: qgemm_mfma.hip (256 x 256 tiles, weights decoded once per call).
// (The round-1 128 x 128 register-decode kernel that lived here was retired in round 6: qgemm_skinny.hip covers its row counts 2-5 x
// faster, matrices with more bit-width sections than the argument block holds included.)
#include "qgemm_prefill.h"
#include "errors.h"
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <mutex>

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
            // 17 .. 128 rows: the weight-stream-bound kernel (qgemm_skinny.hip), the following jobs over the SAME staged rows
            // (q | k | v, gate | up) in the same launch; more rows (or EXL2_PREFILL_MFMA_MIN_ROWS <= rows: tests): qgemm_mfma.hip
            if (rows <= 128 && !(mfma_min_rows > 0 && rows >= mfma_min_rows))
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
                // (declined: K beyond the kernel's LDS / buffer window -- the many-row kernel takes any row count)
            }
            { const int rc = qgemm_mfma_launch(p, gptq, stream); if (rc) return rc; }
        }
    }
    return 0;
}
