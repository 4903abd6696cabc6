// modules.hip -- fused module forwards (QAttn, QMLP) and HIP-graph helpers, + C ABI.
//
// Reference: QAttn (q_attn.cu:84-345, bindings ext_qattn.cpp:24-191), QMLP (q_mlp.cu:17-236, bindings
// ext_qmlp.cpp:87-118).  The reference runs rms_norm -> q_gemm x3 -> rope as 5 launches (replayed from a CUDA graph) and
// norm -> gate, up q_gemm -> act_mul -> down q_gemm as 5 more.  Here, for rows <= 16 (decode):
//   attention part 1 = ONE launch (RMSNorm folded into the activation staging of a fused q|k|v kernel) + RoPE,
//   MLP              = TWO launches (RMSNorm folded into gate|up; SiLU(gate)*up folded into down's staging, residual
//                      added in down's epilogue).
// Whole-token graphs are captured by the host with exl2_graph_* around these calls (all positions are read on the
// device), instead of the reference's per-module graphs with patched kernel arguments (graph.cu:141-164).
#include "qmatrix.h"
#include "qgemv_flat.h"
#include "qgemv_lean.h"
#include "chain_sync.h"
#include "moe.h"
#include "errors.h"
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>

// A/B and test switches of the MoE block (EXL2_MOE_*, EXL2_DEBUG_ROUTE): read ONCE per process -- q_moe_mlp_forward_ runs per layer and
// token on the eager route behind the drop-in (round-5 advisor: no getenv per call) -- unless EXL2_ENV_DYNAMIC is set when the library is
// first used (tests/conftest.py sets it: the tests flip these switches inside one process)
#define ENV_SET(name) ([]() -> bool { static const bool dyn = getenv("EXL2_ENV_DYNAMIC") != nullptr; static const bool v = getenv(name) != nullptr; \
                                      return dyn ? getenv(name) != nullptr : v; }())

int qgemv_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream);

extern "C" {
int exl2_rms_norm(const void* x, const void* w, void* y, float epsilon, int rows, int dim,
                  int add_residual, int input_fp32, int output_fp32, void* stream);
int exl2_rope_qk(void* x_q, void* x_k, const void* sin, const void* cos, int batch_size,
                 int rows_per_batch_q, int rows_per_batch_k, int head_dim, int num_heads_q, int num_heads_k,
                 int past_len, const int* past_lens, int neox_style, int sincos_size, void* stream);
int exl2_act_mul(void* x, const void* y, int rows, int width, int act_gelu,
                 const void* r_weights, int r_weights_stride, void* stream);
int exl2_moe_route(const void* x, const void* gate, void* logits, int rows, int hidden, int num_experts, int topk, void* stream);
int exl2_publish_rows(const void* x, int rows, int hidden, const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, void* stream);
// (library-internal, moe.hip: not part of include/exl2_hip.h -- the boundary is q_moe_mlp_forward_)
int exl2_moe_front(const void* x, const void* norm_w, const void* gate, const void* perm, void* xn, void* xg, void* logits,
                   int rows, int hidden, int num_experts, int topk, float eps, void* stream);
}

static void fill_job(GemvJob& j, const QMatrix* qm, const f16* a, f16* c, int a_mode, int c_mode)
{
    memset(&j, 0, sizeof(j));
    j.m = qm->dev;
    j.a = a; j.c = c;
    j.lda = qm->height; j.ldc = qm->width;
    j.a_mode = a_mode; j.c_mode = c_mode;
}

#define LAUNCH_JOBS(jobs, n, M, gptq, stream, what) do { \
    const int _rc = qgemv_launch(jobs, n, M, gptq, stream); \
    if (_rc < 0) return _rc;            /* the launcher has left its own message (staging OOM, capture, ...) */ \
    if (_rc > 0) EXL2_FAIL(EXL2_E_INVALID, "%s: launch configuration rejected (%d)", what, _rc); \
    HIP_TRY(hipGetLastError()); } while (0)

// ---- QAttn ----------------------------------------------------------------------------------------------------------

struct QAttn
{
    const f16* layernorm; const f16* layernorm_bias; bool layernorm_is_rms; bool headnorm_is_rms; float norm_epsilon;
    QMatrix* q_proj; QMatrix* k_proj; QMatrix* v_proj; QMatrix* o_proj;
    f16* temp_state; f16* temp_dq;
    int max_rows, hidden_size, num_heads, num_kv_heads, head_dim, max_seq_len;
    bool has_residual; int rope_style; int sincos_size;
    const f16* q_norm; const f16* k_norm; const f16* post_layernorm; const f16* post_layernorm_bias;
    bool residual_fp32; bool use_graphs;
    bool chain_ok; f16* norm_w_perm;            // chained decode: layernorm gathered through q/k/v's shared q_perm (owned)
    bool qkv_same_perm;                         // q, k, v share one act-order permutation (prefill: rows staged once)
};

struct QMLP
{
    const f16* layernorm; const f16* layernorm_bias; bool layernorm_is_rms; float norm_epsilon;
    QMatrix* gate; QMatrix* up; QMatrix* down;
    f16* temp_state; f16* temp_a; f16* temp_b; f16* temp_dq;
    int max_rows; bool act_gelu; bool has_residual;
    const f16* post_layernorm; const f16* post_layernorm_bias; bool residual_fp32; bool use_graphs;
    bool chain_ok; f16* norm_w_perm;            // chained decode: layernorm gathered through gate/up's shared q_perm (owned)
    bool gu_same_perm;                          // gate and up share one act-order permutation (prefill: rows staged once)
};

// true when every matrix carries the same act-order permutation (or none does)
static bool same_perm(QMatrix* const* ms, int n)
{
    const int K = ms[0]->height;
    bool any = false, all = true;
    for (int i = 0; i < n; i++) { if (ms[i]->q_perm) any = true; else all = false; if (ms[i]->height != K) return false; }
    if (!any) return true;
    if (!all) return false;
    std::vector<u16> a(K), b(K);
    if (hipMemcpy(a.data(), ms[0]->q_perm, (size_t)K * 2, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (int i = 1; i < n; i++)
    {
        if (ms[i]->q_perm == ms[0]->q_perm) continue;
        if (hipMemcpy(b.data(), ms[i]->q_perm, (size_t)K * 2, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (memcmp(a.data(), b.data(), (size_t)K * 2) != 0) return false;
    }
    return true;
}

// norm weight in the packed order of `qm` (make time; small): dst[i] = w[perm[i]]
static f16* permuted_norm(const f16* w, const QMatrix* qm)
{
    const int K = qm->height;
    std::vector<u16> wh(K), ph(K), out(K);
    if (hipMemcpy(wh.data(), w, (size_t)K * 2, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    if (qm->q_perm)
    {
        if (hipMemcpy(ph.data(), qm->q_perm, (size_t)K * 2, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        for (int i = 0; i < K; i++) out[i] = wh[ph[i] < K ? ph[i] : 0];
    }
    else out = wh;
    f16* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)K * 2) != hipSuccess) return nullptr;
    if (hipMemcpy(d, out.data(), (size_t)K * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return d;
}

#define MOE_MAX_EXPERTS 16
struct QMoEMLP;
static bool hidden_ok(const QMoEMLP* m);
struct QMoEMLP
{
    const f16* layernorm; float norm_epsilon;
    const f16* gate; int num_experts, num_experts_per_token;
    QMatrix* w1[MOE_MAX_EXPERTS]; QMatrix* w2[MOE_MAX_EXPERTS]; QMatrix* w3[MOE_MAX_EXPERTS];
    f16* temp_state; f16* temp_a; f16* temp_b; f16* temp_logits;
    int max_rows, hidden; bool act_gelu;
    bool group_ok;                              // every expert's w1 / w3 share one act-order permutation: grouped launches
    // one row (round 6): the selected experts on the lean kernel -- argument blocks planned here, at load time, per expert; the front
    // kernel of a step copies the selected ones to the slots the gate|up and the down launch read (qgemv_lean.h: LeanGroupPlan)
    bool lean_ok; LeanGroupPlan lean_gu, lean_dn;
    bool lean_rows_ok[LEAN_MAX_M + 1]; LeanGroupPlan lean_gu_r[LEAN_MAX_M + 1], lean_dn_r[LEAN_MAX_M + 1];      // 2-4 rows: one launch over all experts each (the reference's fused form covers <= 4 rows, q_mlp.cu:316-436)
    bool lean_sum_ok; LeanGroupPlan lean_dn2;      // top-2: both selected experts' down projections as ONE pair launch that adds the weighted sum to x (no combine launch)
};

// ---- small kernels of the grouped MoE path -----------------------------------------------------------------------------------
// rows of src gathered through a u16 permutation: dst[r, i] = src[r, perm[i]] (activations into the experts' packed K order)
KERNEL void __launch_bounds__(256) gather_rows_f16_kernel(const f16* src, const u16* perm, f16* dst, int K)
{
    const int row = bid_y();
    const int i = bid_x() * 256 + tid();
    if (i < K) dst[(size_t)row * K + i] = src[(size_t)row * K + (perm ? (int)perm[i] : i)];
}
// x[r, :] += sum over the experts e row r is routed to of part[e, r, :] (already weighted), fp32 sum in expert order, one rounding.
// co (round 6, the block inside a chained step): the finished rows are published for the next consumer as a chained producer's
// epilogue does (qgemv_lean.hip: ep_finish) -- xp = x * that consumer's norm weight in its packed order (one rounding, saturated),
// one partial sum of squares of x per workgroup
struct MoeChainOut { f16* xp; const u16* invperm; const f16* w; float* ss; int ldxp, tiled; };
KERNEL void __launch_bounds__(256) moe_combine_kernel(f16* x, const f16* part, const f16* weights, int rows, int hidden, int E, const MoeChainOut co)
{
    SHARED float sq_part[4];
    const int row = bid_y();
    const int o = bid_x() * 256 + tid();
    const bool on = o * 8 < hidden;
    float sq = 0.0f;
    if (on)
    {
        // two round trips to memory, not one per expert: (1) the row's weights, x, the permutation entries; (2) the selected experts'
        // parts and the next consumer's norm weights -- then the sum in expert order as before (a part that is not selected is not
        // added: -0 + 0 would flip a sign bit)
        f16x8 xv = ((const f16x8*)(x + (size_t)row * hidden))[o];
        u32 mask = 0;
        #pragma unroll
        for (int e = 0; e < MOE_MAX_EXPERTS; e++) if (e < E && as_u16(weights[(size_t)row * E + e]) != 0) mask |= 1u << e;
        u32 xi[8];
        const bool vec_perm = co.xp && co.invperm && (((size_t)co.invperm) & 15) == 0;
        if (vec_perm)
        {
            const u32x4 pv = ((const u32x4*)co.invperm)[o];
            xi[0] = pv.x & 0xFFFFu; xi[1] = pv.x >> 16; xi[2] = pv.y & 0xFFFFu; xi[3] = pv.y >> 16;
            xi[4] = pv.z & 0xFFFFu; xi[5] = pv.z >> 16; xi[6] = pv.w & 0xFFFFu; xi[7] = pv.w >> 16;
        }
        else
        {
            #pragma unroll
            for (int k = 0; k < 8; k++) xi[k] = (co.xp && co.invperm) ? (u32)co.invperm[o * 8 + k] : (u32)(o * 8 + k);
        }
        const f16x8 z8 = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
        f16x8 pv[MOE_MAX_EXPERTS];
        #pragma unroll
        for (int e = 0; e < MOE_MAX_EXPERTS; e++) pv[e] = ((mask >> e) & 1u) ? ((const f16x8*)(part + ((size_t)e * rows + row) * hidden))[o] : z8;
        f16 wn[8];
        #pragma unroll
        for (int k = 0; k < 8; k++) wn[k] = (co.xp && co.w) ? co.w[xi[k]] : (f16)1.0f;
        float acc[8];
        #pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (float)xv[k];
        #pragma unroll
        for (int e = 0; e < MOE_MAX_EXPERTS; e++)
        {
            if ((mask >> e) & 1u)
            {
                #pragma unroll
                for (int k = 0; k < 8; k++) acc[k] += (float)pv[e][k];
            }
        }
        #pragma unroll
        for (int k = 0; k < 8; k++) xv[k] = (f16)acc[k];
        ((f16x8*)(x + (size_t)row * hidden))[o] = xv;
        if (co.xp)
        {
            #pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const float f = fmaxf(-65504.0f, fminf((float)xv[k], 65504.0f));
                const f16 yw = co.w ? (f16)fmaxf(-65504.0f, fminf(f * (float)wn[k], 65504.0f)) : xv[k];
                f16* const dst = co.tiled ? co.xp + ((size_t)(xi[k] >> 3) * 16 + row) * 8 + (xi[k] & 7) : co.xp + (size_t)row * co.ldxp + xi[k];
                *dst = yw;
                sq = fmaf(f, f, sq);
            }
        }
    }
    if (co.xp && co.ss)
    {
        sq = wave_allreduce_add(sq);
        if (lane_id() == 0) sq_part[wave_id()] = sq;
        block_sync();
        if (tid() == 0) co.ss[(size_t)row * gdim_x() + bid_x()] = sq_part[0] + sq_part[1] + sq_part[2] + sq_part[3];
    }
}

static bool hidden_ok(const QMoEMLP* m) { return (m->hidden & 7) == 0; }

// ---- overlapped chain (chain_sync.h) ---------------------------------------------------------------------------------------------
struct ChainSyncState { u32* flags; int n_blocks; void* stream[2]; int next; bool open; };
static thread_local ChainSyncState g_chain = {nullptr, 0, {nullptr, nullptr}, 0, false};

bool chain_sync_active() { return g_chain.open; }
static thread_local int g_chain_tiled = 0;      // per host thread, like g_chain: one decoder per thread may switch its layout
bool chain_xp_tiled() { return g_chain_tiled != 0; }

static u32* chain_block(int k) { return g_chain.flags + (size_t)k * SYNC_BLOCK_WORDS; }

int chain_sync_next(ChainLaunch* out)
{
    EXL2_REQUIRE(g_chain.open, "chain: no overlapped chain is open");
    EXL2_REQUIRE(g_chain.next < g_chain.n_blocks - 1, "chain: more than %d launches in one overlapped chain", g_chain.n_blocks - 1);
    const int k = g_chain.next;
    out->wait = k > 0 ? chain_block(k - 1) : nullptr;
    out->signal = chain_block(k);
    static const bool gates = []() { const char* e = getenv("EXL2_CHAIN_GATES"); return !(e && e[0] == '0'); }();
    out->arrive = gates ? chain_block(k) : nullptr;  // every workgroup reports on entry (sync_report_entry): the gate below counts them
    out->stream = g_chain.stream[k & 1];
    return EXL2_OK;
}

// one wave, in the OTHER stream, behind launch k - 1 and ahead of launch k + 1: holds launch k + 1 back until every workgroup of
// launch k has entered (chain_sync.h); zeroes the entry counters again
KERNEL void __launch_bounds__(64) chain_gate_kernel(u32* producer_block, u32 target) { sync_gate_wait(producer_block, target); }
// end of a chain: nobody waits for the last launch's "go" -- back to zero for the next replay (runs behind it in its stream)
KERNEL void __launch_bounds__(64) chain_tail_kernel(u32* block)
{
    if (lane_id() < SYNC_GO_COPIES) store_relaxed_agent((u32*)sync_go_word(block, lane_id()), 0u);
}

int chain_sync_done(u32 arrivals)
{
    EXL2_REQUIRE(arrivals > 0, "chain: a launch of an overlapped chain must say how many workgroups it has");
    const int k = g_chain.next;
    // EXL2_CHAIN_GATES=0 (measurement only): no gates -- the entry counters are then cleared by a gate-less clear kernel at the end
    static const bool gates = []() { const char* e = getenv("EXL2_CHAIN_GATES"); return !(e && e[0] == '0'); }();
    if (gates) LAUNCH(chain_gate_kernel, dim3(1, 1, 1), dim3(64, 1, 1), 0, g_chain.stream[(k + 1) & 1], chain_block(k), arrivals);
    HIP_TRY(hipGetLastError());
    g_chain.next++;
    return EXL2_OK;
}

static long long g_route_lean = 0, g_route_flat = 0;        // launches taken by qgemv_lean.hip / qgemv_flat.hip (exl2_chain_route_counts)

static bool lean_enabled()
{
    const char* e = getenv("EXL2_LEAN");
    return !(e && e[0] == '0');
}

// One chained launch.  While an overlapped chain is open (chain_sync.h) the launch takes its stream and its counters from it.
static int flat_try(FlatIn& in, void* stream, int* wgs, const char* what)
{
    ChainLaunch cl = {nullptr, nullptr, nullptr, stream};
    const bool overlapped = chain_sync_active();
    if (overlapped)
    {
        const int e = chain_sync_next(&cl);
        if (e) return e;
        in.sync_wait = cl.wait; in.sync_signal = cl.signal; in.sync_arrive = cl.arrive;
    }
    int n_wgs = 0;
    // round 3: the lean kernel (qgemv_lean.hip) takes what it covers (<= LEAN_MAX_M rows, serial chain); EXL2_LEAN=0 keeps
    // the round-2 kernel for A/B runs
    int rc = 1;
    if (lean_enabled()) rc = qgemv_lean_launch(in, cl.stream, &n_wgs);
    if (rc == 0) g_route_lean++;
    // (the overlapped chain's protocol -- entry reports, sharded counters -- is the lean kernel's: what it declines ends the chain)
    if (rc == 1 && !overlapped) { rc = qgemv_flat_launch(in, cl.stream, &n_wgs); if (rc == 0) g_route_flat++; }
    if (rc > 0) EXL2_FAIL(EXL2_E_INVALID, "%s: shape not covered by the chained decode kernel", what);
    if (rc < 0) EXL2_FAIL(EXL2_E_INVALID, "%s: launch configuration rejected (%d)", what, rc);
    HIP_TRY(hipGetLastError());
    // (the gate of the next launch counts this launch's whole grid: the lean kernel puts its matrices on blockIdx.y)
    if (overlapped) { const int e = chain_sync_done((u32)n_wgs * (u32)(in.pair ? 1 : in.n_mats)); if (e) return e; }
    if (wgs) *wgs = n_wgs;
    return EXL2_OK;
}
#define FLAT_TRY(in, stream, wgs, what) do { const int _rc = flat_try(in, stream, wgs, what); if (_rc) return _rc; } while (0)

extern "C" {

// make_q_moe_mlp (ext_qmlp.h; call site moe_mlp.py:114-133: w1 = gate proj, w3 = up proj, w2 = down proj)
// the one-row route's plans: gate|up as a pair per expert (output in the expert's down_proj order), down weighted by the routing weight;
// the same buffers as the grouped route at rows = 1
static void moe_plan_lean(QMoEMLP* m)
{
    m->lean_ok = false;
    const int E = m->num_experts, hidden = m->hidden, inter = m->w1[0]->width, rows = 1;
    if (!(m->group_ok && 2 * MAX_GEMV_ROWS <= m->max_rows && (long long)E * rows <= m->max_rows &&
          (long long)E * rows * hidden <= (long long)m->max_rows * inter) || m->num_experts_per_token > MOE_MAX_SEL) return;
    if (getenv("EXL2_MOE_NO_LEAN_PLAN")) return;
    f16* const xg = m->temp_state + (size_t)MAX_GEMV_ROWS * hidden;
    FlatIn ins[MOE_MAX_EXPERTS];
    for (int e = 0; e < E; e++)
    {
        FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
        in.qm[0] = m->w1[e]; in.qm[1] = m->w3[e];
        in.c[0] = m->temp_a + (size_t)e * rows * inter; in.c[1] = in.c[0]; in.ldc[0] = inter; in.ldc[1] = inter;
        in.c_invperm[0] = m->w2[e]->q_perm ? m->w2[e]->q_invperm : nullptr;
        in.n_mats = 2; in.pair = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = xg; in.lda = hidden; in.c_mode = C_STORE;
        in.act_gelu = m->act_gelu ? 1 : 0;
    }
    if (qgemv_lean_group_plan(ins, E, nullptr, m->num_experts_per_token, &m->lean_gu) != 0) return;
    const f16* scale[MOE_MAX_EXPERTS];
    for (int e = 0; e < E; e++)
    {
        FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
        in.qm[0] = m->w2[e]; in.c[0] = m->temp_b + (size_t)e * rows * hidden; in.ldc[0] = hidden;
        in.n_mats = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = m->temp_a + (size_t)e * rows * inter; in.lda = inter; in.c_mode = C_STORE;
        scale[e] = m->temp_logits + e;
    }
    if (qgemv_lean_group_plan(ins, E, scale, m->num_experts_per_token, &m->lean_dn) != 0) { qgemv_lean_group_free(&m->lean_gu); return; }
    m->lean_ok = true;
    // 2-4 rows: every expert's blocks for that row count; a launch covers all experts, workgroups of experts without a row leave at entry
    // (built, parity-green on the MI355X, measured SLOWER than the grouped launches of qgemv_flat.hip at Mixtral's widths -- bs = 2 / 3 / 4:
    // 537 / 749 / 983 against 577 / 836 / 1104 tok/s, profiles/r09fg_moe_rows_2_4.txt: with 4-7 experts active both routes stream at
    // ~5 TB/s and the lean launches' 16-wave pair workgroups are alone on their CU -- so it is planned and taken only under
    // EXL2_MOE_LEAN_ROWS=1, set when the module is made)
    const char* const rows_env = getenv("EXL2_MOE_LEAN_ROWS");
    for (int r = 2; r <= LEAN_MAX_M && rows_env && atoi(rows_env) != 0; r++)
    {
        if (!(r <= MAX_GEMV_ROWS && (long long)E * r <= m->max_rows && (long long)E * r * hidden <= (long long)m->max_rows * inter)) break;
        for (int e = 0; e < E; e++)
        {
            FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
            in.qm[0] = m->w1[e]; in.qm[1] = m->w3[e];
            in.c[0] = m->temp_a + (size_t)e * r * inter; in.c[1] = in.c[0]; in.ldc[0] = inter; in.ldc[1] = inter;
            in.c_invperm[0] = m->w2[e]->q_perm ? m->w2[e]->q_invperm : nullptr;
            in.n_mats = 2; in.pair = 1; in.M = r; in.a_mode = A_DIRECT; in.a = xg; in.lda = hidden; in.c_mode = C_STORE;
            in.act_gelu = m->act_gelu ? 1 : 0;
            scale[e] = m->temp_logits + e;
        }
        if (qgemv_lean_group_plan(ins, E, scale, 1, &m->lean_gu_r[r], E, 0) != 0) continue;
        for (int e = 0; e < E; e++)
        {
            FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
            in.qm[0] = m->w2[e]; in.c[0] = m->temp_b + (size_t)e * r * hidden; in.ldc[0] = hidden;
            in.n_mats = 1; in.M = r; in.a_mode = A_DIRECT; in.a = m->temp_a + (size_t)e * r * inter; in.lda = inter; in.c_mode = C_STORE;
        }
        if (qgemv_lean_group_plan(ins, E, scale, 1, &m->lean_dn_r[r], E, 1) != 0) { qgemv_lean_group_free(&m->lean_gu_r[r]); continue; }
        m->lean_rows_ok[r] = true;
    }
    if (m->num_experts_per_token == 2 && !getenv("EXL2_MOE_NO_SUM_PLAN"))
    {
        for (int e = 0; e < E; e++)
        {
            FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
            in.qm[0] = m->w2[e]; in.qm[1] = m->w2[e]; in.ldc[0] = hidden; in.ldc[1] = hidden;
            in.n_mats = 2; in.pair = 1; in.pair_sum = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = m->temp_a + (size_t)e * rows * inter; in.lda = inter;
            in.c_mode = C_ACCUM;
        }
        m->lean_sum_ok = qgemv_lean_group_plan(ins, E, scale, 1, &m->lean_dn2) == 0;
    }
}

int exl2_make_q_moe_mlp(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                        float norm_epsilon, const void* gate, int num_experts, int num_experts_per_token,
                        void* const* w1, void* const* w2, void* const* w3, void* temp_state, void* temp_gathered_state,
                        void* temp_a, void* temp_b, void* temp_logits, void* temp_dq, int max_rows, int act_gelu)
{
    (void)temp_gathered_state; (void)temp_dq; (void)layernorm_bias;
    EXL2_REQUIRE(handle && gate && w1 && w2 && w3, "make_q_moe_mlp: null argument");
    EXL2_REQUIRE(layernorm && layernorm_is_rms, "make_q_moe_mlp: only RMSNorm pre-norm is built");
    EXL2_REQUIRE(num_experts == 4 || num_experts == 8 || num_experts == 16,
                 "make_q_moe_mlp: %d experts (the fused path covers 4, 8, 16 like q_mlp.cu:333)", num_experts);
    EXL2_REQUIRE(num_experts_per_token >= 1 && num_experts_per_token <= num_experts, "make_q_moe_mlp: bad experts per token");
    EXL2_REQUIRE(temp_state && temp_a && temp_b && temp_logits, "make_q_moe_mlp: temp buffers required");
    QMoEMLP* m = (QMoEMLP*)calloc(1, sizeof(QMoEMLP));
    if (!m) EXL2_FAIL(EXL2_E_OOM, "make_q_moe_mlp: host out of memory");
    m->layernorm = (const f16*)layernorm; m->norm_epsilon = norm_epsilon; m->gate = (const f16*)gate;
    m->num_experts = num_experts; m->num_experts_per_token = num_experts_per_token;
    for (int i = 0; i < num_experts; i++)
    {
        m->w1[i] = (QMatrix*)w1[i]; m->w2[i] = (QMatrix*)w2[i]; m->w3[i] = (QMatrix*)w3[i];
        if (!m->w1[i] || !m->w2[i] || !m->w3[i]) { free(m); EXL2_FAIL(EXL2_E_INVALID, "make_q_moe_mlp: expert %d has a null projection", i); }
    }
    m->temp_state = (f16*)temp_state; m->temp_a = (f16*)temp_a; m->temp_b = (f16*)temp_b; m->temp_logits = (f16*)temp_logits;
    m->max_rows = max_rows; m->hidden = m->w1[0]->height; m->act_gelu = act_gelu;
    {
        // grouped launches (qgemv_flat.hip) need one packed input order for every expert's gate / up projection -- the
        // quantizer gives them one (conversion/quantize.py:190-192: w1.i and w3.i reuse w1.0's Hessian)
        QMatrix* all[2 * MOE_MAX_EXPERTS];
        for (int i = 0; i < num_experts; i++) { all[2 * i] = m->w1[i]; all[2 * i + 1] = m->w3[i]; }
        m->group_ok = num_experts <= 16 && same_perm(all, 2 * num_experts) && hidden_ok(m);
    }
    moe_plan_lean(m);
    *handle = m;
    return EXL2_OK;
}

int exl2_free_q_moe_mlp(void* handle)
{
    QMoEMLP* m = (QMoEMLP*)handle;
    if (m && m->lean_ok) { qgemv_lean_group_free(&m->lean_gu); qgemv_lean_group_free(&m->lean_dn); }
    if (m && m->lean_sum_ok) qgemv_lean_group_free(&m->lean_dn2);
    for (int r = 2; m && r <= LEAN_MAX_M; r++) if (m->lean_rows_ok[r]) { qgemv_lean_group_free(&m->lean_gu_r[r]); qgemv_lean_group_free(&m->lean_dn_r[r]); }
    free(handle);
    return EXL2_OK;
}

// q_moe_mlp_forward_ (ext_qmlp.cpp:245-272 -> QMoEMLP::forward_, q_mlp.cu:318-402): in place on x [rows, hidden].
// The reference takes rows <= 4; here any row count runs in passes of 16 rows (one MFMA row block): each expert's
// weights stream once per pass for all the rows routed to it, launches whose rows all have zero weight exit at once.
static int moe_forward(void* handle, void* x_, int rows, const MoeChainOut& co, int* npart_out, void* stream)
{
    EXL2_REQUIRE(handle && x_, "q_moe_mlp_forward_: null argument");
    if (npart_out) *npart_out = 0;                  // (0: the route taken had no combine launch -- the caller publishes the rows itself)
    QMoEMLP* m = (QMoEMLP*)handle;
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= m->max_rows, "q_moe_mlp_forward_: %d rows exceed max_rows %d", rows, m->max_rows);
    const int E = m->num_experts, hidden = m->hidden;
    f16* x = (f16*)x_;
    const int inter = m->w1[0]->width;
    // decode-sized row counts whose experts share their act-order: norm + router + top-k + gather in ONE launch (moe.hip:
    // moe_front_kernel, bit-identical to the four kernels it replaces); otherwise the separate kernels
    bool have_xg = false;
    f16* const xg = m->temp_state + (size_t)MAX_GEMV_ROWS * hidden;       // normalised rows in the experts' packed order
    const bool xg_room = m->group_ok && rows <= MAX_GEMV_ROWS && 2 * MAX_GEMV_ROWS <= m->max_rows;
    int front = 1;
    // ---- one row on the lean kernel: front (+ the selected experts' argument blocks) -> gate|up -> down -> combine
    if (rows == 1 && m->lean_ok && !ENV_SET("EXL2_MOE_NO_LEAN") && !ENV_SET("EXL2_MOE_UNFUSED_FRONT") && !ENV_SET("EXL2_MOE_SERIAL") && !ENV_SET("EXL2_MOE_NO_GROUP"))
    {
        MoeCopy cp; memset(&cp, 0, sizeof(cp));
        const bool sum_route = m->lean_sum_ok && !ENV_SET("EXL2_MOE_NO_SUM") && !(co.xp && co.tiled);
        const LeanGroupPlan& dn = sum_route ? m->lean_dn2 : m->lean_dn;
        cp.src[0] = (const u32x4*)m->lean_gu.table_src; cp.dst[0] = (u32x4*)m->lean_gu.table_sel; cp.units[0] = m->lean_gu.block_bytes / 16;
        cp.src[1] = (const u32x4*)dn.table_src; cp.dst[1] = (u32x4*)dn.table_sel; cp.units[1] = dn.block_bytes / 16;
        if (sum_route) { cp.sum[1] = 1; cp.b_lo[1] = dn.b_lo; cp.b_hi[1] = dn.b_hi; cp.b2_lo[1] = dn.b2_lo; cp.b2_hi[1] = dn.b2_hi; }
        cp.n_sel = m->num_experts_per_token;
        front = moe_front_launch(x, m->layernorm, m->gate, m->w1[0]->q_perm, m->temp_state, xg, m->temp_logits, rows, hidden, E,
                                 m->num_experts_per_token, m->norm_epsilon, &cp, stream);
        if (front < 0) return front;
        if (front == 0)
        {
            if (ENV_SET("EXL2_DEBUG_ROUTE")) fprintf(stderr, "q_moe_mlp route: lean%s rows=%d experts=%d\n", sum_route ? " (down pair sums)" : "", rows, E);
            if (sum_route)
            {
                LeanGroupDyn dyn; memset(&dyn, 0, sizeof(dyn));
                dyn.c = x; dyn.ldc = hidden; dyn.xp_out = co.xp; dyn.xp_invperm = co.invperm; dyn.xp_w = co.w; dyn.ss_out = co.xp ? co.ss : nullptr; dyn.ldxp = co.ldxp;
                if (qgemv_lean_group_launch(&m->lean_gu, stream) != 0 || qgemv_lean_group_launch(&dn, stream, &dyn) != 0)
                    EXL2_FAIL(EXL2_E_INVALID, "q_moe_mlp_forward_: a planned lean launch was not taken");
                HIP_TRY(hipGetLastError());
                if (npart_out) *npart_out = co.xp ? dn.grid_x : 0;
                return EXL2_OK;
            }
            if (qgemv_lean_group_launch(&m->lean_gu, stream) != 0 || qgemv_lean_group_launch(&m->lean_dn, stream) != 0)
                EXL2_FAIL(EXL2_E_INVALID, "q_moe_mlp_forward_: a planned lean launch was not taken");
            LAUNCH(moe_combine_kernel, dim3((unsigned)((hidden / 8 + 255) / 256), (unsigned)rows, 1), dim3(256), 0, stream,
                   x, (const f16*)m->temp_b, (const f16*)m->temp_logits, rows, hidden, E, co);
            if (npart_out) *npart_out = (hidden / 8 + 255) / 256;
            HIP_TRY(hipGetLastError());
            return EXL2_OK;
        }
        front = 1;
    }
    // ---- 2-4 rows on the lean kernel: front -> gate|up over all experts -> down over all experts -> combine
    if (rows >= 2 && rows <= LEAN_MAX_M && m->lean_rows_ok[rows] && !ENV_SET("EXL2_MOE_NO_LEAN") && !ENV_SET("EXL2_MOE_UNFUSED_FRONT") && !ENV_SET("EXL2_MOE_SERIAL") &&
        !ENV_SET("EXL2_MOE_NO_GROUP") && !(co.xp && co.tiled))
    {
        front = exl2_moe_front(x, m->layernorm, m->gate, m->w1[0]->q_perm, m->temp_state, xg, m->temp_logits, rows, hidden, E,
                               m->num_experts_per_token, m->norm_epsilon, stream);
        if (front < 0) return front;
        if (front == 0)
        {
            if (ENV_SET("EXL2_DEBUG_ROUTE")) fprintf(stderr, "q_moe_mlp route: lean rows=%d experts=%d\n", rows, E);
            if (qgemv_lean_group_launch(&m->lean_gu_r[rows], stream) != 0 || qgemv_lean_group_launch(&m->lean_dn_r[rows], stream) != 0)
                EXL2_FAIL(EXL2_E_INVALID, "q_moe_mlp_forward_: a planned lean launch was not taken");
            LAUNCH(moe_combine_kernel, dim3((unsigned)((hidden / 8 + 255) / 256), (unsigned)rows, 1), dim3(256), 0, stream,
                   x, (const f16*)m->temp_b, (const f16*)m->temp_logits, rows, hidden, E, co);
            if (npart_out) *npart_out = (hidden / 8 + 255) / 256;
            HIP_TRY(hipGetLastError());
            return EXL2_OK;
        }
        have_xg = false; front = 1;
    }
    if (xg_room && !ENV_SET("EXL2_MOE_UNFUSED_FRONT"))
        front = exl2_moe_front(x, m->layernorm, m->gate, m->w1[0]->q_perm, m->temp_state, xg, m->temp_logits, rows, hidden, E,
                               m->num_experts_per_token, m->norm_epsilon, stream);
    if (front < 0) return front;
    if (front == 0) have_xg = true;
    else
    {
        { const int rc = exl2_rms_norm(x, m->layernorm, m->temp_state, m->norm_epsilon, rows, hidden, 0, 0, 0, stream); if (rc) return rc; }
        { const int rc = exl2_moe_route(m->temp_state, m->gate, m->temp_logits, rows, hidden, E, m->num_experts_per_token, stream); if (rc) return rc; }
    }
    auto make_xg = [&]() {
        if (!have_xg)
            LAUNCH(gather_rows_f16_kernel, dim3((unsigned)((hidden + 255) / 256), (unsigned)rows, 1), dim3(256), 0, stream,
                   (const f16*)m->temp_state, (const u16*)m->w1[0]->q_perm, xg, hidden);
        have_xg = true;
    };
    // ---- grouped route (rows <= 16): all experts' gate|up in ONE launch (SiLU * up in the epilogue, written in each down
    // projection's packed order), all experts' down in ONE launch (weighted partial outputs), one combine.  Replaces the
    // per-expert launch loop below (3 launches per expert: q_mlp.cu:318-402 has the same shape, moe_mlp.py:255-323 a
    // Python loop above 4 rows).
    if (m->group_ok && rows <= MAX_GEMV_ROWS && 2 * MAX_GEMV_ROWS <= m->max_rows && (long long)E * rows <= m->max_rows &&
        (long long)E * rows * hidden <= (long long)m->max_rows * inter && !ENV_SET("EXL2_MOE_SERIAL") && !ENV_SET("EXL2_MOE_NO_GROUP"))
    {
        if (ENV_SET("EXL2_DEBUG_ROUTE")) fprintf(stderr, "q_moe_mlp route: grouped rows=%d experts=%d\n", rows, E);
        make_xg();
        FlatIn ins[MOE_MAX_EXPERTS];
        for (int e = 0; e < E; e++)
        {
            FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
            in.qm[0] = m->w1[e]; in.qm[1] = m->w3[e];
            in.c[0] = m->temp_a + (size_t)e * rows * inter; in.c[1] = in.c[0]; in.ldc[0] = inter; in.ldc[1] = inter;
            in.c_invperm[0] = m->w2[e]->q_perm ? m->w2[e]->q_invperm : nullptr;
            in.n_mats = 2; in.pair = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = xg; in.lda = hidden; in.c_mode = C_STORE;
            in.act_gelu = m->act_gelu ? 1 : 0;
        }
        int rc = qgemv_flat_group_launch(ins, E, m->temp_logits, E, 0, 0, stream);
        if (rc == 0)
        {
            for (int e = 0; e < E; e++)
            {
                FlatIn& in = ins[e]; memset(&in, 0, sizeof(in));
                in.qm[0] = m->w2[e]; in.c[0] = m->temp_b + (size_t)e * rows * hidden; in.ldc[0] = hidden;
                in.n_mats = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = m->temp_a; in.lda = inter; in.c_mode = C_STORE;
            }
            rc = qgemv_flat_group_launch(ins, E, m->temp_logits, E, 1, (long long)rows * inter, stream);
            if (rc < 0) EXL2_FAIL(EXL2_E_INVALID, "q_moe_mlp_forward_: grouped down launch rejected (%d)", rc);
            if (rc == 0)
            {
                LAUNCH(moe_combine_kernel, dim3((unsigned)((hidden / 8 + 255) / 256), (unsigned)rows, 1), dim3(256), 0, stream,
                       x, (const f16*)m->temp_b, (const f16*)m->temp_logits, rows, hidden, E, co);
            if (npart_out) *npart_out = (hidden / 8 + 255) / 256;
                HIP_TRY(hipGetLastError());
                return EXL2_OK;
            }
            // (the gate|up launch wrote only temp_a: the serial route below recomputes everything)
        }
        else if (rc < 0) EXL2_FAIL(EXL2_E_INVALID, "q_moe_mlp_forward_: grouped gate|up launch rejected (%d)", rc);
    }
    // ---- batched route (round 5; rows <= 16 the grouped launch declined: Mixtral's 16 x 14336 down_proj rows do not fit its LDS):
    // the experts on the streaming / phased kernels like the serial loop below, but (1) gate and up read the rows gathered ONCE into
    // the experts' shared packed order (no row pre-pass per launch: it was 8 of the 18 stage_rows launches of a Mixtral layer, 7.3 us
    // each -- profiles/history/r05g_mixtral_b16_kernel_stats.csv), (2) two experts' gate | up and four experts' down per launch
    // (MAX_FUSED_MATS jobs: 6 launches + 2 pre-passes instead of 16 + 16), (3) every expert's SiLU(gate) * up and weighted down
    // output in its own rows of the scratch, summed by moe_combine_kernel (fp32, expert order) instead of 8 read-modify-writes of x.
    if (m->group_ok && rows <= MAX_GEMV_ROWS && (long long)E * rows <= m->max_rows && (E & 3) == 0 &&
        (long long)E * rows * ((long long)inter + hidden) <= (long long)m->max_rows * inter &&
        (long long)(MAX_GEMV_ROWS + rows) * hidden <= (long long)m->max_rows * hidden && !ENV_SET("EXL2_MOE_SERIAL") && !ENV_SET("EXL2_MOE_UNBATCHED"))
    {
        if (ENV_SET("EXL2_DEBUG_ROUTE")) fprintf(stderr, "q_moe_mlp route: batched rows=%d experts=%d\n", rows, E);
        make_xg();
        f16* const dout = m->temp_b + (size_t)E * rows * inter;              // [E][rows][hidden] weighted down outputs
        for (int e0 = 0; e0 < E; e0 += 2)
        {
            GemvJob jobs[4];
            for (int i = 0; i < 4; i++)
            {
                const int e = e0 + (i >> 1);
                f16* const out = ((i & 1) ? m->temp_b : m->temp_a) + (size_t)e * rows * inter;
                fill_job(jobs[i], (i & 1) ? m->w3[e] : m->w1[e], xg, out, A_PLAIN, C_STORE);
                jobs[i].m.perm = nullptr;                                    // (the rows ARE in packed order)
                jobs[i].r_weights = m->temp_logits + e; jobs[i].r_stride = E; jobs[i].mul_r_weights = 0;
                if (m->w2[e]->dev.perm && m->w2[e]->q_invperm) jobs[i].c_invperm = m->w2[e]->q_invperm;
            }
            LAUNCH_JOBS(jobs, 4, rows, m->w1[e0]->is_gptq, stream, "q_moe_mlp_forward_");
        }
        for (int e0 = 0; e0 < E; e0 += 4)
        {
            GemvJob jobs[4];
            for (int i = 0; i < 4; i++)
            {
                const int e = e0 + i;
                fill_job(jobs[i], m->w2[e], m->temp_a + (size_t)e * rows * inter, dout + (size_t)e * rows * hidden,
                         m->act_gelu ? A_GELU_MUL : A_SILU_MUL, C_STORE);
                jobs[i].a2 = m->temp_b + (size_t)e * rows * inter;
                jobs[i].r_weights = m->temp_logits + e; jobs[i].r_stride = E; jobs[i].mul_r_weights = 1;
                if (m->w2[e]->dev.perm && m->w2[e]->q_invperm) jobs[i].m.perm = nullptr;
            }
            LAUNCH_JOBS(jobs, 4, rows, m->w2[e0]->is_gptq, stream, "q_moe_mlp_forward_");
        }
        LAUNCH(moe_combine_kernel, dim3((unsigned)((hidden / 8 + 255) / 256), (unsigned)rows, 1), dim3(256), 0, stream,
               x, (const f16*)dout, (const f16*)m->temp_logits, rows, hidden, E, co);
        if (npart_out) *npart_out = (hidden / 8 + 255) / 256;
        HIP_TRY(hipGetLastError());
        return EXL2_OK;
    }
    for (int r0 = 0; r0 < rows; r0 += MAX_GEMV_ROWS)
    {
        const int nr = rows - r0 < MAX_GEMV_ROWS ? rows - r0 : MAX_GEMV_ROWS;
        const f16* ts = m->temp_state + (size_t)r0 * hidden;
        f16* ta = m->temp_a + (size_t)r0 * inter;
        f16* tb = m->temp_b + (size_t)r0 * inter;
        for (int e = 0; e < E; e++)
        {
            const f16* rw = m->temp_logits + (size_t)r0 * E + e;
            GemvJob jobs[2];
            fill_job(jobs[0], m->w1[e], ts, ta, A_PLAIN, C_STORE);
            fill_job(jobs[1], m->w3[e], ts, tb, A_PLAIN, C_STORE);
            const bool scatter = m->w2[e]->dev.perm && m->w2[e]->q_invperm;
            for (int i = 0; i < 2; i++)
            {
                jobs[i].r_weights = rw; jobs[i].r_stride = E; jobs[i].mul_r_weights = 0;
                if (scatter) jobs[i].c_invperm = m->w2[e]->q_invperm;
            }
            LAUNCH_JOBS(jobs, 2, nr, m->w1[e]->is_gptq, stream, "q_moe_mlp_forward_");
            GemvJob d;
            fill_job(d, m->w2[e], ta, x + (size_t)r0 * hidden, m->act_gelu ? A_GELU_MUL : A_SILU_MUL, C_ACCUM);
            d.a2 = tb;
            d.r_weights = rw; d.r_stride = E; d.mul_r_weights = 1;
            if (scatter) d.m.perm = nullptr;
            LAUNCH_JOBS(&d, 1, nr, m->w2[e]->is_gptq, stream, "q_moe_mlp_forward_");
        }
    }
    return EXL2_OK;
}

int exl2_q_moe_mlp_forward(void* handle, void* x, int rows, void* stream)
{
    MoeChainOut co; memset(&co, 0, sizeof(co));
    return moe_forward(handle, x, rows, co, nullptr, stream);
}

// the block inside a chained decode step: q_moe_mlp_forward_ + the hand-off for the next consumer (xp_out = x * next_norm_w in its
// packed order, ss_out = *npart_out partial sums of squares per row), made by the combine launch where the route has one, by
// exl2_publish_rows behind the block otherwise
int exl2_q_moe_mlp_forward_chain(void* handle, void* x, int rows, const void* next_invperm, const void* next_norm_w, void* xp_out,
                                 float* ss_out, int* npart_out, void* stream)
{
    EXL2_REQUIRE(handle && x && xp_out && ss_out && npart_out, "q_moe_mlp_forward_chain: null argument");
    QMoEMLP* m = (QMoEMLP*)handle;
    MoeChainOut co;
    co.xp = (f16*)xp_out; co.invperm = (const u16*)next_invperm; co.w = (const f16*)next_norm_w; co.ss = ss_out; co.ldxp = m->hidden;
    co.tiled = chain_xp_tiled() ? 1 : 0;
    int npart = 0;
    const int rc = moe_forward(handle, x, rows, co, &npart, stream);
    if (rc) return rc;
    if (npart == 0)
    {
        const int rc2 = exl2_publish_rows(x, rows, m->hidden, next_invperm, next_norm_w, xp_out, ss_out, stream);
        if (rc2) return rc2;
        npart = 1;
    }
    *npart_out = npart;
    return EXL2_OK;
}

// make_q_attn (ext_qattn.cpp:24-104); argument order = SURVEY.md A.5
int exl2_make_q_attn(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                     int headnorm_is_rms, float norm_epsilon, void* q_q_proj, void* q_k_proj, void* q_v_proj,
                     void* q_o_proj, void* temp_state, void* temp_dq, int max_rows, int hidden_size, int num_heads,
                     int num_kv_heads, int head_dim, int max_seq_len, int has_residual, int rope_style, int sincos_size,
                     const void* q_norm, const void* k_norm, const void* post_layernorm,
                     const void* post_layernorm_bias, int residual_fp32, int use_graphs)
{
    EXL2_REQUIRE(handle && q_q_proj && q_k_proj && q_v_proj && q_o_proj, "make_q_attn: null projection handle");
    EXL2_REQUIRE(!layernorm || layernorm_is_rms, "make_q_attn: only RMSNorm pre-norm is built (LayerNorm archs out of scope)");
    EXL2_REQUIRE(!q_norm && !k_norm, "make_q_attn: q/k head norms are not built (non-Llama archs out of scope)");
    EXL2_REQUIRE(!residual_fp32, "make_q_attn: fp32 residual stream is not built");
    // (temp_state may be absent: the reference's tensor-parallel loader makes the handle without scratch, attn.py:259-272,
    // and never runs a forward on it; the forwards below refuse such a handle)
    QAttn* a = (QAttn*)calloc(1, sizeof(QAttn));
    if (!a) EXL2_FAIL(EXL2_E_OOM, "make_q_attn: host out of memory");
    a->layernorm = (const f16*)layernorm; a->layernorm_bias = (const f16*)layernorm_bias;
    a->layernorm_is_rms = layernorm_is_rms; a->headnorm_is_rms = headnorm_is_rms; a->norm_epsilon = norm_epsilon;
    a->q_proj = (QMatrix*)q_q_proj; a->k_proj = (QMatrix*)q_k_proj; a->v_proj = (QMatrix*)q_v_proj; a->o_proj = (QMatrix*)q_o_proj;
    a->temp_state = (f16*)temp_state; a->temp_dq = (f16*)temp_dq;
    a->max_rows = max_rows; a->hidden_size = hidden_size; a->num_heads = num_heads; a->num_kv_heads = num_kv_heads;
    a->head_dim = head_dim; a->max_seq_len = max_seq_len; a->has_residual = has_residual; a->rope_style = rope_style;
    a->sincos_size = sincos_size;
    a->post_layernorm = (const f16*)post_layernorm; a->post_layernorm_bias = (const f16*)post_layernorm_bias;
    a->residual_fp32 = residual_fp32; a->use_graphs = use_graphs;
    if (!(a->q_proj->height == hidden_size && a->k_proj->height == hidden_size && a->v_proj->height == hidden_size))
    {
        free(a);
        EXL2_FAIL(EXL2_E_INVALID, "make_q_attn: projection heights do not match hidden_size %d", hidden_size);
    }
    {
        // chained decode (qgemv_flat.hip): plain pre-RMSNorm residual block whose q/k/v share one act-order permutation
        QMatrix* qkv[3] = {a->q_proj, a->k_proj, a->v_proj};
        const bool gq = a->q_proj->is_gptq;
        a->chain_ok = a->layernorm && a->layernorm_is_rms && !a->post_layernorm && a->has_residual && !a->layernorm_bias
                      && a->k_proj->is_gptq == gq && a->v_proj->is_gptq == gq && a->o_proj->is_gptq == gq
                      && a->o_proj->height == num_heads * head_dim && a->o_proj->width == hidden_size && same_perm(qkv, 3);
        a->qkv_same_perm = a->chain_ok || same_perm(qkv, 3);
        if (a->chain_ok) { a->norm_w_perm = permuted_norm(a->layernorm, a->q_proj); if (!a->norm_w_perm) a->chain_ok = false; }
    }
    *handle = a;
    return EXL2_OK;
}

int exl2_free_q_attn(void* handle)
{
    QAttn* a = (QAttn*)handle;
    if (a && a->norm_w_perm) (void)hipFree(a->norm_w_perm);
    free(handle);
    return EXL2_OK;
}

// q_attn_forward_1 (ext_qattn.cpp:115-159 -> q_attn.cu:247-317): q,k,v = proj(rmsnorm(x)); RoPE(q, k) in place.
// apply_rope = 0 skips the rotation (the caller fuses it with the KV append, exl2_rope_kv_append).
int exl2_q_attn_forward_1(void* handle, const void* x, int batch_size, int q_len, int past_len, const int* past_lens,
                          void* temp_q, void* temp_k, void* temp_v, const void* sin, const void* cos, int apply_rope,
                          void* stream)
{
    EXL2_REQUIRE(handle && x && temp_q && temp_k && temp_v, "q_attn_forward_1: null argument");
    QAttn* a = (QAttn*)handle;
    const int rows = batch_size * q_len;
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(a->temp_state || !a->layernorm, "q_attn_forward_1: handle was made without scratch (tensor-parallel load)");
    // temp_state (and the caller's temp_q/k/v) are sized for max_rows (attn.py:377-379 keeps chunks inside it)
    EXL2_REQUIRE(rows <= a->max_rows, "q_attn_forward_1: %d rows exceed max_rows %d", rows, a->max_rows);
    const bool gptq = a->q_proj->is_gptq;
    EXL2_REQUIRE(a->k_proj->is_gptq == gptq && a->v_proj->is_gptq == gptq, "q_attn_forward_1: mixed EXL2/GPTQ projections");
    // RMSNorm rides in the staging of the rows for every row count: <= 16 rows in the decode kernel's prologue, above that in
    // the row pre-pass of the prefill kernels (stage_rows_kernel: act-order gather + norm, once for q | k | v)
    GemvJob jobs[3];
    const int mode = a->layernorm ? A_RMSNORM : A_PLAIN;
    fill_job(jobs[0], a->q_proj, (const f16*)x, (f16*)temp_q, mode, C_STORE);
    fill_job(jobs[1], a->k_proj, (const f16*)x, (f16*)temp_k, mode, C_STORE);
    fill_job(jobs[2], a->v_proj, (const f16*)x, (f16*)temp_v, mode, C_STORE);
    for (int i = 0; i < 3; i++) { jobs[i].norm_w = a->layernorm; jobs[i].norm_eps = a->norm_epsilon; }
    jobs[1].rows_as_prev = jobs[2].rows_as_prev = a->qkv_same_perm ? 1 : 0;
    LAUNCH_JOBS(jobs, 3, rows, gptq, stream, "q_attn_forward_1");
    if (apply_rope && a->rope_style != 0)
    {
        EXL2_REQUIRE(sin && cos, "q_attn_forward_1: sin/cos tables missing");
        return exl2_rope_qk(temp_q, temp_k, sin, cos, batch_size, q_len * a->num_heads, q_len * a->num_kv_heads,
                            a->head_dim, a->num_heads, a->num_kv_heads, past_len, past_lens, a->rope_style == 2,
                            a->sincos_size, stream);
    }
    return EXL2_OK;
}

// q_attn_forward_2 (ext_qattn.cpp:161-191 -> q_attn.cu:319-345): x (+)= attn_output * Wo
int exl2_q_attn_forward_2(void* handle, void* x, const void* attn_output, int batch_size, int q_len, void* stream)
{
    EXL2_REQUIRE(handle && x && attn_output, "q_attn_forward_2: null argument");
    QAttn* a = (QAttn*)handle;
    const int rows = batch_size * q_len;
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= a->max_rows, "q_attn_forward_2: %d rows exceed max_rows %d", rows, a->max_rows);
    EXL2_REQUIRE(a->temp_state || !a->post_layernorm, "q_attn_forward_2: handle was made without scratch (tensor-parallel load)");
    GemvJob j;
    if (!a->post_layernorm)
    {
        fill_job(j, a->o_proj, (const f16*)attn_output, (f16*)x, A_PLAIN, a->has_residual ? C_ACCUM : C_STORE);
        LAUNCH_JOBS(&j, 1, rows, a->o_proj->is_gptq, stream, "q_attn_forward_2");
        return EXL2_OK;
    }
    fill_job(j, a->o_proj, (const f16*)attn_output, a->temp_state, A_PLAIN, C_STORE);
    LAUNCH_JOBS(&j, 1, rows, a->o_proj->is_gptq, stream, "q_attn_forward_2");
    // add_residual is unconditional in the post-norm branch, as in the reference (q_attn.cu:339, q_mlp.cu:230)
    return exl2_rms_norm(a->temp_state, a->post_layernorm, x, a->norm_epsilon, rows, a->hidden_size, 1, 0, 0, stream);
}

// make_q_mlp (ext_qmlp.h; call site mlp.py:204-223)
int exl2_make_q_mlp(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                    float norm_epsilon, void* q_gate, void* q_up, void* q_down, void* temp_state, void* temp_a,
                    void* temp_b, void* temp_dq, int max_rows, int act_gelu, int has_residual,
                    const void* post_layernorm, const void* post_layernorm_bias, int residual_fp32, int use_graphs)
{
    EXL2_REQUIRE(handle && q_up && q_down, "make_q_mlp: null projection handle");
    EXL2_REQUIRE(!layernorm || layernorm_is_rms, "make_q_mlp: only RMSNorm pre-norm is built (LayerNorm archs out of scope)");
    EXL2_REQUIRE(!residual_fp32, "make_q_mlp: fp32 residual stream is not built");
    // (scratch may be absent -- tensor-parallel load, mlp.py:169-190: the forwards refuse such a handle)
    QMLP* m = (QMLP*)calloc(1, sizeof(QMLP));
    if (!m) EXL2_FAIL(EXL2_E_OOM, "make_q_mlp: host out of memory");
    m->layernorm = (const f16*)layernorm; m->layernorm_bias = (const f16*)layernorm_bias; m->layernorm_is_rms = layernorm_is_rms;
    m->norm_epsilon = norm_epsilon; m->gate = (QMatrix*)q_gate; m->up = (QMatrix*)q_up; m->down = (QMatrix*)q_down;
    m->temp_state = (f16*)temp_state; m->temp_a = (f16*)temp_a; m->temp_b = (f16*)temp_b; m->temp_dq = (f16*)temp_dq;
    m->max_rows = max_rows; m->act_gelu = act_gelu; m->has_residual = has_residual;
    m->post_layernorm = (const f16*)post_layernorm; m->post_layernorm_bias = (const f16*)post_layernorm_bias;
    m->residual_fp32 = residual_fp32; m->use_graphs = use_graphs;
    if (m->gate)
    {
        QMatrix* gu[2] = {m->gate, m->up};
        const bool gq = m->up->is_gptq;
        m->chain_ok = m->temp_a && m->layernorm && m->layernorm_is_rms && !m->layernorm_bias && !m->post_layernorm && m->has_residual
                      && m->gate->is_gptq == gq && m->down->is_gptq == gq && m->gate->width == m->up->width
                      && m->down->height == m->up->width && m->down->width == m->up->height && same_perm(gu, 2);
        m->gu_same_perm = m->chain_ok || same_perm(gu, 2);
        if (m->chain_ok) { m->norm_w_perm = permuted_norm(m->layernorm, m->up); if (!m->norm_w_perm) m->chain_ok = false; }
    }
    *handle = m;
    return EXL2_OK;
}

int exl2_free_q_mlp(void* handle)
{
    QMLP* m = (QMLP*)handle;
    if (m && m->norm_w_perm) (void)hipFree(m->norm_w_perm);
    free(handle);
    return EXL2_OK;
}

// q_mlp_forward_ (ext_qmlp.cpp:87-118 -> q_mlp.cu:153-236): x (+)= act(n Wg) * (n Wu) Wd, n = rmsnorm(x); in place on x
int exl2_q_mlp_forward(void* handle, void* x, int rows, void* stream)
{
    EXL2_REQUIRE(handle && x, "q_mlp_forward_: null argument");
    QMLP* m = (QMLP*)handle;
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= m->max_rows, "q_mlp_forward_: %d rows exceed max_rows %d", rows, m->max_rows);
    EXL2_REQUIRE(m->temp_a && (!m->gate || m->temp_b) && (m->temp_state || !(m->layernorm || m->post_layernorm)),
                 "q_mlp_forward_: handle was made without scratch (tensor-parallel load)");
    const int hidden = m->up->height;
    const bool gptq = m->up->is_gptq;
    const bool skinny = rows <= MAX_GEMV_ROWS;
    f16* down_dst = m->post_layernorm ? m->temp_state : (f16*)x;
    const int down_mode = (m->post_layernorm || !m->has_residual) ? C_STORE : C_ACCUM;
    GemvJob jobs[2];

    const f16* ns = (const f16*)x;
    const int in_mode = m->layernorm ? A_RMSNORM : A_PLAIN;     // norm in the staging of the rows (decode prologue / prefill pre-pass)
    int n = 0;
    if (m->gate) { fill_job(jobs[n], m->gate, ns, m->temp_a, in_mode, C_STORE); n++; fill_job(jobs[n], m->up, ns, m->temp_b, in_mode, C_STORE); n++; }
    else         { fill_job(jobs[n], m->up, ns, m->temp_a, in_mode, C_STORE); n++; }
    for (int i = 0; i < n; i++) { jobs[i].norm_w = m->layernorm; jobs[i].norm_eps = m->norm_epsilon; }
    if (n == 2) jobs[1].rows_as_prev = m->gu_same_perm ? 1 : 0;
    // decode-shaped calls: gate / up write their columns straight into down's packed (act-order) row order, so down
    // stages a contiguous row instead of gathering through q_perm (temp_a / temp_b have no other reader)
    const bool scatter = skinny && m->down->dev.perm && m->down->q_invperm;
    if (scatter) for (int i = 0; i < n; i++) jobs[i].c_invperm = m->down->q_invperm;
    LAUNCH_JOBS(jobs, n, rows, gptq, stream, "q_mlp_forward_");

    // activation folded into the staging of down's input (no intermediate round trip, one launch less) for every row count
    GemvJob d;
    {
        const int amode = m->gate ? (m->act_gelu ? A_GELU_MUL : A_SILU_MUL) : (m->act_gelu ? A_GELU : A_SILU);
        fill_job(d, m->down, m->temp_a, down_dst, amode, down_mode);
        d.a2 = m->temp_b;
        if (scatter) d.m.perm = nullptr;
    }
    LAUNCH_JOBS(&d, 1, rows, m->down->is_gptq, stream, "q_mlp_forward_");
    if (m->post_layernorm)
        return exl2_rms_norm(m->temp_state, m->post_layernorm, x, m->norm_epsilon, rows, hidden, 1, 0, 0, stream);
    return EXL2_OK;
}


// ---- chained decode (qgemv_flat.hip) -----------------------------------------------------------------------------------------

int exl2_chain_route_counts(long long* lean, long long* flat, int reset)
{
    if (lean) *lean = g_route_lean;
    if (flat) *flat = g_route_flat;
    if (reset) { g_route_lean = 0; g_route_flat = 0; }
    return EXL2_OK;
}

int exl2_chain_overlap_begin(void* flags, int n_blocks, void* stream_a, void* stream_b)
{
    EXL2_REQUIRE(flags && n_blocks > 2, "chain_overlap_begin: no counters");
    EXL2_REQUIRE(!g_chain.open, "chain_overlap_begin: a chain is already open on this thread");
    g_chain.flags = (u32*)flags; g_chain.n_blocks = n_blocks; g_chain.stream[0] = stream_a; g_chain.stream[1] = stream_b;
    g_chain.next = 0; g_chain.open = true;
    return EXL2_OK;
}

// Closes the chain.  Returns the number of launches; the last one went to stream (n - 1) & 1, and so does the tail kernel.
int exl2_chain_overlap_end(int* n_launches)
{
    EXL2_REQUIRE(g_chain.open, "chain_overlap_end: no chain is open");
    g_chain.open = false;
    if (n_launches) *n_launches = g_chain.next;
    if (g_chain.next > 0)
    {
        LAUNCH(chain_tail_kernel, dim3(1, 1, 1), dim3(64, 1, 1), 0, g_chain.stream[(g_chain.next - 1) & 1], chain_block(g_chain.next - 1));
        HIP_TRY(hipGetLastError());
    }
    return EXL2_OK;
}

int exl2_q_attn_chain_info(void* handle, int* capable, const void** in_invperm, const void** o_invperm, const void** norm_w_perm)
{
    EXL2_REQUIRE(handle, "q_attn_chain_info: null handle");
    QAttn* a = (QAttn*)handle;
    if (capable) *capable = a->chain_ok ? 1 : 0;
    if (in_invperm) *in_invperm = a->q_proj->q_perm ? a->q_proj->q_invperm : nullptr;
    if (o_invperm) *o_invperm = a->o_proj->q_perm ? a->o_proj->q_invperm : nullptr;
    if (norm_w_perm) *norm_w_perm = a->norm_w_perm;
    return EXL2_OK;
}

int exl2_q_mlp_chain_info(void* handle, int* capable, const void** in_invperm, const void** norm_w_perm)
{
    EXL2_REQUIRE(handle, "q_mlp_chain_info: null handle");
    QMLP* m = (QMLP*)handle;
    if (capable) *capable = m->chain_ok ? 1 : 0;
    if (in_invperm) *in_invperm = m->up->q_perm ? m->up->q_invperm : nullptr;
    if (norm_w_perm) *norm_w_perm = m->norm_w_perm;
    return EXL2_OK;
}

int exl2_q_matrix_perm_info(void* q_matrix, const void** perm, const void** invperm)
{
    EXL2_REQUIRE(q_matrix, "q_matrix_perm_info: null handle");
    QMatrix* q = (QMatrix*)q_matrix;
    if (perm) *perm = q->q_perm;
    if (invperm) *invperm = q->q_perm ? q->q_invperm : nullptr;
    return EXL2_OK;
}

int exl2_q_attn_forward_1_chain(void* handle, const void* xp, const float* ss, int npart, int rows,
                                void* temp_q, void* temp_k, void* temp_v, void* stream)
{
    EXL2_REQUIRE(handle && xp && ss && temp_q && temp_k && temp_v, "q_attn_forward_1_chain: null argument");
    QAttn* a = (QAttn*)handle;
    EXL2_REQUIRE(a->chain_ok, "q_attn_forward_1_chain: module is not chain-capable");
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= a->max_rows && rows <= MAX_GEMV_ROWS, "q_attn_forward_1_chain: %d rows", rows);
    FlatIn in; memset(&in, 0, sizeof(in));
    QMatrix* ms[3] = {a->q_proj, a->k_proj, a->v_proj};
    void* cs[3] = {temp_q, temp_k, temp_v};
    for (int i = 0; i < 3; i++) { in.qm[i] = ms[i]; in.c[i] = (f16*)cs[i]; in.ldc[i] = ms[i]->width; }
    in.n_mats = 3; in.M = rows; in.a_mode = A_NORM_PRE; in.a = (const f16*)xp; in.lda = a->hidden_size;
    in.ss = ss; in.npart = npart; in.eps = a->norm_epsilon; in.c_mode = C_STORE;
    in.a_tiled = chain_xp_tiled() ? 1 : 0;
    FLAT_TRY(in, stream, nullptr, "q_attn_forward_1_chain");
    return EXL2_OK;
}

// q_attn_forward_1's full contract (ext_qattn.cpp:115-159: projections of the normalised rows, then RoPE on q and k in place) from a
// published hand-off: what the module chain behind the operator boundary (dropin/_exl2_fast.cpp) calls -- the reference host reads the
// rotated q / k (k straight in its cache rows, attn.py:1088-1091), so the rotation cannot be left to the attention launch here.
int exl2_q_attn_forward_1_chain_rope(void* handle, const void* xp, const float* ss, int npart, int batch_size, int q_len, int past_len,
                                     const int* past_lens, void* temp_q, void* temp_k, void* temp_v, const void* sin, const void* cos,
                                     void* stream)
{
    EXL2_REQUIRE(handle, "q_attn_forward_1_chain_rope: null handle");
    QAttn* a = (QAttn*)handle;
    const int rc = exl2_q_attn_forward_1_chain(handle, xp, ss, npart, batch_size * q_len, temp_q, temp_k, temp_v, stream);
    if (rc != EXL2_OK || batch_size * q_len <= 0 || a->rope_style == 0) return rc;
    EXL2_REQUIRE(sin && cos, "q_attn_forward_1_chain_rope: sin/cos tables missing");
    return exl2_rope_qk(temp_q, temp_k, sin, cos, batch_size, q_len * a->num_heads, q_len * a->num_kv_heads,
                        a->head_dim, a->num_heads, a->num_kv_heads, past_len, past_lens, a->rope_style == 2,
                        a->sincos_size, stream);
}

int exl2_q_attn_forward_2_chain(void* handle, void* x, const void* attn_out_packed, int rows, const void* next_invperm,
                                const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out, void* stream)
{
    EXL2_REQUIRE(handle && x && attn_out_packed, "q_attn_forward_2_chain: null argument");
    QAttn* a = (QAttn*)handle;
    EXL2_REQUIRE(a->chain_ok, "q_attn_forward_2_chain: module is not chain-capable");
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= a->max_rows && rows <= MAX_GEMV_ROWS, "q_attn_forward_2_chain: %d rows", rows);
    FlatIn in; memset(&in, 0, sizeof(in));
    in.qm[0] = a->o_proj; in.c[0] = (f16*)x; in.ldc[0] = a->o_proj->width;
    in.n_mats = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = (const f16*)attn_out_packed; in.lda = a->o_proj->height;
    in.c_mode = C_ACCUM;
    in.xp_out = (f16*)xp_out; in.xp_invperm = (const u16*)next_invperm; in.xp_w = (const f16*)next_norm_w;
    in.ss_out = xp_out ? ss_out : nullptr; in.ldxp = a->hidden_size;
    in.xp_tiled = (xp_out && chain_xp_tiled()) ? 1 : 0;
    int wgs = 0;
    FLAT_TRY(in, stream, &wgs, "q_attn_forward_2_chain");
    if (npart_out) *npart_out = wgs;
    return EXL2_OK;
}

// part: 1 = gate | up only, 2 = down only, 3 = both (exl2_q_mlp_forward_chain).  Row groups (model.py): gate | up reads K = hidden,
// down K = intermediate -- the rows whose staged copy fits in LDS differ, so a step may group the rows differently for the two.
static int q_mlp_chain_part(void* handle, int part, void* x, const void* xp, const float* ss, int npart, int rows, int row0,
                            const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out, void* stream)
{
    EXL2_REQUIRE(handle && x, "q_mlp_forward_chain: null argument");
    QMLP* m = (QMLP*)handle;
    EXL2_REQUIRE(m->chain_ok, "q_mlp_forward_chain: module is not chain-capable");
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(row0 >= 0 && row0 + rows <= m->max_rows && rows <= MAX_GEMV_ROWS, "q_mlp_forward_chain: rows %d + %d", row0, rows);
    const int hidden = m->up->height, inter = m->up->width;
    f16* const act = m->temp_a + (size_t)row0 * inter;              // SiLU(gate) * up of rows [row0, row0 + rows), down's packed order
    // 5 .. 16 rows, both halves in this call: SiLU(gate) * up goes to down_proj in the layout the matrix cores read (qgemv_flat.h:
    // FlatIn.a_tiled) -- down reads its A operands straight from memory then (the lean kernel's XMEM form), no staged copy, whatever K.
    // EXL2_MLP_TILED: 0 never, 1 (default) where the rows' staged copy would not fit the LDS, 2 from 5 rows up
    const int tiled_on = []() { const char* e = getenv("EXL2_MLP_TILED"); return e ? atoi(e) : 1; }();     // (per call: tests switch it)
    const int xmem_env = []() { const char* e = getenv("EXL2_LEAN_XMEM"); return e ? atoi(e) : 1; }();
    bool tiled = part == 3 && rows > 4 && m->max_rows >= 16 && row0 == 0 && tiled_on > 0 && xmem_env > 0 && lean_enabled() && !chain_sync_active()
                 && (tiled_on >= 2 || (size_t)rows * (size_t)(inter + 8) * 2 > 150u * 1024u);
    if (tiled)
    {
        // the tiled hand-off exists in the lean kernel's XMEM form only (one register pass: K up to ~12 k at 4 bits on 16 waves),
        // and the flat kernel reads no tiled layout: ask for down_proj's plan BEFORE gate | up writes temp_a that way
        // (rows >= M of a tiled buffer stay unwritten: MFMA rows are independent, nothing reads them)
        FlatIn probe; memset(&probe, 0, sizeof(probe));
        probe.qm[0] = m->down; probe.c[0] = (f16*)x; probe.ldc[0] = hidden;
        probe.n_mats = 1; probe.M = rows; probe.a_mode = A_DIRECT; probe.a = act; probe.lda = inter; probe.c_mode = C_ACCUM;
        probe.a_tiled = 1; probe.plan_only = 1;
        probe.xp_out = (f16*)xp_out; probe.xp_invperm = (const u16*)next_invperm; probe.xp_w = (const f16*)next_norm_w;
        probe.ss_out = xp_out ? ss_out : nullptr; probe.ldxp = hidden;
        probe.xp_tiled = (xp_out && chain_xp_tiled()) ? 1 : 0;
        int w = 0;
        if (qgemv_lean_launch(probe, stream, &w) != 0) tiled = false;
    }
    if (part & 1)
    {
        EXL2_REQUIRE(xp && ss, "q_mlp_forward_chain: null argument");
        // gate | up from (xp, ss); SiLU(gate) * up in the epilogue, written in down's packed order
        FlatIn in; memset(&in, 0, sizeof(in));
        in.qm[0] = m->gate; in.qm[1] = m->up; in.c[0] = act; in.c[1] = act; in.ldc[0] = inter; in.ldc[1] = inter;
        in.c_invperm[0] = m->down->q_perm ? m->down->q_invperm : nullptr;
        in.n_mats = 2; in.pair = 1; in.M = rows; in.a_mode = A_NORM_PRE; in.a = (const f16*)xp; in.lda = hidden;
        in.ss = ss; in.npart = npart; in.eps = m->norm_epsilon; in.c_mode = C_STORE;
        in.act_gelu = m->act_gelu ? 1 : 0;
        in.c_tiled = tiled ? 1 : 0;
        in.a_tiled = chain_xp_tiled() ? 1 : 0;
        FLAT_TRY(in, stream, nullptr, "q_mlp_forward_chain (gate|up)");
    }
    if (part & 2)
    {
        FlatIn in; memset(&in, 0, sizeof(in));
        in.qm[0] = m->down; in.c[0] = (f16*)x; in.ldc[0] = hidden;
        in.n_mats = 1; in.M = rows; in.a_mode = A_DIRECT; in.a = act; in.lda = inter; in.c_mode = C_ACCUM;
        in.a_tiled = tiled ? 1 : 0;
        in.xp_out = (f16*)xp_out; in.xp_invperm = (const u16*)next_invperm; in.xp_w = (const f16*)next_norm_w;
        in.ss_out = xp_out ? ss_out : nullptr; in.ldxp = hidden;
        in.xp_tiled = (xp_out && chain_xp_tiled()) ? 1 : 0;
        int wgs = 0;
        FLAT_TRY(in, stream, &wgs, "q_mlp_forward_chain (down)");
        if (npart_out) *npart_out = wgs;
    }
    return EXL2_OK;
}

int exl2_q_mlp_forward_chain(void* handle, void* x, const void* xp, const float* ss, int npart, int rows,
                             const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out, void* stream)
{
    EXL2_REQUIRE(handle && x && xp && ss, "q_mlp_forward_chain: null argument");
    return q_mlp_chain_part(handle, 3, x, xp, ss, npart, rows, 0, next_invperm, next_norm_w, xp_out, ss_out, npart_out, stream);
}

int exl2_q_mlp_forward_chain_part(void* handle, int part, int row0, void* x, const void* xp, const float* ss, int npart, int rows,
                                  const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out,
                                  void* stream)
{
    EXL2_REQUIRE(part == 1 || part == 2, "q_mlp_forward_chain_part: part %d", part);
    return q_mlp_chain_part(handle, part, x, xp, ss, npart, rows, row0, next_invperm, next_norm_w, xp_out, ss_out, npart_out, stream);
}

// reference: none (the reference has no chained decode); exllamav2_amd/model.py: GreedyGraphDecoder.step_chain brackets a step of
// 5 .. 16 rows with (1) / (0)
int exl2_chain_set_tiled(int on)
{
    g_chain_tiled = on ? 1 : 0;
    return EXL2_OK;
}

int exl2_gemm_half_q_half_chain(const void* xp, const float* ss, int npart, float eps,
                                void* q_matrix, void* c, int rows, void* stream)
{
    EXL2_REQUIRE(xp && ss && q_matrix && c, "gemm_half_q_half_chain: null argument");
    QMatrix* q = (QMatrix*)q_matrix;
    if (rows <= 0) return EXL2_OK;
    EXL2_REQUIRE(rows <= MAX_GEMV_ROWS, "gemm_half_q_half_chain: %d rows", rows);
    FlatIn in; memset(&in, 0, sizeof(in));
    in.qm[0] = q; in.c[0] = (f16*)c; in.ldc[0] = q->width;
    in.n_mats = 1; in.M = rows; in.a_mode = A_NORM_PRE; in.a = (const f16*)xp; in.lda = q->height;
    in.ss = ss; in.npart = npart; in.eps = eps; in.c_mode = C_STORE;
    in.a_tiled = chain_xp_tiled() ? 1 : 0;
    FLAT_TRY(in, stream, nullptr, "gemm_half_q_half_chain");
    return EXL2_OK;
}

}  // extern "C"
