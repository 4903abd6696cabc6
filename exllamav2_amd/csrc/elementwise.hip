// elementwise.hip -- RMSNorm, RoPE, act(gate)*up for gfx950, + their C ABI.
//
// Reference: rms_norm.cu:33-175 (numerics: clamp +-65504, fp32 sum of squares, rsqrtf, (x*w)*r in fp32, RN to fp16),
// rope.cu:10-174 (fp16 hfma2 rotation from sin/cos tables), q_mlp_activation.cuh:54-112 (act_mul), bindings
// ext_norm.cpp:22-110, ext_rope.cpp:21-62.  All three are HBM/launch-latency bound: 16-byte accesses, one pass where the
// row fits in registers.
#include "hw.h"
#include "errors.h"
#include "chain_sync.h"

// ---- RMSNorm --------------------------------------------------------------------------------------------------------

struct RmsArgs
{
    const void* x; const f16* w; void* y;
    float eps, r_dim;
    int rows, dim;
    int add_residual, input_fp32, output_fp32;
};

DEV float clampf(float f) { return fmaxf(-65504.0f, fminf(f, 65504.0f)); }

KERNEL void __launch_bounds__(256) rms_norm_kernel(const RmsArgs a)
{
    SHARED float part[4];
    const int row = bid_x();
    const int t = tid();
    const int lane = lane_id();
    const int wv = wave_id();
    const int dim8 = a.dim >> 3;          // host guarantees dim % 8 == 0

    float ss = 0.0f;
    if (!a.input_fp32)
    {
        const f16x8* xr = (const f16x8*)((const f16*)a.x + (size_t)row * a.dim);
        for (int i = t; i < dim8; i += 256)
        {
            const f16x8 v = xr[i];
            #pragma unroll
            for (int e = 0; e < 8; e++) { const float f = clampf((float)v[e]); ss = fmaf(f, f, ss); }
        }
    }
    else
    {
        const f32x4* xr = (const f32x4*)((const float*)a.x + (size_t)row * a.dim);
        for (int i = t; i < dim8 * 2; i += 256)
        {
            const f32x4 v = xr[i];
            #pragma unroll
            for (int e = 0; e < 4; e++) ss = fmaf(v[e], v[e], ss);       // fp32 input is not clamped (rms_norm.cu:92-99)
        }
    }
    ss = wave_allreduce_add(ss);
    if (lane == 0) part[wv] = ss;
    block_sync();
    ss = part[0] + part[1] + part[2] + part[3];
    const float rmf = fast_rsqrt(ss * a.r_dim + a.eps);

    const f16x8* wr = (const f16x8*)a.w;
    for (int i = t; i < dim8; i += 256)
    {
        float f[8];
        if (!a.input_fp32)
        {
            const f16x8 v = ((const f16x8*)((const f16*)a.x + (size_t)row * a.dim))[i];
            #pragma unroll
            for (int e = 0; e < 8; e++) f[e] = clampf((float)v[e]);
        }
        else
        {
            const f32x4* xr = (const f32x4*)((const float*)a.x + (size_t)row * a.dim);
            const f32x4 v0 = xr[2 * i], v1 = xr[2 * i + 1];
            #pragma unroll
            for (int e = 0; e < 4; e++) { f[e] = v0[e]; f[4 + e] = v1[e]; }
        }
        const f16x8 wv8 = wr[i];
        float n[8];
        #pragma unroll
        for (int e = 0; e < 8; e++) n[e] = f[e] * (float)wv8[e] * rmf;
        if (!a.output_fp32)
        {
            f16x8* yr = (f16x8*)((f16*)a.y + (size_t)row * a.dim);
            f16x8 o;
            #pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (f16)n[e];
            if (a.add_residual) o = yr[i] + o;                             // __hadd2(y, n) (rms_norm.cu:139)
            yr[i] = o;
        }
        else
        {
            f32x4* yr = (f32x4*)((float*)a.y + (size_t)row * a.dim);
            f32x4 o0 = {n[0], n[1], n[2], n[3]}, o1 = {n[4], n[5], n[6], n[7]};
            if (a.add_residual) { o0 += yr[2 * i]; o1 += yr[2 * i + 1]; }
            yr[2 * i] = o0; yr[2 * i + 1] = o1;
        }
    }
}

// ---- RoPE -----------------------------------------------------------------------------------------------------------

struct RopeArgs
{
    f16* x_q; f16* x_k;                 // [batch, rows_per_batch, head_dim] each (x_k nullable)
    const f16* sin; const f16* cos;     // [max_seq_len, sincos_size]
    const int* past_lens;               // nullable
    int rows_q, rows_k;                 // rows per batch
    int head_dim, heads_q, heads_k;
    int past_len, neox, sincos_size;
};

DEV void rope_rows(f16* x, int rows_per_batch, int num_heads, const RopeArgs& a)
{
    // one thread per (row, pair of rotation pairs): NeoX pairs (c, c + half); GPT-J pairs (2i, 2i+1)
    const int b = bid_z();
    const int per_row = a.neox ? (a.sincos_size >> 2) : (a.sincos_size >> 1);     // half2 units
    const int idx = bid_x() * nthreads() + tid();
    if (idx >= rows_per_batch * per_row) return;
    const int row = idx / per_row;
    const int col = (idx - row * per_row) * 2;

    int past = a.past_len;
    if (past == -1) { past = a.past_lens[b]; past = past > 0 ? past : 0; }        // rope.cu:39-47
    else if (a.past_lens) past += a.past_lens[b];
    int srow = past + row / num_heads;
    srow = srow > 0 ? srow : 0;

    f16* xr = x + ((size_t)b * rows_per_batch + row) * a.head_dim;
    const f16* sr = a.sin + (size_t)srow * a.sincos_size;
    const f16* cr = a.cos + (size_t)srow * a.sincos_size;
    if (a.neox)
    {
        const int half_dim = a.sincos_size >> 1;
        const f16x2 c2 = *(const f16x2*)(cr + col);
        const f16x2 s2 = *(const f16x2*)(sr + col);
        const f16x2 l = *(const f16x2*)(xr + col);
        const f16x2 r = *(const f16x2*)(xr + col + half_dim);
        const f16x2 ls = r * (-s2);
        const f16x2 rs = l * s2;
        *(f16x2*)(xr + col) = h2_fma(l, c2, ls);
        *(f16x2*)(xr + col + half_dim) = h2_fma(r, c2, rs);
    }
    else
    {
        const f16x2 c01 = *(const f16x2*)(cr + col);
        f16x2 s01 = *(const f16x2*)(sr + col);
        s01.x = -s01.x;
        const f16x2 x01 = *(const f16x2*)(xr + col);
        const f16x2 x10 = {x01.y, x01.x};
        *(f16x2*)(xr + col) = h2_fma(x10, s01, x01 * c01);
    }
}

KERNEL void __launch_bounds__(256) rope_kernel(const RopeArgs a)
{
    if (bid_y() == 0) rope_rows(a.x_q, a.rows_q, a.heads_q, a);
    else              rope_rows(a.x_k, a.rows_k, a.heads_k, a);
}

// ---- act(x) * y -----------------------------------------------------------------------------------------------------

struct ActMulArgs { f16* x; const f16* y; int rows, width; int gelu; const f16* r_weights; int r_stride; };

DEV f16 act_silu(f16 g)
{
    const float gf = (float)g;
    return (f16)(gf / (1.0f + fast_exp(-gf)));
}
DEV f16 act_gelu(f16 g)
{
    // q_mlp_activation.cuh:23-34 (tanh form, fp32)
    const float x = (float)g;
    const float t = 0.797884560803f * (x + 0.044715f * x * x * x);
    return (f16)(0.5f * x * (1.0f + tanhf(t)));
}

KERNEL void __launch_bounds__(256) act_mul_kernel(const ActMulArgs a)
{
    const int row = bid_y();
    const int i = bid_x() * 256 + tid();
    if (i >= (a.width >> 3)) return;
    if (a.r_weights && as_u16(a.r_weights[(size_t)row * a.r_stride]) == 0) return;
    f16x8* xr = (f16x8*)(a.x + (size_t)row * a.width);
    const f16x8 g = xr[i];
    const f16x8 u = ((const f16x8*)(a.y + (size_t)row * a.width))[i];
    f16x8 o;
    #pragma unroll
    for (int e = 0; e < 8; e++)
    {
        f16 v = (a.gelu ? act_gelu(g[e]) : act_silu(g[e])) * u[e];
        v = v > (f16)65504.0f ? (f16)65504.0f : v;
        v = v < (f16)-65504.0f ? (f16)-65504.0f : v;
        o[e] = v;
    }
    xr[i] = o;
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------

extern "C" {

int exl2_rms_norm(const void* x, const void* w, void* y, float epsilon, int rows, int dim,
                  int add_residual, int input_fp32, int output_fp32, void* stream)
{
    EXL2_REQUIRE(x && w && y, "rms_norm: null argument");
    EXL2_REQUIRE(dim > 0 && dim % 8 == 0, "rms_norm: dim %d must be a multiple of 8", dim);
    if (rows <= 0) return EXL2_OK;
    RmsArgs a;
    a.x = x; a.w = (const f16*)w; a.y = y; a.eps = epsilon; a.r_dim = 1.0f / (float)dim;
    a.rows = rows; a.dim = dim; a.add_residual = add_residual; a.input_fp32 = input_fp32; a.output_fp32 = output_fp32;
    LAUNCH(rms_norm_kernel, dim3((unsigned)rows), dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// rope_cuda_qk (rope.cu:220-273): x_k may be null (rope_cuda, :176-218)
int exl2_rope_qk(void* x_q, void* x_k, const void* sin, const void* cos, int batch_size,
                 int rows_per_batch_q, int rows_per_batch_k, int head_dim, int num_heads_q, int num_heads_k,
                 int past_len, const int* past_lens, int neox_style, int sincos_size, void* stream)
{
    EXL2_REQUIRE(x_q && sin && cos, "rope: null argument");
    EXL2_REQUIRE(sincos_size > 0 && sincos_size % 4 == 0 && sincos_size <= head_dim, "rope: bad sincos_size %d", sincos_size);
    EXL2_REQUIRE(past_len != -1 || past_lens, "rope: past_len == -1 needs past_lens");
    if (batch_size <= 0) return EXL2_OK;
    RopeArgs a;
    a.x_q = (f16*)x_q; a.x_k = (f16*)x_k; a.sin = (const f16*)sin; a.cos = (const f16*)cos; a.past_lens = past_lens;
    a.rows_q = rows_per_batch_q; a.rows_k = x_k ? rows_per_batch_k : 0;
    a.head_dim = head_dim; a.heads_q = num_heads_q; a.heads_k = num_heads_k > 0 ? num_heads_k : 1;
    a.past_len = past_len; a.neox = neox_style; a.sincos_size = sincos_size;
    const int per_row = neox_style ? (sincos_size >> 2) : (sincos_size >> 1);
    const int rows = rows_per_batch_q > a.rows_k ? rows_per_batch_q : a.rows_k;
    const long long work = (long long)rows * per_row;
    dim3 grid((unsigned)((work + 255) / 256), x_k ? 2u : 1u, (unsigned)batch_size);
    LAUNCH(rope_kernel, grid, dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_act_mul(void* x, const void* y, int rows, int width, int act_gelu,
                 const void* r_weights, int r_weights_stride, void* stream)
{
    EXL2_REQUIRE(x && y, "act_mul: null argument");
    EXL2_REQUIRE(width % 8 == 0, "act_mul: width %d must be a multiple of 8", width);
    if (rows <= 0) return EXL2_OK;
    ActMulArgs a;
    a.x = (f16*)x; a.y = (const f16*)y; a.rows = rows; a.width = width; a.gelu = act_gelu;
    a.r_weights = (const f16*)r_weights; a.r_stride = r_weights_stride;
    dim3 grid((unsigned)(((width >> 3) + 255) / 256), (unsigned)rows, 1);
    LAUNCH(act_mul_kernel, grid, dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"

// ---- decode-loop utilities: keep the whole greedy step on the device ----------------------------
// embedding gather (reference: CPU nn.Embedding + H2D copy, model.py:1014-1016), greedy argmax (test_inference.py:607),
// and the per-step increment of the device-side sequence lengths.

KERNEL void __launch_bounds__(256) embed_rows_kernel(const f16* table, const int* ids, f16* out, int hidden, int vocab)
{
    const int row = bid_x();
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f16x8* src = (const f16x8*)(table + (size_t)id * hidden);
    f16x8* dst = (f16x8*)(out + (size_t)row * hidden);
    for (int i = tid(); i < (hidden >> 3); i += 256) dst[i] = src[i];
}

// embedding row -> x, x times the first consumer's norm weight in that consumer's packed order, and the row's sum of squares
// (npart = 1): what the chained decode (qgemv_flat.h: A_NORM_PRE) expects from the producer of a residual stream
KERNEL void __launch_bounds__(256) embed_rows_chain_kernel(const f16* table, const int* ids, f16* out, int hidden, int vocab,
                                                           const u16* invperm, const f16* next_w, f16* xp, float* ss, int tiled)
{
    SHARED float part[4];
    const int row = bid_x();
    int id = ids ? ids[row] : row;                       // (no ids: row r of `table` -- exl2_publish_rows)
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f16x8* src = (const f16x8*)(table + (size_t)id * hidden);
    f16x8* dst = out ? (f16x8*)(out + (size_t)row * hidden) : nullptr;
    f16* xr = xp + (size_t)row * hidden;
    float sq = 0.0f;
    for (int i = tid(); i < (hidden >> 3); i += 256)
    {
        const f16x8 v = src[i];
        if (dst) dst[i] = v;
        u32 idx[8];
        if (invperm)
        {
            const u32x4 pv = ((const u32x4*)invperm)[i];
            idx[0] = pv.x & 0xFFFF; idx[1] = pv.x >> 16; idx[2] = pv.y & 0xFFFF; idx[3] = pv.y >> 16;
            idx[4] = pv.z & 0xFFFF; idx[5] = pv.z >> 16; idx[6] = pv.w & 0xFFFF; idx[7] = pv.w >> 16;
        }
        else
        {
            #pragma unroll
            for (int e = 0; e < 8; e++) idx[e] = (u32)(8 * i + e);
        }
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            const float f = fmaxf(-65504.0f, fminf((float)v[e], 65504.0f));
            sq = fmaf(f, f, sq);
            // (tiled: the layout the lean kernel's XMEM form reads -- qgemv_flat.h: FlatIn.a_tiled)
            f16* const dstp = tiled ? xp + ((size_t)(idx[e] >> 3) * 16 + row) * 8 + (idx[e] & 7) : xr + idx[e];
            *dstp = next_w ? (f16)fmaxf(-65504.0f, fminf(f * (float)next_w[idx[e]], 65504.0f)) : v[e];
        }
    }
    sq = wave_allreduce_add(sq);
    if (lane_id() == 0) part[wave_id()] = sq;
    block_sync();
    if (tid() == 0) ss[row] = (part[0] + part[1]) + (part[2] + part[3]);
}

KERNEL void __launch_bounds__(256) gather_f16_kernel(const f16* src, const u16* perm, f16* dst, int n)
{
    const int i = bid_x() * 256 + tid();
    if (i < n) dst[i] = src[perm ? (int)perm[i] : i];
}

KERNEL void __launch_bounds__(1024) argmax_rows_kernel(const f16* logits, int* out_ids, int vocab, int ld,
                                                       int* history, int* hist_pos, int hist_stride, int pos_inc)
{
    SHARED float best_v[16];
    SHARED int best_i[16];
    const int row = bid_x();
    const f16* lr = logits + (size_t)row * ld;
    float bv = -3.0e38f; int bi = 0;
    // 16-byte loads when the row allows it; a thread visits its octets in ascending order and inside an octet the
    // elements in ascending order, so `>` keeps the lowest index among equal values
    const bool vec = ((ld & 7) == 0) && ((((size_t)logits) & 15) == 0);
    const int n8 = vec ? (vocab >> 3) : 0;
    for (int o = tid(); o < n8; o += nthreads())
    {
        const f16x8 v8 = ((const f16x8*)lr)[o];
        #pragma unroll
        for (int e = 0; e < 8; e++) { const float v = (float)v8[e]; if (v > bv) { bv = v; bi = o * 8 + e; } }
    }
    for (int i = n8 * 8 + tid(); i < vocab; i += nthreads())
    {
        const float v = (float)lr[i];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    // wave reduction: larger value wins, ties -> lower index (torch.argmax returns the first maximum)
    for (int mask = 1; mask < 64; mask <<= 1)
    {
        const float ov = shfl_xor_f32(bv, mask);
        const int oi = (int)shfl_xor_u32((u32)bi, mask);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane_id() == 0) { best_v[wave_id()] = bv; best_i[wave_id()] = bi; }
    block_sync();
    if (tid() == 0)
    {
        const int nw = nthreads() >> 6;
        for (int w = 1; w < nw; w++)
            if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) { bv = best_v[w]; bi = best_i[w]; }
        out_ids[row] = bi;
        // token log at the position read on the device; pos_inc != 0: this launch also advances the position (the decode
        // loop's `cache_seqlens += 1`, one launch less per step)
        if (hist_pos)
        {
            const int pos = hist_pos[row] + pos_inc;
            if (pos_inc) hist_pos[row] = pos;
            if (history) history[(size_t)row * hist_stride + pos] = bi;
        }
    }
}

KERNEL void __launch_bounds__(64) add_i32_kernel(int* p, int n, int v)
{
    const int i = bid_x() * 64 + tid();
    if (i < n) p[i] += v;
}

extern "C" {

int exl2_embed_rows(const void* table, const int* ids, void* out, int rows, int hidden, int vocab, void* stream)
{
    EXL2_REQUIRE(table && ids && out, "embed_rows: null argument");
    EXL2_REQUIRE(hidden % 8 == 0, "embed_rows: hidden %d must be a multiple of 8", hidden);
    if (rows <= 0) return EXL2_OK;
    LAUNCH(embed_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const f16*)table, ids, (f16*)out, hidden, vocab);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_embed_rows_chain(const void* table, const int* ids, void* x, int rows, int hidden, int vocab,
                          const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, void* stream)
{
    EXL2_REQUIRE(table && ids && x && xp_out && ss_out, "embed_rows_chain: null argument");
    EXL2_REQUIRE(hidden % 8 == 0, "embed_rows_chain: hidden %d must be a multiple of 8", hidden);
    EXL2_REQUIRE(!next_invperm || (((size_t)next_invperm) & 15) == 0, "embed_rows_chain: invperm must be 16-byte aligned");
    if (rows <= 0) return EXL2_OK;
    LAUNCH(embed_rows_chain_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const f16*)table, ids, (f16*)x, hidden, vocab,
           (const u16*)next_invperm, (const f16*)next_norm_w, (f16*)xp_out, ss_out, chain_xp_tiled() ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// The hand-off a chained producer would have left, made from rows that are already in memory: xp_out = x * next_norm_w in the
// consumer's packed order, ss_out[row] = sum of squares (npart = 1).  What the module chain behind the operator boundary
// (dropin/_exl2_fast.cpp) runs when a module's input did not come from the module it expected.
int exl2_publish_rows(const void* x, int rows, int hidden, const void* next_invperm, const void* next_norm_w,
                      void* xp_out, float* ss_out, void* stream)
{
    EXL2_REQUIRE(x && xp_out && ss_out, "publish_rows: null argument");
    EXL2_REQUIRE(hidden % 8 == 0, "publish_rows: hidden %d must be a multiple of 8", hidden);
    EXL2_REQUIRE(!next_invperm || (((size_t)next_invperm) & 15) == 0, "publish_rows: invperm must be 16-byte aligned");
    if (rows <= 0) return EXL2_OK;
    LAUNCH(embed_rows_chain_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const f16*)x, (const int*)nullptr, (f16*)nullptr, hidden, rows,
           (const u16*)next_invperm, (const f16*)next_norm_w, (f16*)xp_out, ss_out, chain_xp_tiled() ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_gather_f16(const void* src, const void* perm, void* dst, int n, void* stream)
{
    EXL2_REQUIRE(src && dst, "gather_f16: null argument");
    if (n <= 0) return EXL2_OK;
    LAUNCH(gather_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const f16*)src, (const u16*)perm, (f16*)dst, n);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_argmax_rows(const void* logits, int* out_ids, int rows, int vocab, int ld,
                     int* history, int* hist_pos, int hist_stride, int pos_inc, void* stream)
{
    EXL2_REQUIRE(logits && out_ids, "argmax_rows: null argument");
    EXL2_REQUIRE(!history || hist_pos, "argmax_rows: history needs hist_pos");
    if (rows <= 0) return EXL2_OK;
    LAUNCH(argmax_rows_kernel, dim3((unsigned)rows), dim3(1024), 0, stream, (const f16*)logits, out_ids, vocab, ld,
           history, hist_pos, hist_stride, pos_inc);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_add_i32(int* p, int n, int value, void* stream)
{
    EXL2_REQUIRE(p, "add_i32: null argument");
    if (n <= 0) return EXL2_OK;
    LAUNCH(add_i32_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, p, n, value);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"
