// attn.hip -- paged / contiguous KV attention for gfx950 that needs no flash-attn, + RoPE-and-append, + C ABI.
//
// Reference: the attention step between q_attn_forward_1 and _2 is third-party in the reference
// (flash_attn_with_kvcache, attn.py:602-613; torch fallback _attn_torch attn.py:869-937).  This kernel implements the
// same contract: softmax(q k^T * scale + causal bottom-right) v with GQA, keys addressed through a
// block table [batch, pages] over a cache viewed as [pages, page_size, kv_heads, head_dim], fp32 softmax.
//
// Decode-shaped design (flash-decoding): grid = (kv_head, split, batch); a workgroup streams its slice of the keys once
// for ALL query heads that share the kv head (GQA) and all q_len query tokens, 16 bytes per lane per K and per V row
// (HDIM/8 lanes per key -> each key/value row is one contiguous 2*HDIM-byte read), every (wave, lane-group) running an
// independent online-softmax stream; streams are merged through LDS, splits through a small combine kernel.  HBM-bound:
// algorithmic bytes = 2 * ctx * kv_heads * head_dim * 2 per layer per token.
#include "hw.h"
#include "errors.h"
#include "chain_sync.h"
#include "attn_merge.h"
#include "attn_fused_body.h"
#include <string.h>
#include <type_traits>

struct AttnArgs
{
    const f16* q;                 // [b, s, H, hd]
    const f16* k_cache;           // [pages, page_size, KVH, hd]  (block_table == null: [b, page_size, KVH, hd])
    const f16* v_cache;
    const int* cache_seqlens;     // [b] or null
    const int* block_table;       // [b, pages_per_seq] or null
    f16* out;                     // [b, s, H, hd]
    float* part_o;                // [b*s*H, nsplit, hd]
    float* part_ml;               // [b*s*H, nsplit, 2]
    int b, s, H, KVH;
    int page_size, page_shift, pages_per_seq;
    int len_const, len_offset;    // total keys = (cache_seqlens ? cache_seqlens[b] : len_const) + len_offset
    int nsplit, causal;
    float scale;
    // flash-attn's window_size = (window_left, .) and softcap (attn.py:590-600: Mistral / Gemma-type checkpoints): query at absolute
    // position p sees keys [p - window_left, p] (window_left < 0: no window); scores = softcap * tanh(q.k * scale / softcap) (0: off)
    int window_left; float softcap;
};

template <int HDIM, int RB>
KERNEL void __launch_bounds__(ATT_WAVES * 64) attn_decode_kernel(const AttnArgs a)
{
    DYN_SMEM(smem);
    constexpr int LPK = HDIM / 8;               // lanes per key
    constexpr int KPW = 64 / LPK;             // keys per wave step
    constexpr int NSTREAM = ATT_WAVES * KPW;
    constexpr int ROWF = HDIM + 2;              // floats per (stream, row) in LDS

    const int kh = bid_x();
    const int split = bid_y();
    const int b = bid_z() / ((a.s * (a.H / a.KVH) + RB - 1) / RB);
    const int rblk = bid_z() % ((a.s * (a.H / a.KVH) + RB - 1) / RB);
    const int G = a.H / a.KVH;
    const int R = a.s * G;                    // query rows sharing this kv head
    const int r0 = rblk * RB;
    const int nrows = min(RB, R - r0);

    const int lane = lane_id();
    const int wv = wave_id();
    const int group = lane / LPK;
    const int dl = lane % LPK;

    const int total = (a.cache_seqlens ? a.cache_seqlens[b] : a.len_const) + a.len_offset;
    int kps = (total + a.nsplit - 1) / a.nsplit;
    kps = (kps + 15) & ~15;
    const int k_start = split * kps;
    const int k_end = min(total, k_start + kps);

    // query fragments + causal limits
    f16x8 qf[RB];
    int limit[RB];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        const int rr = r0 + (r < nrows ? r : 0);
        const int j = rr / G, g = rr - j * G;
        qf[r] = *(const f16x8*)(a.q + (((size_t)b * a.s + j) * a.H + kh * G + g) * HDIM + dl * 8);
        limit[r] = a.causal ? (total - a.s + j + 1) : total;
    }
    // sliding window: row r sees keys >= lo[r]; keys below every row's bound are not even read (the split starts behind them)
    int lo[RB], lo_min = 0x7fffffff;
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        const int rr = r0 + (r < nrows ? r : 0);
        const int pos = total - a.s + rr / G;                 // the query's absolute position
        lo[r] = a.window_left >= 0 ? max(0, pos - a.window_left) : 0;
        lo_min = min(lo_min, lo[r]);
    }
    const float cap = a.softcap, inv_cap = cap > 0.0f ? 1.0f / cap : 0.0f;

    float m[RB], l[RB], o[RB][8];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        m[r] = NEG_BIG; l[r] = 0.0f;
        #pragma unroll
        for (int e = 0; e < 8; e++) o[r][e] = 0.0f;
    }

    const size_t row_stride = (size_t)a.KVH * HDIM;          // elements between consecutive token slots
    const int k_first = max(k_start, lo_min & ~(KPW - 1));   // (aligned down: the waves' key groups stay where they were)
    for (int base = k_first + wv * KPW; base < k_end; base += ATT_WAVES * KPW)
    {
        const int kpos = base + group;
        const bool in_range = kpos < k_end;
        const int kp = in_range ? kpos : k_start;           // keep the address valid, mask the score
        size_t tok;
        if (a.block_table)
            tok = (size_t)a.block_table[(size_t)b * a.pages_per_seq + (kp >> a.page_shift)] * a.page_size
                  + (kp & (a.page_size - 1));
        else
            tok = (size_t)b * a.page_size + kp;
        const size_t off = tok * row_stride + (size_t)kh * HDIM + dl * 8;
        const f16x8 kf = ld_nt((const f16x8*)(a.k_cache + off));
        const f16x8 vf = ld_nt((const f16x8*)(a.v_cache + off));

        #pragma unroll
        for (int r = 0; r < RB; r++)
        {
            if (r < nrows)
            {
                float d = 0.0f;
                #pragma unroll
                for (int e = 0; e < 4; e++)
                    d = dot2_f32_f16((f16x2){qf[r][2 * e], qf[r][2 * e + 1]}, (f16x2){kf[2 * e], kf[2 * e + 1]}, d);
                d = group_allreduce_add<LPK>(d);
                float sc = d * a.scale;
                if (cap > 0.0f) sc = cap * tanhf(sc * inv_cap);
                const bool valid = in_range && kpos < limit[r] && kpos >= lo[r];
                const float m_new = valid ? fmaxf(m[r], sc) : m[r];
                const float alpha = fast_exp(m[r] - m_new);
                const float p = valid ? fast_exp(sc - m_new) : 0.0f;
                m[r] = m_new;
                l[r] = l[r] * alpha + p;
                #pragma unroll
                for (int e = 0; e < 8; e++) o[r][e] = o[r][e] * alpha + p * (float)vf[e];
            }
        }
    }

    // merge the NSTREAM independent softmax streams of this workgroup
    float* st = (float*)smem;
    const int stream = wv * KPW + group;
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        if (r < nrows)
        {
            float* p = st + ((size_t)stream * RB + r) * ROWF;
            #pragma unroll
            for (int e = 0; e < 8; e++) p[dl * 8 + e] = o[r][e];
            if (dl == 0) { p[HDIM] = m[r]; p[HDIM + 1] = l[r]; }
        }
    }
    block_sync();
    for (int idx = tid(); idx < nrows * HDIM; idx += nthreads())
    {
        const int r = idx / HDIM, d = idx - r * HDIM;
        float M = NEG_BIG;
        for (int s2 = 0; s2 < NSTREAM; s2++) M = fmaxf(M, st[((size_t)s2 * RB + r) * ROWF + HDIM]);
        float L = 0.0f, O = 0.0f;
        for (int s2 = 0; s2 < NSTREAM; s2++)
        {
            const float* p = st + ((size_t)s2 * RB + r) * ROWF;
            const float w = fast_exp(p[HDIM] - M);
            L += p[HDIM + 1] * w;
            O += p[d] * w;
        }
        const int rr = r0 + r;
        const int j = rr / G, g = rr - j * G;
        const size_t qrow = ((size_t)b * a.s + j) * a.H + kh * G + g;
        if (a.nsplit == 1)
        {
            a.out[qrow * HDIM + d] = (f16)(L > 0.0f ? O / L : 0.0f);
        }
        else
        {
            a.part_o[(qrow * a.nsplit + split) * HDIM + d] = O;
            if (d == 0)
            {
                a.part_ml[(qrow * a.nsplit + split) * 2 + 0] = M;
                a.part_ml[(qrow * a.nsplit + split) * 2 + 1] = L;
            }
        }
    }
}

KERNEL void __launch_bounds__(256) attn_combine_kernel(const AttnArgs a, int hd)
{
    const size_t qrow = bid_x();
    for (int d = tid(); d < hd; d += nthreads())
    {
        a.out[qrow * hd + d] = (f16)merge_split_partials<false>(a.part_o, a.part_ml, qrow, a.nsplit, a.nsplit, hd, d);
    }
}

// ---- one-launch decode step: the body lives in attn_fused_body.h (a device function since round 6) ----
template <int HDIM, int RB, bool DEP = false>
KERNEL void __launch_bounds__(ATT_WAVES * 64) attn_fused_kernel(const FusedArgs a)
{
    DYN_SMEM(smem);
    attn_fused_body<HDIM, RB, DEP>(a, bid_x(), bid_y(), bid_z(), smem);
}

// ---- RoPE on q / new k + append of new k, v into the (paged) cache at device-side positions ---------------------------

struct RopeAppendArgs
{
    f16* q; f16* k_new; const f16* v_new;       // [b, s, H|KVH, hd]
    f16* k_cache; f16* v_cache;                 // nullable: no append
    const f16* sin; const f16* cos;             // [max_seq, sincos_size]
    const int* past_lens;                       // [b] or null
    const int* block_table;                     // [b, pages_per_seq] or null
    int b, s, H, KVH, hd;
    int past_len, neox, sincos_size, rope;      // rope == 0: no rotation (append only)
    int page_size, page_shift, pages_per_seq;
};

KERNEL void __launch_bounds__(64) rope_append_kernel(const RopeAppendArgs a)
{
    // one 64-thread workgroup per (token, head slot); slot < H: q ; < H + KVH: k ; else v
    const int slot = bid_x();
    const int j = bid_y();
    const int b = bid_z();
    int past = a.past_len;
    if (past == -1) { past = a.past_lens[b]; past = past > 0 ? past : 0; }
    else if (a.past_lens) past += a.past_lens[b];
    const int pos = past + j;

    size_t tok = 0;
    if (a.k_cache)
    {
        if (a.block_table)
            tok = (size_t)a.block_table[(size_t)b * a.pages_per_seq + (pos >> a.page_shift)] * a.page_size
                  + (pos & (a.page_size - 1));
        else
            tok = (size_t)b * a.page_size + pos;
    }

    const int t = tid();
    if (slot >= a.H + a.KVH)
    {
        if (!a.v_cache) return;
        const int h = slot - a.H - a.KVH;
        const f16* src = a.v_new + (((size_t)b * a.s + j) * a.KVH + h) * a.hd;
        f16* dst = a.v_cache + (tok * a.KVH + h) * a.hd;
        for (int i = t; i < (a.hd >> 3); i += 64) ((f16x8*)dst)[i] = ((const f16x8*)src)[i];
        return;
    }
    const bool is_k = slot >= a.H;
    const int h = is_k ? slot - a.H : slot;
    f16* x = is_k ? a.k_new + (((size_t)b * a.s + j) * a.KVH + h) * a.hd
                  : a.q + (((size_t)b * a.s + j) * a.H + h) * a.hd;
    f16* dst = (is_k && a.k_cache) ? a.k_cache + (tok * a.KVH + h) * a.hd : nullptr;

    const int srow = pos > 0 ? pos : 0;
    const f16* sr = a.sin + (size_t)srow * a.sincos_size;
    const f16* cr = a.cos + (size_t)srow * a.sincos_size;
    const int half_dim = a.sincos_size >> 1;
    // rotate: one thread per rotation pair
    for (int c = t; c < (a.hd >> 1); c += 64)
    {
        if (a.neox)
        {
            // pairs (c, c + half_dim) for c < half_dim; columns >= sincos_size pass through
            int c0, c1;
            bool rot = a.rope && c < half_dim;
            if (c < half_dim) { c0 = c; c1 = c + half_dim; }
            else { c0 = a.sincos_size + 2 * (c - half_dim); c1 = c0 + 1; }      // unrotated tail, two columns per thread
            f16 l = x[c0], r = x[c1];
            if (rot)
            {
                const f16 cs = cr[c], sn = sr[c];
                const f16 ls = r * (-sn);
                const f16 rs = l * sn;
                const f16 l2 = h_fma(l, cs, ls);
                const f16 r2 = h_fma(r, cs, rs);
                l = l2; r = r2;
                x[c0] = l; x[c1] = r;
            }
            if (dst) { dst[c0] = l; dst[c1] = r; }
        }
        else
        {
            const int c0 = 2 * c, c1 = 2 * c + 1;
            f16 x0 = x[c0], x1 = x[c1];
            if (a.rope && c0 < a.sincos_size)
            {
                const f16 r0 = h_fma(x1, -sr[c0], x0 * cr[c0]);
                const f16 r1 = h_fma(x0, sr[c1], x1 * cr[c1]);
                x0 = r0; x1 = r1;
                x[c0] = x0; x[c1] = x1;
            }
            if (dst) { dst[c0] = x0; dst[c1] = x1; }
        }
    }
}

// Round 6: the same work for prompts (thousands of tokens) with 16-byte accesses -- the kernel above moves two bytes per lane (0.37 ms
// per layer of an 8 x 2048 prefill for 0.8 GB of traffic: 2.2 TB/s).  One 256-thread workgroup per token; a unit = 8 rotation pairs
// of a q / k head (columns [8 u, 8 u + 8) and their partners half a head away: two 16-byte loads, the sin / cos rows as 16-byte loads,
// two or four 16-byte stores) or 8 elements of a v head.  NeoX pairing over the whole head (sincos_size == head_dim), head_dim a
// multiple of 16; the fp16 operations per element are those of rope_append_kernel in the same order: bit-identical results.
KERNEL void __launch_bounds__(256) rope_append_rows_kernel(const RopeAppendArgs a)
{
    const int j = bid_x(), b = bid_y();
    int past = a.past_len;
    if (past == -1) { past = a.past_lens[b]; past = past > 0 ? past : 0; }
    else if (a.past_lens) past += a.past_lens[b];
    const int pos = past + j;
    size_t tok = 0;
    if (a.k_cache)
    {
        if (a.block_table) tok = (size_t)a.block_table[(size_t)b * a.pages_per_seq + (pos >> a.page_shift)] * a.page_size + (pos & (a.page_size - 1));
        else tok = (size_t)b * a.page_size + pos;
    }
    const int srow = pos > 0 ? pos : 0;
    const f16* const sr = a.sin + (size_t)srow * a.hd;
    const f16* const cr = a.cos + (size_t)srow * a.hd;
    const int half = a.hd >> 1, upr = half >> 3;                       // units per rotated head
    const int n_rot = (a.H + a.KVH) * upr;
    const int upv = a.hd >> 3;
    const int n_all = n_rot + ((a.v_new && a.v_cache) ? a.KVH * upv : 0);
    const size_t row = (size_t)b * a.s + j;
    for (int unit = tid(); unit < n_all; unit += 256)
    {
        if (unit >= n_rot)
        {
            const int v = unit - n_rot, h = v / upv, c8 = v - h * upv;
            ((f16x8*)(a.v_cache + (tok * a.KVH + h) * a.hd))[c8] = ((const f16x8*)(a.v_new + (row * a.KVH + h) * a.hd))[c8];
            continue;
        }
        const int hs = unit / upr, c8 = unit - hs * upr;
        const bool is_k = hs >= a.H;
        const int h = is_k ? hs - a.H : hs;
        f16* const x = is_k ? a.k_new + (row * a.KVH + h) * a.hd : a.q + (row * a.H + h) * a.hd;
        f16x8 l = ((const f16x8*)x)[c8], r = ((const f16x8*)(x + half))[c8];
        if (a.rope)
        {
            const f16x8 cs = ((const f16x8*)cr)[c8], sn = ((const f16x8*)sr)[c8];
            #pragma unroll
            for (int e = 0; e < 8; e++)
            {
                const f16 ls = r[e] * (-sn[e]);
                const f16 rs = l[e] * sn[e];
                const f16 l2 = h_fma(l[e], cs[e], ls);
                const f16 r2 = h_fma(r[e], cs[e], rs);
                l[e] = l2; r[e] = r2;
            }
            ((f16x8*)x)[c8] = l; ((f16x8*)(x + half))[c8] = r;
        }
        if (is_k && a.k_cache)
        {
            f16* const dst = a.k_cache + (tok * a.KVH + h) * a.hd;
            ((f16x8*)dst)[c8] = l; ((f16x8*)(dst + half))[c8] = r;
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------

static int ilog2_exact(int x) { int s = 0; while ((1 << s) < x) s++; return (1 << s) == x ? s : -1; }

template <int HDIM>
static void launch_decode(const AttnArgs& a, int rb, dim3 grid, void* stream)
{
    const int lpk = HDIM / 8, kpw = 64 / lpk;
    const size_t lds = (size_t)ATT_WAVES * kpw * rb * (HDIM + 2) * 4;
    static bool attr_done[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr_done))
    {
        (void)hipFuncSetAttribute((const void*)attn_decode_kernel<HDIM, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_decode_kernel<HDIM, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    switch (rb)
    {
        case 1: LAUNCH((attn_decode_kernel<HDIM, 1>), grid, dim3(ATT_WAVES * 64), lds, stream, a); break;
        case 2: LAUNCH((attn_decode_kernel<HDIM, 2>), grid, dim3(ATT_WAVES * 64), lds, stream, a); break;
        case 4: LAUNCH((attn_decode_kernel<HDIM, 4>), grid, dim3(ATT_WAVES * 64), lds, stream, a); break;
        default: LAUNCH((attn_decode_kernel<HDIM, 8>), grid, dim3(ATT_WAVES * 64), lds, stream, a); break;
    }
}

extern "C" {

// bytes of fp32 scratch exl2_paged_attn needs for (rows = b * s * H, nsplit)
long long exl2_paged_attn_scratch_bytes(int rows, int head_dim, int nsplit)
{
    return nsplit <= 1 ? 0 : (long long)rows * nsplit * (head_dim + 2) * 4;
}

// Attention over a paged (block_table != null, page_size a power of two) or contiguous (block_table == null, cache
// [b, page_size, KVH, hd]) fp16 KV cache that already holds all keys.  Keys per sequence = (cache_seqlens ?
// cache_seqlens[i] : len_const) + len_offset; query token j attends keys [0, total - q_len + j + 1) when causal.
// nsplit <= 0 picks a split count from the grid size.
// window_left >= 0: flash-attn's sliding window (a query at absolute position p sees keys [p - window_left, p]); softcap > 0: scores =
// softcap * tanh(q.k * scale / softcap) -- the two keyword arguments the reference passes for Mistral / Gemma-type checkpoints
// (attn.py:590-600).
int exl2_paged_attn_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                       const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       int window_left, float softcap, void* stream);
int exl2_paged_attn(const void* q, const void* k_cache, const void* v_cache, void* out,
                    const int* cache_seqlens, const int* block_table,
                    int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                    int page_size, int pages_per_seq, int len_const, int len_offset,
                    float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes, void* stream)
{
    return exl2_paged_attn_ex(q, k_cache, v_cache, out, cache_seqlens, block_table, batch, q_len, num_heads, num_kv_heads, head_dim, page_size,
                              pages_per_seq, len_const, len_offset, softmax_scale, causal, nsplit, scratch, scratch_bytes, -1, 0.0f, stream);
}
int exl2_paged_attn_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                       const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       int window_left, float softcap, void* stream)
{
    EXL2_REQUIRE(q && k_cache && v_cache && out, "paged_attn: null argument");
    EXL2_REQUIRE(window_left < 0 || causal, "paged_attn: a sliding window needs causal attention");
    EXL2_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256, "paged_attn: head_dim %d unsupported (64/128/256)", head_dim);
    EXL2_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attn: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const f16*)q; a.k_cache = (const f16*)k_cache; a.v_cache = (const f16*)v_cache; a.out = (f16*)out;
    a.cache_seqlens = cache_seqlens; a.block_table = block_table;
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads;
    a.page_size = page_size; a.pages_per_seq = pages_per_seq;
    a.page_shift = ilog2_exact(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "paged_attn: page_size %d must be a power of two", page_size);
    a.len_const = len_const; a.len_offset = len_offset; a.causal = causal; a.scale = softmax_scale;
    a.window_left = window_left; a.softcap = softcap > 0.0f ? softcap : 0.0f;

    const int G = num_heads / num_kv_heads;
    const int R = q_len * G;
    const int rb = R >= 8 ? 8 : (R >= 4 ? 4 : (R >= 2 ? 2 : 1));
    const int rblocks = (R + rb - 1) / rb;
    if (nsplit <= 0)
    {
        // ~2 workgroups per CU
        const long long base = (long long)num_kv_heads * batch * rblocks;
        nsplit = (int)((512 + base - 1) / base);
        if (nsplit > 16) nsplit = 16;
        if (nsplit < 1) nsplit = 1;
    }
    const long long need = exl2_paged_attn_scratch_bytes(batch * q_len * num_heads, head_dim, nsplit);
    if (need > scratch_bytes || (need > 0 && !scratch)) nsplit = 1;        // no scratch: single pass
    a.nsplit = nsplit;
    if (nsplit > 1)
    {
        a.part_o = (float*)scratch;
        a.part_ml = a.part_o + (size_t)batch * q_len * num_heads * nsplit * head_dim;
    }
    dim3 grid((unsigned)num_kv_heads, (unsigned)nsplit, (unsigned)(batch * rblocks));
    if (head_dim == 64) launch_decode<64>(a, rb, grid, stream);
    else if (head_dim == 128) launch_decode<128>(a, rb, grid, stream);
    else launch_decode<256>(a, rb, grid, stream);
    if (nsplit > 1)
        LAUNCH(attn_combine_kernel, dim3((unsigned)(batch * q_len * num_heads)), dim3(head_dim < 256 ? head_dim : 256), 0,
               stream, a, head_dim);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// One-launch decode-step attention (see attn_fused_kernel).  Returns 1 (nothing launched) when the shape is outside what
// the fused kernel covers -- the caller then uses exl2_rope_kv_append + exl2_paged_attn.  q / k_new are NOT modified.
// `counters`: >= batch * kv_heads * row_blocks zeroed u32, left zeroed.  Positions: past = past_const + cache_seqlens[b].
static int attn_decode_fused_impl(const void* q, const void* k_new, const void* v_new, void* k_cache, void* v_cache, void* out,
                           const void* sin, const void* cos, const int* cache_seqlens, const int* block_table,
                           int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                           int page_size, int pages_per_seq, int past_const, float softmax_scale,
                           int rope_style, int sincos_size, int nsplit, void* scratch, long long scratch_bytes,
                           void* counters, int n_counters, const void* out_invperm, void* out_natural, void* stream)
{
    EXL2_REQUIRE(q && k_new && v_new && k_cache && v_cache && out, "attn_decode_fused: null argument");
    EXL2_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "attn_decode_fused: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    EXL2_REQUIRE(rope_style == 0 || (sin && cos), "attn_decode_fused: sin/cos tables missing");
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    if (!(head_dim == 64 || head_dim == 128 || head_dim == 256)) return 1;
    if (!(rope_style == 0 || (rope_style == 2 && (sincos_size <= 0 || sincos_size == head_dim)))) return 1;
    const int G = num_heads / num_kv_heads;
    const int R = q_len * G;
    if (R > 32) return 1;
    if (block_table && pages_per_seq > ATT_MAX_PAGES) return 1;      // (the kernel keeps a split's page ids in LDS)
    // few KV heads (GQA): prefer more, lighter workgroups -- 4 query rows each (the KV stream is small and re-read per
    // row block) and up to 64 splits; with many KV heads 8 rows share one pass over the keys
    const bool gqa_small = (long long)num_kv_heads * batch < 64;
    int rb_ = (R >= 8 && !gqa_small) ? 8 : (R >= 4 ? 4 : (R >= 2 ? 2 : 1));
    // round 6, few KV heads: two query rows per workgroup (twice the workgroups over the same keys, which the L2 serves) -- same box,
    // Mixtral 3.5 bpw bs=1: 434 -> 444 tok/s at ctx <= 36, 390 -> 408 at 1920, 349 -> 360 at 8000, 278 -> 270 at 30000; TinyLlama GPTQ
    // 1450 -> 1534 (profiles/r09bc_attention_row_blocks.txt).  The row block is fixed when a step is captured, the context is not:
    // two rows where the cache cannot hold more than 16384 tokens per sequence
    if (gqa_small && rb_ > 2 && (long long)page_size * pages_per_seq <= 16384) rb_ = 2;
    { static const int rb_env = []() { const char* e = getenv("EXL2_ATT_RB"); return e ? atoi(e) : 0; }();        // (A/B: rows per block of the one-launch decode attention)
      if ((rb_env == 1 || rb_env == 2 || rb_env == 4 || rb_env == 8) && rb_env <= rb_) rb_ = rb_env; }
    const int rb = rb_;
    const int rblocks = (R + rb - 1) / rb;
    if (!counters || (long long)batch * num_kv_heads * rblocks > n_counters) return 1;
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const f16*)q; a.k_new = (const f16*)k_new; a.v_new = (const f16*)v_new;
    a.k_cache = (f16*)k_cache; a.v_cache = (f16*)v_cache; a.out = (f16*)out;
    a.sin = (const f16*)sin; a.cos = (const f16*)cos;
    a.cache_seqlens = cache_seqlens; a.block_table = block_table; a.counters = (u32*)counters;
    a.out_invperm = (const u16*)out_invperm;
    a.out_nat = out_invperm ? (f16*)out_natural : nullptr;      // (without a permutation `out` IS the natural-order copy)
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads;
    a.page_size = page_size; a.pages_per_seq = pages_per_seq; a.page_shift = ilog2_exact(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "attn_decode_fused: page_size %d must be a power of two", page_size);
    a.past_const = past_const; a.rope = rope_style != 0; a.scale = softmax_scale;
    {
        // keys one workgroup takes before the step is split over several (a split costs the partial-result hand-off)
        // (read per call: tools/attn_bench.py sweeps it inside one process)
        const char* e = getenv("EXL2_ATT_KPS"); const int v = e ? atoi(e) : 0;
        a.keys_per_split_min = v >= 16 ? v : ATT_KPS_DEFAULT;
        // few KV heads: two slopes -- a split per 64 keys up to 16 splits, then one per 256 keys (measured: see fused_active_splits)
        if (v < 16 && gqa_small && !getenv("EXL2_ATT_ONE_SLOPE")) { a.keys_per_split_min = 64; a.split_short_cap = 16; a.keys_per_split_long = 256; }
    }
    if (nsplit <= 0)
    {
        const long long base = (long long)num_kv_heads * batch * rblocks;
        // (few KV heads: 4-row workgroups hold 66 KB of LDS, two per CU -- a grid beyond ~768 of them makes the workgroups of unused splits
        // queue in front of the working ones: 48 splits x 2 row blocks x 8 KV heads measured better than 64 x 2 x 8 at every context)
        // (many KV heads: up to 32 splits since round 5 -- with the batched merge 32 slices beat 16 from ~4000 keys on by 5-10 %)
        nsplit = (int)(((gqa_small ? 768 : 1024) + base - 1) / base);
        if (nsplit > (gqa_small ? 64 : 32)) nsplit = gqa_small ? 64 : 32;
        if (nsplit < 1) nsplit = 1;
        // (measurement switch, read per call: the grid's number of splits -- below the policy's figure: what do the workgroups of unused
        // splits cost a short-context launch?  above it, up to 64: more, shorter key slices at a long context)
        const char* e = getenv("EXL2_ATT_NSPLIT_MAX"); const int cap = e ? atoi(e) : 0;
        if (cap >= 1) { const long long fill = (long long)(4096 + base - 1) / base; nsplit = cap < 64 ? cap : 64; if (nsplit > fill) nsplit = (int)fill; if (nsplit < 1) nsplit = 1; }
    }
    long long need = exl2_paged_attn_scratch_bytes(batch * q_len * num_heads, head_dim, nsplit);
    while (nsplit > 1 && (need > scratch_bytes || !scratch)) { nsplit /= 2; need = exl2_paged_attn_scratch_bytes(batch * q_len * num_heads, head_dim, nsplit); }
    a.nsplit = nsplit;
    if (nsplit > 1)
    {
        a.part_o = (float*)scratch;
        a.part_ml = a.part_o + (size_t)batch * q_len * num_heads * nsplit * head_dim;
    }
    dim3 grid((unsigned)num_kv_heads, (unsigned)nsplit, (unsigned)(batch * rblocks));
    const int lpk = head_dim / 8, kpw = 64 / lpk;
    const size_t lds = (size_t)ATT_WAVES * kpw * rb * (head_dim + 2) * 4 + 16 + (size_t)ATT_MAX_PAGES * 4;
    const bool overlapped = chain_sync_active();
    if (overlapped)
    {
        ChainLaunch cl;
        const int e = chain_sync_next(&cl);
        if (e) return e;
        a.sync_wait = cl.wait; a.sync_signal = cl.signal; a.sync_total = (u32)(batch * num_kv_heads * rblocks);
        a.sync_arrive = cl.arrive;
        stream = cl.stream;
    }
#define FUSED_CASE(HDIM_, RB_) do { if (overlapped) LAUNCH((attn_fused_kernel<HDIM_, RB_, true>), grid, dim3(ATT_WAVES * 64), lds, stream, a); \
                                    else LAUNCH((attn_fused_kernel<HDIM_, RB_, false>), grid, dim3(ATT_WAVES * 64), lds, stream, a); } while (0)
#define FUSED_HD(HDIM_) \
    do { static bool attr_done[EXL2_MAX_DEVICES] = {false}; \
         if (exl2_first_on_device(attr_done)) { (void)hipFuncSetAttribute((const void*)attn_fused_kernel<HDIM_, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                           (void)hipFuncSetAttribute((const void*)attn_fused_kernel<HDIM_, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                           (void)hipFuncSetAttribute((const void*)attn_fused_kernel<HDIM_, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                           (void)hipFuncSetAttribute((const void*)attn_fused_kernel<HDIM_, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
         switch (rb) { case 1: FUSED_CASE(HDIM_, 1); break; case 2: FUSED_CASE(HDIM_, 2); break; \
                       case 4: FUSED_CASE(HDIM_, 4); break; default: FUSED_CASE(HDIM_, 8); break; } } while (0)
    if (head_dim == 64) FUSED_HD(64);
    else if (head_dim == 128) FUSED_HD(128);
    else FUSED_HD(256);
#undef FUSED_HD
#undef FUSED_CASE
    HIP_TRY(hipGetLastError());
    if (overlapped) { const int e = chain_sync_done((u32)(grid.x * grid.y * grid.z)); if (e) return e; }
    return EXL2_OK;
}

int exl2_attn_decode_fused(const void* q, const void* k_new, const void* v_new, void* k_cache, void* v_cache, void* out,
                           const void* sin, const void* cos, const int* cache_seqlens, const int* block_table,
                           int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                           int page_size, int pages_per_seq, int past_const, float softmax_scale,
                           int rope_style, int sincos_size, int nsplit, void* scratch, long long scratch_bytes,
                           void* counters, int n_counters, const void* out_invperm, void* stream)
{
    return attn_decode_fused_impl(q, k_new, v_new, k_cache, v_cache, out, sin, cos, cache_seqlens, block_table, batch, q_len, num_heads,
                                  num_kv_heads, head_dim, page_size, pages_per_seq, past_const, softmax_scale, rope_style, sincos_size,
                                  nsplit, scratch, scratch_bytes, counters, n_counters, out_invperm, nullptr, stream);
}

// The same launch with TWO copies of the output: `out` through out_invperm (o_proj's packed order, what exl2_q_attn_forward_2_chain
// reads) and `out_natural` in flash-attn's order -- the tensor the reference host receives from flash_attn_func (attn.py:960-977)
// and hands to q_attn_forward_2 (attn.py:1195-1203): the module chain behind the operator boundary (dropin/_exl2_fast.cpp) recognises
// that tensor and reads the packed copy instead.
int exl2_attn_decode_fused_dual(const void* q, const void* k_new, const void* v_new, void* k_cache, void* v_cache, void* out,
                                const void* sin, const void* cos, const int* cache_seqlens, const int* block_table,
                                int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                                int page_size, int pages_per_seq, int past_const, float softmax_scale,
                                int rope_style, int sincos_size, int nsplit, void* scratch, long long scratch_bytes,
                                void* counters, int n_counters, const void* out_invperm, void* out_natural, void* stream)
{
    EXL2_REQUIRE(!out_invperm || out_natural, "attn_decode_fused_dual: out_natural missing");
    return attn_decode_fused_impl(q, k_new, v_new, k_cache, v_cache, out_invperm ? out : (out_natural ? out_natural : out), sin, cos,
                                  cache_seqlens, block_table, batch, q_len, num_heads,
                                  num_kv_heads, head_dim, page_size, pages_per_seq, past_const, softmax_scale, rope_style, sincos_size,
                                  nsplit, scratch, scratch_bytes, counters, n_counters, out_invperm, out_natural, stream);
}

// RoPE on q and k_new in place (rope.cu numerics) and, when caches are given, append of the rotated k_new and of v_new at
// positions past_len (+ past_lens[b]) + j through the block table -- the "k = new_k, v = new_v" half of
// flash_attn_with_kvcache, with positions read on the device so the launch can live in a HIP graph.
int exl2_rope_kv_append(void* q, void* k_new, const void* v_new, void* k_cache, void* v_cache,
                        const void* sin, const void* cos, int batch, int q_len, int num_heads, int num_kv_heads,
                        int head_dim, int past_len, const int* past_lens, const int* block_table,
                        int page_size, int pages_per_seq, int rope_style, int sincos_size, void* stream)
{
    EXL2_REQUIRE(q && k_new, "rope_kv_append: null argument");
    EXL2_REQUIRE(rope_style == 0 || (sin && cos), "rope_kv_append: sin/cos tables missing");
    EXL2_REQUIRE(past_len != -1 || past_lens, "rope_kv_append: past_len == -1 needs past_lens");
    EXL2_REQUIRE(head_dim % 8 == 0, "rope_kv_append: bad head_dim");
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    RopeAppendArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (f16*)q; a.k_new = (f16*)k_new; a.v_new = (const f16*)v_new; a.k_cache = (f16*)k_cache; a.v_cache = (f16*)v_cache;
    a.sin = (const f16*)sin; a.cos = (const f16*)cos; a.past_lens = past_lens; a.block_table = block_table;
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads; a.hd = head_dim;
    a.past_len = past_len; a.neox = rope_style == 2; a.rope = rope_style != 0;      // ROPE_STYLE_* q_attn.cuh:13-15
    a.sincos_size = sincos_size > 0 ? sincos_size : head_dim;
    a.page_size = page_size; a.pages_per_seq = pages_per_seq; a.page_shift = ilog2_exact(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "rope_kv_append: page_size must be a power of two");
    // prompts: one workgroup per token with 16-byte accesses (NeoX over the whole head, or no rotation at all); a decode step's few
    // rows keep the head-per-workgroup kernel (more, smaller workgroups)
    static const int rows_min = []() { const char* e = getenv("EXL2_ROPE_ROWS_MIN"); return e ? atoi(e) : 64; }();
    const bool aligned = ((((size_t)q) | ((size_t)k_new) | ((size_t)v_new) | ((size_t)k_cache) | ((size_t)v_cache) | ((size_t)sin) | ((size_t)cos)) & 15) == 0;
    if ((long long)batch * q_len >= rows_min && head_dim % 16 == 0 && aligned && (!a.rope || (a.neox && a.sincos_size == head_dim)) && batch <= 65535)
    {
        LAUNCH(rope_append_rows_kernel, dim3((unsigned)q_len, (unsigned)batch, 1), dim3(256), 0, stream, a);
        HIP_TRY(hipGetLastError());
        return EXL2_OK;
    }
    const int slots = num_heads + num_kv_heads + ((v_new && v_cache) ? num_kv_heads : 0);
    LAUNCH(rope_append_kernel, dim3((unsigned)slots, (unsigned)q_len, (unsigned)batch), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"
