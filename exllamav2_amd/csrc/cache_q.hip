// cache_q.hip -- quantized KV-cache codec (Q4 / Q6 / Q8) for gfx950 + C ABI.
//
// Reference: cache_q.cuh:4-185 (block = 512 consecutive fp16 elements, 256 threads x half2; 32-point Walsh-Hadamard
// over lanes; absmax per 32 contiguous elements; RTN to 4 or 8 bits; fp16 scale), addressing cache.cu:143-223 (pack,
// contiguous + paged) and :324-401 (unpack), bindings ext_cache.cpp:80-269.
// The butterflies run over xor distances < 32, i.e. inside each half of a wave64, through ds_swizzle (no LDS memory);
// arithmetic is fp16 exactly as the reference (v_pk_add_f16 / v_pk_fma_f16), so packed bytes and scales are
// reproducible bit for bit by the oracle.
#include "hw.h"
#include "errors.h"
#include "cache_q_pack.h"
#include <string.h>

#define QBLOCK 512

template <int WBITS>
DEV void fp16_to_q_block(int t, const f16* in, u8* out, f16* scales, size_t block_offset)
{
    q_pack_lane<WBITS>(t, ((const f16x2*)(in + block_offset))[t], out, scales, block_offset);
}

template <int WBITS>
DEV void q_to_fp16_block(int t, const u8* in, const f16* scales, f16* out, size_t block_offset)
{
    const f16 scale = scales[block_offset / 32 + (t >> 4)];
    f16x2 w;
    if constexpr (WBITS == 4)
    {
        const u32 q = ((const u32*)(in + block_offset / 2))[t >> 2];
        const int sh = (t & 3) * 8;
        w.x = (f16)(float)((int)((q >> sh) & 0xF) - 8);
        w.y = (f16)(float)((int)((q >> (sh + 4)) & 0xF) - 8);
    }
    else
    {
        const u32 q = ((const u32*)(in + block_offset))[t >> 1];
        const int sh = (t & 1) * 16;
        w.x = (f16)(float)((int)((q >> sh) & 0xFF) - 128);
        w.y = (f16)(float)((int)((q >> (sh + 8)) & 0xFF) - 128);
    }
    w = w * h2_dup(scale);
    w = wht32(w, t);
    w = w * h2_dup((f16)0.03125f);
    ((f16x2*)(out + block_offset))[t] = w;
}

struct QKVArgs
{
    const void* k_in; void* k_out; void* k_scales;
    const void* v_in; void* v_out; void* v_scales;
    const int* cache_seqlens; const int* block_table;
    int dim, offset, stride, blocks_x;          // contiguous mode (element units)
    int pages_per_seq, page_size, q_len;        // paged mode
    int wbits_k, wbits_v;
};

template <int DIR>      // 0: fp16 -> q ; 1: q -> fp16
DEV void codec_block(const QKVArgs& a, int kv, size_t block_offset)
{
    const int t = tid();
    const int wbits = kv ? a.wbits_v : a.wbits_k;
    if (DIR == 0)
    {
        const f16* in = (const f16*)(kv ? a.v_in : a.k_in);
        u8* out = (u8*)(kv ? a.v_out : a.k_out);
        f16* sc = (f16*)(kv ? a.v_scales : a.k_scales);
        if (wbits == 4) fp16_to_q_block<4>(t, in, out, sc, block_offset);
        else            fp16_to_q_block<8>(t, in, out, sc, block_offset);
    }
    else
    {
        const u8* in = (const u8*)(kv ? a.v_in : a.k_in);
        f16* out = (f16*)(kv ? a.v_out : a.k_out);
        const f16* sc = (const f16*)(kv ? a.v_scales : a.k_scales);
        if (wbits == 4) q_to_fp16_block<4>(t, in, sc, out, block_offset);
        else            q_to_fp16_block<8>(t, in, sc, out, block_offset);
    }
}

// contiguous: grid (width / 512, batch, 2)   (cache.cu:196-223, 375-401)
template <int DIR>
KERNEL void __launch_bounds__(256) kv_codec_kernel(const QKVArgs a)
{
    const int kv = bid_z();
    const size_t block_offset = (size_t)a.offset + (size_t)bid_y() * a.stride + (size_t)bid_x() * QBLOCK;
    codec_block<DIR>(a, kv, block_offset);
}

// paged: grid (pages_per_seq, blocks per page chunk, 2 * batch)   (cache.cu:143-195, 324-373)
template <int DIR>
KERNEL void __launch_bounds__(256) kv_codec_paged_kernel(const QKVArgs a)
{
    const int kv = bid_z() & 1;
    const int y = bid_z() >> 1;
    const int seqlen = a.cache_seqlens[y];
    // pack: only the pages the appended tokens land in have work -- the grid covers those, counted from the page of
    // `seqlen` (the reference launches one column of blocks per page of the sequence and lets them exit, cache.cu:143-195;
    // at 16k tokens that is > 30 000 idle workgroups per layer per step)
    const int x = DIR == 0 ? seqlen / a.page_size + bid_x() : bid_x();
    if (x >= a.pages_per_seq) return;
    const int page = a.block_table[a.pages_per_seq * y + x];
    const int vx_a = a.page_size * x;
    int px_a, px_b;
    if (DIR == 0) { px_a = seqlen - vx_a; px_b = px_a + a.q_len; }        // tokens being appended
    else          { px_a = 0; px_b = seqlen - vx_a; }                      // everything valid in this page
    if (a.dim % QBLOCK)
    {
        while (((long long)px_a * a.dim) % QBLOCK) px_a--;
        while (((long long)px_b * a.dim) % QBLOCK) px_b++;
    }
    px_a = px_a > 0 ? px_a : 0;
    px_b = px_b < a.page_size ? px_b : a.page_size;
    const long long block_a = ((long long)page * a.page_size + px_a) * a.dim;
    const long long block_b = ((long long)page * a.page_size + px_b) * a.dim;
    for (long long j = block_a + (long long)bid_y() * QBLOCK; j < block_b; j += (long long)gdim_y() * QBLOCK)
        codec_block<DIR>(a, kv, (size_t)j);
}

// ---- RoPE on q / new k + Q4 pack of the new k, v rows straight from registers (decode steps over a Q4 cache) --------------------
//
// What the Q4 decode path ran per layer before: rope_append_kernel (attn.hip: rotate q and k in place, copy rotated k and v into the
// cache's fp16 staging pages) and kv_codec_paged_kernel<0> (read the staging pages back, pack) -- two launches of ~5 us for a few
// KB.  Here one wave per (token, head row): q rows are rotated in place; k rows are rotated in place (the attention kernel attends
// over the step's own keys in fp16, attn_q4.hip) and packed from the registers that hold them; v rows are packed.  Same fp16
// arithmetic in the same order as the two kernels it replaces (rope.cu numerics as restated in rope_append_kernel; q_pack_lane is
// the codec's own function): codes and scales are bit-identical (tests/test_ops.py).  head_dim 128, full rotary only.
struct RopeQ4Args
{
    f16* q; f16* k_new; const f16* v_new;       // [b, s, H|KVH, 128]
    u8* k_codes; f16* k_scales; u8* v_codes; f16* v_scales;
    const f16* sin; const f16* cos;             // [max_seq, 128]
    const int* past_lens; const int* block_table;
    int b, s, H, KVH;
    int past_len, neox, rope;
    int page_size, page_shift, pages_per_seq;
};

KERNEL void __launch_bounds__(64) rope_quant_q4_kernel(const RopeQ4Args a)
{
    constexpr int HDIM = 128;
    const int slot = bid_x();                   // < H: q ; < H + KVH: k ; else v
    const int j = bid_y();
    const int b = bid_z();
    const int t = tid();
    int past = a.past_len;
    if (past == -1) { past = a.past_lens[b]; past = past > 0 ? past : 0; }
    else if (a.past_lens) past += a.past_lens[b];
    const int pos = past + j;
    const bool is_q = slot < a.H, is_v = slot >= a.H + a.KVH;
    const int h = is_q ? slot : (is_v ? slot - a.H - a.KVH : slot - a.H);
    size_t tok = 0;
    if (!is_q)
    {
        if (a.block_table)
            tok = (size_t)a.block_table[(size_t)b * a.pages_per_seq + (pos >> a.page_shift)] * a.page_size + (pos & (a.page_size - 1));
        else
            tok = (size_t)b * a.page_size + pos;
    }
    const size_t cache_off = (tok * a.KVH + h) * HDIM;            // element offset of this head row in the cache
    if (is_v)
    {
        const f16x2 w = ((const f16x2*)(a.v_new + (((size_t)b * a.s + j) * a.KVH + h) * HDIM))[t];
        q_pack_lane<4>(t, w, a.v_codes, a.v_scales, cache_off);
        return;
    }
    f16* x = is_q ? a.q + (((size_t)b * a.s + j) * a.H + h) * HDIM : a.k_new + (((size_t)b * a.s + j) * a.KVH + h) * HDIM;
    f16x2 w = ((const f16x2*)x)[t];
    if (a.rope)
    {
        const int srow = pos > 0 ? pos : 0;
        const f16* sr = a.sin + (size_t)srow * HDIM;
        const f16* cr = a.cos + (size_t)srow * HDIM;
        w = rope_lane_pair128(w, t, sr, cr, a.neox != 0);
        ((f16x2*)x)[t] = w;
    }
    if (!is_q) q_pack_lane<4>(t, w, a.k_codes, a.k_scales, cache_off);
}

static int ilog2_exact_cq(int x) { int s = 0; while ((1 << s) < x) s++; return (1 << s) == x ? s : -1; }

static int wbits_pair(int wbits, int* k, int* v)
{
    if (wbits == 4) { *k = 4; *v = 4; return 0; }
    if (wbits == 6) { *k = 8; *v = 4; return 0; }       // cache.cu:259-276: Q6 = 8-bit keys, 4-bit values
    if (wbits == 8) { *k = 8; *v = 8; return 0; }
    return -1;
}

template <int DIR>
static int kv_codec(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                    int batch_size, int dim, int seq_stride_tokens, int offset, int width, int page_size,
                    const int* cache_seqlens, const int* block_table, int pages_per_seq, int wbits, void* stream)
{
    QKVArgs a;
    memset(&a, 0, sizeof(a));
    EXL2_REQUIRE(wbits_pair(wbits, &a.wbits_k, &a.wbits_v) == 0, "q cache: wbits must be 4, 6 or 8 (got %d)", wbits);
    a.k_in = k_in; a.k_out = k_out; a.k_scales = k_scales; a.v_in = v_in; a.v_out = v_out; a.v_scales = v_scales;
    a.dim = dim;
    if (page_size)
    {
        EXL2_REQUIRE(cache_seqlens && block_table, "q cache: paged mode needs cache_seqlens and block_table");
        a.cache_seqlens = cache_seqlens; a.block_table = block_table;
        a.pages_per_seq = pages_per_seq; a.page_size = page_size; a.q_len = width;
        long long per_page_blocks = ((long long)page_size * dim + QBLOCK - 1) / QBLOCK;
        int pages_x = pages_per_seq;
        if (DIR == 0)
        {
            // appended tokens: at most (width - 1) / page_size + 2 pages, width * dim / 512 (+ alignment) blocks each
            const int touched = (width > 0 ? (width - 1) / page_size : 0) + 2;
            if (touched < pages_x) pages_x = touched;
            const long long need = ((long long)width * dim + QBLOCK - 1) / QBLOCK + 2;
            if (need < per_page_blocks) per_page_blocks = need;
        }
        if (per_page_blocks > 256) per_page_blocks = 256;
        dim3 grid((unsigned)pages_x, (unsigned)per_page_blocks, (unsigned)(2 * batch_size));
        if (DIR == 0) LAUNCH(kv_codec_paged_kernel<0>, grid, dim3(256), 0, stream, a);
        else          LAUNCH(kv_codec_paged_kernel<1>, grid, dim3(256), 0, stream, a);
    }
    else
    {
        // ext_cache.cpp:148-155: widen [offset, offset + width) tokens to whole 512-element blocks
        if (dim % QBLOCK)
        {
            while (((long long)offset * dim) % QBLOCK) offset--;
            while (((long long)width * dim) % QBLOCK) width++;
        }
        a.offset = offset * dim; a.stride = seq_stride_tokens * dim;
        const int blocks = (int)(((long long)width * dim) / QBLOCK);
        if (blocks <= 0 || batch_size <= 0) return EXL2_OK;
        dim3 grid((unsigned)blocks, (unsigned)batch_size, v_in ? 2u : 1u);
        if (DIR == 0) LAUNCH(kv_codec_kernel<0>, grid, dim3(256), 0, stream, a);
        else          LAUNCH(kv_codec_kernel<1>, grid, dim3(256), 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

extern "C" {

// fp16_to_q_kv (ext_cache.cpp:80-173).  Tensors are [batch | pages, seq | page_size, kv_heads, head_dim]; `dim` =
// kv_heads * head_dim, `seq_stride_tokens` = size(1).  page_size == 0: contiguous range [offset, offset + width) tokens
// of every batch row; page_size > 0: `width` = q_len tokens appended at cache_seqlens through block_table.
int exl2_fp16_to_q_kv(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                      int batch_size, int dim, int seq_stride_tokens, int offset, int width, int page_size,
                      const int* cache_seqlens, const int* block_table, int pages_per_seq, int wbits, void* stream)
{
    EXL2_REQUIRE(k_in && k_out && k_scales, "fp16_to_q_kv: null argument");
    return kv_codec<0>(k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, dim, seq_stride_tokens, offset, width,
                       page_size, cache_seqlens, block_table, pages_per_seq, wbits, stream);
}

// q_to_fp16_kv (ext_cache.cpp:175-269)
int exl2_q_to_fp16_kv(const void* k_in, void* k_out, const void* k_scales, const void* v_in, void* v_out,
                      const void* v_scales, int batch_size, int dim, int seq_stride_tokens, int offset, int width,
                      int page_size, const int* cache_seqlens, const int* block_table, int pages_per_seq, int wbits,
                      void* stream)
{
    EXL2_REQUIRE(k_in && k_out && k_scales, "q_to_fp16_kv: null argument");
    return kv_codec<1>(k_in, k_out, (void*)k_scales, v_in, v_out, (void*)v_scales, batch_size, dim, seq_stride_tokens,
                       offset, width, page_size, cache_seqlens, block_table, pages_per_seq, wbits, stream);
}

// RoPE on q and k_new in place + Q4 pack of the rotated k_new and of v_new into the cache at positions past_len (+ past_lens[b]) + j
// through the block table (exl2_rope_kv_append's conventions; block_table == NULL: row b of a [batch, page_size = max_seq_len] cache).
// One launch for exl2_rope_kv_append + exl2_fp16_to_q_kv (paged, wbits 4) of a decode step.  Returns 1 without launching for shapes
// it does not cover (head_dim != 128, partial rotary): the caller takes the two entry points it replaces.
int exl2_rope_quant_append_q4(void* q, void* k_new, const void* v_new, void* k_codes, void* k_scales, void* v_codes, void* v_scales,
                              const void* sin, const void* cos, int batch, int q_len, int num_heads, int num_kv_heads,
                              int head_dim, int past_len, const int* past_lens, const int* block_table,
                              int page_size, int pages_per_seq, int rope_style, int sincos_size, void* stream)
{
    EXL2_REQUIRE(q && k_new && v_new && k_codes && k_scales && v_codes && v_scales, "rope_quant_append_q4: null argument");
    EXL2_REQUIRE(rope_style == 0 || (sin && cos), "rope_quant_append_q4: sin/cos tables missing");
    EXL2_REQUIRE(past_len != -1 || past_lens, "rope_quant_append_q4: past_len == -1 needs past_lens");
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    if (head_dim != 128 || (rope_style != 0 && sincos_size > 0 && sincos_size != head_dim)) return 1;
    RopeQ4Args a;
    memset(&a, 0, sizeof(a));
    a.q = (f16*)q; a.k_new = (f16*)k_new; a.v_new = (const f16*)v_new;
    a.k_codes = (u8*)k_codes; a.k_scales = (f16*)k_scales; a.v_codes = (u8*)v_codes; a.v_scales = (f16*)v_scales;
    a.sin = (const f16*)sin; a.cos = (const f16*)cos; a.past_lens = past_lens; a.block_table = block_table;
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads;
    a.past_len = past_len; a.neox = rope_style == 2; a.rope = rope_style != 0;      // ROPE_STYLE_* q_attn.cuh:13-15
    a.page_size = page_size; a.pages_per_seq = pages_per_seq; a.page_shift = ilog2_exact_cq(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "rope_quant_append_q4: page_size must be a power of two");
    LAUNCH(rope_quant_q4_kernel, dim3((unsigned)(num_heads + 2 * num_kv_heads), (unsigned)q_len, (unsigned)batch), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"
