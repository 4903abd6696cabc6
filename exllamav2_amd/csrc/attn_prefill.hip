// attn_prefill.hip -- MFMA flash attention for prefill-shaped steps (many query rows per sequence) over the FP16 KV cache.
//
// Replaces, on the product path, what the reference does for long queries without flash-attn: `_attn_torch`
// (attn.py:869-937: torch SDPA / matmul-softmax-matmul with a lower-right causal mask, GQA by head repetition), and what it
// gets from flash-attn with it (`flash_attn_func` / `flash_attn_with_kvcache` with q_len > 1, attn.py:602-613, 960-977).
// Same contract as exl2_paged_attn (attn.hip): q [b, s, H, hd] already rotated, K/V already appended to the cache; the cache
// is [pages, page_size, KVH, hd] behind a block table or [b, T, KVH, hd] without one; total keys of sequence b =
// len_const + cache_seqlens[b] + len_offset, query row j sits at position total - s + j (lower-right aligned causal mask).
//
// Shape of the kernel (gfx950, 64-wide waves, v_mfma_f32_16x16x32_f16):
//   * one workgroup = 64 query rows of one (sequence, head): 4 waves x 16 rows; grid (ceil(s / 64), H, b) -- a 2048-token
//     prefill of a 32-head model is 1024 workgroups per sequence, >> 256 CUs;
//   * keys in tiles of 32: K and V tiles row-major in LDS (16-byte coalesced loads, 16-byte LDS writes); the next tile's
//     global loads are in flight while the current one is computed;
//   * scores are computed transposed, S^T = K Q^T (A = K rows from LDS, B = Q rows held in registers for the whole kernel):
//     the accumulator layout of S^T -- lane (q = lane % 16, g = lane / 16) holds keys 4g .. 4g+3 of a 16-key block -- IS the
//     B-operand layout of the second product O^T = V^T P^T once the 32 keys of a tile are taken in the order
//     {4g .. 4g+3 of block 0, 4g .. 4g+3 of block 1}: the probabilities never leave their registers, and V^T's A operand
//     is two transposing LDS reads (ds_read_b64_tr_b16: lane (feature i, g) gets keys 4g .. 4g+3 of feature column i) per 16
//     output features.  (The first version wrote V transposed with 2-byte LDS stores: 32-way bank conflicts, 100 TFLOP/s.)
//   * online softmax per query = per accumulator column: 8 local values + two cross-lane steps (lanes l, l^16, l^32, l^48),
//     fp32 statistics, exp in fp32, P rounded to fp16 for the MFMA (as flash-attn does);
//   * O^T accumulators: hd / 16 blocks of 4 fp32 per lane; final 1 / l, 8-byte stores.
// Bound: MFMA (2 * s * T * hd FLOP per head) -- 16 MFMAs per 32 keys per wave against ~60 VALU ops of softmax.
#include "hw.h"
#include "errors.h"
#include <string.h>

#ifndef FP_RPW
#define FP_RPW 2                   // 16-row query blocks per wave: every K / V fragment read from LDS feeds FP_RPW MFMAs
#endif
#define FP_WAVES_C 4
#define FP_BQ (FP_WAVES_C * 16 * FP_RPW)
#ifndef FP_BK
#define FP_BK 64
#endif
#define FP_WAVES 4
#ifndef FP_OCC
#define FP_OCC (FP_RPW == 1 ? 4 : 2)   // waves per SIMD the register allocation is held to (= workgroups of 4 waves per CU)
#endif
#ifndef FP_QK_BATCH
#define FP_QK_BATCH 0              // (round-5 experiment, RPW = 1 only, measured slower: DESIGN.md 3.5)
#endif
#define FP_NEG_BIG (-1.0e30f)

struct FlashArgs
{
    const f16* q; const f16* k_cache; const f16* v_cache; f16* out;
    const int* cache_seqlens; const int* block_table;
    int b, s, H, KVH;
    int page_size, page_shift, pages_per_seq;
    int len_const, len_offset;
    float scale;
    int causal;
};
struct FlashWS { int window_left; float softcap; };     // flash-attn's window_size[0] (< 0: none) and softcap (0: off): csrc/attn.hip AttnArgs


// WS: the instantiation with flash-attn's sliding window / softcap (a template parameter, not run-time tests: as run-time tests they cost
// the plain kernel 30 scalar spills and 30 % of its speed -- 0.83 -> 1.09 ms at 8 x 2048 x 32 heads, round 6)
template <int HDIM, bool WS = false>
KERNEL void __launch_bounds__(FP_WAVES * 64, FP_OCC) flash_prefill_kernel(const FlashArgs a, const FlashWS ws)
{
    DYN_SMEM(smem);
    constexpr int KSTR = HDIM + 8;                 // halfs per K row (272-byte rows at hd 128: conflict-free 16-byte reads)
    constexpr int VSTR = HDIM + 16;                // halfs per V row: rows 8 banks apart (mod 64) -> conflict-free tr reads
    constexpr int KK = HDIM / 32;                  // MFMAs (k = 32 features) per 16-key score block
    constexpr int DB = HDIM / 16;                  // 16-feature output blocks
    constexpr int CHUNKS = FP_BK * HDIM / 8;       // 16-byte pieces of a K (or V) tile
    constexpr int CPT = (CHUNKS + FP_WAVES * 64 - 1) / (FP_WAVES * 64);
    f16* k_lds = (f16*)smem;
    f16* v_lds = k_lds + FP_BK * KSTR;

    const int t = tid(), lane = lane_id(), wv = wave_id();
    const int qi = lane & 15, g = lane >> 4;
    const int q0 = bid_x() * FP_BQ, h = bid_y(), b = bid_z();
    const int kh = h / (a.H / a.KVH);

    int total = a.len_const + a.len_offset;
    if (a.cache_seqlens) { const int p = a.cache_seqlens[b]; total += p > 0 ? p : 0; }
    const int qpos0 = total - a.s;                                            // position of query row 0
    const int kend = a.causal ? min(total, qpos0 + q0 + FP_BQ) : total;       // keys this workgroup can see
    const int n_tiles = (kend + FP_BK - 1) / FP_BK;
    const size_t row_stride = (size_t)a.KVH * HDIM;

    // Cache slot of a tile's first key: ONE page look-up per tile, made two tiles ahead (round 5).  A tile of FP_BK keys never straddles
    // a page (page sizes are powers of two >= FP_BK, checked by the host), so its rows are base + row.  Before, every 16-byte piece of
    // a tile looked its page up itself behind `if (block_table)`: a dependent load per piece, and the compiler's wait for it --
    // `s_waitcnt vmcnt(0)` where the two paths meet, executed with or without a table -- drained the pieces already requested: a tile's
    // eight row loads left in four dependent round trips instead of together, and the kernel waited ~6 us per tile for ~0.4 us of MFMAs.
    auto tile_base = [&](int tile) -> size_t {
        const int kp = tile * FP_BK;
        if (a.block_table)
            return (size_t)a.block_table[(size_t)b * a.pages_per_seq + (kp >> a.page_shift)] * a.page_size + (kp & (a.page_size - 1));
        return (size_t)b * a.page_size + kp;
    };

    // this wave's RPW x 16 query rows as B operands of S^T = K Q^T: lane (qi, g) holds Q[row 16 n + qi][32 kk + 8 g .. + 8] of block n.
    // RPW = 2 (round 5): every K / V fragment read from LDS feeds TWO MFMAs -- at 16 rows per wave the kernel was bound by LDS bandwidth
    // (a wave re-read the whole 32 KB K + V tile for 16 query rows: 640 KB per CU and tile round against 128 B per cycle)
    constexpr int RPW = FP_RPW;
    int qrow[RPW], qpos[RPW];
    f16x8 qb[RPW][KK];
    #pragma unroll
    for (int n = 0; n < RPW; n++)
    {
        qrow[n] = q0 + (wv * RPW + n) * 16 + qi;
        const int qrow_c = qrow[n] < a.s ? qrow[n] : a.s - 1;
        const f16* qp = a.q + (((size_t)b * a.s + qrow_c) * a.H + h) * HDIM + 8 * g;
        #pragma unroll
        for (int kk = 0; kk < KK; kk++) qb[n][kk] = *(const f16x8*)(qp + 32 * kk);
        qpos[n] = qpos0 + qrow[n];                                            // last key this query may see (causal)
    }
    const int wave_kmax = a.causal ? qpos0 + q0 + (wv * RPW + RPW) * 16 - 1 : total - 1;   // nothing beyond it matters to this wave

    f32x4 ot[RPW][DB];
    float m_run[RPW], l_run[RPW];
    #pragma unroll
    for (int n = 0; n < RPW; n++)
    {
        #pragma unroll
        for (int d = 0; d < DB; d++) ot[n][d] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        m_run[n] = FP_NEG_BIG; l_run[n] = 0.0f;
    }

    // tile loads: thread -> 16-byte pieces c = t + 256 i; piece c = (key row c / (HDIM / 8), feature octet c % (HDIM / 8))
    f16x8 kreg[CPT], vreg[CPT];
    static_assert(CHUNKS % (FP_WAVES * 64) == 0, "a tile is a whole number of 16-byte pieces per thread");
    auto fetch = [&](int tile, size_t base) {
        // (every request unconditional, all of a tile's 2 CPT loads back to back)
        const int last = total - 1 - tile * FP_BK;                           // rows beyond the sequence re-read its last key (masked below)
        #pragma unroll
        for (int i = 0; i < CPT; i++)
        {
            const int c = t + i * FP_WAVES * 64;
            const int row = c / (HDIM / 8), oct = c % (HDIM / 8);
            const size_t off = (base + (size_t)(row < last ? row : last)) * row_stride + (size_t)kh * HDIM + oct * 8;
            kreg[i] = *(const f16x8*)(a.k_cache + off);
            vreg[i] = *(const f16x8*)(a.v_cache + off);
        }
    };
    auto stage = [&]() {
        #pragma unroll
        for (int i = 0; i < CPT; i++)
        {
            const int c = t + i * FP_WAVES * 64;
            const int row = c / (HDIM / 8), oct = c % (HDIM / 8);
            *(f16x8*)(k_lds + row * KSTR + oct * 8) = kreg[i];
            *(f16x8*)(v_lds + row * VSTR + oct * 8) = vreg[i];
        }
    };

    // (the look-up runs two tiles ahead of the tile in use, the rows one tile ahead; a tile index past the end looks tile 0 up again)
    // sliding window: every row of this workgroup sits at position >= qpos0 + q0, so no row sees a key below that minus the window
    int tile_lo = 0;
    float cap = 0.0f, inv_cap = 0.0f;
    size_t base_next;
    if constexpr (WS)
    {
        tile_lo = ws.window_left >= 0 ? max(0, qpos0 + q0 - ws.window_left) / FP_BK : 0;
        cap = ws.softcap; inv_cap = cap > 0.0f ? 1.0f / cap : 0.0f;
        base_next = n_tiles > tile_lo + 1 ? tile_base(tile_lo + 1) : 0;
        if (n_tiles > tile_lo) fetch(tile_lo, tile_base(tile_lo));
    }
    else
    {
        base_next = n_tiles > 1 ? tile_base(1) : 0;
        if (n_tiles > 0) fetch(0, tile_base(0));
    }
    for (int tile = WS ? tile_lo : 0; tile < n_tiles; tile++)
    {
        block_sync();                                                         // everybody is done with the previous tile
        stage();
        block_sync();
        if (tile + 1 < n_tiles)
        {
            fetch(tile + 1, base_next);                                       // in flight during the MFMAs below
            base_next = tile_base(tile + 2 < n_tiles ? tile + 2 : 0);
        }
        const int k0 = tile * FP_BK;
        if (k0 > wave_kmax) continue;                                         // fully masked for this wave's rows

        // S^T: FP_BK / 16 blocks of 16 keys; one LDS read of a K fragment per RPW MFMAs
        constexpr int NBLK = FP_BK / 16;
        f32x4 st[RPW][NBLK];
        #pragma unroll
        for (int blk = 0; blk < NBLK; blk++)
        {
            #pragma unroll
            for (int n = 0; n < RPW; n++) st[n][blk] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const f16* kp = k_lds + (16 * blk + qi) * KSTR + 8 * g;
            #pragma unroll
            for (int kk = 0; kk < KK; kk++)
            {
                const f16x8 kf = *(const f16x8*)(kp + 32 * kk);
                #pragma unroll
                for (int n = 0; n < RPW; n++) st[n][blk] = mfma_16x16x32_f16(kf, qb[n][kk], st[n][blk]);
            }
        }
        // mask + online softmax for column qi of each query block (lane holds keys k0 + 16 blk + 4 g + r)
        f16x8 pb[RPW][NBLK / 2];                     // B operands: 32 keys each = {block 2n keys 4g..4g+3, block 2n+1 keys 4g..4g+3}
        float alpha[RPW];
        #pragma unroll
        for (int n = 0; n < RPW; n++)
        {
            float sc[4 * NBLK];
            float m_loc = FP_NEG_BIG;
            #pragma unroll
            for (int blk = 0; blk < NBLK; blk++)
                #pragma unroll
                for (int r = 0; r < 4; r++)
                {
                    const int kpos = k0 + 16 * blk + 4 * g + r;
                    float v;
                    if constexpr (WS)
                    {
                        const bool valid = kpos < total && (!a.causal || kpos <= qpos[n]) && (ws.window_left < 0 || kpos >= qpos[n] - ws.window_left);
                        float sv = st[n][blk][r] * a.scale;
                        if (cap > 0.0f) sv = cap * tanhf(sv * inv_cap);
                        v = valid ? sv : FP_NEG_BIG;
                    }
                    else
                    {
                        const bool valid = kpos < total && (!a.causal || kpos <= qpos[n]);
                        v = valid ? st[n][blk][r] * a.scale : FP_NEG_BIG;
                    }
                    sc[blk * 4 + r] = v;
                    m_loc = fmaxf(m_loc, v);
                }
            m_loc = fmaxf(m_loc, shfl_xor_f32(m_loc, 16));
            m_loc = fmaxf(m_loc, shfl_xor_f32(m_loc, 32));
            const float m_new = fmaxf(m_run[n], m_loc);
            alpha[n] = fast_exp(m_run[n] - m_new);
            float p_sum = 0.0f;
            #pragma unroll
            for (int e = 0; e < 4 * NBLK; e++)
            {
                const float p = sc[e] > 0.5f * FP_NEG_BIG ? fast_exp(sc[e] - m_new) : 0.0f;
                p_sum += p;
                pb[n][e / 8][e % 8] = (f16)p;
            }
            p_sum += shfl_xor_f32(p_sum, 16);
            p_sum += shfl_xor_f32(p_sum, 32);
            l_run[n] = l_run[n] * alpha[n] + p_sum;
            m_run[n] = m_new;
        }
        // O^T = alpha O^T + V^T P^T: A = V^T rows (feature 16 db + qi), keys {4g..4g+3, 16+4g..16+4g+3} -- transposing reads:
        // this lane points at key row 4 g + qi / 4, features 16 db + 4 (qi % 4) .. + 3 and receives column qi of the block;
        // one pair of reads per RPW MFMAs
        const f16* vbase = v_lds + (4 * g + (qi >> 2)) * VSTR + 4 * (qi & 3);
        #pragma unroll
        for (int d = 0; d < DB; d++)
        {
            #pragma unroll
            for (int n = 0; n < RPW; n++) { ot[n][d][0] *= alpha[n]; ot[n][d][1] *= alpha[n]; ot[n][d][2] *= alpha[n]; ot[n][d][3] *= alpha[n]; }
            #pragma unroll
            for (int m2 = 0; m2 < NBLK / 2; m2++)
            {
                const f16x4 lo = lds_read_tr16_b64(vbase + (32 * m2) * VSTR + 16 * d), hi = lds_read_tr16_b64(vbase + (32 * m2 + 16) * VSTR + 16 * d);
                const f16x8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                #pragma unroll
                for (int n = 0; n < RPW; n++) ot[n][d] = mfma_16x16x32_f16(va, pb[n][m2], ot[n][d]);
            }
        }
    }

    // O^T[feature 16 d + 4 g + r][query qi] / l -> out[query][feature]: 4 consecutive features per lane
    #pragma unroll
    for (int n = 0; n < RPW; n++)
    {
        if (qrow[n] < a.s)
        {
            const float inv = l_run[n] > 0.0f ? 1.0f / l_run[n] : 0.0f;
            f16* op = a.out + (((size_t)b * a.s + qrow[n]) * a.H + h) * HDIM + 4 * g;
            #pragma unroll
            for (int d = 0; d < DB; d++)
            {
                const f16x4 y = {(f16)(ot[n][d][0] * inv), (f16)(ot[n][d][1] * inv), (f16)(ot[n][d][2] * inv), (f16)(ot[n][d][3] * inv)};
                *(f16x4*)(op + 16 * d) = y;
            }
        }
    }
}

static int fp_ilog2_exact(int x) { int s = 0; while ((1 << s) < x) s++; return (1 << s) == x ? s : -1; }

extern "C" {

// Prefill-shaped attention over the FP16 cache (see the header of this file).  Returns 1 (nothing launched) for a head_dim
// outside {64, 128, 256}: the caller then uses exl2_paged_attn.
int exl2_flash_prefill_ex(const void* q, const void* k_cache, const void* v_cache, void* out, const int* cache_seqlens,
                          const int* block_table, int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                          int page_size, int pages_per_seq, int len_const, int len_offset, float softmax_scale, int causal,
                          int window_left, float softcap, void* stream);
int exl2_flash_prefill(const void* q, const void* k_cache, const void* v_cache, void* out, const int* cache_seqlens,
                       const int* block_table, int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset, float softmax_scale, int causal,
                       void* stream)
{
    return exl2_flash_prefill_ex(q, k_cache, v_cache, out, cache_seqlens, block_table, batch, q_len, num_heads, num_kv_heads, head_dim, page_size,
                                 pages_per_seq, len_const, len_offset, softmax_scale, causal, -1, 0.0f, stream);
}
// the same with flash-attn's sliding window and softcap (exl2_paged_attn_ex, csrc/attn.hip)
int exl2_flash_prefill_ex(const void* q, const void* k_cache, const void* v_cache, void* out, const int* cache_seqlens,
                          const int* block_table, int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                          int page_size, int pages_per_seq, int len_const, int len_offset, float softmax_scale, int causal,
                          int window_left, float softcap, void* stream)
{
    EXL2_REQUIRE(window_left < 0 || causal, "flash_prefill: a sliding window needs causal attention");
    EXL2_REQUIRE(q && k_cache && v_cache && out, "flash_prefill: null argument");
    EXL2_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "flash_prefill: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    if (!(head_dim == 64 || head_dim == 128 || head_dim == 256)) return 1;
    FlashArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const f16*)q; a.k_cache = (const f16*)k_cache; a.v_cache = (const f16*)v_cache; a.out = (f16*)out;
    a.cache_seqlens = cache_seqlens; a.block_table = block_table;
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads;
    a.page_size = page_size; a.pages_per_seq = pages_per_seq; a.page_shift = fp_ilog2_exact(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "flash_prefill: page_size %d must be a power of two", page_size);
    if (block_table && page_size < FP_BK) return 1;            // (a tile of FP_BK keys must not straddle a page: not covered, the caller takes exl2_paged_attn)
    a.len_const = len_const; a.len_offset = len_offset; a.scale = softmax_scale; a.causal = causal;
    FlashWS w; w.window_left = window_left; w.softcap = softcap > 0.0f ? softcap : 0.0f;
    dim3 grid((unsigned)((q_len + FP_BQ - 1) / FP_BQ), (unsigned)num_heads, (unsigned)batch);
    const size_t lds = ((size_t)FP_BK * (head_dim + 8) + (size_t)FP_BK * (head_dim + 16)) * sizeof(f16);
    {
        // head_dim 256 asks for 68,608 bytes of dynamic LDS: above the 64 KB default limit like the other large kernels
        static bool attr[EXL2_MAX_DEVICES] = {false};
        if (exl2_first_on_device(attr))
        {
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)flash_prefill_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    }
    const bool ws = w.window_left >= 0 || w.softcap > 0.0f;
    switch (head_dim)
    {
        case 64:  if (ws) LAUNCH((flash_prefill_kernel<64, true>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  else LAUNCH((flash_prefill_kernel<64, false>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  break;
        case 128: if (ws) LAUNCH((flash_prefill_kernel<128, true>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  else LAUNCH((flash_prefill_kernel<128, false>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  break;
        default:  if (ws) LAUNCH((flash_prefill_kernel<256, true>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  else LAUNCH((flash_prefill_kernel<256, false>), grid, dim3(FP_WAVES * 64), lds, stream, a, w);
                  break;
    }
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

}  // extern "C"
