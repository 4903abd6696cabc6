// attn_q4.hip -- decode attention straight from the Q4 KV cache.
//
// Reference behaviour: ExLlamaV2Cache_Q4.get_kv_state (cache.py:472-514) unpacks the WHOLE live range of K and V to an
// fp16 temp for every layer of every decode step (q_to_fp16_kv, cache.cu:324-401), attention then reads that temp:
// O(ctx) bytes written and read again per layer per token.  This kernel reads the 4-bit codes and scales directly:
// 144 B per key per head (hd = 128) instead of 512 B, and nothing is unpacked to memory.
//
// The cache format (cache_q.cuh:4-185, restated in cache_q.hip): per 64 consecutive elements, even- and odd-indexed
// elements are each rotated by a 32-point Walsh-Hadamard transform H, then every 32 contiguous elements share one fp16
// scale s and store codes c in [0, 15]:   x = (1/32) H d,  d = (c - 8) s.   H is symmetric, so
//     q . x = (1/32) (H q) . d            and            sum_keys p x = (1/32) H (sum_keys p d):
// the QUERY is rotated once per workgroup, scores are dot products of the rotated query with raw codes, the weighted
// values accumulate in the rotated domain and are rotated back once at the end.  (The reference rounds the unpacked
// x to fp16 element by element; this path keeps fp32 sums -- same values within that rounding.)
#include "hw.h"
#include "errors.h"
#include "attn_merge.h"
#include "cache_q_pack.h"
#include <string.h>
#include <stdlib.h>

#define AQ_WAVES 4
#define AQ_UNROLL 4
#define AQ_MAX_PAGES 256          // block-table entries of one split kept in LDS
#define AQ_NEG_BIG (-1.0e30f)

struct AttnQ4Args
{
    const f16* q;                 // [b, s, H, hd]  (already rotated by RoPE)
    const u8* k_codes; const f16* k_scales;     // [pages | b, page_size | T, KVH, hd/2] , [.., KVH, hd/32]
    const u8* v_codes; const f16* v_scales;
    const f16* k_new; const f16* v_new;          // nullable: [b, s, KVH, hd] fp16 (k rotated): keys >= total - s come from here
    const int* cache_seqlens; const int* block_table;
    f16* out; float* part_o; float* part_ml;
    // FUSED form (template parameter; head_dim 128): q and k_new arrive UNROTATED -- the launch applies RoPE to them on the way in
    // (positions total - s + j) and packs the rotated k_new and v_new into the codes: the whole decode step in one launch
    const f16* sin; const f16* cos; int rope, neox;
    int pack_first;               // A/B switch (EXL2_Q4_PACK_FIRST=1): the step's rows are packed by split 0 in front of its attention (first form)
    u32* counters;                // nullable: [b, KVH, row blocks] zeroed tickets -- the last split of a row block to finish merges (no combine launch)
    const u16* out_invperm;       // nullable: feature n of a token row is stored at out[row, out_invperm[n]] (the consumer's packed order)
    int b, s, H, KVH;
    int page_size, page_shift, pages_per_seq;
    int len_const, len_offset, nsplit, causal;
    float scale;
};

// where feature d of query row qrow = (token row) * H + head goes (exl2_attn_decode_fused's convention)
DEV size_t q4_out_index(const AttnQ4Args& a, size_t qrow, int hd, int d)
{
    if (!a.out_invperm) return qrow * hd + d;
    const size_t tok = qrow / a.H;
    const int head = (int)(qrow - tok * a.H);
    return tok * ((size_t)a.H * hd) + a.out_invperm[head * hd + d];
}

DEV int q4_eff_splits(int total, int nsplit)
{
    const int e = (total + 255) >> 8;                       // ~256 keys per split at least
    return e < 1 ? 1 : (e > nsplit ? nsplit : e);
}

// RoPE on 8 consecutive elements [c, c + 8) of a head row (`own`), `partner` = the elements HALF = head_dim / 2 columns away (NeoX; unused
// for GPT-J): rope_append_kernel's fp16 operations (attn.hip), element by element
template <int HALF = 64>
DEV f16x8 rope8_128(f16x8 own, f16x8 partner, int c, const f16* sr, const f16* cr, bool neox)
{
    f16x8 y;
    if (neox)
    {
        const int c0 = c & (HALF - 1);
        const f16x8 cs = *(const f16x8*)(cr + c0), sn = *(const f16x8*)(sr + c0);
        #pragma unroll
        for (int e = 0; e < 8; e++)
            y[e] = c < HALF ? h_fma(own[e], cs[e], partner[e] * (-sn[e])) : h_fma(own[e], cs[e], partner[e] * sn[e]);
    }
    else
    {
        const f16x8 cs = *(const f16x8*)(cr + c), sn = *(const f16x8*)(sr + c);
        #pragma unroll
        for (int e = 0; e < 8; e += 2)
        {
            y[e] = h_fma(own[e + 1], -sn[e], own[e] * cs[e]);
            y[e + 1] = h_fma(own[e], sn[e + 1], own[e + 1] * cs[e + 1]);
        }
    }
    return y;
}

template <int HDIM, int RB, bool FUSED = false>
KERNEL void __launch_bounds__(AQ_WAVES * 64) attn_q4_decode_kernel(const AttnQ4Args a)
{
    static_assert(!FUSED || HDIM == 128 || HDIM == 64, "the one-launch form is built for head_dim 128 and 64");
    DYN_SMEM(smem);
    constexpr int LPK = HDIM / 16;              // lanes per key: 16 elements (8 code bytes) each
    constexpr int KPW = 64 / LPK;
    constexpr int NSTREAM = AQ_WAVES * KPW;
    constexpr int ROWF = HDIM + 2;

    const int kh = bid_x();
    const int split = bid_y();
    const int G = a.H / a.KVH;
    const int R = a.s * G;
    const int rblocks = (R + RB - 1) / RB;
    const int b = bid_z() / rblocks;
    const int rblk = bid_z() % rblocks;
    const int r0 = rblk * RB;
    const int nrows = min(RB, R - r0);

    const int lane = lane_id();
    const int wv = wave_id();
    const int group = lane / LPK;
    const int u = lane % LPK;                   // elements [16 u, 16 u + 16) of the head

    const int total = (a.cache_seqlens ? a.cache_seqlens[b] : a.len_const) + a.len_offset;
    // the grid is fixed (HIP graph) but the sequence length is not: use as many splits as the length deserves
    const int eff = q4_eff_splits(total, a.nsplit);
    if constexpr (FUSED)
    {
        // ---- the step's own rows into the cache: one workgroup per (sequence, kv head), one wave per row (k rotated first).  Nobody
        // in this launch reads what is written here: the codes serve keys < total - s, the step's own keys are attended in fp16.
        // The grid is sized for long sequences (HIP graph), so unless the sequence uses every split there is a workgroup with
        // nothing to attend over -- the first idle split packs, beside the attention instead of in front of it; else split 0.
        // (head_dim 64, round 6: a wave packs 128 elements = the rows of TWO adjacent kv heads of a token -- the workgroup of the even head
        // packs for both; the host takes this form for an even number of kv heads only)
        const int pack_split = (eff < a.nsplit && !a.pack_first) ? eff : 0;
        if (split == pack_split && rblk == 0 && (HDIM == 128 || (kh & 1) == 0))
        {
            for (int task = wv; task < 2 * a.s; task += AQ_WAVES)
            {
                const int j = task >> 1;
                const bool is_v = task & 1;
                const int pos = total - a.s + j;
                size_t tok;
                if (a.block_table) tok = (size_t)a.block_table[(size_t)b * a.pages_per_seq + (pos >> a.page_shift)] * a.page_size + (pos & (a.page_size - 1));
                else tok = (size_t)b * a.page_size + pos;
                const size_t cache_off = (tok * a.KVH + kh) * HDIM;
                const size_t src = (((size_t)b * a.s + j) * a.KVH + kh) * HDIM;
                if (is_v) q_pack_lane<4>(lane, ((const f16x2*)(a.v_new + src))[lane], (u8*)a.v_codes, (f16*)a.v_scales, cache_off);
                else
                {
                    f16x2 w = ((const f16x2*)(a.k_new + src))[lane];
                    const int srow = pos > 0 ? pos : 0;
                    if (a.rope) w = rope_lane_pair<HDIM>(w, lane, a.sin + (size_t)srow * HDIM, a.cos + (size_t)srow * HDIM, a.neox != 0);
                    q_pack_lane<4>(lane, w, (u8*)a.k_codes, (f16*)a.k_scales, cache_off);
                }
            }
        }
    }
    if (split >= eff) return;
    int kps = (total + eff - 1) / eff;
    kps = (kps + 15) & ~15;
    const int k_start = split * kps;
    const int k_end = min(total, k_start + kps);

    // ---- block-table entries of this split and the raw query rows into LDS ------------------------------------------------
    float* qh_lds = (float*)smem;                                       // [RB][HDIM] rotated rows
    f16* qraw_lds = (f16*)(qh_lds + RB * HDIM);                         // [RB][HDIM] raw rows
    int* pg_lds = (int*)(qraw_lds + RB * HDIM);                         // [AQ_MAX_PAGES] page of key (k_start >> shift) + i
    const int pg0 = a.block_table ? (k_start >> a.page_shift) : 0;
    if (a.block_table)
    {
        const int npg = ((k_end > k_start ? k_end - 1 : k_start) >> a.page_shift) - pg0 + 1;
        for (int i = tid(); i < npg && i < AQ_MAX_PAGES; i += nthreads())
            pg_lds[i] = a.block_table[(size_t)b * a.pages_per_seq + pg0 + i];
    }
    for (int idx = tid(); idx < nrows * (HDIM / 8); idx += nthreads())
    {
        const int r = idx / (HDIM / 8), o8 = idx - r * (HDIM / 8);
        const int rr = r0 + r, j = rr / G, g = rr - j * G;
        const f16* qrow_p = a.q + (((size_t)b * a.s + j) * a.H + kh * G + g) * HDIM;
        f16x8 qv = *(const f16x8*)(qrow_p + o8 * 8);
        if constexpr (FUSED)
        {
            if (a.rope)
            {
                const int pos = total - a.s + j, srow = pos > 0 ? pos : 0;
                const f16x8 qpart = *(const f16x8*)(qrow_p + ((o8 * 8) ^ (HDIM / 2)));
                qv = rope8_128<HDIM / 2>(qv, qpart, o8 * 8, a.sin + (size_t)srow * HDIM, a.cos + (size_t)srow * HDIM, a.neox != 0);
            }
        }
        ((f16x8*)qraw_lds)[idx] = qv;
    }
    block_sync();
    // ---- rotate the query rows: qh[64 g + 2 t + comp] = sum_t' H[t, t'] q[64 g + 2 t' + comp], H[t,t'] = (-1)^popc(t & t')
    for (int idx = tid(); idx < nrows * HDIM; idx += nthreads())
    {
        const int r = idx / HDIM, e = idx - r * HDIM;
        const f16* qr = qraw_lds + r * HDIM;
        const int span = e >> 6, t = (e & 63) >> 1, comp = e & 1;
        float acc = 0.0f;
        #pragma unroll 8
        for (int t2 = 0; t2 < 32; t2++)
        {
            const float v = (float)qr[span * 64 + 2 * t2 + comp];
            acc += (__builtin_popcount(t & t2) & 1) ? -v : v;
        }
        qh_lds[idx] = acc;
    }
    block_sync();
    f16x2 qf[RB][8], qp[RB][8];                                         // rotated / plain query slices of this lane
    int limit[RB];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        const int rs = r < nrows ? r : 0;
        #pragma unroll
        for (int i = 0; i < 8; i++)
            qf[r][i] = (f16x2){(f16)qh_lds[rs * HDIM + 16 * u + 2 * i], (f16)qh_lds[rs * HDIM + 16 * u + 2 * i + 1]};
        const int rr = r0 + rs, j = rr / G;
        const f16* qr = qraw_lds + rs * HDIM + 16 * u;
        #pragma unroll
        for (int i = 0; i < 8; i++) qp[r][i] = (f16x2){qr[2 * i], qr[2 * i + 1]};
        limit[r] = a.causal ? (total - a.s + j + 1) : total;
    }
    block_sync();

    float m[RB], l[RB], o[RB][16];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        m[r] = AQ_NEG_BIG; l[r] = 0.0f;
        #pragma unroll
        for (int e = 0; e < 16; e++) o[r][e] = 0.0f;
    }

    const size_t tok_codes = (size_t)a.KVH * (HDIM / 2);                // bytes per token slot
    const size_t tok_scales = (size_t)a.KVH * (HDIM / 32);
    const int past = a.k_new ? total - a.s : total;                      // keys [past, total) come from k_new / v_new
    const int k_old_end = min(k_end, past);
    // Codes stay biased: a pair is the half2 (1024 + c_even, 1024 + c_odd) straight from v_perm + or; the "- 8" is applied
    // once per batch through sum(qh) on the K side and through sum(p s) on the V side (fp32, exact enough: the biased
    // sums are ~130x the centred ones, 2^-24 relative each).  The running maximum is updated once per batch of AQ_UNROLL
    // steps, so the 16 accumulators are rescaled once per batch, not per key.
    constexpr int STEP = AQ_WAVES * KPW;
    float qsum[RB], osub[RB];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        float t = 0.0f;
        #pragma unroll
        for (int i = 0; i < 8; i++) t += (float)qf[r][i].x + (float)qf[r][i].y;
        qsum[r] = t * 1032.0f;
        osub[r] = 0.0f;
    }
    for (int base0 = k_start + wv * KPW; base0 < k_old_end; base0 += AQ_UNROLL * STEP)
    {
        u32x2 kcs[AQ_UNROLL], vcs[AQ_UNROLL];
        f16 kss[AQ_UNROLL], vss[AQ_UNROLL];
        #pragma unroll
        for (int un = 0; un < AQ_UNROLL; un++)
        {
            const int kpos = base0 + un * STEP + group;
            const int kp = kpos < k_old_end ? kpos : k_start;
            size_t tok;
            if (a.block_table)
            {
                const int pi = (kp >> a.page_shift) - pg0;
                const int pg = pi < AQ_MAX_PAGES ? pg_lds[pi] : a.block_table[(size_t)b * a.pages_per_seq + (kp >> a.page_shift)];
                tok = (size_t)pg * a.page_size + (kp & (a.page_size - 1));
            }
            else
                tok = (size_t)b * a.page_size + kp;
            const size_t co = tok * tok_codes + (size_t)kh * (HDIM / 2) + u * 8;
            const size_t so = tok * tok_scales + (size_t)kh * (HDIM / 32) + (u >> 1);
            kcs[un] = ld_nt((const u32x2*)(a.k_codes + co));
            vcs[un] = ld_nt((const u32x2*)(a.v_codes + co));
            kss[un] = a.k_scales[so];
            vss[un] = a.v_scales[so];
        }
        // scores of the batch
        float sc[RB][AQ_UNROLL];
        #pragma unroll
        for (int un = 0; un < AQ_UNROLL; un++)
        {
            const u32 l0 = kcs[un].x & 0x0F0F0F0Fu, h0 = (kcs[un].x >> 4) & 0x0F0F0F0Fu;
            const u32 l1 = kcs[un].y & 0x0F0F0F0Fu, h1 = (kcs[un].y >> 4) & 0x0F0F0F0Fu;
            f16x2 kd[8];
            kd[0] = as_h2(byte_perm(h0, l0, 0x0C040C00u) | 0x64006400u); kd[1] = as_h2(byte_perm(h0, l0, 0x0C050C01u) | 0x64006400u);
            kd[2] = as_h2(byte_perm(h0, l0, 0x0C060C02u) | 0x64006400u); kd[3] = as_h2(byte_perm(h0, l0, 0x0C070C03u) | 0x64006400u);
            kd[4] = as_h2(byte_perm(h1, l1, 0x0C040C00u) | 0x64006400u); kd[5] = as_h2(byte_perm(h1, l1, 0x0C050C01u) | 0x64006400u);
            kd[6] = as_h2(byte_perm(h1, l1, 0x0C060C02u) | 0x64006400u); kd[7] = as_h2(byte_perm(h1, l1, 0x0C070C03u) | 0x64006400u);
            const float ks = (float)kss[un] * (1.0f / 32.0f);
            const int kpos = base0 + un * STEP + group;
            #pragma unroll
            for (int r = 0; r < RB; r++)
            {
                float d = 0.0f;
                #pragma unroll
                for (int i = 0; i < 8; i++) d = dot2_f32_f16(qf[r][i], kd[i], d);
                d = (d - qsum[r]) * ks;
                if constexpr (LPK == 4) d = quad_allreduce_add(d);
                else if constexpr (LPK == 8) d = row8_allreduce_add(d);
                else d = row16_allreduce_add(d);
                const bool valid = r < nrows && kpos < k_old_end && kpos < limit[r];
                sc[r][un] = valid ? d * a.scale : AQ_NEG_BIG;
            }
        }
        // one maximum / rescale per batch
        float pw[RB][AQ_UNROLL];
        #pragma unroll
        for (int r = 0; r < RB; r++)
        {
            float m_new = m[r];
            #pragma unroll
            for (int un = 0; un < AQ_UNROLL; un++) m_new = fmaxf(m_new, sc[r][un]);
            const float alpha = fast_exp(m[r] - m_new);
            float ps = 0.0f;
            #pragma unroll
            for (int un = 0; un < AQ_UNROLL; un++)
            {
                const float p = sc[r][un] > 0.5f * AQ_NEG_BIG ? fast_exp(sc[r][un] - m_new) : 0.0f;
                ps += p;
                pw[r][un] = p * (float)vss[un];
            }
            m[r] = m_new;
            l[r] = l[r] * alpha + ps;
            osub[r] *= alpha;
            #pragma unroll
            for (int e = 0; e < 16; e++) o[r][e] *= alpha;
        }
        // weighted biased codes
        #pragma unroll
        for (int un = 0; un < AQ_UNROLL; un++)
        {
            const u32 l0 = vcs[un].x & 0x0F0F0F0Fu, h0 = (vcs[un].x >> 4) & 0x0F0F0F0Fu;
            const u32 l1 = vcs[un].y & 0x0F0F0F0Fu, h1 = (vcs[un].y >> 4) & 0x0F0F0F0Fu;
            f16x2 vd[8];
            vd[0] = as_h2(byte_perm(h0, l0, 0x0C040C00u) | 0x64006400u); vd[1] = as_h2(byte_perm(h0, l0, 0x0C050C01u) | 0x64006400u);
            vd[2] = as_h2(byte_perm(h0, l0, 0x0C060C02u) | 0x64006400u); vd[3] = as_h2(byte_perm(h0, l0, 0x0C070C03u) | 0x64006400u);
            vd[4] = as_h2(byte_perm(h1, l1, 0x0C040C00u) | 0x64006400u); vd[5] = as_h2(byte_perm(h1, l1, 0x0C050C01u) | 0x64006400u);
            vd[6] = as_h2(byte_perm(h1, l1, 0x0C060C02u) | 0x64006400u); vd[7] = as_h2(byte_perm(h1, l1, 0x0C070C03u) | 0x64006400u);
            float vfl[16];
            #pragma unroll
            for (int i = 0; i < 8; i++) { vfl[2 * i] = (float)vd[i].x; vfl[2 * i + 1] = (float)vd[i].y; }
            #pragma unroll
            for (int r = 0; r < RB; r++)
            {
                const float pv = pw[r][un];
                osub[r] += pv;
                #pragma unroll
                for (int e = 0; e < 16; e++) o[r][e] = fmaf(pv, vfl[e], o[r][e]);
            }
        }
    }
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        #pragma unroll
        for (int e = 0; e < 16; e++) o[r][e] -= 1032.0f * osub[r];
    }

    // ---- the step's own keys / values, still fp16 (the reference attends over them before they are quantised) -------------
    for (int base = max(k_start, past) + wv * KPW; base < k_end; base += AQ_WAVES * KPW)
    {
        const int kpos = base + group;
        const bool in_range = kpos < k_end;
        const int kp = in_range ? kpos : k_end - 1;
        const size_t src = (((size_t)b * a.s + (kp - past)) * a.KVH + kh) * HDIM + 16 * u;
        f16x2 kn[8];
        float w[16];
        #pragma unroll
        for (int i = 0; i < 8; i++)
        {
            kn[i] = (f16x2){a.k_new[src + 2 * i], a.k_new[src + 2 * i + 1]};
            w[2 * i] = (float)a.v_new[src + 2 * i]; w[2 * i + 1] = (float)a.v_new[src + 2 * i + 1];
        }
        if constexpr (FUSED)
        {
            if (a.rope)
            {
                // the lane's 16 elements [16 u, 16 u + 16) of the UNROTATED row, their NeoX partners 64 columns away
                const int srow = kp > 0 ? kp : 0;
                const f16* sr = a.sin + (size_t)srow * HDIM;
                const f16* cr = a.cos + (size_t)srow * HDIM;
                const size_t psrc = src - 16 * u + ((16 * u) ^ (HDIM / 2));
                #pragma unroll
                for (int hblk = 0; hblk < 2; hblk++)
                {
                    f16x8 own, part;
                    #pragma unroll
                    for (int e = 0; e < 8; e++) own[e] = e & 1 ? kn[4 * hblk + e / 2].y : kn[4 * hblk + e / 2].x;
                    part = *(const f16x8*)(a.k_new + psrc + 8 * hblk);
                    const f16x8 y = rope8_128<HDIM / 2>(own, part, 16 * u + 8 * hblk, sr, cr, a.neox != 0);
                    #pragma unroll
                    for (int e = 0; e < 4; e++) kn[4 * hblk + e] = (f16x2){y[2 * e], y[2 * e + 1]};
                }
            }
        }
        // rotate v into the accumulation domain: H over the pair index t = 8 (u & 3) + i, per component
        #pragma unroll
        for (int mbit = 1; mbit < 8; mbit <<= 1)
        {
            #pragma unroll
            for (int i = 0; i < 8; i++)
            {
                if (!(i & mbit))
                {
                    #pragma unroll
                    for (int c = 0; c < 2; c++)
                    {
                        const float x0 = w[2 * i + c], x1 = w[2 * (i | mbit) + c];
                        w[2 * i + c] = x0 + x1; w[2 * (i | mbit) + c] = x0 - x1;
                    }
                }
            }
        }
        #pragma unroll
        for (int lbit = 1; lbit < 4; lbit <<= 1)
        {
            #pragma unroll
            for (int e = 0; e < 16; e++)
            {
                const float p = lbit == 1 ? shfl_xor_f32(w[e], 1) : shfl_xor_f32(w[e], 2);
                w[e] = (u & lbit) ? p - w[e] : w[e] + p;
            }
        }
        #pragma unroll
        for (int r = 0; r < RB; r++)
        {
            if (r < nrows)
            {
                float d = 0.0f;
                #pragma unroll
                for (int i = 0; i < 8; i++) d = dot2_f32_f16(qp[r][i], kn[i], d);
                if constexpr (LPK == 4) d = quad_allreduce_add(d);
                else if constexpr (LPK == 8) d = row8_allreduce_add(d);
                else d = row16_allreduce_add(d);
                const float sc = d * a.scale;
                const bool valid = in_range && kpos < limit[r];
                const float m_new = valid ? fmaxf(m[r], sc) : m[r];
                const float alpha = fast_exp(m[r] - m_new);
                const float p = (valid ? fast_exp(sc - m_new) : 0.0f);
                m[r] = m_new;
                l[r] = l[r] * alpha + p;
                #pragma unroll
                for (int e = 0; e < 16; e++) o[r][e] = o[r][e] * alpha + p * w[e];
            }
        }
    }

    // ---- merge the streams of this workgroup (rotated domain), rotate back, store / emit partials --------------------------
    block_sync();                                                       // every wave is done with the page ids in LDS
    float* st = (float*)smem;                                           // [NSTREAM][RB][ROWF]
    const int stream = wv * KPW + group;
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        if (r < nrows)
        {
            float* p = st + ((size_t)stream * RB + r) * ROWF;
            #pragma unroll
            for (int e = 0; e < 16; e++) p[u * 16 + e] = o[r][e];
            if (u == 0) { p[HDIM] = m[r]; p[HDIM + 1] = l[r]; }
        }
    }
    block_sync();
    float* mg = st + (size_t)NSTREAM * RB * ROWF;                       // merged rotated rows [RB][HDIM + 2]
    for (int idx = tid(); idx < nrows * HDIM; idx += nthreads())
    {
        const int r = idx / HDIM, d = idx - r * HDIM;
        float M = AQ_NEG_BIG;
        for (int s2 = 0; s2 < NSTREAM; s2++) M = fmaxf(M, st[((size_t)s2 * RB + r) * ROWF + HDIM]);
        float L = 0.0f, O = 0.0f;
        for (int s2 = 0; s2 < NSTREAM; s2++)
        {
            const float* p = st + ((size_t)s2 * RB + r) * ROWF;
            const float w = fast_exp(p[HDIM] - M);
            L += p[HDIM + 1] * w;
            O += p[d] * w;
        }
        mg[r * ROWF + d] = O;
        if (d == 0) { mg[r * ROWF + HDIM] = M; mg[r * ROWF + HDIM + 1] = L; }
    }
    block_sync();
    for (int idx = tid(); idx < nrows * HDIM; idx += nthreads())
    {
        const int r = idx / HDIM, e = idx - r * HDIM;
        const int span = e >> 6, t = (e & 63) >> 1, comp = e & 1;
        float acc = 0.0f;
        for (int t2 = 0; t2 < 32; t2++)
        {
            const float v = mg[r * ROWF + span * 64 + 2 * t2 + comp];
            acc += (__builtin_popcount(t & t2) & 1) ? -v : v;
        }
        acc *= (1.0f / 32.0f);
        const float M = mg[r * ROWF + HDIM], L = mg[r * ROWF + HDIM + 1];
        const int rr = r0 + r, j = rr / G, g = rr - j * G;
        const size_t qrow = ((size_t)b * a.s + j) * a.H + kh * G + g;
        if (eff == 1)
        {
            a.out[q4_out_index(a, qrow, HDIM, e)] = (f16)(L > 0.0f ? acc / L : 0.0f);
        }
        else if (a.counters)
        {
            // (agent-scope stores, read back with agent-scope loads by the merging workgroup: attn.hip's hand-off)
            store_agent_f32(a.part_o + (qrow * a.nsplit + split) * HDIM + e, acc);
            if (e == 0)
            {
                store_agent_f32(a.part_ml + (qrow * a.nsplit + split) * 2 + 0, M);
                store_agent_f32(a.part_ml + (qrow * a.nsplit + split) * 2 + 1, L);
            }
        }
        else
        {
            a.part_o[(qrow * a.nsplit + split) * HDIM + e] = acc;
            if (e == 0)
            {
                a.part_ml[(qrow * a.nsplit + split) * 2 + 0] = M;
                a.part_ml[(qrow * a.nsplit + split) * 2 + 1] = L;
            }
        }
    }
    if (eff == 1 || !a.counters) return;
    // hand-off: the last split of this (sequence, kv head, row block) to arrive merges all of them
    u32* ticket_lds = (u32*)(mg + RB * ROWF);
    wait_vmcnt0();
    block_sync();
    u32* counter = a.counters + ((size_t)b * a.KVH + kh) * rblocks + rblk;
    if (tid() == 0) *ticket_lds = ticket_add_agent(counter, 1u);
    block_sync();
    if (*ticket_lds != (u32)(eff - 1)) return;
    for (int idx = tid(); idx < nrows * HDIM; idx += nthreads())
    {
        const int r = idx / HDIM, d = idx - r * HDIM;
        const int rr = r0 + r, j = rr / G, g = rr - j * G;
        const size_t qrow = ((size_t)b * a.s + j) * a.H + kh * G + g;
        a.out[q4_out_index(a, qrow, HDIM, d)] = (f16)merge_split_partials<true>(a.part_o, a.part_ml, qrow, a.nsplit, eff, HDIM, d);
    }
    if (tid() == 0) store_relaxed_agent(counter, 0u);
}

KERNEL void __launch_bounds__(256) attn_q4_combine_kernel(const AttnQ4Args a, int hd)
{
    const size_t qrow = bid_x();
    const int b = (int)(qrow / ((size_t)a.s * a.H));
    const int total = (a.cache_seqlens ? a.cache_seqlens[b] : a.len_const) + a.len_offset;
    const int eff = q4_eff_splits(total, a.nsplit);
    if (eff == 1) return;                                   // the single split stored the result itself
    for (int d = tid(); d < hd; d += nthreads())
    {
        a.out[q4_out_index(a, qrow, hd, d)] = (f16)merge_split_partials<false>(a.part_o, a.part_ml, qrow, a.nsplit, eff, hd, d);
    }
}

static int ilog2_exact_q4(int x) { int s = 0; while ((1 << s) < x) s++; return (1 << s) == x ? s : -1; }

template <int HDIM>
static void launch_q4(const AttnQ4Args& a, int rb, dim3 grid, void* stream, bool fused = false)
{
    static bool attr_done_fused[EXL2_MAX_DEVICES] = {false};
    const int lpk = HDIM / 16, kpw = 64 / lpk;
    size_t lds = ((size_t)AQ_WAVES * kpw * rb + rb) * (HDIM + 2) * 4 + 16;      // (+ the ticket)
    const size_t pro = (size_t)rb * HDIM * 4 + (size_t)rb * HDIM * 2 + AQ_MAX_PAGES * 4;     // prologue: rotated + raw rows + pages
    if (lds < pro) lds = pro;
    static bool attr_done[EXL2_MAX_DEVICES] = {false};
    if (exl2_first_on_device(attr_done))
    {
        (void)hipFuncSetAttribute((const void*)attn_q4_decode_kernel<HDIM, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_q4_decode_kernel<HDIM, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if constexpr (HDIM == 128 || HDIM == 64)
    {
        if (fused)
        {
            if (exl2_first_on_device(attr_done_fused))
            {
                (void)hipFuncSetAttribute((const void*)attn_q4_decode_kernel<HDIM, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)attn_q4_decode_kernel<HDIM, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            switch (rb)
            {
                case 1: LAUNCH((attn_q4_decode_kernel<HDIM, 1, true>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
                case 2: LAUNCH((attn_q4_decode_kernel<HDIM, 2, true>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
                default: LAUNCH((attn_q4_decode_kernel<HDIM, 4, true>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
            }
            return;
        }
    }
    switch (rb)
    {
        case 1: LAUNCH((attn_q4_decode_kernel<HDIM, 1>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
        case 2: LAUNCH((attn_q4_decode_kernel<HDIM, 2>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
        default: LAUNCH((attn_q4_decode_kernel<HDIM, 4>), grid, dim3(AQ_WAVES * 64), lds, stream, a); break;
    }
}

// Attention over a Q4 KV cache.  k_new / v_new == NULL: the cache holds all `total` keys.  Otherwise the last q_len keys
// (the step's own, k_new already rotated) are taken from these fp16 tensors -- the reference attends over the step's K/V
// before quantising them (cache.py:517-556 runs after attention) -- and only keys < total - q_len come from the codes.
// Same addressing / length conventions as exl2_paged_attn; codes [.., KVH, hd/2] uint8, scales [.., KVH, hd/32] fp16.
// Returns 1 without launching for shapes it does not cover (caller unpacks with exl2_q_to_fp16_kv + exl2_paged_attn).
static int paged_attn_q4_impl(const void* q, const void* k_codes, const void* k_scales, const void* v_codes, const void* v_scales,
                       const void* k_new, const void* v_new, void* out, const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       const void* out_invperm, void* counters, int n_counters, void* stream,
                       const void* sin = nullptr, const void* cos = nullptr, int rope_style = -1, int sincos_size = 0)
{
    const bool fused = rope_style >= 0;                     // the one-launch form: RoPE + pack of the step's rows inside this launch
    EXL2_REQUIRE(q && k_codes && k_scales && v_codes && v_scales && out, "paged_attn_q4: null argument");
    EXL2_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged_attn_q4: heads %d not a multiple of kv heads %d", num_heads, num_kv_heads);
    if (batch <= 0 || q_len <= 0) return EXL2_OK;
    if (!(head_dim == 64 || head_dim == 128 || head_dim == 256)) return 1;
    if (fused && (!(head_dim == 128 || (head_dim == 64 && num_kv_heads % 2 == 0)) || !k_new || !v_new || (rope_style != 0 && (!sin || !cos || (sincos_size > 0 && sincos_size != head_dim))))) return 1;
    const int G = num_heads / num_kv_heads;
    const int R = q_len * G;
    if (R > 64) return 1;
    AttnQ4Args a;
    memset(&a, 0, sizeof(a));
    a.q = (const f16*)q; a.k_codes = (const u8*)k_codes; a.k_scales = (const f16*)k_scales;
    a.v_codes = (const u8*)v_codes; a.v_scales = (const f16*)v_scales; a.out = (f16*)out;
    EXL2_REQUIRE((k_new == nullptr) == (v_new == nullptr), "paged_attn_q4: k_new and v_new go together");
    a.k_new = (const f16*)k_new; a.v_new = (const f16*)v_new; a.out_invperm = (const u16*)out_invperm;
    a.cache_seqlens = cache_seqlens; a.block_table = block_table;
    a.b = batch; a.s = q_len; a.H = num_heads; a.KVH = num_kv_heads;
    a.page_size = page_size; a.pages_per_seq = pages_per_seq; a.page_shift = ilog2_exact_q4(page_size);
    EXL2_REQUIRE(!block_table || a.page_shift >= 0, "paged_attn_q4: page_size %d must be a power of two", page_size);
    a.len_const = len_const; a.len_offset = len_offset; a.causal = causal; a.scale = softmax_scale;
    if (fused) { a.sin = (const f16*)sin; a.cos = (const f16*)cos; a.rope = rope_style != 0; a.neox = rope_style == 2; }      // ROPE_STYLE_* q_attn.cuh:13-15
    if (fused) { const char* e = getenv("EXL2_Q4_PACK_FIRST"); a.pack_first = (e && atoi(e)) ? 1 : 0; }
    // Query rows per workgroup.  Four rows share one pass over the keys -- but a workgroup's own dependent chain (rotate the rows,
    // fill registers, merge 32 streams per row, rotate back) grows with its rows, and up to a few thousand keys that chain is the
    // launch: the 70B shape (8 rows per kv head) measures 91.1 / 92.8 / 94.2 tok/s at 4 / 2 / 1 rows per workgroup with an empty
    // cache and 87.5 / 88.6 / 89.3 with 1920 tokens cached (profiles/history/r05y_ab_q4_rb*.txt).  The length is device-side (HIP graph),
    // the CAPACITY of the sequence is not: one row per workgroup while the sequence cannot exceed 4096 keys, four beyond
    // (unmeasured there: eight passes over the keys against two).  EXL2_Q4_RB=1|2|4 overrides.
    const long long capacity = block_table ? (long long)pages_per_seq * page_size : (long long)page_size;
    int rb = R >= 4 ? 4 : (R >= 2 ? 2 : 1);
    if (capacity <= 4096) rb = 1;
    if (const char* e = getenv("EXL2_Q4_RB")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) rb = v < R ? v : (R >= 4 ? 4 : (R >= 2 ? 2 : 1)); }
    const int rblocks = (R + rb - 1) / rb;
    if (nsplit <= 0)
    {
        const long long base = (long long)num_kv_heads * batch * rblocks;
        // 144 B per key: the stream is latency-bound, not bandwidth-bound -- many short splits (4 waves each)
        nsplit = (int)((1024 + base - 1) / base);
        if (nsplit > 32) nsplit = 32;
        if (nsplit < 1) nsplit = 1;
    }
    long long need = nsplit <= 1 ? 0 : (long long)batch * q_len * num_heads * nsplit * (head_dim + 2) * 4;
    while (nsplit > 1 && (need > scratch_bytes || !scratch)) { nsplit /= 2; need = nsplit <= 1 ? 0 : (long long)batch * q_len * num_heads * nsplit * (head_dim + 2) * 4; }
    if (need > scratch_bytes || (need > 0 && !scratch)) nsplit = 1;
    a.nsplit = nsplit;
    if (nsplit > 1)
    {
        a.part_o = (float*)scratch;
        a.part_ml = a.part_o + (size_t)batch * q_len * num_heads * nsplit * head_dim;
    }
    // tickets given: the last split of a row block merges inside the launch (the combine launch cost ~5 us per layer whether or not
    // the length needed more than one split)
    if (nsplit > 1 && counters && (long long)n_counters >= (long long)batch * num_kv_heads * rblocks) a.counters = (u32*)counters;
    dim3 grid((unsigned)num_kv_heads, (unsigned)nsplit, (unsigned)(batch * rblocks));
    if (head_dim == 64) launch_q4<64>(a, rb, grid, stream, fused);
    else if (head_dim == 128) launch_q4<128>(a, rb, grid, stream, fused);
    else launch_q4<256>(a, rb, grid, stream);
    if (nsplit > 1 && !a.counters)
        LAUNCH(attn_q4_combine_kernel, dim3((unsigned)(batch * q_len * num_heads)), dim3(head_dim < 256 ? head_dim : 256), 0,
               stream, a, head_dim);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

extern "C" {

int exl2_paged_attn_q4(const void* q, const void* k_codes, const void* k_scales, const void* v_codes, const void* v_scales,
                       const void* k_new, const void* v_new, void* out, const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       const void* out_invperm, void* stream)
{
    return paged_attn_q4_impl(q, k_codes, k_scales, v_codes, v_scales, k_new, v_new, out, cache_seqlens, block_table, batch, q_len,
                              num_heads, num_kv_heads, head_dim, page_size, pages_per_seq, len_const, len_offset, softmax_scale, causal,
                              nsplit, scratch, scratch_bytes, out_invperm, nullptr, 0, stream);
}

// The same with `counters`: n_counters >= batch * q_len * num_heads (one per query row always suffices; fewer than the launch needs: the combine launch runs instead) u32 tickets, ZERO before the first
// call (every launch leaves them zero): the split partials are merged by the last split to finish, inside the launch -- one launch
// instead of two.  (exl2_attn_decode_fused's counters serve: same indexing, never used by both at once on a stream.)
int exl2_paged_attn_q4_merged(const void* q, const void* k_codes, const void* k_scales, const void* v_codes, const void* v_scales,
                              const void* k_new, const void* v_new, void* out, const int* cache_seqlens, const int* block_table,
                              int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                              int page_size, int pages_per_seq, int len_const, int len_offset,
                              float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                              const void* out_invperm, void* counters, int n_counters, void* stream)
{
    return paged_attn_q4_impl(q, k_codes, k_scales, v_codes, v_scales, k_new, v_new, out, cache_seqlens, block_table, batch, q_len,
                              num_heads, num_kv_heads, head_dim, page_size, pages_per_seq, len_const, len_offset, softmax_scale, causal,
                              nsplit, scratch, scratch_bytes, out_invperm, counters, n_counters, stream);
}

// The whole decode step over a Q4 cache in ONE launch: RoPE on q and k_new (NOT modified in memory: rotated on the way into the
// kernel), Q4 pack of the rotated k_new and of v_new into the codes / scales at positions past + j (past = cache_seqlens[b], or
// past_const without cache_seqlens), attention over the codes for keys below `past` and over the step's own rows in fp16, split
// merge by ticket == exl2_rope_kv_append + exl2_fp16_to_q_kv + exl2_paged_attn_q4 (+ its combine launch).  Returns 1 without
// launching for shapes it does not cover (head_dim != 128, partial rotary, > 64 query rows per kv head): the caller takes
// exl2_rope_quant_append_q4 + exl2_paged_attn_q4_merged.
int exl2_attn_q4_decode_fused(const void* q, const void* k_new, const void* v_new, void* k_codes, void* k_scales, void* v_codes,
                              void* v_scales, void* out, const void* sin, const void* cos, const int* cache_seqlens,
                              const int* block_table, int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                              int page_size, int pages_per_seq, int past_const, float softmax_scale, int rope_style, int sincos_size,
                              int nsplit, void* scratch, long long scratch_bytes, void* counters, int n_counters,
                              const void* out_invperm, void* stream)
{
    EXL2_REQUIRE(rope_style >= 0 && rope_style <= 2, "attn_q4_decode_fused: rope_style %d", rope_style);
    return paged_attn_q4_impl(q, k_codes, k_scales, v_codes, v_scales, k_new, v_new, out, cache_seqlens, block_table, batch, q_len,
                              num_heads, num_kv_heads, head_dim, page_size, pages_per_seq, past_const, q_len, softmax_scale, 1,
                              nsplit, scratch, scratch_bytes, out_invperm, counters, n_counters, stream, sin, cos, rope_style, sincos_size);
}

}  // extern "C"
