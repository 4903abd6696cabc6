// attn_fused_body.h -- the one-launch decode-step attention (RoPE + append + split-KV attention + merge) as a device function.
// Included by attn.hip (the kernel around it + host code).  A device function since round 6, when it was also run as the TAIL of the
// chained q | k | v launch (the last tile of a kv head to finish attends for that head): measured 14-21 % SLOWER than the two launches
// and taken out again (profiles/r06_attention_fold_experiment.txt, r06_attention_fold.patch).
// Reference: flash_attn_with_kvcache(k = new_k, v = new_v, ...) as attn.py:602-613 calls it.
#pragma once
#include "hw.h"
#include "chain_sync.h"
#include "attn_merge.h"

#define ATT_WAVES 4
#define ATT_UNROLL 4
#define ATT_MAX_PAGES 2048               // page ids of one split's key range kept in LDS (attn_fused_kernel): 8 KB, 512 K keys at 256 per page
#ifndef ATT_KPS_DEFAULT
#define ATT_KPS_DEFAULT 128
#endif
#define NEG_BIG (-1.0e30f)


template <int LPK> DEV float group_allreduce_add(float v)
{
    if constexpr (LPK == 8) return row8_allreduce_add(v);
    else if constexpr (LPK == 16) return row16_allreduce_add(v);
    else { v = row16_allreduce_add(v); return v + as_f32(swz_xor_u32<16>(f32_bits(v))); }
}


// ---- one-launch decode step: RoPE(q, new k) + append(new k, v) + split-KV attention + combine -----------------------------
//
// The three launches above (rope_append -> attn_decode -> attn_combine) cost ~4-5 us each on MI355X whatever their size
// (launch-bound), i.e. more than the K/V stream itself at short contexts.  This kernel does the whole
// flash_attn_with_kvcache(k=new_k, v=new_v, ...) contract (attn.py:602-613) in one launch:
//   * q rows are rotated in registers (rope.cu NeoX numerics; the rotation partner d +- HDIM/2 lives in lane +- LPK/2);
//   * keys at positions >= past come from k_new / v_new, are rotated, used, and written to the cache by the workgroup
//     whose key slice owns them (row block 0 only, so every cache slot has exactly one writer);
//   * the number of active splits is chosen ON THE DEVICE from the sequence length (the grid is fixed inside a HIP graph);
//   * splits meet through a ticket counter per (batch, kv head, row block): the last arriver merges the partials
//     (agent-scope release/acquire, cdna_hip_programming.md G16) and resets the counter for the next launch.
struct FusedArgs
{
    const f16* q; const f16* k_new; const f16* v_new;   // [b, s, H|KVH, hd], un-rotated
    f16* k_cache; f16* v_cache;
    const f16* sin; const f16* cos;                     // [max_seq, HDIM]
    const int* cache_seqlens; const int* block_table;
    f16* out; float* part_o; float* part_ml; u32* counters;
    const u16* out_invperm;                             // nullable: feature n of a token row is stored at out[row, out_invperm[n]]
    f16* out_nat;                                       // nullable: a second copy of the output in natural order (exl2_attn_decode_fused_dual)
    int b, s, H, KVH;                                   // (the consumer's packed order, i.e. o_proj's act-order: qgemv_flat.hip)
    int page_size, page_shift, pages_per_seq;
    int past_const, nsplit, rope, keys_per_split_min;
    int split_short_cap, keys_per_split_long;           // two-slope split count (fused_active_splits); keys_per_split_long = 0: one slope
    float scale;
    // overlapped chain (chain_sync.h / hw.h): q / k_new / v_new are read behind the wait, with agent-scope loads; the
    // workgroup that writes a group's final output arrives once (sync_total = batch * kv_heads * row_blocks per launch)
    const u32* sync_wait; u32* sync_signal; u32 sync_total;
    u32* sync_arrive;                                   // every workgroup of the grid reports here on entry (the next launch's gate)
};

// where feature d of query row qrow = (token row) * H + head goes
// Number of splits a launch of `total` keys uses (the grid holds a.nsplit of them; the rest leave at entry).  One slope: a split per
// keys_per_split_min keys.  Two slopes (few KV heads -- grouped-query models; round 5, profiles/history/r05m_attn_sweep2_merge.jsonl): a split
// per keys_per_split_min keys up to split_short_cap, beyond that a split per keys_per_split_long keys -- short contexts want many short
// slices early (the launch is latency-bound), long ones few enough that the slices stay long (each split pays its own start-up and the
// merge reads every partial).
DEV int fused_active_splits(const FusedArgs& a, int total)
{
    int eff = (total + a.keys_per_split_min - 1) / a.keys_per_split_min;
    if (a.keys_per_split_long > 0)
    {
        if (eff > a.split_short_cap) eff = a.split_short_cap;
        const int e2 = (total + a.keys_per_split_long - 1) / a.keys_per_split_long;
        if (e2 > eff) eff = e2;
    }
    return eff < 1 ? 1 : (eff > a.nsplit ? a.nsplit : eff);
}

template <int HDIM> DEV size_t fused_out_index(const FusedArgs& a, size_t qrow, int d)
{
    if (!a.out_invperm) return qrow * HDIM + d;
    const size_t tok = qrow / a.H;
    const int head = (int)(qrow - tok * a.H);
    return tok * ((size_t)a.H * HDIM) + a.out_invperm[head * HDIM + d];
}

// rotary table rows of position pos for lane dl (columns [8 dl, 8 dl + 8) and their partner columns): requested ...
template <int LPK> DEV void rope_neox_load(const f16* sin, const f16* cos, int pos, int dl, f16x8& cs, f16x8& sn)
{
    constexpr int HL = LPK / 2;
    const size_t off = (size_t)pos * (LPK * 8) + (size_t)(dl % HL) * 8;
    cs = *(const f16x8*)(cos + off);
    sn = *(const f16x8*)(sin + off);
}
// ... applied: lane dl holds columns [8 dl, 8 dl + 8); its partner columns (+- HDIM/2) sit in lane dl ^ (LPK/2)
template <int LPK> DEV f16x8 rope_neox_apply(f16x8 x, f16x8 cs, f16x8 sn, int dl)
{
    constexpr int HL = LPK / 2;
    u32x4 u = __builtin_bit_cast(u32x4, x);
    u32x4 pu;
    #pragma unroll
    for (int i = 0; i < 4; i++) pu[i] = swz_xor_u32<HL>(u[i]);
    const f16x8 partner = __builtin_bit_cast(f16x8, pu);
    if (dl < HL) sn = -sn;                               // left half: l' = l cos + r (-sin); right: r' = r cos + l sin
    const f16x8 t = partner * sn;
    return __builtin_elementwise_fma(x, cs, t);
}
template <int LPK> DEV f16x8 rope_neox_frag(f16x8 x, const f16* sin, const f16* cos, int pos, int dl)
{
    f16x8 cs, sn;
    rope_neox_load<LPK>(sin, cos, pos, dl, cs, sn);
    return rope_neox_apply<LPK>(x, cs, sn, dl);
}

// DEP: a launch of the overlapped chain (chain_sync.h; experimental) -- a TEMPLATE parameter since round 5, like the lean kernel's: as
// run-time tests the polling loop and the agent-scope alternatives of every level-1 load sat in the ordinary launch's code, and the
// compiler drained the first request (`vmcnt(0)`) in front of them: one round trip at the head of every launch.
// kh / split / zidx = what blockIdx.x / .y / .z are for the kernel; NT = threads taking part (ATT_WAVES * 64); smem: dynamic LDS.
template <int HDIM, int RB, bool DEP = false>
DEV void attn_fused_body(const FusedArgs& a, const int kh, const int split, const int zidx, u8* const smem)
{
    constexpr int NT = ATT_WAVES * 64;
    constexpr int LPK = HDIM / 8;
    constexpr int KPW = 64 / LPK;
    constexpr int NSTREAM = ATT_WAVES * KPW;
    constexpr int ROWF = HDIM + 2;

    const int G = a.H / a.KVH;
    const int R = a.s * G;
    const int rblocks = (R + RB - 1) / RB;
    const int b = zidx / rblocks;
    const int rblk = zidx % rblocks;
    const int r0 = rblk * RB;
    const int nrows = min(RB, R - r0);

    const int lane = lane_id();
    const int wv = wave_id();
    const int group = lane / LPK;
    const int dl = lane % LPK;

    constexpr bool dep = DEP;                                  // overlapped chain: producer may still be running
    if constexpr (DEP) if (a.sync_arrive && tid() == 0) sync_report_entry(a.sync_arrive, (u32)((zidx * gdim_y() + split) * gdim_x() + kh));

    // ---- request level 1: everything whose address does not depend on the cache length goes out TOGETHER -- the length
    // itself, the first page of the sequence (split 0 starts at key 0), the query rows, the new key / value row this stream
    // takes first, the output position of the element this thread finalises.  (One dependent round trip each before:
    // length -> page -> keys -> new key -> rotary rows -> output position, 6.9 us per launch at 64 keys;
    // profiles/history/r03_kernel_stats.csv.)
    // (UNCONDITIONAL loads from an address that is valid either way: behind `if (ptr) v = *ptr` the compiler waits for the load where the
    // two paths meet -- `s_waitcnt vmcnt(0)` right behind the request, in front of every other level-1 request: one whole round trip
    // at the head of every launch, found in the gfx950 code in round 5)
    // (a non-temporal load = a VECTOR load the scheduler hoists with the other level-1 requests: as an ordinary load of a uniform
    // address the compiler made it a scalar load and sank it behind the wait for the vector loads -- a dependent level of its own
    // again; as a volatile load it is waited for on the spot)
    const int p_ld = ld_nt(a.cache_seqlens ? a.cache_seqlens + b : (const int*)a.q);
    const bool spec = a.block_table != nullptr && split == 0;
    const int tab_spec = ld_nt(spec ? a.block_table + (size_t)b * a.pages_per_seq : (const int*)a.q);
    if constexpr (DEP) if (a.sync_wait)
    {
        // overlapped chain: q / k_new / v_new are read behind the wait; workgroups of unused splits leave before it
        const int p_raw_e = a.cache_seqlens ? p_ld : 0;
        const int total_e = a.past_const + (p_raw_e > 0 ? p_raw_e : 0) + a.s;
        if (split >= fused_active_splits(a, total_e)) return;
        if (wv == 0) sync_wait_go(a.sync_wait, kh + split);
        block_sync();
    }
    f16x8 qf[RB];
    int limit[RB];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        const int rr = r0 + (r < nrows ? r : 0);
        const int j = rr / G, g = rr - j * G;
        const f16* qp = a.q + (((size_t)b * a.s + j) * a.H + kh * G + g) * HDIM + dl * 8;
        qf[r] = dep ? load_agent_f16x8(qp) : *(const f16x8*)qp;
        limit[r] = j + 1;
    }
    const int j0 = wv * KPW + group;                            // the new key this stream takes first
    const int jn = j0 < a.s ? j0 : a.s - 1;
    const size_t src0 = (((size_t)b * a.s + jn) * a.KVH + kh) * HDIM + dl * 8;
    f16x8 kn0 = dep ? load_agent_f16x8(a.k_new + src0) : *(const f16x8*)(a.k_new + src0);
    const f16x8 vn0 = dep ? load_agent_f16x8(a.v_new + src0) : *(const f16x8*)(a.v_new + src0);
    int out_pre = 0;
    if (a.out_invperm && tid() < nrows * HDIM)
    {
        const int r = tid() / HDIM, d = tid() - r * HDIM;
        const int rr = r0 + r;
        const int g = rr - (rr / G) * G;
        out_pre = (int)a.out_invperm[(kh * G + g) * HDIM + d];
    }

    const int p_raw = a.cache_seqlens ? p_ld : 0;
    // (tab_spec tied to the length through an opaque zero: needed HERE, so its request cannot be sunk -- as a scalar load -- into the
    // block-table branch behind the wait for the level-1 vector loads)
    u32 opaque_zero = 0;
    pin_scalar(opaque_zero);
    const int past = a.past_const + (p_raw > 0 ? p_raw : 0) + (int)((u32)tab_spec & opaque_zero);
    const int total = past + a.s;
    const int eff = fused_active_splits(a, total);
    if (split >= eff) return;
    int kps = (total + eff - 1) / eff;
    kps = (kps + 15) & ~15;
    const int k_start = split * kps;
    const int k_end = min(total, k_start + kps);
    #pragma unroll
    for (int r = 0; r < RB; r++) limit[r] += past;

    // Page ids of the split's key range live in LDS (round 5).  Before, every key step looked its page up in memory and the compiler's
    // wait for that look-up -- `s_waitcnt vmcnt(0)` -- also drained the K / V rows already requested: the steps of a batch went out one
    // round trip after the other instead of together.  ONE look-up path only: with a memory look-up kept as a run-time alternative the
    // compiler waits for it on the LDS path too.  A split inside the sequence's first page (every short-context launch) takes the entry
    // that was requested speculatively at level 1: no further dependent load, one barrier.  (The host refuses tables wider than
    // ATT_MAX_PAGES: exl2_attn_decode_fused returns 1, the caller takes the three-launch route.)
    int* const pg_lds = (int*)(smem + (size_t)NSTREAM * RB * ROWF * 4 + 16);
    const int pg0 = a.block_table ? (k_start >> a.page_shift) : 0;
    if (a.block_table)
    {
        const int npg = (((k_end > k_start ? k_end : k_start + 1) - 1) >> a.page_shift) - pg0 + 1;
        // (an EMPTY split -- k_start >= total once eff is clamped and kps rounded up -- may sit one page past the row: the index is clamped,
        // the entry is never used; round-5 advisor finding)
        if (spec && npg == 1 && pg0 == 0) { if (tid() == 0) pg_lds[0] = tab_spec; }
        else for (int i = tid(); i < npg; i += NT) pg_lds[i] = a.block_table[(size_t)b * a.pages_per_seq + min(pg0 + i, a.pages_per_seq - 1)];
        block_sync();
    }
    auto slot_of = [&](const int kp) -> size_t
    {
        if (a.block_table) return (size_t)pg_lds[(kp >> a.page_shift) - pg0] * a.page_size + (kp & (a.page_size - 1));
        return (size_t)b * a.page_size + kp;
    };
    const size_t row_stride = (size_t)a.KVH * HDIM;
    const int k_old_end = min(k_end, past);
    constexpr int STEP = ATT_WAVES * KPW;
    constexpr int UNR = RB <= 2 ? 2 * ATT_UNROLL : ATT_UNROLL;   // keys of UNR steps requested together (128 = a whole split at 1-2 rows)

    // ---- request level 2: what needs the length -- the first batch of cached keys, the rotary rows of the queries and of
    // the new key -- again together, before anything is used
    // Two half-batches (round 5): while one is used the other is in flight.  EVERY request is unconditional -- a position beyond the
    // split's cached keys reads the split's first row (a cache hit; its score is masked) -- so that the compiler's count of the loads in
    // flight stays exact and the wait in front of a half-batch is `vmcnt(loads of the other half)`, not `vmcnt(0)`.
    constexpr int HB = UNR / 2;
    f16x8 kfa[HB], vfa[HB], kfb[HB], vfb[HB];
    const int base_first = k_start + wv * KPW;
    auto request_half = [&](f16x8 (&kf)[HB], f16x8 (&vf)[HB], const int base0)
    {
        // (all cache slots first, then all rows: whatever the page look-up waits for, it waits ONCE per half-batch, in front of its
        // loads, not between them)
        size_t slot[HB];
        #pragma unroll
        for (int u = 0; u < HB; u++)
        {
            const int kpos = base0 + u * STEP + group;
            slot[u] = slot_of(kpos < k_old_end ? kpos : k_start);   // keep the address valid, mask the score
        }
        #pragma unroll
        for (int u = 0; u < HB; u++)
        {
            const size_t off = slot[u] * row_stride + (size_t)kh * HDIM + dl * 8;
            kf[u] = ld_nt((const f16x8*)(a.k_cache + off));
            vf[u] = ld_nt((const f16x8*)(a.v_cache + off));
        }
    };
    const bool any_old = base_first < k_old_end;
    if (any_old) { request_half(kfa, vfa, base_first); request_half(kfb, vfb, base_first + HB * STEP); }
    const bool pre_new = k_start <= past;                       // the stream's first new key is row j0 of this step
    if (a.rope)
    {
        f16x8 cs_q[RB], sn_q[RB], cs_k, sn_k;
        #pragma unroll
        for (int r = 0; r < RB; r++) rope_neox_load<LPK>(a.sin, a.cos, limit[r] - 1, dl, cs_q[r], sn_q[r]);
        rope_neox_load<LPK>(a.sin, a.cos, past + jn, dl, cs_k, sn_k);
        #pragma unroll
        for (int r = 0; r < RB; r++) qf[r] = rope_neox_apply<LPK>(qf[r], cs_q[r], sn_q[r], dl);
        kn0 = rope_neox_apply<LPK>(kn0, cs_k, sn_k, dl);
    }

    float m[RB], l[RB], o[RB][8];
    #pragma unroll
    for (int r = 0; r < RB; r++)
    {
        m[r] = NEG_BIG; l[r] = 0.0f;
        #pragma unroll
        for (int e = 0; e < 8; e++) o[r][e] = 0.0f;
    }

    auto attend = [&](const f16x8 kf, const f16x8 vf, const int kpos, const bool in_range)
    {
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
                const float sc = d * a.scale;
                const bool valid = in_range && kpos < limit[r];
                const float m_new = valid ? fmaxf(m[r], sc) : m[r];
                const float alpha = fast_exp(m[r] - m_new);
                const float p = valid ? fast_exp(sc - m_new) : 0.0f;
                m[r] = m_new;
                l[r] = l[r] * alpha + p;
                #pragma unroll
                for (int e = 0; e < 8; e++) o[r][e] = o[r][e] * alpha + p * (float)vf[e];
            }
        }
    };
    // keys already in the cache: the K/V rows of UNR steps are requested together (one round trip per batch instead of one
    // per step); the first batch is already in flight
#define ATTEND_HALF(KF, VF, BASE) \
    _Pragma("unroll") \
    for (int u = 0; u < HB; u++) \
    { \
        const int kpos_ = (BASE) + u * STEP + group; \
        if ((BASE) + u * STEP < k_old_end) attend(KF[u], VF[u], kpos_, kpos_ < k_old_end); \
    }
    for (int base0 = base_first; base0 < k_old_end; base0 += UNR * STEP)
    {
        // half A is used with half B in flight; A is re-requested (next batch) before B is used
        ATTEND_HALF(kfa, vfa, base0)
        if (base0 + HB * STEP >= k_old_end) break;
        request_half(kfa, vfa, base0 + UNR * STEP);
        ATTEND_HALF(kfb, vfb, base0 + HB * STEP)
        if (base0 + UNR * STEP >= k_old_end) break;
        request_half(kfb, vfb, base0 + UNR * STEP + HB * STEP);
    }
#undef ATTEND_HALF
    // keys of this step: rotate, use, append (the stream's first one was requested and rotated above)
    const int base_new = max(k_start, past) + wv * KPW;
    for (int base = base_new; base < k_end; base += ATT_WAVES * KPW)
    {
        const int kpos = base + group;
        const bool in_range = kpos < k_end;
        const int kp = in_range ? kpos : k_end - 1;
        f16x8 kfn, vfn;
        if (pre_new && base == base_new) { kfn = kn0; vfn = vn0; }
        else
        {
            const size_t src = (((size_t)b * a.s + (kp - past)) * a.KVH + kh) * HDIM + dl * 8;
            kfn = dep ? load_agent_f16x8(a.k_new + src) : *(const f16x8*)(a.k_new + src);
            vfn = dep ? load_agent_f16x8(a.v_new + src) : *(const f16x8*)(a.v_new + src);
            if (a.rope) kfn = rope_neox_frag<LPK>(kfn, a.sin, a.cos, kp, dl);
        }
        if (in_range && rblk == 0 && a.k_cache)
        {
            const size_t off = slot_of(kp) * row_stride + (size_t)kh * HDIM + dl * 8;
            *(f16x8*)(a.k_cache + off) = kfn;
            *(f16x8*)(a.v_cache + off) = vfn;
        }
        attend(kfn, vfn, kpos, in_range);
    }

    // merge the NSTREAM independent softmax streams of this workgroup
    float* st = (float*)smem;
    u32* ticket_lds = (u32*)(st + (size_t)NSTREAM * RB * ROWF);
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
    for (int idx = tid(); idx < nrows * HDIM; idx += NT)
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
        if (eff == 1)
        {
            const f16 y = (f16)(L > 0.0f ? O / L : 0.0f);
            const size_t oi = (a.out_invperm && idx == tid()) ? (qrow / a.H) * ((size_t)a.H * HDIM) + (size_t)out_pre : fused_out_index<HDIM>(a, qrow, d);
            if (DEP && a.sync_signal) store_agent_f16(a.out + oi, y);
            else a.out[oi] = y;
            if (a.out_nat) a.out_nat[qrow * HDIM + d] = y;
        }
        else
        {
            // partial results travel as agent-scope (write-through) stores and are read back with agent-scope loads: the
            // hand-off then needs no L2 write-back / invalidate, only "my stores have completed" before the ticket
            store_agent_f32(a.part_o + (qrow * a.nsplit + split) * HDIM + d, O);
            if (d == 0)
            {
                store_agent_f32(a.part_ml + (qrow * a.nsplit + split) * 2 + 0, M);
                store_agent_f32(a.part_ml + (qrow * a.nsplit + split) * 2 + 1, L);
            }
        }
    }
    // (overlapped chain: the outputs above / below are agent-scope stores; one signal once the workgroup's have completed)
    auto signal_done = [&]() {
        if (!DEP || !a.sync_signal) return;
        wait_vmcnt0();
        block_sync();
        if (wv == 0) sync_arrive_publish(a.sync_signal, a.sync_total, a.sync_wait);
    };
    if (eff == 1) { signal_done(); return; }

    // hand-off: the last split to arrive merges
    wait_vmcnt0();
    block_sync();
    u32* counter = a.counters + ((size_t)b * a.KVH + kh) * rblocks + rblk;
    if (tid() == 0) *ticket_lds = ticket_add_agent(counter, 1u);
    block_sync();
    if (*ticket_lds != (u32)(eff - 1)) return;
    for (int idx = tid(); idx < nrows * HDIM; idx += NT)
    {
        const int r = idx / HDIM, d = idx - r * HDIM;
        const int rr = r0 + r;
        const int j = rr / G, g = rr - j * G;
        const size_t qrow = ((size_t)b * a.s + j) * a.H + kh * G + g;
        const f16 y = (f16)merge_split_partials<true>(a.part_o, a.part_ml, qrow, a.nsplit, eff, HDIM, d);
        if (DEP && a.sync_signal) store_agent_f16(a.out + fused_out_index<HDIM>(a, qrow, d), y);
        else a.out[fused_out_index<HDIM>(a, qrow, d)] = y;
        if (a.out_nat) a.out_nat[qrow * HDIM + d] = y;
    }
    if (tid() == 0) store_relaxed_agent(counter, 0u);
    signal_done();
}

