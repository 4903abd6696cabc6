// attn_merge.h -- merge of the split partials of the split-KV decode attention kernels (attn.hip, attn_q4.hip).
#pragma once

// The `eff` partials (running maximum M_s, sum L_s, weighted value O_s[d]) of one (query row, feature d) are requested EIGHT AT A
// TIME and folded with the online-softmax update: ceil(eff / 8) round trips.  Rounds 1-4 ran `for s: M = max(M, load)` and then
// `for s: three loads, accumulate` -- one load-and-wait per loop trip (the gfx950 code had `s_waitcnt vmcnt(0)` inside both loops)
// = 2 eff dependent round trips of ~0.6 us: a 64-split launch of a grouped-query model spent ~75 of its 80 us in the merge
// (profiles/history/r05l_attn_sweep2.jsonl).  AGENT: the partials were written by other workgroups of the SAME launch (agent-scope loads;
// hw.h explains why no fence).
template <bool AGENT>
DEV float merge_split_partials(const float* part_o, const float* part_ml, size_t qrow, int nsplit, int eff, int hd, int d)
{
    float M = -1.0e30f, L = 0.0f, O = 0.0f;
    for (int s0 = 0; s0 < eff; s0 += 8)
    {
        float mv[8], lv[8], ov[8];
        #pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const int s2 = s0 + u < eff ? s0 + u : eff - 1;
            const float* ml = part_ml + (qrow * (size_t)nsplit + s2) * 2;
            const float* po = part_o + (qrow * (size_t)nsplit + s2) * hd + d;
            if constexpr (AGENT) { mv[u] = load_agent_f32(ml); lv[u] = load_agent_f32(ml + 1); ov[u] = load_agent_f32(po); }
            else { mv[u] = ml[0]; lv[u] = ml[1]; ov[u] = po[0]; }
        }
        #pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (s0 + u < eff)
            {
                const float m_new = fmaxf(M, mv[u]);
                const float alpha = fast_exp(M - m_new), w = fast_exp(mv[u] - m_new);
                L = L * alpha + lv[u] * w;
                O = O * alpha + ov[u] * w;
                M = m_new;
            }
        }
    }
    return L > 0.0f ? O / L : 0.0f;
}
