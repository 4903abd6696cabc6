// cache_q_pack.h -- the Q4 / Q8 KV-cache packer on registers, and RoPE on a head row held one f16 pair per lane: shared by the
// codec (cache_q.hip), the RoPE + pack launch of a decode step (cache_q.hip: rope_quant_q4_kernel) and the one-launch Q4 decode
// attention (attn_q4.hip).  Reference numerics: cache_q.cuh:4-185 (Walsh-Hadamard over 32 lanes, absmax per 32 elements, RTN, fp16
// scale), rope.cu as restated in attn.hip: rope_append_kernel.
#pragma once
#include "hw.h"

DEV f16x2 wht32(f16x2 w, int t)
{
    u32 wi = as_u32(w);
    #define WHT_STEP(M) { const u32 p = swz_xor_u32<M>(wi); if (t & M) wi ^= 0x80008000u; wi = as_u32(as_h2(wi) + as_h2(p)); }
    WHT_STEP(1) WHT_STEP(2) WHT_STEP(4) WHT_STEP(8) WHT_STEP(16)
    #undef WHT_STEP
    return as_h2(wi);
}

DEV f16 hmax(f16 a, f16 b) { return a > b ? a : b; }
DEV f16 habs(f16 a) { return as_h((u16)(as_u16(a) & 0x7FFF)); }

DEV int rint_clamp(f16 v, int hi)
{
    const float f = (float)v;
    int q = (f != f) ? 0 : (int)rintf(f);          // __half2int_rn(NaN) == 0
    q = q < 0 ? 0 : q;
    return q > hi ? hi : q;
}

// pack one 512-element block; t = thread in block (0..255).  Everything below the load works on 32-lane halves of a wave (64
// consecutive elements), so ONE wave with t = lane packs any 128-element piece whose offset is a multiple of 128 (q_pack_lane's
// other caller: rope_quant_q4_kernel, one wave per head row of 128)
template <int WBITS>
DEV void q_pack_lane(int t, f16x2 w, u8* out, f16* scales, size_t block_offset)
{
    w = wht32(w, t);

    f16 am = hmax(habs(w.x), habs(w.y));
    am = hmax(am, as_h((u16)swz_xor_u32<8>((u32)as_u16(am))));
    am = hmax(am, as_h((u16)swz_xor_u32<4>((u32)as_u16(am))));
    am = hmax(am, as_h((u16)swz_xor_u32<2>((u32)as_u16(am))));
    am = hmax(am, as_h((u16)swz_xor_u32<1>((u32)as_u16(am))));
    const f16x2 am2 = h2_dup(am);

    if constexpr (WBITS == 4)
    {
        f16x2 n = h2_div_rn(w, am2);
        n = h2_fma(n, h2_dup((f16)8.0f), h2_dup((f16)8.0f));
        u32 q = (u32)rint_clamp(n.x, 15) | ((u32)rint_clamp(n.y, 15) << 4);
        q |= shfl_idx_u32(q, lane_id() + 1) << 8;          // lanes t % 2 == 0 now hold 2 bytes
        q |= shfl_idx_u32(q, lane_id() + 2) << 16;         // lanes t % 4 == 0 hold 4 bytes
        if ((t & 3) == 0) ((u32*)(out + block_offset / 2))[t >> 2] = q;
        if ((t & 15) == 0) scales[block_offset / 32 + (t >> 4)] = am * (f16)0.125f;
    }
    else
    {
        f16x2 n = h2_div_rn(w, am2);
        n = h2_fma(n, h2_dup((f16)128.0f), h2_dup((f16)128.0f));
        u32 q = (u32)rint_clamp(n.x, 255) | ((u32)rint_clamp(n.y, 255) << 8);
        q |= shfl_idx_u32(q, lane_id() + 1) << 16;
        if ((t & 1) == 0) ((u32*)(out + block_offset))[t >> 1] = q;
        if ((t & 15) == 0) scales[block_offset / 32 + (t >> 4)] = am * (f16)0.0078125f;
    }
}

// RoPE on head rows held as elements (2 t, 2 t + 1) in lane t of one wave: HDIM = 128 -- one row per wave; HDIM = 64 -- TWO rows (two
// adjacent kv heads of one token: lanes 0..31 and 32..63), same position.  sr / cr = the position's sin / cos rows.  Same fp16 operations
// in the same order as rope_append_kernel (attn.hip): NeoX pairs (c, c + HDIM / 2) sit in lanes t and t ^ (HDIM / 4).
template <int HDIM>
DEV f16x2 rope_lane_pair(f16x2 w, int t, const f16* sr, const f16* cr, bool neox)
{
    constexpr int LPH = HDIM / 2;                                   // lanes per head row
    const int th = t & (LPH - 1);
    if (neox)
    {
        const f16x2 p = as_h2(shfl_xor_u32(as_u32(w), LPH / 2));
        const int c = 2 * (th & (LPH / 2 - 1));
        const f16x2 cs = *(const f16x2*)(cr + c), sn = *(const f16x2*)(sr + c);
        if (th < LPH / 2)
        {
            // l = own, r = partner:  l' = fma(l, cos, r * (-sin))
            w.x = h_fma(w.x, cs.x, p.x * (-sn.x));
            w.y = h_fma(w.y, cs.y, p.y * (-sn.y));
        }
        else
        {
            // r = own, l = partner:  r' = fma(r, cos, l * sin)
            w.x = h_fma(w.x, cs.x, p.x * sn.x);
            w.y = h_fma(w.y, cs.y, p.y * sn.y);
        }
    }
    else
    {
        const int c0 = 2 * th;
        const f16x2 cs = *(const f16x2*)(cr + c0), sn = *(const f16x2*)(sr + c0);
        const f16 r0 = h_fma(w.y, -sn.x, w.x * cs.x);
        const f16 r1 = h_fma(w.x, sn.y, w.y * cs.y);
        w.x = r0; w.y = r1;
    }
    return w;
}
DEV f16x2 rope_lane_pair128(f16x2 w, int t, const f16* sr, const f16* cr, bool neox) { return rope_lane_pair<128>(w, t, sr, cr, neox); }
