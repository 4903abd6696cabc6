// qlayout.h -- the private ("tile16") weight layout this library converts EXL2 / GPTQ tensors into at
// make_q_matrix time, and the register-level decoders the kernels use.
//
// On disk (reference format, SURVEY.md A.1 / pack_tensor.cu:118-248): q_weight int32 [R, N]; a run of 32 K-rows at
// b bits is b consecutive word-rows; the 32 codes of one column form one LSB-first bitstream over those b words.
//
// Private layout (ours; only the on-disk format is contract -- the reference also re-shuffles in place,
// q_matrix.cu:189-195):
//   * columns are cut into tiles of 16; K is cut per bit-width section into super-chunks of 128 rows (4 chunks of 32);
//   * one super-chunk of one tile is owned by one wavefront: lane l = (j = l >> 4, c = l & 15) owns column c of the tile
//     and rows {32 q + 8 j + e : q = 0..3, e = 0..7} of the super-chunk = exactly the 8 K-values of the MFMA
//     16x16x32 B-fragment of chunk q.  That is 32 codes = b dwords per lane, for every b;
//   * the 32 codes of a lane are 16 "pairs" P = 4 q + i holding (e = 2 i) in the LOW 16-bit half and (e = 2 i + 1) in
//     the HIGH half of a dword at the same bit offset, so `(w >> s) & mask | 0x64006400` yields the half2
//     (1024 + code_lo * m, 1024 + code_hi * m) and one v_pk_add / v_pk_fma turns it into exact (code - zero) values;
//   * in memory a super-chunk of a tile is [piece][lane][words of the piece]; pieces are the 16/12/8/4-byte vector
//     loads the lane issues (b = 4: x4 | 8: x4,x4 | 6: x4,x2 | 5: x4,x1 | 3: x3 | 2: x2), so every load
//     instruction of a wave reads one contiguous run;
//   * a section occupies the same words it occupied on disk: [tile][super-chunk][...]; when a section has
//     (chunks % 4) != 0 its last, partial super-chunk lives (padded) in a small side buffer.
#pragma once
#include "hw.h"

#define TILE_N 16
#define SUPER_ROWS 128
#define MAGIC_H2 0x64006400u

// ---- where pair P of a lane's super-chunk lives: regular pairs (word, bit offset within each 16-bit half) -----------
// returns word index, *off = bit offset, or -1 for the "extra" pairs whose bits are scattered (3/5/6-bit)
HD int pair_slot(int bits, int P, int* off)
{
    switch (bits)
    {
        case 2: *off = 8 * ((P >> 2) & 1) + 2 * (P & 3); return P >> 3;
        case 3: if (P == 15) return -1; *off = 3 * (P % 5); return P / 5;
        case 4: *off = 4 * (P & 3); return P >> 2;
        case 5: if (P == 15) return -1; *off = 5 * (P % 3); return P / 3;
        case 6: if (P >= 12) return -1; *off = 6 * (P & 1); return P >> 1;
        case 8: *off = 8 * (P & 1); return P >> 1;
    }
    return -1;
}

// Insert one code into a lane's word array.  hi = 0: low half (even e), 1: high half (odd e).
HD void put_code(u32* w, int bits, int P, int hi, u32 code)
{
    int off;
    int word = pair_slot(bits, P, &off);
    const int hs = hi * 16;
    if (word >= 0) { w[word] |= code << (off + hs); return; }
    if (bits == 3)          // P == 15: bit t of the code -> bit 15 of the half in word t
    {
        for (int t = 0; t < 3; t++) w[t] |= ((code >> t) & 1u) << (15 + hs);
    }
    else if (bits == 5)     // P == 15: bit t -> bit 15 of the half in word t (t = 0..4)
    {
        for (int t = 0; t < 5; t++) w[t] |= ((code >> t) & 1u) << (15 + hs);
    }
    else if (bits == 6)     // P = 12 + x: low 4 bits -> bits [12,16) of the half in word x;
    {                       //             high 2 bits -> bits [12 + 2 (x & 1), +2) of the half in word 4 + (x >> 1)
        const int x = P - 12;
        w[x] |= (code & 0xFu) << (12 + hs);
        w[4 + (x >> 1)] |= ((code >> 4) & 3u) << (12 + 2 * (x & 1) + hs);
    }
}

// Generic (slow) inverse of put_code: used by tests and as the executable definition of the layout.
HD u32 get_code(const u32* w, int bits, int P, int hi)
{
    int off;
    int word = pair_slot(bits, P, &off);
    const int hs = hi * 16;
    if (word >= 0) return (w[word] >> (off + hs)) & ((1u << bits) - 1u);
    u32 code = 0;
    if (bits == 3)      { for (int t = 0; t < 3; t++) code |= ((w[t] >> (15 + hs)) & 1u) << t; }
    else if (bits == 5) { for (int t = 0; t < 5; t++) code |= ((w[t] >> (15 + hs)) & 1u) << t; }
    else if (bits == 6)
    {
        const int x = P - 12;
        code = (w[x] >> (12 + hs)) & 0xFu;
        code |= ((w[4 + (x >> 1)] >> (12 + 2 * (x & 1) + hs)) & 3u) << 4;
    }
    return code;
}

// ---- load pieces -----------------------------------------------------------------------------------------------------
// word index (within a super-chunk of a tile, 64 * bits words) of lane `lane`'s word `i`
HD int lane_word_index(int bits, int lane, int i)
{
    // pieces: first piece = min(bits, 4) words (3 for bits == 3, 2 for bits == 2), second piece = the rest
    const int p0 = bits < 4 ? bits : 4;
    if (i < p0) return lane * p0 + i;
    const int p1 = bits - p0;
    return 64 * p0 + lane * p1 + (i - p0);
}

// ---- descriptor of one run of super-chunks (device + host) ----------------------------------------------------------
struct QDesc
{
    u32 base_word;      // word offset of (tile 0, super-chunk 0 of this run) in the weight buffer / tail buffer
    u32 tile_stride;    // words between consecutive 16-column tiles
    u16 n_super;        // super-chunks in this run (<= QDESC_MAX_SUPER)
    u16 k_base;         // first packed K-row of the run
    u8  bits;
    u8  nvalid_last;    // valid 32-row chunks in the LAST super-chunk of the run (1..4)
    u8  in_tail;        // 0: weight buffer, 1: tail buffer
    u8  pad0;
    u32 sc_prefix;      // index of this run's first super-chunk among all super-chunks of the matrix (K order)
};
#define QDESC_MAX_SUPER 16

// One bit-width section of the matrix (or its partial last super-chunk): a contiguous stream per 16-column tile.
// Unlike QDesc these are not split for phased staging; they travel in the kernel arguments (scalar registers).
struct QRun
{
    u32 base_word;      // word offset of (tile 0, super-chunk 0) in the weight buffer / tail buffer
    u32 tile_stride;    // words between consecutive tiles
    u16 n_super;        // super-chunks per tile in this run
    u16 k_base;         // first packed K row
    u8  bits;
    u8  nvalid_last;    // valid chunks of the last super-chunk (4 unless this is a tail run)
    u8  in_tail;
    u8  pad;
};
#define MAX_RUNS 12

#ifndef QLAYOUT_NO_DEVICE_DECODERS

// ---- register-level decoders -----------------------------------------------------------------------------------------
// raw(bits) -> exact (code - zero) as half2.  `zero` may differ per chunk (GPTQ); EXL2 passes 2^(b-1).
DEV f16x2 dq_direct(u32 x, u32 mask, f16x2 sub)        { return as_h2(and_or(x, mask, MAGIC_H2)) - sub; }
DEV f16x2 dq_scaled(u32 x, u32 mask, f16x2 mul, f16x2 add) { return h2_fma(as_h2(and_or(x, mask, MAGIC_H2)), mul, add); }

// constants for one chunk: sub = 1024 + z ; addM = -(1024 / M + z)
struct ZC { f16x2 sub, a4, a8, a16, a32, a64; };
DEV ZC make_zc(f16 z)
{
    ZC r;
    r.sub = h2_dup((f16)1024.0f + z);
    r.a4  = h2_dup(-((f16)256.0f + z));
    r.a8  = h2_dup(-((f16)128.0f + z));
    r.a16 = h2_dup(-((f16)64.0f + z));
    r.a32 = h2_dup(-((f16)32.0f + z));
    r.a64 = h2_dup(-((f16)16.0f + z));
    return r;
}
#define H2C(v) ((f16x2){(f16)(v), (f16)(v)})

// decode all 16 pairs of a super-chunk: p[4 q + i] = (code(e = 2i) - z_q, code(e = 2i+1) - z_q)
template <int BITS> DEV void dequant_super(const u32* w, const ZC* zc, f16x2* p);

template <> DEV void dequant_super<4>(const u32* w, const ZC* zc, f16x2* p)
{
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const u32 x = w[q], y = x >> 8;
        p[4 * q + 0] = dq_direct(x, 0x000F000Fu, zc[q].sub);
        p[4 * q + 1] = dq_scaled(x, 0x00F000F0u, H2C(0.0625f), zc[q].a16);
        p[4 * q + 2] = dq_direct(y, 0x000F000Fu, zc[q].sub);
        p[4 * q + 3] = dq_scaled(y, 0x00F000F0u, H2C(0.0625f), zc[q].a16);
    }
}

template <> DEV void dequant_super<8>(const u32* w, const ZC* zc, f16x2* p)
{
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const u32 x0 = w[2 * q], x1 = w[2 * q + 1];
        p[4 * q + 0] = dq_direct(x0,      0x00FF00FFu, zc[q].sub);
        p[4 * q + 1] = dq_direct(x0 >> 8, 0x00FF00FFu, zc[q].sub);
        p[4 * q + 2] = dq_direct(x1,      0x00FF00FFu, zc[q].sub);
        p[4 * q + 3] = dq_direct(x1 >> 8, 0x00FF00FFu, zc[q].sub);
    }
}

template <> DEV void dequant_super<2>(const u32* w, const ZC* zc, f16x2* p)
{
    #pragma unroll
    for (int h = 0; h < 2; h++)          // word h holds chunks 2h (byte 0 of each half) and 2h+1 (byte 1)
    {
        #pragma unroll
        for (int b = 0; b < 2; b++)
        {
            const int q = 2 * h + b;
            const u32 x = b ? (w[h] >> 8) : w[h];
            p[4 * q + 0] = dq_direct(x, 0x00030003u, zc[q].sub);
            p[4 * q + 1] = dq_scaled(x, 0x000C000Cu, H2C(0.25f),     zc[q].a4);
            p[4 * q + 2] = dq_scaled(x, 0x00300030u, H2C(0.0625f),   zc[q].a16);
            p[4 * q + 3] = dq_scaled(x, 0x00C000C0u, H2C(0.015625f), zc[q].a64);
        }
    }
}

template <> DEV void dequant_super<3>(const u32* w, const ZC* zc, f16x2* p)
{
    // pair P < 15: word P / 5, offset 3 * (P % 5); chunk of pair P is P >> 2
    #pragma unroll
    for (int t = 0; t < 3; t++)
    {
        const u32 x = w[t], y = x >> 9;
        const int P = 5 * t;
        p[P + 0] = dq_direct(x, 0x00070007u, zc[(P + 0) >> 2].sub);
        p[P + 1] = dq_scaled(x, 0x00380038u, H2C(0.125f),    zc[(P + 1) >> 2].a8);
        p[P + 2] = dq_scaled(x, 0x01C001C0u, H2C(0.015625f), zc[(P + 2) >> 2].a64);
        p[P + 3] = dq_direct(y, 0x00070007u, zc[(P + 3) >> 2].sub);
        p[P + 4] = dq_scaled(y, 0x00380038u, H2C(0.125f),    zc[(P + 4) >> 2].a8);
    }
    u32 e = (w[0] >> 15) & 0x00010001u;
    e |= (w[1] >> 14) & 0x00020002u;
    e |= (w[2] >> 13) & 0x00040004u;
    p[15] = as_h2(e | MAGIC_H2) - zc[3].sub;
}

template <> DEV void dequant_super<5>(const u32* w, const ZC* zc, f16x2* p)
{
    // pair P < 15: word P / 3, offset 5 * (P % 3)
    #pragma unroll
    for (int t = 0; t < 5; t++)
    {
        const u32 x = w[t];
        const int P = 3 * t;
        p[P + 0] = dq_direct(x,       0x001F001Fu, zc[(P + 0) >> 2].sub);
        p[P + 1] = dq_scaled(x,       0x03E003E0u, H2C(0.03125f), zc[(P + 1) >> 2].a32);
        p[P + 2] = dq_direct(x >> 10, 0x001F001Fu, zc[(P + 2) >> 2].sub);
    }
    u32 e = (w[0] >> 15) & 0x00010001u;
    e |= (w[1] >> 14) & 0x00020002u;
    e |= (w[2] >> 13) & 0x00040004u;
    e |= (w[3] >> 12) & 0x00080008u;
    e |= (w[4] >> 11) & 0x00100010u;
    p[15] = as_h2(e | MAGIC_H2) - zc[3].sub;
}

template <> DEV void dequant_super<6>(const u32* w, const ZC* zc, f16x2* p)
{
    // pair P < 12: word P / 2, offset 6 * (P & 1)
    #pragma unroll
    for (int t = 0; t < 6; t++)
    {
        const u32 x = w[t];
        const int P = 2 * t;
        p[P + 0] = dq_direct(x,      0x003F003Fu, zc[(P + 0) >> 2].sub);
        p[P + 1] = dq_direct(x >> 6, 0x003F003Fu, zc[(P + 1) >> 2].sub);
    }
    // extras P = 12 + x (all in chunk 3): low nibble from word x bits [12,16), high 2 bits from word 4 + (x >> 1)
    #pragma unroll
    for (int x = 0; x < 4; x++)
    {
        u32 e = (w[x] >> 12) & 0x000F000Fu;
        e |= (w[4 + (x >> 1)] >> (8 + 2 * (x & 1))) & 0x00300030u;
        p[12 + x] = as_h2(e | MAGIC_H2) - zc[3].sub;
    }
}

// ---- loading a lane's words of one super-chunk (sc_ptr = first word of the super-chunk of this tile) -----------------
template <int BITS> struct LaneWords { u32 w[BITS]; };

template <int BITS> DEV void load_lane_words(const u32* sc_ptr, int lane, LaneWords<BITS>& r)
{
    if constexpr (BITS == 4)
    {
        const u32x4 v = ld_nt((const u32x4*)sc_ptr + lane);
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    }
    else if constexpr (BITS == 8)
    {
        const u32x4 v0 = ld_nt((const u32x4*)sc_ptr + lane);
        const u32x4 v1 = ld_nt((const u32x4*)(sc_ptr + 256) + lane);
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y; r.w[6] = v1.z; r.w[7] = v1.w;
    }
    else if constexpr (BITS == 6)
    {
        const u32x4 v0 = ld_nt((const u32x4*)sc_ptr + lane);
        const u32x2 v1 = ld_nt((const u32x2*)(sc_ptr + 256) + lane);
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y;
    }
    else if constexpr (BITS == 5)
    {
        const u32x4 v0 = ld_nt((const u32x4*)sc_ptr + lane);
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = ld_nt(sc_ptr + 256 + lane);
    }
    else if constexpr (BITS == 3)
    {
        const u32* p = sc_ptr + lane * 3;
        r.w[0] = ld_nt(p); r.w[1] = ld_nt(p + 1); r.w[2] = ld_nt(p + 2);
    }
    else
    {
        const u32x2 v = ld_nt((const u32x2*)sc_ptr + lane);
        r.w[0] = v.x; r.w[1] = v.y;
    }
}

#endif  // QLAYOUT_NO_DEVICE_DECODERS
