// qgemv_flat.hip -- decode q_gemm for a CHAIN of modules: the producer of an activation vector leaves it in the form its
// consumer's prologue wants, so a launch starts streaming weights one memory round trip after it starts.
//
// Replaces gemm_half_q_half_kernel (exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565) + rms_norm_kernel (rms_norm.cu:33-175)
// + act_mul_kernel (q_mlp_activation.cuh:54-112) on the decode path, as composed by QAttn::forward_cuda_1 / _2
// (q_attn.cu:153-345) and QMLP::forward_run_ (q_mlp.cu:153-236).
//
// Why another kernel next to qgemv_stream.hip: in-kernel timestamps (profiles/r01_trace_mlp.txt) put 4.6-5.8 us of a
// 10-14 us launch BEFORE the first weight is decoded -- argument fetch, a DMA round trip, an LDS->LDS pass that applies
// the act-order permutation + RMSNorm (or recomputes SiLU(gate)*up in every one of the 256 workgroups), three barriers --
// and 1.4-1.7 us after the last (a full barrier, the LDS combine, a dependent residual load).  None of that is bytes.
// Here:
//   * PROLOGUE = one round trip.  Mode A_DIRECT: the input rows already are in this matrix' packed (act-order) K order
//     (the producer scattered them: attention output -> o_proj, SiLU(gate)*up -> down_proj) and are copied global -> LDS by
//     LDS-DMA, nothing else.  Mode A_NORM_PRE: the producer left the residual stream permuted (`xp`) plus one partial
//     sum of squares per producer workgroup (`ss`); the norm weight was permuted at make time; a thread loads its octets
//     of xp / w, every wave reduces the partials (fixed order), the normalised octet goes to LDS.  One LDS barrier.
//   * The weight ring is filled BEFORE the prologue data is waited for: the weight addresses depend on nothing.
//   * WORK SPLIT: the 16-column tiles of all fused matrices form one flat list; a workgroup takes a contiguous range of
//     them ("slots", <= 16) and its waves split the slots' total decode cost (super-chunks x bits) into equal contiguous
//     ranges -- a wave may finish one tile and start the next (two partial sums).  Every CU gets the same number of tiles
//     +-1 whatever the mix of matrices (q|k|v: 3 tiles on each of 256 CUs instead of 4 on 192).
//   * gate and up tiles with the same index sit in the same workgroup (pair mode): the epilogue computes
//     SiLU(gate)*up once and writes it in down's packed order -- down's prologue is A_DIRECT.
//   * EPILOGUE: partial sums meet in a dedicated LDS area (one barrier, fixed order = deterministic); the residual / bias /
//     scatter indices were prefetched in the prologue, so nothing after the barrier waits for memory; a residual-adding
//     launch also publishes x in the NEXT consumer's packed order and its workgroup's partial sum of squares.
#include "qgemv_common.h"
#include "qgemv_flat.h"
#include "errors.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#define FLAT_MAX_SLOTS 16
#define FLAT_WAVES 16
#define FLAT_MINORS 8             // bit-width runs (full runs and partial super-chunks) besides the main one

#ifdef EXL2_TRACE
#define FTRACE(i) do { if (args.trace && lane_id() == 0) args.trace[((size_t)bid_x() * 16 + wave_id()) * 16 + (i)] = realtime_stamp(); } while (0)
#else
#define FTRACE(i) do { } while (0)
#endif

// ---- arguments ----------------------------------------------------------------------------------------------------------
// The 16 waves of a workgroup share ONE scalar unit and every kernel-argument word costs a memory round trip the first
// time it is touched: the first version of this kernel planned its work split on the device (loops over slots and runs,
// dependent argument loads) and spent 8 us before its first barrier.  So: the split is plain arithmetic on a few words,
// and everything the first weight request needs sits in `hot`, fetched with the first batch of scalar loads; the rest
// (`cold`: minor runs, epilogue pointers) is fetched while the first requests are in flight.
struct FpMatHot
{
    const u32* main_base;         // (tile 0, item 0) of the main (largest full) run
    const f16* sc_tab; const f16* zp_tab;
    int n_tiles, unit0, G, cg_units;
    u32 main_stride;              // words between consecutive tiles of the main run
    int main_F;                   // items (super-chunks) of the main run
    int main_meta;                // bits | chunk0 << 8
    int pad;
};
struct FpMinor { u32 base_off; u32 tile_stride; u32 n_chunk; u32 meta; };     // 16 bytes: word offset of (tile 0, item 0) in qw / tail;
                                                                                // n_super | chunk0 << 16; bits | nvalid << 8 | in_tail << 16
struct alignas(64) FpMatCold
{
    const u32* qw; const u32* tail;           // with minor[0..2]: one 64-byte block = one scalar load for the usual matrix
    FpMinor minor[FLAT_MINORS];
    f16* c; const u16* c_invperm; const f16* bias;
    int ldc, n_minor;
};

struct FlatArgs
{
    const f16* a;                 // A_DIRECT: rows in packed order [M, lda]; A_NORM_PRE: xp [M, lda]
    const f16* norm_w;            // A_NORM_PRE: norm weight in packed order [K]
    const float* ss;              // A_NORM_PRE: partial sums of squares [M, npart]
    f16* xp_out; const u16* xp_invperm; float* ss_out;      // chain-out (nullable): x in the next consumer's order + partials
    u64* trace;
    int npart, lda, K, M, a_mode, n_mats, pair, a_stride;
    int c_mode, act_gelu, ldxp, any_bias, wgs;
    int units_lo, units_rem;      // a workgroup owns lo units (pairs in pair mode), the first `rem` workgroups one more
    int S[2], slot_mul[2], slice_mul[2];      // per class (lo / lo + 1 units): waves per slot, floor(w / S) and floor(x / Sm) multipliers
    u32 lds_minor_off, minor_wave_bytes;      // wave-private staging of the minor items
    float eps;
    u32 lds_sc_off, sc_piece, lds_zp_off, lds_cg_off, cg_stride, lds_red_off;
    FpMatHot hot[FLAT_MAX_MATS];
    FpMatCold cold[FLAT_MAX_MATS];
};

// ---- the weight stream: a wave-private ring of items in LDS, filled by LDS-DMA -------------------------------------------
// An item = one super-chunk (128 K-rows) of one 16-column tile = 256 * bits bytes, contiguous in the tile16 layout.  A wave's
// work is a short list of segments (its slice of the main run, its slices of the minor runs, any bit widths); their items
// go through ONE ring: item j is copied global -> LDS (global_load_lds, no registers, non-temporal) D items ahead of its
// decode, whatever segment it belongs to -- so a change of bit width costs no memory round trip (with a register ring the
// next segment started cold: +1.5-2 us on every wave, profiles/r02_trace_flat_v3.txt), and the ring costs no VGPRs.
DEV int item_dma_instrs(int bits) { return (16 * bits + 63) >> 6; }           // 1 KiB per wave-instruction

// copy one item (16 * bits units of 16 bytes) to an LDS slot
DEV void ring_issue(const u32* src, u8* slot, int bits, int lane)
{
    const int units = 16 * bits;
    for (int base = 0; base < units; base += 64)
        if (base + lane < units) dma_to_lds16_nt(src + (size_t)(base + lane) * 4, slot + (size_t)base * 16);
}

template <int BITS> DEV void ring_issue_t(const u32* src, u8* slot, int lane)
{
    constexpr int units = 16 * BITS;
    #pragma unroll
    for (int base = 0; base < units; base += 64)
        if (base + 64 <= units || lane < units - base) dma_to_lds16_nt(src + (size_t)(base + lane) * 4, slot + (size_t)base * 16);
}

// a lane's words of an item that sits in LDS in its memory layout ([piece][lane][words], qlayout.h)
template <int BITS> DEV void lds_lane_words(const u32* slot, int lane, LaneWords<BITS>& r)
{
    if constexpr (BITS == 4)
    {
        const u32x4 v = ((const u32x4*)slot)[lane];
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    }
    else if constexpr (BITS == 8)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        const u32x4 v1 = ((const u32x4*)(slot + 256))[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y; r.w[6] = v1.z; r.w[7] = v1.w;
    }
    else if constexpr (BITS == 6)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        const u32x2 v1 = ((const u32x2*)(slot + 256))[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y;
    }
    else if constexpr (BITS == 5)
    {
        const u32x4 v0 = ((const u32x4*)slot)[lane];
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = slot[256 + lane];
    }
    else if constexpr (BITS == 3)
    {
        const u32* p = slot + lane * 3;
        r.w[0] = p[0]; r.w[1] = p[1]; r.w[2] = p[2];
    }
    else
    {
        const u32x2 v = ((const u32x2*)slot)[lane];
        r.w[0] = v.x; r.w[1] = v.y;
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the counter completes in issue order); n > 8 waits for <= 8 (stricter)
DEV void wait_vmcnt_dyn(int n)
{
    switch (n)
    {
        case 0: wait_vmcnt_le<0>(); break;
        case 1: wait_vmcnt_le<1>(); break;
        case 2: wait_vmcnt_le<2>(); break;
        case 3: wait_vmcnt_le<3>(); break;
        case 4: wait_vmcnt_le<4>(); break;
        case 5: wait_vmcnt_le<5>(); break;
        case 6: wait_vmcnt_le<6>(); break;
        case 7: wait_vmcnt_le<7>(); break;
        default: wait_vmcnt_le<8>(); break;
    }
}

// ---- streaming a segment through a register ring (qgemv_common.h: stream_items) ----------------------------------------------
#ifndef RING_DEPTH4
#define RING_DEPTH4 4
#endif
#define RING_DEPTH 4
template <int BITS> struct FlatDepth { static constexpr int v = BITS == 8 ? 3 : (BITS <= 4 ? RING_DEPTH4 : RING_DEPTH); };
struct Seg { const u32* ptr; int n; int chunk0; int bits; int nvalid; };

template <int BITS, bool GPTQ>
DEV void flat_stream(const Seg& s, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if (s.nvalid != 4)
    {
        LaneWords<BITS> w;
        load_lane_words<BITS>(s.ptr, lane, w);
        gemv_super<BITS, GPTQ, false>(w, ph, s.chunk0, s.nvalid, lane, acc);
        return;
    }
    LaneWords<BITS> b[FlatDepth<BITS>::v];
    stream_items<BITS, GPTQ, FlatDepth<BITS>::v>(s.ptr, s.n, s.chunk0, ph, lane, acc, b, false);
}

template <bool GPTQ>
DEV void flat_stream_any(const Seg& s, const PhaseCtx& ph, int lane, f32x4& acc)
{
    if constexpr (GPTQ) flat_stream<4, true>(s, ph, lane, acc);
    else switch (s.bits)
    {
        case 4: flat_stream<4, false>(s, ph, lane, acc); break;
        case 8: flat_stream<8, false>(s, ph, lane, acc); break;
        case 6: flat_stream<6, false>(s, ph, lane, acc); break;
        case 5: flat_stream<5, false>(s, ph, lane, acc); break;
        case 3: flat_stream<3, false>(s, ph, lane, acc); break;
        default: flat_stream<2, false>(s, ph, lane, acc); break;
    }
}

template <bool GPTQ>
KERNEL void __launch_bounds__(1024) qgemv_flat_kernel(const FlatArgs args)
{
    DYN_SMEM(smem);
    FTRACE(0);
    const int lane = lane_id();
    const int wv = uniform(wave_id());
    const int t = tid();
    const int nw = FLAT_WAVES;
    const int M = args.M, K = args.K;
    const int oct = K >> 3;
    const int b = bid_x();

    // ---- this workgroup's units, this wave's slot and K slice: a handful of scalar ops -----------------------------------
    const int cls = b < args.units_rem ? 1 : 0;
    const int cnt = args.units_lo + cls;                                        // units (pairs) of this workgroup
    const int start = b * args.units_lo + (b < args.units_rem ? b : args.units_rem);
    const int nslots = args.pair ? 2 * cnt : cnt;
    const int S = cls ? args.S[1] : args.S[0];
    const int slot = (wv * (cls ? args.slot_mul[1] : args.slot_mul[0])) >> 8;  // floor(wv / S)
    const int r = wv - slot * S;
    const int smul = cls ? args.slice_mul[1] : args.slice_mul[0];               // floor(x / Sm) = (x * smul) >> 16
    const bool active = slot < nslots;
    // slot (relative) -> matrix, tile
    auto slot_mat = [&](int s) -> int {
        if (args.pair) return s & 1;
        const int u = start + s;
        return (u >= args.hot[1].unit0 ? 1 : 0) + (u >= args.hot[2].unit0 ? 1 : 0) + (u >= args.hot[3].unit0 ? 1 : 0);
    };
    auto slot_tile = [&](int s, int j) -> int {
        if (args.pair) return start + (s >> 1);
        const int u0j = j == 0 ? args.hot[0].unit0 : j == 1 ? args.hot[1].unit0 : j == 2 ? args.hot[2].unit0 : args.hot[3].unit0;
        return start + s - u0j;
    };
    // first tile of matrix j in this workgroup and how many (its table piece)
    auto mat_piece = [&](int j, int& tile0, int& ntile) {
        const FpMatHot& h = args.hot[j];
        if (args.pair) { tile0 = start; ntile = cnt; return; }
        const int lo = start > h.unit0 ? start : h.unit0;
        const int e = h.unit0 + h.n_tiles, hi = start + cnt < e ? start + cnt : e;
        tile0 = lo - h.unit0; ntile = hi > lo ? hi - lo : 0;
    };

    f16* a_lds = (f16*)smem;
    float* red = (float*)(smem + args.lds_red_off);

    // the wave's slice of its slot's main run
    const int mj = active ? slot_mat(slot) : 0;
    const int mtile = active ? slot_tile(slot, mj) : 0;
    // the matrix' minor-run records: requested now (one 64-byte scalar load), used after the ring fill has been issued
    const u32x4* cold_blk = (const u32x4*)&args.cold[mj];
    const u32x4 cb0 = cold_blk[0], cb1 = cold_blk[1], cb2 = cold_blk[2], cb3 = cold_blk[3];
    const int n_minor = active ? (mj == 0 ? args.hot[0].pad : mj == 1 ? args.hot[1].pad : mj == 2 ? args.hot[2].pad : args.hot[3].pad) : 0;
    Seg first; first.n = 0; first.ptr = nullptr; first.bits = 4; first.nvalid = 4; first.chunk0 = 0;
    if (active)
    {
        const u32* mb = mj == 0 ? args.hot[0].main_base : mj == 1 ? args.hot[1].main_base : mj == 2 ? args.hot[2].main_base : args.hot[3].main_base;
        const u32 ms = mj == 0 ? args.hot[0].main_stride : mj == 1 ? args.hot[1].main_stride : mj == 2 ? args.hot[2].main_stride : args.hot[3].main_stride;
        const int mF = mj == 0 ? args.hot[0].main_F : mj == 1 ? args.hot[1].main_F : mj == 2 ? args.hot[2].main_F : args.hot[3].main_F;
        const int mm = mj == 0 ? args.hot[0].main_meta : mj == 1 ? args.hot[1].main_meta : mj == 2 ? args.hot[2].main_meta : args.hot[3].main_meta;
        const int i0 = (r * mF * smul) >> 16, i1 = ((r + 1) * mF * smul) >> 16;
        first.bits = mm & 0xFF;
        first.n = i1 - i0;
        first.chunk0 = (mm >> 8) + 4 * i0;
        first.ptr = mb + (size_t)mtile * ms + (size_t)i0 * (64u * first.bits);
    }
    const bool early = first.n > 0;
    FTRACE(12);

    // ---- issue: prologue inputs -------------------------------------------------------------------------------------------
    f16x8 xr = {0, 0, 0, 0, 0, 0, 0, 0}, wr = {0, 0, 0, 0, 0, 0, 0, 0};
    // rows x octets <= one per thread and whole waves per row: each thread owns one octet, no loop
    const bool one_pass = args.a_mode == A_NORM_PRE && M * oct <= nw * 64 && (oct & 63) == 0 && args.npart <= 256;
    const int row1 = one_pass ? t / oct : 0, oc1 = one_pass ? t - row1 * oct : 0;
    float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (args.a_mode == A_DIRECT)
    {
        const f16* a = args.a; const int lda = args.lda;
        for (int rr = 0; rr < M; rr++)
            dma_units16([&](int u) { return (const void*)(a + (size_t)rr * lda + (size_t)u * 8); }, a_lds + (size_t)rr * args.a_stride, oct, wv, nw, lane, rr % nw);
    }
    else if (one_pass && t < M * oct)
    {
        xr = *(const f16x8*)(args.a + (size_t)row1 * args.lda + (size_t)oc1 * 8);
        wr = *(const f16x8*)(args.norm_w + (size_t)oc1 * 8);
        // partial sums of squares: every wave reduces its row's partials itself (fixed order), no LDS round trip
        const float* sp = args.ss + (size_t)row1 * args.npart;
        #pragma unroll
        for (int i = 0; i < 4; i++) if (lane + 64 * i < args.npart) ssp[i] = sp[lane + 64 * i];
    }
    // tables: matrix j = wave & 3 is served by the four waves with that residue; a matrix' scale tables of this
    // workgroup's tiles are one contiguous piece of its [tile][G][16] table, the chunk -> group map sits behind the table
    {
        const int j = wv & 3, sub = wv >> 2;
        if (j < args.n_mats)
        {
            const FpMatHot& h = args.hot[j];
            int tile0, ntile;
            mat_piece(j, tile0, ntile);
            if (ntile > 0)
            {
                const int units = ntile * h.G * 2;
                const f16* st = h.sc_tab + (size_t)tile0 * h.G * 16;
                u8* dst = smem + args.lds_sc_off + (size_t)j * args.sc_piece;
                for (int base = sub * 64; base < units; base += 4 * 64)
                    if (base + lane < units) dma_to_lds16(st + (size_t)(base + lane) * 8, dst + (size_t)base * 16);
                if constexpr (GPTQ)
                {
                    const f16* zt = h.zp_tab + (size_t)tile0 * h.G * 16;
                    u8* zd = smem + args.lds_zp_off + (size_t)j * args.sc_piece;
                    for (int base = sub * 64; base < units; base += 4 * 64)
                        if (base + lane < units) dma_to_lds16(zt + (size_t)(base + lane) * 8, zd + (size_t)base * 16);
                }
                if (sub == 3)
                {
                    const u8* cg = (const u8*)(h.sc_tab + (size_t)h.n_tiles * h.G * 16);
                    for (int base = 0; base < h.cg_units; base += 64)
                        if (base + lane < h.cg_units) dma_to_lds16(cg + (size_t)(base + lane) * 16, smem + args.lds_cg_off + (size_t)j * args.cg_stride + (size_t)base * 16);
                }
            }
        }
    }
    FTRACE(1);

    // ---- prologue arithmetic: activations into LDS (runs between the early and the late part of the ring fill) -----------
    auto prologue = [&]() {
        if (args.a_mode != A_NORM_PRE) return;
        if (one_pass)
        {
            if (t < M * oct)
            {
                float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
                ss = wave_allreduce_add(ss);
                const float rms = fast_rsqrt(ss * (1.0f / (float)K) + args.eps);
                f16x8 v;
                #pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    const float f = fmaxf(-65504.0f, fminf((float)xr[e], 65504.0f));
                    v[e] = (f16)((f * (float)wr[e]) * rms);
                }
                *(f16x8*)(a_lds + (size_t)row1 * args.a_stride + (size_t)oc1 * 8) = v;
            }
            return;
        }
        // many rows: row by row, every wave reduces the row's partials itself
        for (int rr = 0; rr < M; rr++)
        {
            float ss = 0.0f;
            const float* sp = args.ss + (size_t)rr * args.npart;
            for (int i = lane; i < args.npart; i += 64) ss += sp[i];
            ss = wave_allreduce_add(ss);
            const float rms = fast_rsqrt(ss * (1.0f / (float)K) + args.eps);
            for (int o = t; o < oct; o += nw * 64)
            {
                const f16x8 x = *(const f16x8*)(args.a + (size_t)rr * args.lda + (size_t)o * 8);
                const f16x8 w = *(const f16x8*)(args.norm_w + (size_t)o * 8);
                f16x8 v;
                #pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    const float f = fmaxf(-65504.0f, fminf((float)x[e], 65504.0f));
                    v[e] = (f16)((f * (float)w[e]) * rms);
                }
                *(f16x8*)(a_lds + (size_t)rr * args.a_stride + (size_t)o * 8) = v;
            }
        }
    };

    // tables of this wave's slot in LDS
    PhaseCtx ph;
    ph.a_lds = a_lds; ph.a_stride = args.a_stride; ph.M = M; ph.phase_k0 = 0;
    auto set_tables = [&]() {
        int tile0, ntile;
        mat_piece(mj, tile0, ntile);
        const int G = mj == 0 ? args.hot[0].G : mj == 1 ? args.hot[1].G : mj == 2 ? args.hot[2].G : args.hot[3].G;
        const size_t off = (size_t)mj * args.sc_piece + (size_t)(mtile - tile0) * G * 32;
        ph.sc_lds = (const f16*)(smem + args.lds_sc_off + off);
        ph.zp_lds = (const f16*)(smem + args.lds_zp_off + off);
        ph.cg_lds = (const u16*)(smem + args.lds_cg_off + (size_t)mj * args.cg_stride);
    };

    // ---- the minor runs (the matrix' other bit widths, partial super-chunks): a few percent of a tile.  Streamed after the
    // main slice through registers they would start cold -- a memory round trip on every wave.  Instead every wave copies
    // ITS share of the slot's minor items (item i of run q belongs to wave (i + 3 q) mod S) into a wave-private LDS area by
    // LDS-DMA right behind its ring fill, and decodes them from there after the main slice: no round trip, no registers,
    // no synchronisation (a wave only reads what it copied itself).
    u8* my_minor = smem + args.lds_minor_off + (size_t)wv * args.minor_wave_bytes;
    auto for_my_minor_items = [&](auto&& fn) {
        // fn(src, bits, nvalid, chunk): the same walk at issue and at decode time
        const u32* qw = (const u32*)(((u64)cb0.y << 32) | cb0.x);
        const u32* tl = (const u32*)(((u64)cb0.w << 32) | cb0.z);
        auto one_run = [&](int q, u32 base_off, u32 tile_stride, u32 n_chunk, u32 meta) {
            const int bits = (int)(meta & 0xFFu), nvalid = (int)((meta >> 8) & 0xFFu);
            const int F = (int)(n_chunk & 0xFFFFu), chunk0 = (int)(n_chunk >> 16);
            const u32* base = (((meta >> 16) & 1u) ? tl : qw) + base_off + (size_t)mtile * tile_stride;
            int i = r - 3 * q; while (i < 0) i += S;                  // first item of run q that is mine
            #pragma nounroll
            for (; i < F; i += S) fn(base + (size_t)i * (64u * bits), bits, nvalid, chunk0 + 4 * i);
        };
        if (n_minor > 0) one_run(0, cb1.x, cb1.y, cb1.z, cb1.w);
        if (n_minor > 1) one_run(1, cb2.x, cb2.y, cb2.z, cb2.w);
        if (n_minor > 2) one_run(2, cb3.x, cb3.y, cb3.z, cb3.w);
        #pragma nounroll
        for (int q = 3; q < n_minor; q++)                               // rare: more than three minor runs
        {
            const FpMinor& mn = args.cold[mj].minor[q];
            one_run(q, mn.base_off, mn.tile_stride, mn.n_chunk, mn.meta);
        }
    };
    auto issue_minors = [&]() -> int {
        int n_dma = 0; u32 off = 0;
        for_my_minor_items([&](const u32* src, int bits, int nvalid, int chunk) {
            (void)nvalid; (void)chunk;
            ring_issue(src, my_minor + off, bits, lane);
            n_dma += item_dma_instrs(bits); off += 256u * bits;
        });
        return n_dma;
    };

    // ---- head: ring fill of the main slice (the weight addresses depend on nothing) around the prologue, then the main
    // slice itself.  One copy per bit width so that the ring lives in typed registers from fill to decode.
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto head = [&](auto bits_tag) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr int DD = FlatDepth<BITS>::v;
        constexpr int LPI = (BITS == 8 || BITS == 6 || BITS == 5) ? 2 : (BITS == 3 ? 3 : 1);
        LaneWords<BITS> b[DD];
        ring_fill<BITS, DD, 0, DD>(b, first.ptr, first.n, lane);
        FTRACE(9);
        const int mdma = issue_minors();
        prologue();
        FTRACE(3);
        // the LDS-DMA copies (rows in A_DIRECT mode, tables) were issued before the ring loads: wait for them only
        wait_vmcnt_dyn(DD * LPI + mdma);
        block_sync_lds();
        FTRACE(4);
        set_tables();
        stream_items<BITS, GPTQ, DD>(first.ptr, first.n, first.chunk0, ph, lane, acc, b, true);
    };
    if (!early)
    {
        (void)issue_minors();
        prologue();
        FTRACE(3);
        wait_vmcnt_le<0>();
        block_sync_lds();
        FTRACE(4);
        if (active) set_tables();
    }
    else if constexpr (GPTQ) head(std::integral_constant<int, 4>());
    else switch (first.bits)
    {
        case 4: head(std::integral_constant<int, 4>()); break;
        case 8: head(std::integral_constant<int, 8>()); break;
        case 6: head(std::integral_constant<int, 6>()); break;
        case 5: head(std::integral_constant<int, 5>()); break;
        case 3: head(std::integral_constant<int, 3>()); break;
        default: head(std::integral_constant<int, 2>()); break;
    }

    // ---- the minor items of this wave, from LDS --------------------------------------------------------------------------------
    if (active)
    {
        wait_vmcnt_le<0>();                                               // (landed long ago: issued before the main slice streamed)
        u32 off = 0;
        for_my_minor_items([&](const u32* src, int bits, int nvalid, int chunk) {
            (void)src;
            const u32* slot_ptr = (const u32*)(my_minor + off);
            off += 256u * bits;
            auto consume = [&](auto bits_tag) {
                constexpr int BITS = decltype(bits_tag)::value;
                LaneWords<BITS> w;
                lds_lane_words<BITS>(slot_ptr, lane, w);
                if (nvalid == 4) gemv_super<BITS, GPTQ, true>(w, ph, chunk, 4, lane, acc);
                else             gemv_super<BITS, GPTQ, false>(w, ph, chunk, nvalid, lane, acc);
            };
            if constexpr (GPTQ) consume(std::integral_constant<int, 4>());
            else switch (bits)
            {
                case 4: consume(std::integral_constant<int, 4>()); break;
                case 8: consume(std::integral_constant<int, 8>()); break;
                case 6: consume(std::integral_constant<int, 6>()); break;
                case 5: consume(std::integral_constant<int, 5>()); break;
                case 3: consume(std::integral_constant<int, 3>()); break;
                default: consume(std::integral_constant<int, 2>()); break;
            }
        });
        // this wave's partial sum of its slot
        const int c = lane & 15, j4 = lane >> 4;
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int row = j4 * 4 + q;
            if (row < M) red[(wv * M + row) * 16 + c] = acc[q];
        }
    }
    FTRACE(5);
    block_sync_lds();
    FTRACE(7);

    // ---- combine (fixed order) + epilogue: wave `row` finalises row `row`; lane -> column lane & 15 of output tiles
    // (lane >> 4) + 4 i.  Single-matrix launches (the ones with a residual) take their output pointers from scalars.
    if (wv >= M) { FTRACE(8); return; }
    const int row = wv;
    const int ep_c = lane & 15;
    const int n_out = cnt;
    auto slot_sum = [&](int s) -> float {
        float v = 0.0f;
        for (int w = s * S; w < s * S + S; w++) v += red[(w * M + row) * 16 + ep_c];      // fixed order: deterministic
        return v;
    };
    float sq = 0.0f;
    #pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int o = (lane >> 4) + 4 * i;
        if (o < n_out)
        {
            const int s = args.pair ? 2 * o : o;
            const int j = slot_mat(s);
            const int n = slot_tile(s, j) * 16 + ep_c;
            f16* cbase = j == 0 ? args.cold[0].c : j == 1 ? args.cold[1].c : j == 2 ? args.cold[2].c : args.cold[3].c;
            const u16* cip = j == 0 ? args.cold[0].c_invperm : j == 1 ? args.cold[1].c_invperm : j == 2 ? args.cold[2].c_invperm : args.cold[3].c_invperm;
            const int cld = j == 0 ? args.cold[0].ldc : j == 1 ? args.cold[1].ldc : j == 2 ? args.cold[2].ldc : args.cold[3].ldc;
            f16* cp = cbase + (size_t)row * cld + (cip ? (int)cip[n] : n);
            f16 y;
            if (args.pair)
            {
                float gv = slot_sum(2 * o), uv = slot_sum(2 * o + 1);
                if (args.any_bias)
                {
                    if (args.cold[0].bias) gv += (float)args.cold[0].bias[n];
                    if (args.cold[1].bias) uv += (float)args.cold[1].bias[n];
                }
                y = clamp_h(act_h((f16)gv, args.act_gelu != 0) * (f16)uv);
            }
            else
            {
                float v = slot_sum(o);
                if (args.any_bias)
                {
                    const f16* bp = j == 0 ? args.cold[0].bias : j == 1 ? args.cold[1].bias : j == 2 ? args.cold[2].bias : args.cold[3].bias;
                    if (bp) v += (float)bp[n];
                }
                if (args.c_mode == C_ACCUM) v += (float)*cp;
                y = (f16)v;
            }
            *cp = y;
            if (args.xp_out)
            {
                args.xp_out[(size_t)row * args.ldxp + (args.xp_invperm ? (int)args.xp_invperm[n] : n)] = y;
                const float f = fmaxf(-65504.0f, fminf((float)y, 65504.0f));
                sq = fmaf(f, f, sq);
            }
        }
    }
    if (args.ss_out)
    {
        sq = wave_allreduce_add(sq);
        if (lane == 0) args.ss_out[(size_t)row * args.wgs + b] = sq;
    }
    FTRACE(8);
}

// ---- host --------------------------------------------------------------------------------------------------------------

#ifdef EXL2_TRACE
static u64* g_ftrace_buf = nullptr;
static int g_ftrace_which = 0, g_ftrace_count = 0;
extern "C" void exl2_debug_set_flat_trace(void* p, int which) { g_ftrace_buf = (u64*)p; g_ftrace_which = which; g_ftrace_count = 0; }
#endif

static inline u32 al16(u32 x) { return (x + 15u) & ~15u; }
static inline bool ptr16(const void* p) { return (((size_t)p) & 15) == 0; }

static int flat_num_cus()
{
    static int n = 0;
    if (n <= 0)
    {
        hipDeviceProp_t prop; int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// floor(x / S) == (x * m) >> sh for every x in [0, xmax]?
static bool mul_ok(int S, int m, int sh, int xmax)
{
    for (int x = 0; x <= xmax; x++) if (((x * m) >> sh) != x / S) return false;
    return true;
}

// returns 0 when launched, 1 when the shape is not covered (caller falls back), < 0 on error.  *wgs_out = grid size
// (= the number of partial sums a chain-out launch publishes per row).
int qgemv_flat_launch(const FlatIn& in, void* stream, int* wgs_out)
{
    if (in.n_mats < 1 || in.n_mats > FLAT_MAX_MATS || in.M < 1 || in.M > MAX_GEMV_ROWS) return 1;
    const QMatrix* q0 = in.qm[0];
    const bool gptq = q0->is_gptq;
    const int K = q0->height, M = in.M;
    if (K & 7) return 1;
    if (!ptr16(in.a) || (in.lda & 7)) return 1;
    if (in.a_mode == A_NORM_PRE && (!ptr16(in.norm_w) || !in.ss || in.npart < 1 || in.npart > 256)) return 1;
    if (in.pair && (in.n_mats != 2 || in.qm[0]->width != in.qm[1]->width)) return 1;
    FlatArgs args;
    memset(&args, 0, sizeof(args));
    int units = 0, g_max = 0, f_max = 0, max_bits = 2;
    u32 cg_max = 0;
    for (int j = 0; j < FLAT_MAX_MATS; j++) args.hot[j].unit0 = 0x7fffffff;
    for (int j = 0; j < in.n_mats; j++)
    {
        const QMatrix* qm = in.qm[j];
        const QMatDev& d = qm->dev;
        if (qm->height != K || qm->is_gptq != gptq || d.n_runs <= 0 || !d.sc_tab || (gptq && !d.zp_tab)) return 1;
        const QRun& mr = d.runs[d.main_run];
        const bool has_main = mr.nvalid_last == 4;                // else: no full super-chunk at all (sections < 128 rows)
        FpMatHot& h = args.hot[j];
        h.main_base = (mr.in_tail ? d.tail : d.qw) + mr.base_word; h.main_stride = mr.tile_stride; h.main_F = has_main ? mr.n_super : 0;
        h.main_meta = (int)mr.bits | (((int)mr.k_base >> 5) << 8);
        h.sc_tab = d.sc_tab; h.zp_tab = d.zp_tab;
        h.n_tiles = d.N / TILE_N; h.G = d.G; h.cg_units = (int)(d.pack_units - (d.pack_cg_off >> 4));
        h.unit0 = in.pair ? 0 : units;
        FpMatCold& e = args.cold[j];
        e.qw = d.qw; e.tail = d.tail; e.c = in.c[j]; e.c_invperm = in.c_invperm[j]; e.bias = d.bias; e.ldc = in.ldc[j];
        e.n_minor = 0;
        for (int i = 0; i < d.n_runs; i++)
        {
            const QRun& run = d.runs[i];
            if ((int)run.bits > max_bits) max_bits = run.bits;
            if (run.nvalid_last == 4 && (int)run.n_super > f_max) f_max = run.n_super;
            if (has_main && i == d.main_run) continue;
            if (e.n_minor >= FLAT_MINORS) return 1;                  // more bit-width runs than this path carries
            FpMinor& mn = e.minor[e.n_minor++];
            mn.base_off = run.base_word; mn.tile_stride = run.tile_stride;
            mn.n_chunk = (u32)run.n_super | ((u32)((int)run.k_base >> 5) << 16);
            mn.meta = (u32)run.bits | ((u32)run.nvalid_last << 8) | (run.in_tail ? (1u << 16) : 0u);
        }
        h.pad = e.n_minor;                                          // (hot word: the minor-run count)
        if (d.bias) args.any_bias = 1;
        units += h.n_tiles;
        if (d.G > g_max) g_max = d.G;
        if ((u32)h.cg_units > cg_max) cg_max = (u32)h.cg_units;
    }
    const int ncu = flat_num_cus();
    // workgroups: one per CU; more when a workgroup would own more than 16 slots (pairs: 8)
    const int per_wg_max = in.pair ? FLAT_MAX_SLOTS / 2 : FLAT_MAX_SLOTS;
    const int items = in.pair ? units / 2 : units;
    int wgs = ncu;
    if (items < wgs) wgs = items;
    if ((items + wgs - 1) / wgs > per_wg_max) wgs = (items + per_wg_max - 1) / per_wg_max;
    const char* fw = getenv("EXL2_FLAT_WGS");
    if (fw && atoi(fw) > 0 && (items + atoi(fw) - 1) / atoi(fw) <= per_wg_max && atoi(fw) <= items) wgs = atoi(fw);
    if (in.ss_out && wgs > 256) return 1;
    const int lo = items / wgs, rem = items % wgs;
    const int spu = in.pair ? 2 : 1;
    u32 minor_wave_bytes = 0;
    for (int c = 0; c < 2; c++)
    {
        const int ns = (lo + c) * spu;                                     // slots of a workgroup of this class
        int S = ns > 0 ? FLAT_WAVES / ns : 1;
        if (S < 1) return 1;
        const char* fsp = getenv("EXL2_FLAT_SPLIT");
        if (fsp && atoi(fsp) > 0 && atoi(fsp) * ns <= FLAT_WAVES) S = atoi(fsp);
        args.S[c] = S;
        args.slot_mul[c] = 256 / S + 1;
        args.slice_mul[c] = 65536 / S + 1;
        if (!mul_ok(S, args.slot_mul[c], 8, FLAT_WAVES - 1) || !mul_ok(S, args.slice_mul[c], 16, (S + 1) * (f_max > 0 ? f_max : 1))) return 1;
        if (ns <= 0) continue;
        // bytes of minor items the busiest wave of a slot stages (the kernel's assignment: item i of run q -> wave (i + 3 q) mod S)
        for (int j = 0; j < in.n_mats; j++)
            for (int r = 0; r < S; r++)
            {
                u32 bytes = 0;
                for (int q = 0; q < args.cold[j].n_minor; q++)
                {
                    const FpMinor& mn = args.cold[j].minor[q];
                    const int F = (int)(mn.n_chunk & 0xFFFFu), bits = (int)(mn.meta & 0xFFu);
                    int i = r - 3 * q; while (i < 0) i += S;
                    for (; i < F; i += S) bytes += 256u * bits;
                }
                if (bytes > minor_wave_bytes) minor_wave_bytes = bytes;
            }
    }
    args.a = in.a; args.norm_w = in.norm_w; args.ss = in.ss; args.npart = in.npart; args.lda = in.lda; args.K = K; args.M = M;
    args.a_mode = in.a_mode; args.n_mats = in.n_mats; args.pair = in.pair; args.eps = in.eps;
    args.a_stride = K + 8; args.c_mode = in.c_mode; args.act_gelu = in.act_gelu;
    args.xp_out = in.xp_out; args.xp_invperm = in.xp_invperm; args.ss_out = in.ss_out; args.ldxp = in.ldxp; args.wgs = wgs;
    args.units_lo = lo; args.units_rem = rem;
    u32 total = al16((u32)M * (u32)args.a_stride * 2);
    args.sc_piece = al16((u32)(lo + 1) * (u32)g_max * 32);
    args.lds_sc_off = total; total += args.sc_piece * in.n_mats;
    args.lds_zp_off = total; total += gptq ? args.sc_piece * in.n_mats : 0;
    args.cg_stride = cg_max * 16;
    args.lds_cg_off = total; total += args.cg_stride * in.n_mats;
    args.lds_red_off = total; total += (u32)FLAT_WAVES * M * 16 * 4;
    args.lds_minor_off = total; args.minor_wave_bytes = al16(minor_wave_bytes); total += FLAT_WAVES * args.minor_wave_bytes;
    if (total > 160 * 1024) return 1;
#ifdef EXL2_TRACE
    args.trace = (g_ftrace_buf && g_ftrace_count++ == g_ftrace_which) ? g_ftrace_buf : nullptr;
#endif
    static bool attr = false;
    if (!attr)
    {
        (void)hipFuncSetAttribute((const void*)qgemv_flat_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qgemv_flat_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    dim3 grid((unsigned)wgs, 1, 1), block(FLAT_WAVES * 64, 1, 1);
    if (gptq) LAUNCH((qgemv_flat_kernel<true>), grid, block, total, stream, args);
    else      LAUNCH((qgemv_flat_kernel<false>), grid, block, total, stream, args);
    if (wgs_out) *wgs_out = wgs;
    return 0;
}
