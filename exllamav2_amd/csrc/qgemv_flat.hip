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
#include <mutex>

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

struct FlatHdr
{
    const f16* a;                 // A_DIRECT: rows in packed order [M, lda]; A_NORM_PRE: xp [M, lda]
    const f16* xp_w;              // chain-out: the next consumer's norm weight in its packed order (nullable = 1); see qgemv_flat.h
    const float* ss;              // A_NORM_PRE: partial sums of squares [M, npart]
    f16* xp_out; const u16* xp_invperm; float* ss_out;      // chain-out (nullable): x in the next consumer's order + partials
    const f16* r_weights;         // grouped (MoE) launches: routing weights [M, r_stride], column = group; nullable
    u64* trace;
    int npart, lda, K, M, a_mode, n_mats, pair, a_stride;
    int c_mode, act_gelu, ldxp, any_bias, wgs;
    int units_lo, units_rem;      // a workgroup owns lo units (pairs in pair mode), the first `rem` workgroups one more
    int S[2], slot_mul[2], slice_mul[2];      // per class (lo / lo + 1 units): waves per slot, floor(w / S) and floor(x / S) multipliers
    u32 lds_minor_off, minor_wave_bytes;      // wave-private staging of the minor items
    float eps;
    u32 lds_sc_off, sc_piece, lds_zp_off, lds_cg_off, cg_stride, lds_red_off;
    int r_stride, mul_r;          // grouped launches: row stride of r_weights; multiply the result by the weight (down projection)
    long long a_gstride;          // grouped launches: elements between the inputs of consecutive groups
    const u32* sync_wait; u32* sync_signal; u32* sync_arrive; u32 sync_total;      // overlapped chain (DEP instantiation; chain_sync.h)
};

// one launch = one set of <= 4 fused matrices ...
struct FlatArgs : FlatHdr
{
    FpMatHot hot_[FLAT_MAX_MATS];
    FpMatCold cold_[FLAT_MAX_MATS];
    DEV const FpMatHot& hot(int j) const { return hot_[j]; }
    DEV const FpMatCold& cold(int j) const { return cold_[j]; }
    DEV int group() const { return 0; }
    static constexpr bool grouped = false;
};

// ... or blockIdx.y = group: every group its own set of matrices, all with the same shapes (the experts of a MoE layer:
// replaces the per-expert launch loop of QMoEMLP::forward_, q_mlp.cu:318-402 / moe_mlp.py:255-323).  The group index is
// arithmetic on the block id, so `hot(j)` stays one batch of scalar loads.
#define FLAT_MAX_GROUPS 16
#define FLAT_GROUP_MATS 2
struct FlatGroupArgs : FlatHdr
{
    FpMatHot hot_[FLAT_MAX_GROUPS * FLAT_GROUP_MATS];
    FpMatCold cold_[FLAT_MAX_GROUPS * FLAT_GROUP_MATS];
    DEV const FpMatHot& hot(int j) const { return hot_[bid_y() * FLAT_GROUP_MATS + (j < FLAT_GROUP_MATS ? j : 0)]; }
    DEV const FpMatCold& cold(int j) const { return cold_[bid_y() * FLAT_GROUP_MATS + (j < FLAT_GROUP_MATS ? j : 0)]; }
    DEV int group() const { return bid_y(); }
    static constexpr bool grouped = true;
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

// BIG: some matrix of the launch has a second run of >= 2 items per wave (e.g. 4 / 3-bit halves): those runs stream through
// the register ring after the main slice.  A separate instantiation: the extra streaming code costs the common case
// (one dominant width + a few percent of others) ~10 % through its sheer size (measured).
// DEP: the launch overlaps its predecessor (chain_sync.h): nothing that depends on the predecessor is touched before the
// wait, which sits between the ring fill and the activation loads; outputs are agent-scope stores followed by a signal.
template <bool GPTQ, typename ARGS, bool BIG, bool DEP = false>
KERNEL void __launch_bounds__(1024) qgemv_flat_kernel(const ARGS args)
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

    // grouped launch: the group's routing weights; a group no row is routed to has nothing to do (wave-uniform scalar loads)
    const f16* rw = nullptr;
    if constexpr (ARGS::grouped) rw = args.r_weights ? args.r_weights + args.group() : nullptr;
    if (ARGS::grouped && rw)
    {
        u32 any = 0;
        for (int rr = 0; rr < M; rr++) any |= (u32)as_u16(rw[(size_t)rr * args.r_stride]);
        if (uniform(any) == 0) return;
    }
    const f16* a_in = ARGS::grouped ? args.a + (size_t)args.group() * (size_t)args.a_gstride : args.a;

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
        if (args.n_mats == 1) return 0;
        const int u = start + s;
        return (u >= args.hot(1).unit0 ? 1 : 0) + (u >= args.hot(2).unit0 ? 1 : 0) + (u >= args.hot(3).unit0 ? 1 : 0);
    };
    auto slot_tile = [&](int s, int j) -> int {
        if (args.pair) return start + (s >> 1);
        const int u0j = j == 0 ? args.hot(0).unit0 : j == 1 ? args.hot(1).unit0 : j == 2 ? args.hot(2).unit0 : args.hot(3).unit0;
        return start + s - u0j;
    };
    // first tile of matrix j in this workgroup and how many (its table piece)
    auto mat_piece = [&](int j, int& tile0, int& ntile) {
        const FpMatHot& h = args.hot(j);
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
    const u32x4* cold_blk = (const u32x4*)&args.cold(mj);
    const u32x4 cb0 = cold_blk[0], cb1 = cold_blk[1], cb2 = cold_blk[2], cb3 = cold_blk[3];
    const int n_minor = active ? (mj == 0 ? args.hot(0).pad : mj == 1 ? args.hot(1).pad : mj == 2 ? args.hot(2).pad : args.hot(3).pad) : 0;
    Seg first; first.n = 0; first.ptr = nullptr; first.bits = 4; first.nvalid = 4; first.chunk0 = 0;
    if (active)
    {
        const u32* mb = mj == 0 ? args.hot(0).main_base : mj == 1 ? args.hot(1).main_base : mj == 2 ? args.hot(2).main_base : args.hot(3).main_base;
        const u32 ms = mj == 0 ? args.hot(0).main_stride : mj == 1 ? args.hot(1).main_stride : mj == 2 ? args.hot(2).main_stride : args.hot(3).main_stride;
        const int mF = mj == 0 ? args.hot(0).main_F : mj == 1 ? args.hot(1).main_F : mj == 2 ? args.hot(2).main_F : args.hot(3).main_F;
        const int mm = mj == 0 ? args.hot(0).main_meta : mj == 1 ? args.hot(1).main_meta : mj == 2 ? args.hot(2).main_meta : args.hot(3).main_meta;
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
    // (a macro, not a lambda: in the DEP instantiation it is expanded inside every copy of `head`, where a nested closure
    // would send xr / wr / ssp through memory)
    // (the fields it needs are read from the argument block once, here: every further mention of `args` inside the six copies
    // of `head` counts against the compiler's use limit for keeping a by-value kernel argument out of the stack)
    const int act_mode = args.a_mode, act_lda = args.lda, act_stride = args.a_stride, act_npart = args.npart;
    const f16* const act_a = args.a; const float* const act_ss = args.ss;
    const u32* const dep_wait = DEP ? args.sync_wait : nullptr;
    if constexpr (DEP) if (args.sync_arrive && t == 0) (void)ticket_add_agent(args.sync_arrive, 1u);      // "this workgroup holds its CU"
#define FLAT_ISSUE_ACTS() do { \
        if (act_mode == A_DIRECT) \
        { \
            for (int rr = 0; rr < M; rr++) \
                dma_units16<DEP>([&](int u) { return (const void*)(a_in + (size_t)rr * act_lda + (size_t)u * 8); }, a_lds + (size_t)rr * act_stride, oct, wv, nw, lane, rr % nw); \
        } \
        else if (one_pass && t < M * oct) \
        { \
            xr = DEP ? load_agent_f16x8(act_a + (size_t)row1 * act_lda + (size_t)oc1 * 8) : *(const f16x8*)(act_a + (size_t)row1 * act_lda + (size_t)oc1 * 8); \
            wr = (f16x8){1, 1, 1, 1, 1, 1, 1, 1};      /* the producer multiplied by the norm weight (qgemv_flat.h: A_NORM_PRE) */ \
            /* partial sums of squares: every wave reduces its row's partials itself (fixed order), no LDS round trip */ \
            const float* sp_ = act_ss + (size_t)row1 * act_npart; \
            _Pragma("unroll") \
            for (int i = 0; i < 4; i++) if (lane + 64 * i < act_npart) ssp[i] = DEP ? load_agent_f32(sp_ + lane + 64 * i) : sp_[lane + 64 * i]; \
        } } while (0)
    // DEP: the predecessor's outputs are read only behind the wait (head, below); everything issued here is static data
#define FLAT_AWAIT_INPUTS() do { FTRACE(10); if (dep_wait) { if (wv == 0) sync_wait_go(dep_wait, bid_x()); block_sync_lds(); } FTRACE(11); FLAT_ISSUE_ACTS(); } while (0)
    if constexpr (!DEP) FLAT_ISSUE_ACTS();
    // tables: matrix j = wave & 3 is served by the four waves with that residue; a matrix' scale tables of this
    // workgroup's tiles are one contiguous piece of its [tile][G][16] table, the chunk -> group map sits behind the table
    {
        const int j = wv & 3, sub = wv >> 2;
        if (j < args.n_mats)
        {
            const FpMatHot& h = args.hot(j);
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
                const float rms = 1.0f;                 // (1 / rms(x) multiplies the finished sums: qgemv_flat.h, A_NORM_PRE)
                (void)ssp;
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
            const float rms = 1.0f;                     // (1 / rms(x) multiplies the finished sums)
            for (int o = t; o < oct; o += nw * 64)
            {
                const f16x8 x = DEP ? load_agent_f16x8(args.a + (size_t)rr * args.lda + (size_t)o * 8) : *(const f16x8*)(args.a + (size_t)rr * args.lda + (size_t)o * 8);
                const f16x8 w = {1, 1, 1, 1, 1, 1, 1, 1};      // (the producer multiplied by the norm weight)
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
        const int G = mj == 0 ? args.hot(0).G : mj == 1 ? args.hot(1).G : mj == 2 ? args.hot(2).G : args.hot(3).G;
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
    // a run's record -> fields.  meta bit 17: a BIG run (>= 2 items per wave): it streams through the register ring after the
    // main slice (its cold start is paid once per many items); small runs and partial super-chunks are staged in LDS.
    const u32* cold_qw = (const u32*)(((u64)cb0.y << 32) | cb0.x);
    const u32* cold_tl = (const u32*)(((u64)cb0.w << 32) | cb0.z);
    struct MinorRec { const u32* base; int F, bits, nvalid, chunk0; bool big; };
    auto minor_rec = [&](int q) -> MinorRec {
        u32 base_off, tile_stride, n_chunk, meta;
        if (q < 3)
        {
            base_off = q == 0 ? cb1.x : q == 1 ? cb2.x : cb3.x; tile_stride = q == 0 ? cb1.y : q == 1 ? cb2.y : cb3.y;
            n_chunk = q == 0 ? cb1.z : q == 1 ? cb2.z : cb3.z; meta = q == 0 ? cb1.w : q == 1 ? cb2.w : cb3.w;
        }
        else
        {
            const FpMinor& mn = args.cold(mj).minor[q];                 // rare: more than three minor runs
            base_off = mn.base_off; tile_stride = mn.tile_stride; n_chunk = mn.n_chunk; meta = mn.meta;
        }
        MinorRec m;
        m.base = (((meta >> 16) & 1u) ? cold_tl : cold_qw) + base_off + (size_t)mtile * tile_stride;
        m.F = (int)(n_chunk & 0xFFFFu); m.bits = (int)(meta & 0xFFu); m.nvalid = (int)((meta >> 8) & 0xFFu);
        m.chunk0 = (int)(n_chunk >> 16); m.big = ((meta >> 17) & 1u) != 0;
        return m;
    };
    // (common case, !BIG: the first three records unrolled from registers -- fewest scalar instructions before barrier 1)
    auto for_my_minor_items = [&](auto&& fn) {
        auto one_run = [&](int q, u32 base_off, u32 tile_stride, u32 n_chunk, u32 meta) {
            const int bits = (int)(meta & 0xFFu), nvalid = (int)((meta >> 8) & 0xFFu);
            const int F = (int)(n_chunk & 0xFFFFu), chunk0 = (int)(n_chunk >> 16);
            const u32* base = (((meta >> 16) & 1u) ? cold_tl : cold_qw) + base_off + (size_t)mtile * tile_stride;
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
            const FpMinor& mn = args.cold(mj).minor[q];
            one_run(q, mn.base_off, mn.tile_stride, mn.n_chunk, mn.meta);
        }
    };
    auto issue_minors = [&]() -> int {
        // LDS-DMA of this wave's share of the small runs: item i of run q belongs to wave (i + 3 q) mod S
        int n_dma = 0; u32 off = 0;
        if (!active) return 0;
        if constexpr (!BIG)
        {
            for_my_minor_items([&](const u32* src, int bits, int nvalid, int chunk) {
                (void)nvalid; (void)chunk;
                ring_issue(src, my_minor + off, bits, lane);
                n_dma += item_dma_instrs(bits); off += 256u * bits;
            });
            return n_dma;
        }
        #pragma nounroll
        for (int q = 0; q < n_minor; q++)
        {
            const MinorRec m = minor_rec(q);
            if (m.big) continue;
            int i = r - 3 * q; while (i < 0) i += S;
            #pragma nounroll
            for (; i < m.F; i += S)
            {
                ring_issue(m.base + (size_t)i * (64u * m.bits), my_minor + off, m.bits, lane);
                n_dma += item_dma_instrs(m.bits); off += 256u * m.bits;
            }
        }
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
        if constexpr (DEP) FLAT_AWAIT_INPUTS();
        prologue();
        FTRACE(3);
        // the LDS-DMA copies (rows in A_DIRECT mode, tables) were issued before the ring loads: wait for them only
        // (DEP: the rows were issued last, everything has to have landed)
        if constexpr (DEP) wait_vmcnt_le<0>(); else wait_vmcnt_dyn(DD * LPI + mdma);
        block_sync_lds();
        FTRACE(4);
        set_tables();
        stream_items<BITS, GPTQ, DD>(first.ptr, first.n, first.chunk0, ph, lane, acc, b, true);
    };
    if (!early)
    {
        (void)issue_minors();
        if constexpr (DEP) FLAT_AWAIT_INPUTS();
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

    // ---- the matrix' other BIG runs: slice r of S of each, through the register ring ---------------------------------------------
    if constexpr (BIG) if (active)
    {
        #pragma nounroll
        for (int q = 0; q < n_minor; q++)
        {
            const MinorRec m = minor_rec(q);
            if (!m.big) continue;
            const int i0 = (r * m.F * smul) >> 16, i1 = ((r + 1) * m.F * smul) >> 16;
            if (i1 <= i0) continue;
            Seg sg;
            sg.bits = m.bits; sg.nvalid = m.nvalid; sg.ptr = m.base + (size_t)i0 * (64u * m.bits); sg.n = i1 - i0; sg.chunk0 = m.chunk0 + 4 * i0;
            flat_stream_any<GPTQ>(sg, ph, lane, acc);
        }
    }

    // ---- the small runs' items of this wave, from LDS ---------------------------------------------------------------------------
    if (active)
    {
        wait_vmcnt_le<0>();                                               // (landed long ago: issued before the main slice streamed)
        u32 off = 0;
        auto decode_item = [&](const u32* slot_ptr, int bits, int nvalid, int chunk) {
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
        };
        if constexpr (!BIG)
        {
            for_my_minor_items([&](const u32* src, int bits, int nvalid, int chunk) {
                (void)src;
                const u32* slot_ptr = (const u32*)(my_minor + off);
                off += 256u * bits;
                decode_item(slot_ptr, bits, nvalid, chunk);
            });
        }
        else
        {
            #pragma nounroll
            for (int q = 0; q < n_minor; q++)
            {
                const MinorRec m = minor_rec(q);
                if (m.big) continue;
                int i = r - 3 * q; while (i < 0) i += S;
                #pragma nounroll
                for (; i < m.F; i += S)
                {
                    const u32* slot_ptr = (const u32*)(my_minor + off);
                    off += 256u * m.bits;
                    decode_item(slot_ptr, m.bits, m.nvalid, m.chunk0 + 4 * i);
                }
            }
        }
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
    // A_NORM_PRE: the activations were x * w (qgemv_flat.h); 1 / rms(x) of this row from the producer's partial sums (fixed order)
    float rms_row = 1.0f;
    if (args.a_mode == A_NORM_PRE)
    {
        float ss = 0.0f;
        const float* sp = args.ss + (size_t)row * args.npart;
        for (int i = lane; i < args.npart; i += 64) ss += DEP ? load_agent_f32(sp + i) : sp[i];
        ss = wave_allreduce_add(ss);
        rms_row = fast_rsqrt(ss * (1.0f / (float)K) + args.eps);
    }
    const f16 rwt = (ARGS::grouped && rw) ? rw[(size_t)row * args.r_stride] : (f16)1.0f;
    const bool row_on = !(ARGS::grouped && rw) || as_u16(rwt) != 0;      // rows the group is not routed to are left alone
    #pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int o = (lane >> 4) + 4 * i;
        if (o < n_out && row_on)
        {
            const int s = args.pair ? 2 * o : o;
            const int j = slot_mat(s);
            const int n = slot_tile(s, j) * 16 + ep_c;
            f16* cbase = j == 0 ? args.cold(0).c : j == 1 ? args.cold(1).c : j == 2 ? args.cold(2).c : args.cold(3).c;
            const u16* cip = j == 0 ? args.cold(0).c_invperm : j == 1 ? args.cold(1).c_invperm : j == 2 ? args.cold(2).c_invperm : args.cold(3).c_invperm;
            const int cld = j == 0 ? args.cold(0).ldc : j == 1 ? args.cold(1).ldc : j == 2 ? args.cold(2).ldc : args.cold(3).ldc;
            f16* cp = cbase + (size_t)row * cld + (cip ? (int)cip[n] : n);
            f16 y;
            if (args.pair)
            {
                float gv = slot_sum(2 * o) * rms_row, uv = slot_sum(2 * o + 1) * rms_row;
                if (args.any_bias)
                {
                    if (args.cold(0).bias) gv += (float)args.cold(0).bias[n];
                    if (args.cold(1).bias) uv += (float)args.cold(1).bias[n];
                }
                y = clamp_h(act_h((f16)gv, args.act_gelu != 0) * (f16)uv);
            }
            else
            {
                float v = slot_sum(o) * rms_row;
                if (args.any_bias)
                {
                    const f16* bp = j == 0 ? args.cold(0).bias : j == 1 ? args.cold(1).bias : j == 2 ? args.cold(2).bias : args.cold(3).bias;
                    if (bp) v += (float)bp[n];
                }
                if (ARGS::grouped && args.mul_r) v *= (float)rwt;
                if (args.c_mode == C_ACCUM) v += DEP ? (float)load_agent_f16(cp) : (float)*cp;
                y = (f16)v;
            }
            if constexpr (DEP) store_agent_f16(cp, y); else *cp = y;
            if (args.xp_out)
            {
                const int xi = args.xp_invperm ? (int)args.xp_invperm[n] : n;
                f16* xo = args.xp_out + (size_t)row * args.ldxp + xi;
                const float f = fmaxf(-65504.0f, fminf((float)y, 65504.0f));
                const f16 yw = args.xp_w ? (f16)fmaxf(-65504.0f, fminf(f * (float)args.xp_w[xi], 65504.0f)) : y;      // x * the consumer's norm weight
                if constexpr (DEP) store_agent_f16(xo, yw); else *xo = yw;
                sq = fmaf(f, f, sq);
            }
        }
    }
    if (args.ss_out)
    {
        sq = wave_allreduce_add(sq);
        if (lane == 0) { if constexpr (DEP) store_agent_f32(args.ss_out + (size_t)row * args.wgs + b, sq); else args.ss_out[(size_t)row * args.wgs + b] = sq; }
    }
    if constexpr (DEP) if (args.sync_signal) sync_arrive_publish(args.sync_signal, args.sync_total, args.sync_wait);    // wgs * M arrivals
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
    static int n[EXL2_MAX_DEVICES] = {0};
    const int dev = exl2_current_device();
    if (n[dev] <= 0)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
        if (n[dev] <= 0) n[dev] = 256;
    }
    return n[dev];
}

// floor(x / S) == (x * m) >> sh for every x in [0, xmax]?
static bool mul_ok(int S, int m, int sh, int xmax)
{
    for (int x = 0; x <= xmax; x++) if (((x * m) >> sh) != x / S) return false;
    return true;
}

// Fills the header and the per-matrix records of ONE set of fused matrices; `hot` / `cold` point at the set's records.
// f_max / g_max / cg_max / max_minor accumulate over sets (a grouped launch plans every group into one header).
struct FlatPlanAcc { int units, g_max, f_max; u32 cg_max; bool any_bias; };

static int flat_fill_mats(const FlatIn& in, FpMatHot* hot, FpMatCold* cold, int n_slots_hot, FlatPlanAcc& acc)
{
    const QMatrix* q0 = in.qm[0];
    const bool gptq = q0->is_gptq;
    const int K = q0->height;
    int units = 0;
    for (int j = 0; j < n_slots_hot; j++) hot[j].unit0 = 0x7fffffff;
    for (int j = 0; j < in.n_mats; j++)
    {
        const QMatrix* qm = in.qm[j];
        const QMatDev& d = qm->dev;
        if (qm->height != K || qm->is_gptq != gptq || d.n_runs <= 0 || !d.sc_tab || (gptq && !d.zp_tab)) return 1;
        const QRun& mr = d.runs[d.main_run];
        const bool has_main = mr.nvalid_last == 4;                // else: no full super-chunk at all (sections < 128 rows)
        FpMatHot& h = hot[j];
        h.main_base = (mr.in_tail ? d.tail : d.qw) + mr.base_word; h.main_stride = mr.tile_stride; h.main_F = has_main ? mr.n_super : 0;
        h.main_meta = (int)mr.bits | (((int)mr.k_base >> 5) << 8);
        h.sc_tab = d.sc_tab; h.zp_tab = d.zp_tab;
        h.n_tiles = d.N / TILE_N; h.G = d.G; h.cg_units = (int)(d.pack_units - (d.pack_cg_off >> 4));
        h.unit0 = in.pair ? 0 : units;
        FpMatCold& e = cold[j];
        e.qw = d.qw; e.tail = d.tail; e.c = in.c[j]; e.c_invperm = in.c_invperm[j]; e.bias = d.bias; e.ldc = in.ldc[j];
        e.n_minor = 0;
        for (int i = 0; i < d.n_runs; i++)
        {
            const QRun& run = d.runs[i];
            if (run.nvalid_last == 4 && (int)run.n_super > acc.f_max) acc.f_max = run.n_super;
            if (has_main && i == d.main_run) continue;
            if (e.n_minor >= FLAT_MINORS) return 1;                  // more bit-width runs than this path carries
            FpMinor& mn = e.minor[e.n_minor++];
            mn.base_off = run.base_word; mn.tile_stride = run.tile_stride;
            mn.n_chunk = (u32)run.n_super | ((u32)((int)run.k_base >> 5) << 16);
            mn.meta = (u32)run.bits | ((u32)run.nvalid_last << 8) | (run.in_tail ? (1u << 16) : 0u);
            // a full run with >= 2 items for every wave even at the finest split (16 waves per slot): register ring, not LDS staging
            if (run.nvalid_last == 4 && (int)run.n_super >= 2 * FLAT_WAVES) mn.meta |= 1u << 17;
        }
        h.pad = e.n_minor;                                          // (hot word: the minor-run count)
        if (d.bias) acc.any_bias = true;
        units += h.n_tiles;
        if (d.G > acc.g_max) acc.g_max = d.G;
        if ((u32)h.cg_units > acc.cg_max) acc.cg_max = (u32)h.cg_units;
    }
    acc.units = units;
    return 0;
}

// bytes of minor items the busiest wave of a slot stages (the kernel's assignment: item i of run q -> wave (i + 3 q) mod S)
static u32 flat_minor_bytes(const FpMatCold* cold, int n_mats, int S)
{
    u32 worst = 0;
    for (int j = 0; j < n_mats; j++)
        for (int r = 0; r < S; r++)
        {
            u32 bytes = 0;
            for (int q = 0; q < cold[j].n_minor; q++)
            {
                const FpMinor& mn = cold[j].minor[q];
                if ((mn.meta >> 17) & 1u) continue;                       // big run: register ring, not staged
                const int F = (int)(mn.n_chunk & 0xFFFFu), bits = (int)(mn.meta & 0xFFu);
                int i = r - 3 * q; while (i < 0) i += S;
                for (; i < F; i += S) bytes += 256u * bits;
            }
            if (bytes > worst) worst = bytes;
        }
    return worst;
}

// the work split + LDS layout shared by all sets of a launch; returns the dynamic LDS size (0: not covered)
static u32 flat_plan(FlatHdr& hdr, const FlatIn& in, const FlatPlanAcc& acc, u32 minor_bytes_of_S(int, void*), void* ctx,
                     bool chain_out, int* wgs_out)
{
    const int K = in.qm[0]->height, M = in.M;
    const bool gptq = in.qm[0]->is_gptq;
    const int ncu = flat_num_cus();
    // workgroups: one per CU; more when a workgroup would own more than 16 slots (pairs: 8)
    const int per_wg_max = in.pair ? FLAT_MAX_SLOTS / 2 : FLAT_MAX_SLOTS;
    const int items = in.pair ? acc.units / 2 : acc.units;
    int wgs = ncu;
    if (items < wgs) wgs = items;
    if ((items + wgs - 1) / wgs > per_wg_max) wgs = (items + per_wg_max - 1) / per_wg_max;
    const char* fw = getenv("EXL2_FLAT_WGS");
    if (fw && atoi(fw) > 0 && (items + atoi(fw) - 1) / atoi(fw) <= per_wg_max && atoi(fw) <= items) wgs = atoi(fw);
    if (chain_out && wgs > 256) return 0;
    const int lo = items / wgs, rem = items % wgs;
    const int spu = in.pair ? 2 : 1;
    u32 minor_wave_bytes = 0;
    for (int c = 0; c < 2; c++)
    {
        const int ns = (lo + c) * spu;                                     // slots of a workgroup of this class
        int S = ns > 0 ? FLAT_WAVES / ns : 1;
        if (S < 1) return 0;
        const char* fsp = getenv("EXL2_FLAT_SPLIT");
        if (fsp && atoi(fsp) > 0 && atoi(fsp) * ns <= FLAT_WAVES) S = atoi(fsp);
        hdr.S[c] = S;
        hdr.slot_mul[c] = 256 / S + 1;
        hdr.slice_mul[c] = 65536 / S + 1;
        if (!mul_ok(S, hdr.slot_mul[c], 8, FLAT_WAVES - 1) || !mul_ok(S, hdr.slice_mul[c], 16, (S + 1) * (acc.f_max > 0 ? acc.f_max : 1))) return 0;
        if (ns <= 0) continue;
        const u32 b = minor_bytes_of_S(S, ctx);
        if (b > minor_wave_bytes) minor_wave_bytes = b;
    }
    hdr.a = in.a; hdr.xp_w = in.xp_w; hdr.ss = in.ss; hdr.npart = in.npart; hdr.lda = in.lda; hdr.K = K; hdr.M = M;
    hdr.a_mode = in.a_mode; hdr.n_mats = in.n_mats; hdr.pair = in.pair; hdr.eps = in.eps;
    hdr.a_stride = K + 8; hdr.c_mode = in.c_mode; hdr.act_gelu = in.act_gelu; hdr.any_bias = acc.any_bias ? 1 : 0;
    hdr.xp_out = in.xp_out; hdr.xp_invperm = in.xp_invperm; hdr.ss_out = in.ss_out; hdr.ldxp = in.ldxp; hdr.wgs = wgs;
    hdr.units_lo = lo; hdr.units_rem = rem;
    u32 total = al16((u32)M * (u32)hdr.a_stride * 2);
    hdr.sc_piece = al16((u32)(lo + 1) * (u32)acc.g_max * 32);
    hdr.lds_sc_off = total; total += hdr.sc_piece * in.n_mats;
    hdr.lds_zp_off = total; total += gptq ? hdr.sc_piece * in.n_mats : 0;
    hdr.cg_stride = acc.cg_max * 16;
    hdr.lds_cg_off = total; total += hdr.cg_stride * in.n_mats;
    hdr.lds_red_off = total; total += (u32)FLAT_WAVES * M * 16 * 4;
    hdr.lds_minor_off = total; hdr.minor_wave_bytes = al16(minor_wave_bytes); total += FLAT_WAVES * hdr.minor_wave_bytes;
    if (total > 160 * 1024) return 0;
#ifdef EXL2_TRACE
    // (two consecutive launches are stamped: producer + consumer of an overlapped hand-off; second buffer behind the first)
    {
        const int idx = g_ftrace_count++;
        hdr.trace = (g_ftrace_buf && (idx == g_ftrace_which || idx == g_ftrace_which + 1)) ? g_ftrace_buf + (size_t)(idx - g_ftrace_which) * 256 * 16 * 16 : nullptr;
    }
#endif
    if (wgs_out) *wgs_out = wgs;
    return total;
}

static bool flat_in_ok(const FlatIn& in)
{
    if (in.n_mats < 1 || in.n_mats > FLAT_MAX_MATS || in.M < 1 || in.M > MAX_GEMV_ROWS) return false;
    if (in.qm[0]->height & 7) return false;
    if (!ptr16(in.a) || (in.lda & 7)) return false;
    if (in.a_mode == A_NORM_PRE && (!in.ss || in.npart < 1 || in.npart > 256)) return false;
    if (in.pair && (in.n_mats != 2 || in.qm[0]->width != in.qm[1]->width)) return false;
    return true;
}

static void flat_attrs()
{
    static bool attr[EXL2_MAX_DEVICES] = {false};
    if (!exl2_first_on_device(attr)) return;
#define FLAT_ATTR(...) (void)hipFuncSetAttribute((const void*)qgemv_flat_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    FLAT_ATTR(false, FlatArgs, false); FLAT_ATTR(false, FlatArgs, true); FLAT_ATTR(true, FlatArgs, false);
    FLAT_ATTR(false, FlatGroupArgs, false); FLAT_ATTR(false, FlatGroupArgs, true); FLAT_ATTR(true, FlatGroupArgs, false);
    FLAT_ATTR(false, FlatArgs, false, true); FLAT_ATTR(false, FlatArgs, true, true); FLAT_ATTR(true, FlatArgs, false, true);
#undef FLAT_ATTR
}

static bool flat_any_big(const FpMatCold* cold, int n)
{
    for (int j = 0; j < n; j++) for (int q = 0; q < cold[j].n_minor; q++) if ((cold[j].minor[q].meta >> 17) & 1u) return true;
    return false;
}

// returns 0 when launched, 1 when the shape is not covered (caller falls back), < 0 on error.  *wgs_out = grid size
// (= the number of partial sums a chain-out launch publishes per row).
int qgemv_flat_launch(const FlatIn& in, void* stream, int* wgs_out)
{
    if (!flat_in_ok(in) || in.a_tiled || in.c_tiled || in.xp_tiled) return 1;      // (the tiled layouts are the lean kernel's)
    FlatArgs args;
    memset(&args, 0, sizeof(args));
    FlatPlanAcc acc; memset(&acc, 0, sizeof(acc));
    if (flat_fill_mats(in, args.hot_, args.cold_, FLAT_MAX_MATS, acc)) return 1;
    struct Ctx { const FpMatCold* cold; int n; } ctx = {args.cold_, in.n_mats};
    int wgs = 0;
    const u32 lds = flat_plan(args, in, acc, [](int S, void* c) { return flat_minor_bytes(((Ctx*)c)->cold, ((Ctx*)c)->n, S); }, &ctx,
                              in.ss_out != nullptr, &wgs);
    if (!lds) return 1;
    flat_attrs();
    dim3 grid((unsigned)wgs, 1, 1), block(FLAT_WAVES * 64, 1, 1);
    const bool big = flat_any_big(args.cold_, in.n_mats);
    if (in.sync_wait || in.sync_signal || in.sync_arrive)
    {
        args.sync_wait = in.sync_wait; args.sync_signal = in.sync_signal; args.sync_total = (u32)wgs * (u32)in.M;
        args.sync_arrive = in.sync_arrive;
        if (in.qm[0]->is_gptq) LAUNCH((qgemv_flat_kernel<true, FlatArgs, false, true>), grid, block, lds, stream, args);
        else if (big)          LAUNCH((qgemv_flat_kernel<false, FlatArgs, true, true>), grid, block, lds, stream, args);
        else                   LAUNCH((qgemv_flat_kernel<false, FlatArgs, false, true>), grid, block, lds, stream, args);
    }
    else if (in.qm[0]->is_gptq) LAUNCH((qgemv_flat_kernel<true, FlatArgs, false>), grid, block, lds, stream, args);      // (GPTQ: one run)
    else if (big)          LAUNCH((qgemv_flat_kernel<false, FlatArgs, true>), grid, block, lds, stream, args);
    else                   LAUNCH((qgemv_flat_kernel<false, FlatArgs, false>), grid, block, lds, stream, args);
    if (wgs_out) *wgs_out = wgs;
    return 0;
}

// Grouped launch: `n_groups` sets of matrices of identical shapes (the experts of a MoE layer) as blockIdx.y of ONE launch.
// Group g reads its input rows at a + g * a_gstride, writes through its own c pointers, and is weighted by column g of
// r_weights [M, r_stride]: a group none of whose rows carries a weight exits at entry (q_gemm_kernel.cuh:189-200), rows
// with a zero weight are not written; mul_r multiplies the result by the weight.
int qgemv_flat_group_launch(const FlatIn* ins, int n_groups, const f16* r_weights, int r_stride, int mul_r, long long a_gstride,
                            void* stream)
{
    if (n_groups < 1 || n_groups > FLAT_MAX_GROUPS || !r_weights) return 1;
    const FlatIn& in0 = ins[0];
    if (in0.n_mats > FLAT_GROUP_MATS || in0.xp_out || in0.ss_out || in0.c_mode != C_STORE) return 1;
    static FlatGroupArgs args;                                        // (large: keep it off the stack; launches copy it)
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    memset(&args, 0, sizeof(args));
    FlatPlanAcc acc; memset(&acc, 0, sizeof(acc));
    int units0 = -1;
    for (int g = 0; g < n_groups; g++)
    {
        const FlatIn& in = ins[g];
        if (!flat_in_ok(in) || in.n_mats != in0.n_mats || in.pair != in0.pair || in.M != in0.M || in.a_mode != A_DIRECT ||
            in.qm[0]->height != in0.qm[0]->height || in.qm[0]->is_gptq != in0.qm[0]->is_gptq) return 1;
        if (flat_fill_mats(in, args.hot_ + g * FLAT_GROUP_MATS, args.cold_ + g * FLAT_GROUP_MATS, FLAT_GROUP_MATS, acc)) return 1;
        if (units0 < 0) units0 = acc.units; else if (acc.units != units0) return 1;       // same shapes in every group
    }
    struct Ctx { const FpMatCold* cold; int n; } ctx = {args.cold_, n_groups * FLAT_GROUP_MATS};
    int wgs = 0;
    const u32 lds = flat_plan(args, in0, acc, [](int S, void* c) { return flat_minor_bytes(((Ctx*)c)->cold, ((Ctx*)c)->n, S); }, &ctx, false, &wgs);
    if (!lds) return 1;
    args.r_weights = r_weights; args.r_stride = r_stride; args.mul_r = mul_r; args.a_gstride = a_gstride;
    flat_attrs();
    dim3 grid((unsigned)wgs, (unsigned)n_groups, 1), block(FLAT_WAVES * 64, 1, 1);
    if (in0.qm[0]->is_gptq) LAUNCH((qgemv_flat_kernel<true, FlatGroupArgs, false>), grid, block, lds, stream, args);
    else if (flat_any_big(args.cold_, n_groups * FLAT_GROUP_MATS)) LAUNCH((qgemv_flat_kernel<false, FlatGroupArgs, true>), grid, block, lds, stream, args);
    else                    LAUNCH((qgemv_flat_kernel<false, FlatGroupArgs, false>), grid, block, lds, stream, args);
    return 0;
}
