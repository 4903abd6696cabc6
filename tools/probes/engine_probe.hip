// engine_probe.hip -- round 6, gates (b) and (c) of the round-5 review: what does a decode layer cost as ONE persistent launch
// built like the hardware guide's weight-streaming engine (MI355X_MICROARCH.md "engine-vs-launches", "prefetch-credit",
// "allgather") around THIS library's 4-bit decode?
//
//   * one workgroup per CU = 1 LOADER wave + NC CONSUMER waves;
//   * the loader walks the CU's share of the whole step's weights in model order (o, gate|up, down, q|k|v of every layer: the tiles
//     c, c + 256, ... of each launch of the chained decode path) and copies it 1 KB item by 1 KB item into an LDS ring with
//     `buffer_load_dwordx4 ... lds` (non-temporal), D copies in flight, never waiting for anything but ring space: the HBM stream
//     runs THROUGH the module boundaries;
//   * consumer w takes the items w, w + NC, ... of the ring (real decode: qlayout.h dequant_super<4> + 4 MFMAs against the
//     activations in LDS), partial sums of a tile meet in LDS, the tile's 16 outputs are published as 8-byte {2 halfs, tag}
//     granules (agent-scope stores);
//   * the input vector of the next phase is gathered by the consumers of EVERY CU with agent-scope loads, re-polling the granules
//     whose tag is not the phase's epoch yet (data-tagged all-gather: no flag, no fence), into LDS.
//
// Variants (template): NC consumers, DEC decode on / off (off: items are released unread = gate (b), the pure stream probe),
// HO hand-offs on / off (off: phases follow each other without any cross-CU dependency = stream + decode throughput).
// Every spin is bounded; a give-up is counted and reported.
//
// Build: hipcc --offload-arch=gfx950 -O3 -I exllamav2_amd/csrc tools/probes/engine_probe.hip -o tools/probes/engine_probe
#include "qlayout.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define NPH 4
#define N_CU 256
#define X_MAX_HALFS 11008
#define SPIN_LIMIT (1 << 15)
#define MAX_TILES_CU 6

struct PhaseDesc
{
    u32 w_off;              // byte offset of the phase's weights inside a layer: [tile][F + 1][1 KB] (item 0 of a tile = its scale table)
    int n_tiles, F;         // tiles of the launch (pair phases: 2 * pairs; pair u = tiles u and pairs + u), items per tile
    int pair, k_in;         // k_in: halfs of the input vector
    int out_off;            // the phase's outputs go to granules [out_off / 2 + ...) of vec[(p + 1) % NPH]
};
struct EngArgs
{
    const u8* w; u64 cu_stride;   // CU c's stream: w + c * cu_stride, its items of the whole step back to back in consumption order
    u64* vec[NPH];          // vec[p]: input vector of phase p as granules {lo: 2 halfs, hi: tag}
    const u32* epoch_ctr;   // bumped by a one-thread kernel behind every launch (graph replay needs no new arguments)
    u32* err;               // [0] give-ups of the gather, [1] of the ring waits, [2] of the consumer syncs
    int layers, pad;
    u64* trace;             // optional: [layer * NPH + p][cu][4] 10 ns stamps of consumer 0: gather start / end, items end, publish end
    PhaseDesc ph[NPH];
};

// flags live in LDS and are polled: typed LDS pointers (a generic volatile pointer becomes a flat access + vmcnt(0), which would
// drain the loader's copies in flight)
typedef volatile __attribute__((address_space(3))) u32 lds_vu32;
typedef volatile __attribute__((address_space(3))) float lds_vf32;
DEV u32 lds_read_u32(lds_vu32* p) { return *p; }
DEV void lds_write_u32(lds_vu32* p, u32 v) { *p = v; }

// one fill = 16 consecutive items (16 KB) of the CU's stream into 16 consecutive ring slots: 4 x (one M0 / one scalar offset, four
// copies with immediate offsets 0 / 1 / 2 / 3 KB -- the immediate advances the memory address AND the LDS address)
#define FILL 16
// THIN: what the loader does while its CU gathers (0 nothing, 1 one fill in flight, 2 pause); GW: consumers that sweep
// RAW: the raw 4-bit feed (profiles/history/r05_raw4_experiment.txt): the matrix cores get the un-subtracted codes (1024 + q / 64 + q), the
// constant part enters as the C operand from a per-item correction table made ONCE per phase and CU from the gathered vector
// TB > 0: the tight consumer -- a consumer takes TURNS of TB consecutive items of one tile (turn t of the phase -> consumer t % NC), reads a
// turn's operands together, and publishes the end of its last finished turn (the loader frees the prefix below the minimum)
template <int NC, int RD, int D, bool DEC, bool HO, bool CHECK = false, int THIN = 0, int GW = NC, bool RAW = false, int TB = 0>
KERNEL void __launch_bounds__((NC + 1) * 64) engine_kernel(const EngArgs a)
{
    constexpr int R = RD * FILL;                                       // ring slots (RD fills); D = fills in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u8* const ring = smem;                                              // [R][1 KB]
    f16* const xbuf = (f16*)(smem + R * 1024);                          // [2][X_MAX_HALFS]
    float* const part = (float*)(smem + R * 1024 + 2 * X_MAX_HALFS * 2);     // [MAX_TILES_CU][NC][16]
    lds_vu32* const flags = (lds_vu32*)((u8*)part + MAX_TILES_CU * NC * 64);
    lds_vu32* const f_filled = flags;                                   // items landed in the ring (count)
    lds_vu32* const f_cons = flags + 16;                                // [NC] items consumer w has released
    lds_vu32* const f_gath = flags + 32;                                // [NC] epoch consumer w has gathered its part of
    lds_vu32* const f_done = flags + 48;                                // [NC] epoch consumer w has finished the items of
    float* const corr = (float*)((u8*)part + MAX_TILES_CU * NC * 64 + 512);  // [2][X_MAX_HALFS / 128] per-item corrections (RAW)
    lds_vu32* const f_gflag = flags + 96;
    lds_vu32* const f_gath2 = flags + 112;                              // [NC] RAW: the correction table of the epoch is made                               // the CU is gathering (consumer 0 sets / clears it)
    lds_vf32* const f_ssq = (lds_vf32*)(flags + 64);                    // [2][NC] partial sums of squares of the gathered vector
    const int lane = lane_id(), wv = uniform(wave_id());
    const int cu = bid_x();
    if (threadIdx.x < 128) flags[threadIdx.x] = 0;
    __syncthreads();
    const u32 epoch0 = a.epoch_ctr[0] * 1024u;

    if (wv == 0)
    {
        // ------------------------------------------------------------------ loader
        // total items of this CU in a step (the stream is padded to whole fills)
        u32 total = 0;
        for (int p = 0; p < NPH; p++)
        {
            const PhaseDesc d = a.ph[p];
            const int units = d.pair ? d.n_tiles / 2 : d.n_tiles;
            total += (u32)(((units - cu + N_CU - 1) / N_CU) * (d.pair ? 2 : 1) * (d.F + 1));
        }
        total *= (u32)a.layers;
        const u32 n_fills = (total + FILL - 1) / FILL;
        const u8* const base = a.w + (u64)cu * a.cu_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        u32 freed = 0;                                                  // items known to be released (a prefix of the stream)
        const u32 vo = (u32)lane * 16u;
        for (u32 f = 0; f < n_fills; f++)
        {
            const u32 g = f * FILL;
            if (g + FILL > freed + (u32)R)
            {
                // the ring looks full: look again; if it is, wait -- everything issued lands meanwhile, say so first
                bool drained = false;
                int spins = 0;
                while (true)
                {
                    u32 v = lane < NC ? (TB > 0 ? lds_read_u32(f_cons + lane) : lds_read_u32(f_cons + lane) * NC + lane) : 0xFFFFFFFFu;
                    #pragma unroll
                    for (int m = 1; m < 16; m <<= 1) { const u32 o = shfl_xor_u32(v, m); v = o < v ? o : v; }
                    freed = uniform(v);
                    if (g + FILL <= freed + (u32)R) break;
                    if (!drained)
                    {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) lds_write_u32(f_filled, g);
                        drained = true;
                    }
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.err + 1, 1u); freed = g; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if constexpr (THIN > 0)
            {
                if (uniform(lds_read_u32(f_gflag)))
                {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) lds_write_u32(f_filled, g);
                    if constexpr (THIN == 2)
                    {
                        int spins = 0;
                        while (uniform(lds_read_u32(f_gflag)) && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            const u32 slot = g % (u32)R;
            #pragma unroll
            for (int q = 0; q < 4; q++)
            {
                __attribute__((address_space(3))) void* const dst = (__attribute__((address_space(3))) void*)(ring + (slot + 4 * q) * 1024);
                const u32 so = (g + 4 * q) * 1024u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, so, 0, 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, so, 1024, 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, so, 2048, 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, so, 3072, 2);
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FILL * (D - 1)) : "memory");
            if (f + 2 > (u32)D && lane == 0) lds_write_u32(f_filled, (f + 2 - D) * FILL);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_write_u32(f_filled, n_fills * FILL);
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int w = wv - 1;
    const int c = lane & 15, j = lane >> 4;
    u32 myk = 0, g0 = 0, filled = 0;
    const ZC zc = make_zc((f16)8.0f);
    const ZC z4[4] = {zc, zc, zc, zc};
    u32 m_lo = 0x000F000Fu, m_hi = 0x00F000F0u, k_lo = 0x64006400u, k_hi = 0x54005400u;
    pin_scalar(m_lo); pin_scalar(m_hi); pin_vector(k_lo); pin_vector(k_hi);
    auto sync_consumers = [&](lds_vu32* f, u32 e) {
        if (lane == 0) lds_write_u32(f + w, e);
        int spins = 0;
        while (true)
        {
            u32 v = lane < NC ? lds_read_u32(f + lane) : 0xFFFFFFFFu;
            #pragma unroll
            for (int m = 1; m < 16; m <<= 1) { const u32 o = shfl_xor_u32(v, m); v = o < v ? o : v; }
            if (uniform(v) >= e) break;
            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.err + 2, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    for (int L = 0; L < a.layers; L++)
    for (int p = 0; p < NPH; p++)
    {
        const PhaseDesc d = a.ph[p];
        const u32 e = epoch0 + (u32)(L * NPH + p) + 1u;                 // tag of this phase's INPUT vector; its outputs carry e + 1
        f16* const xb = xbuf + (size_t)(e & 1u) * X_MAX_HALFS;
        float rs = 1.0f;
        if constexpr (HO)
        {
            // ---- gather the input vector: consumer w takes granules [lo, hi), 16 x 64 at a time, re-polling what is not there yet
            const u64 tr0 = realtime_stamp();
            if (THIN > 0 && w == 0 && lane == 0) lds_write_u32(f_gflag, 1u);
            const int gn = d.k_in / 2;
            const int q = ((gn + GW - 1) / GW + 63) & ~63;
            const int lo = w < GW ? w * q : gn, hi = (lo + q < gn) ? lo + q : gn;
            const u64* const src = a.vec[p];
            float ssq = 0.0f;
            const bool first = (L == 0 && p == 0);                      // (the step's input is whatever the buffer holds)
            for (int c0 = lo; c0 < hi; c0 += 1024)
            {
                u32 pend = 0;
                #pragma unroll
                for (int t = 0; t < 16; t++) if (c0 + t * 64 + lane < hi) pend |= 1u << t;
                int spins = 0;
                while (true)
                {
                    u64 v[16];
                    #pragma unroll
                    for (int t = 0; t < 16; t++) if ((pend >> t) & 1u) v[t] = __hip_atomic_load(src + c0 + t * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    #pragma unroll
                    for (int t = 0; t < 16; t++)
                        if (((pend >> t) & 1u) && ((u32)(v[t] >> 32) == e || first))
                        {
                            const u32 dat = (u32)v[t];
                            *(u32*)(xb + 2 * (c0 + t * 64 + lane)) = dat;
                            const f16x2 h = as_h2(dat);
                            ssq += (float)h.x * (float)h.x + (float)h.y * (float)h.y;
                            pend &= ~(1u << t);
                        }
                    if (wave_ballot(pend != 0) == 0) break;
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.err + 0, 1u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            ssq = wave_allreduce_add(ssq);
            if (lane == 0) f_ssq[(e & 1u) * 16 + w] = ssq;
            sync_consumers(f_gath, e);
            if (THIN > 0 && w == 0 && lane == 0) lds_write_u32(f_gflag, 0u);
            if (a.trace && w == 0 && lane == 0) { u64* t = a.trace + ((size_t)(L * NPH + p) * N_CU + cu) * 4; t[0] = tr0; t[1] = realtime_stamp(); }
            if constexpr (RAW)
            {
                // corr[item] = -(1032 Sa + 72 Sb): Sa = the item's x at k % 8 in {0, 1, 4, 5} (the positions fed as 1024 + q), Sb = the others (64 + q)
                float* const cr = corr + (e & 1u) * (X_MAX_HALFS / 128);
                const int groups = d.k_in / 8;
                for (int g8 = w * 64 + lane; g8 < ((groups + 63) & ~63); g8 += NC * 64)
                {
                    float v = 0.0f;
                    if (g8 < groups)
                    {
                        const f16x8 xv = *(const f16x8*)(xb + g8 * 8);
                        const float sa = (float)xv[0] + (float)xv[1] + (float)xv[4] + (float)xv[5];
                        const float sb = (float)xv[2] + (float)xv[3] + (float)xv[6] + (float)xv[7];
                        v = -(1032.0f * sa + 72.0f * sb);
                    }
                    v = row16_allreduce_add(v);
                    if ((lane & 15) == 0 && g8 < groups) cr[g8 >> 4] = v;
                }
                sync_consumers(f_gath2, e);
            }
            float tot = 0.0f;
            for (int t = 0; t < NC; t++) tot += f_ssq[(e & 1u) * 16 + t];
            rs = rsqrtf(tot / (float)d.k_in + 1e-5f);
        }
        // ---- my items of this phase
        const int units = d.pair ? d.n_tiles / 2 : d.n_tiles;
        const int my_units = (units - cu + N_CU - 1) / N_CU;
        const int my_tiles = my_units * (d.pair ? 2 : 1);
        const int per_tile = d.F + 1;
        const u32 n = (u32)(my_tiles * per_tile);
        if constexpr (TB > 0)
        {
            static_assert(RAW && DEC, "the tight consumer is the raw 4-bit feed");
            u32 gt = 0, mine = (u32)w;                                  // turn counter of the phase, my next turn
            const float* const cr = corr + (e & 1u) * (X_MAX_HALFS / 128);
            for (int ti = 0; ti < my_tiles; ti++)
            {
                const u32 tb = g0 + (u32)(ti * per_tile);               // the tile's scale item; its F weight items follow
                float accs = 0.0f;
                for (int t0 = 0; t0 < d.F; t0 += TB, gt++)
                {
                    if (gt != mine) continue;
                    mine += NC;
                    const int nb = d.F - t0 < TB ? d.F - t0 : TB;
                    const u32 first = tb + 1u + (u32)t0, last = first + (u32)nb - 1u;
                    if (filled <= last)
                    {
                        int spins = 0;
                        while (true)
                        {
                            filled = uniform(lds_read_u32(f_filled));
                            if (filled > last) break;
                            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.err + 1, 1u); filled = last + 1; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    const u32 s0 = first % (u32)R;
                    u32x4 wq[TB]; float cs[TB];
                    #pragma unroll
                    for (int i = 0; i < TB; i++)
                    {
                        u32 sl = s0 + (u32)i; sl = sl >= (u32)R ? sl - (u32)R : sl;
                        wq[i] = *(const u32x4*)(ring + sl * 1024 + lane * 16);
                        cs[i] = cr[t0 + (i < nb ? i : 0)];
                    }
                    #pragma unroll
                    for (int i = 0; i < TB; i++)
                    {
                        const u32 ww[4] = {wq[i].x, wq[i].y, wq[i].z, wq[i].w};
                        f32x4 pt = {cs[i], cs[i], cs[i], cs[i]};
                        const f16* const arow = xb + (t0 + (i < nb ? i : 0)) * 128 + 8 * j;
                        #pragma unroll
                        for (int qq = 0; qq < 4; qq++)
                        {
                            const u32 xw = ww[qq], yw = xw >> 8;
                            const u32x4 bw = {(xw & m_lo) | k_lo, (xw & m_hi) | k_hi, (yw & m_lo) | k_lo, (yw & m_hi) | k_hi};
                            const f16x8 b = __builtin_bit_cast(f16x8, bw);
                            const f16x8 av = *(const f16x8*)(arow + qq * 32);
                            pt = mfma_16x16x32_f16(av, b, pt);
                        }
                        accs = fmaf(i < nb ? 0.0078125f : 0.0f, pt[0], accs);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) lds_write_u32(f_cons + w, last + 1u);
                }
                if (lane < 16) part[(ti * NC + w) * 16 + c] = accs;
            }
            if (lane == 0) lds_write_u32(f_cons + w, g0 + n);          // nothing of this phase is mine any more
        }
        else
        {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        int cur_tile = -1;
        u32 x = g0 + ((u32)w + NC - g0 % NC) % NC;
        for (; x < g0 + n; x += NC)
        {
            const int rel = (int)(x - g0), ti = rel / per_tile, k = rel - ti * per_tile;
            if (ti != cur_tile)
            {
                if (cur_tile >= 0 && lane < 16) part[(cur_tile * NC + w) * 16 + c] = acc[0];
                cur_tile = ti; acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (filled <= x)
            {
                int spins = 0;
                while (true)
                {
                    filled = uniform(lds_read_u32(f_filled));
                    if (filled > x) break;
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.err + 1, 1u); filled = x + 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if constexpr (CHECK)
            {
                const u32 got = *(const u32*)(ring + (x % (u32)R) * 1024 + lane * 16);
                if (got != (((u32)cu << 20) | x) + (u32)lane) atomicAdd(a.err + 3, 1u);
            }
            if (DEC && RAW && k > 0)
            {
                const u32 slot = x % (u32)R;
                const u32x4 wq = *(const u32x4*)(ring + slot * 1024 + lane * 16);
                const u32 ww[4] = {wq.x, wq.y, wq.z, wq.w};
                const float cs = corr[(e & 1u) * (X_MAX_HALFS / 128) + (k - 1)];
                f32x4 pt = {cs, cs, cs, cs};
                const f16* const arow = xb + (k - 1) * 128 + 8 * j;
                #pragma unroll
                for (int qq = 0; qq < 4; qq++)
                {
                    const u32 xw = ww[qq], yw = xw >> 8;
                    const u32x4 bw = {(xw & m_lo) | k_lo, (xw & m_hi) | k_hi, (yw & m_lo) | k_lo, (yw & m_hi) | k_hi};
                    const f16x8 b = __builtin_bit_cast(f16x8, bw);
                    const f16x8 av = *(const f16x8*)(arow + qq * 32);
                    pt = mfma_16x16x32_f16(av, b, pt);
                }
                acc[0] = fmaf(0.0078125f, pt[0], acc[0]);
            }
            else if (DEC && k > 0)
            {
                const u32 slot = x % (u32)R;
                const u32x4 wq = *(const u32x4*)(ring + slot * 1024 + lane * 16);
                const u32 ww[4] = {wq.x, wq.y, wq.z, wq.w};
                f16x2 pp[16];
                dequant_super<4>(ww, z4, pp);
                f32x4 pt = {0.0f, 0.0f, 0.0f, 0.0f};
                const f16* const arow = xb + (k - 1) * 128 + 8 * j;
                #pragma unroll
                for (int qq = 0; qq < 4; qq++)
                {
                    const f16x8 b = {pp[4 * qq].x, pp[4 * qq].y, pp[4 * qq + 1].x, pp[4 * qq + 1].y, pp[4 * qq + 2].x, pp[4 * qq + 2].y, pp[4 * qq + 3].x, pp[4 * qq + 3].y};
                    const f16x8 av = *(const f16x8*)(arow + qq * 32);
                    pt = mfma_16x16x32_f16(av, b, pt);
                }
                #pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = fmaf(0.0078125f, pt[i], acc[i]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            myk++;
            if (lane == 0) lds_write_u32(f_cons + w, myk);
        }
        if (cur_tile >= 0 && lane < 16) part[(cur_tile * NC + w) * 16 + c] = acc[0];
        }
        // (a consumer whose share held no item of a tile leaves that slot of `part` stale: every tile has F + 1 >= 33 > NC items)
        sync_consumers(f_done, e + 1);
        if (a.trace && w == 0 && lane == 0) a.trace[((size_t)(L * NPH + p) * N_CU + cu) * 4 + 2] = realtime_stamp();
        // ---- reduce + epilogue + publish: unit ui -> consumer ui % NC
        for (int ui = w; ui < my_units; ui += NC)
        {
            float v = 0.0f, v2 = 0.0f;
            if (lane < 16)
            {
                const int t0 = d.pair ? 2 * ui : ui;
                for (int t = 0; t < NC; t++) v += part[(t0 * NC + t) * 16 + c];
                if (d.pair) for (int t = 0; t < NC; t++) v2 += part[((t0 + 1) * NC + t) * 16 + c];
            }
            v *= rs * 0.02f;
            if (d.pair) { v2 *= rs * 0.02f; v = v / (1.0f + __expf(-v)) * v2; }
            v = fminf(fmaxf(v, -4.0f), 4.0f);
            const float vn = shfl_idx_f32(v, lane + 1);
            if constexpr (HO)
            {
                if (lane < 16 && (lane & 1) == 0)
                {
                    const f16x2 h = {(f16)v, (f16)vn};
                    const int unit = cu + ui * N_CU;
                    const u64 gr = ((u64)(e + 1u) << 32) | as_u32(h);
                    __hip_atomic_store(a.vec[(p + 1) % NPH] + (d.out_off + unit * 16 + lane) / 2, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            else if (lane == 0 && v == 123.456f) a.err[3] = 1;
        }
        if (a.trace && w == 0 && lane == 0) a.trace[((size_t)(L * NPH + p) * N_CU + cu) * 4 + 3] = realtime_stamp();
        g0 += n;
    }
}

__global__ void bump_kernel(u32* ctr) { ctr[0] += 1; }
// word 4 l of item x of CU c = (c << 20 | x) + l (the CHECK variant reads it back from the ring); everything else 0x37373737
__global__ void init_kernel(u32* w, u64 cu_words, u32 items)
{
    u32* const b = w + (u64)blockIdx.x * cu_words;
    for (u64 i = threadIdx.x; i < (u64)items * 256; i += blockDim.x)
    {
        const u32 x = (u32)(i >> 8), o = (u32)(i & 255);
        b[i] = (o & 3) == 0 ? ((blockIdx.x << 20) | x) + (o >> 2) : 0x37373737u;
    }
}

// the launches baseline of the same content: lean_probe.hip (26.9 us per layer), production 35.8 us (profiles/history/r05_kernel_stats.csv)
typedef void (*EngFn)(const EngArgs);
struct Variant { const char* name; EngFn fn; int nc, rd; };
#define V(NC, RD, D, DEC, HO) {"NC" #NC " RD" #RD " D" #D " dec" #DEC " ho" #HO, engine_kernel<NC, RD, D, DEC, HO>, NC, RD}
#define VT(NC, RD, D, THIN, GW) {"NC" #NC " RD" #RD " D" #D " dec1 ho1 thin" #THIN " gw" #GW, engine_kernel<NC, RD, D, true, true, false, THIN, GW>, NC, RD}
#define VT0(NC, RD, D, THIN, GW) {"NC" #NC " RD" #RD " D" #D " dec0 ho1 thin" #THIN " gw" #GW, engine_kernel<NC, RD, D, false, true, false, THIN, GW>, NC, RD}
#define VR(NC, RD, D, HO, GW) {"NC" #NC " RD" #RD " D" #D " RAW ho" #HO " gw" #GW, engine_kernel<NC, RD, D, true, HO, false, 0, GW, true>, NC, RD}
#define VB(NC, RD, D, HO, TB) {"NC" #NC " RD" #RD " D" #D " RAW TB" #TB " ho" #HO, engine_kernel<NC, RD, D, true, HO, false, 0, NC, true, TB>, NC, RD}
#define VCHK(NC, RD, D) {"NC" #NC " RD" #RD " D" #D " CHECK", engine_kernel<NC, RD, D, false, false, true>, NC, RD}

int main(int argc, char** argv)
{
    const int layers = 32;
    EngArgs a; memset(&a, 0, sizeof(a));
    // o 256 tiles x 32 items, gate|up 688 pairs x 32, down 256 x 86, q|k|v 768 x 32 (Llama-2-7B at 4 bits; item = 1 KB)
    const int tiles[NPH] = {256, 1376, 256, 768}, F[NPH] = {32, 32, 86, 32}, pair[NPH] = {0, 1, 0, 0}, k_in[NPH] = {4096, 4096, 11008, 4096};
    u64 off = 0; u32 cu_items = 0;
    for (int p = 0; p < NPH; p++)
    {
        a.ph[p].w_off = 0; a.ph[p].n_tiles = tiles[p]; a.ph[p].F = F[p]; a.ph[p].pair = pair[p]; a.ph[p].k_in = k_in[p]; a.ph[p].out_off = 0;
        off += (u64)tiles[p] * (F[p] + 1) * 1024;
        const int units = pair[p] ? tiles[p] / 2 : tiles[p];
        cu_items += (u32)(((units + N_CU - 1) / N_CU) * (pair[p] ? 2 : 1) * (F[p] + 1));          // (CU 0: the longest stream)
    }
    cu_items = (cu_items * layers + FILL + 15) / 16 * 16;
    a.cu_stride = (u64)cu_items * 1024; a.layers = layers;
    u8* w; CK(hipMalloc(&w, a.cu_stride * N_CU));
    hipLaunchKernelGGL(init_kernel, dim3(N_CU), dim3(1024), 0, 0, (u32*)w, a.cu_stride / 4, cu_items);
    CK(hipDeviceSynchronize());
    a.w = w;
    for (int p = 0; p < NPH; p++) { CK(hipMalloc(&a.vec[p], 16384 * 8)); CK(hipMemset(a.vec[p], 0, 16384 * 8)); }
    u32* ctr; CK(hipMalloc(&ctr, 64)); CK(hipMemset(ctr, 0, 64));
    u32* err; CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
    a.epoch_ctr = ctr; a.err = err;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, ctr);
    printf("layer = %.1f MB (%d layers); launches baseline of the same content: 26.9 us (lean_probe), production 35.8 us per layer\n", off / 1e6, layers);

    // RD = fills of 16 KB in the ring, D = fills in flight
    std::vector<Variant> vars = {
        VCHK(3, 6, 2), V(3, 6, 2, 0, 0), V(11, 6, 3, 1, 1), VR(11, 6, 3, 0, 11), VR(11, 6, 3, 1, 11),
        VB(3, 6, 3, 0, 4), VB(7, 6, 3, 0, 4), VB(11, 6, 3, 0, 4), VB(15, 6, 3, 0, 2), VB(7, 6, 3, 0, 2), VB(11, 6, 3, 0, 2),
        VB(3, 6, 3, 1, 4), VB(7, 6, 3, 1, 4), VB(11, 6, 3, 1, 4), VB(15, 6, 3, 1, 2), VB(7, 6, 3, 1, 2), VB(11, 6, 3, 1, 2),
    };
    u64* trace; CK(hipMalloc(&trace, (size_t)layers * NPH * N_CU * 4 * 8));
    const bool want_trace = getenv("ENGINE_TRACE") != nullptr;
    const char* only = argc > 1 ? argv[1] : nullptr;
    for (const Variant& v : vars)
    {
        if (only && !strstr(v.name, only)) continue;
        const size_t lds = (size_t)v.rd * FILL * 1024 + 2 * X_MAX_HALFS * 2 + MAX_TILES_CU * v.nc * 64 + 512 + 1024;
        CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemsetAsync(err, 0, 64, st));
        auto run = [&]() {
            hipLaunchKernelGGL(v.fn, dim3(N_CU), dim3((v.nc + 1) * 64), lds, st, a);
            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, st, ctr);
        };
        a.trace = nullptr;
        run(); CK(hipStreamSynchronize(st));
        const int reps = 5;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++) run();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        u32 herr[4]; CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
        const double us_layer = ms * 1e3 / reps / layers;
        printf("%-24s %7.2f us per layer  %5.2f TB/s  %.3f of 8 TB/s   (LDS %zu KB; give-ups gather %u ring %u sync %u; check mismatches %u)\n", v.name, us_layer,
               off / 1e6 / us_layer, off / 1e6 / us_layer / 8.0, lds / 1024, herr[0], herr[1], herr[2], herr[3]);
        if (want_trace && strstr(v.name, "ho1"))
        {
            // one traced launch: where an edge's time goes (averages over the phases of layers 4 .. 27; 10 ns stamps shared by all XCDs)
            a.trace = trace; CK(hipMemsetAsync(trace, 0, (size_t)layers * NPH * N_CU * 4 * 8, st));
            run(); CK(hipStreamSynchronize(st));
            std::vector<u64> t((size_t)layers * NPH * N_CU * 4);
            CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
            const char* pn[NPH] = {"o", "gate|up", "down", "q|k|v"};
            for (int p = 0; p < NPH; p++)
            {
                double g_med = 0, after_pub = 0, it_min = 0, it_med = 0, it_max = 0, pub = 0, span = 0; int cnt = 0;
                for (int L = 4; L < 28; L++)
                {
                    const size_t ph = (size_t)L * NPH + p, prev = ph - 1;
                    u64 pub_last = 0, g_end_max = 0, it_end_min = ~0ull, it_end_max = 0, g_end_min = ~0ull;
                    std::vector<double> gd, itd;
                    for (int c = 0; c < N_CU; c++)
                    {
                        const u64* x = &t[(ph * N_CU + c) * 4]; const u64* y = &t[(prev * N_CU + c) * 4];
                        if (y[3] > pub_last) pub_last = y[3];
                        if (x[1] > g_end_max) g_end_max = x[1];
                        if (x[1] < g_end_min) g_end_min = x[1];
                        if (x[2] < it_end_min) it_end_min = x[2];
                        if (x[2] > it_end_max) it_end_max = x[2];
                        gd.push_back((double)(x[1] - x[0])); itd.push_back((double)(x[2] - x[1]));
                        pub += (double)(x[3] - x[2]) / N_CU;
                    }
                    std::sort(gd.begin(), gd.end()); std::sort(itd.begin(), itd.end());
                    g_med += gd[N_CU / 2]; after_pub += (double)(g_end_max - pub_last); it_min += itd[0]; it_med += itd[N_CU / 2]; it_max += itd[N_CU - 1];
                    span += (double)(it_end_max - g_end_min);
                    cnt++;
                }
                const double k = 0.01 / cnt;
                printf("    %-8s gather (median CU) %5.2f us; last gather end - last publish of the producer phase %5.2f us; items min / median / max %5.2f / %5.2f / %5.2f us; "
                       "first gather end -> last items end %5.2f us; reduce + publish %4.2f us\n", pn[p], g_med * k, after_pub * k, it_min * k, it_med * k, it_max * k, span * k, pub * k);
            }
            a.trace = nullptr;
        }
        fflush(stdout);
    }
    return 0;
}
