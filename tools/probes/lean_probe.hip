// lean_probe.hip -- round 3: what does a decode q_gemm launch cost when NOTHING but the necessary sits between wave entry and
// the first weight request?  A skeleton of the kernel the round is about to build, with the real 4-bit decode
// (qlayout.h: dequant_super<4> + 4 MFMAs + scale) but a trivial work split (every wave the same number of items), over the
// four per-layer shapes of Llama-2-7B (o 256 tiles x 32 items, gate|up 1376 x 32, down 256 x 88, q|k|v 768 x 32; item = 1 KB).
//
// Knobs (template parameters, one instantiation per combination used in the table at the bottom):
//   WAVES   waves per workgroup (4 / 8 / 16)  -> workgroups per CU follow from the register count
//   NI      items per wave, ALL requested at wave entry (the whole slice in flight: no ring)
//   XMODE   0: wave-private LDS-DMA of the wave's own x slice (no barrier at all)
//           1: workgroup-wide LDS-DMA of the whole x row + one LDS barrier
//           2: A fragments straight from global memory (no LDS)
//           3: wave-private, plain loads + ds_write (no LDS-DMA: the compiler keeps counting vmcnt)
//   COMB    0: the S waves of a tile sit in one workgroup: partial sums meet in LDS (one barrier)
//           1: partial sums meet in global memory: write-through stores + ticket, the last arriver sums in slot order
// Timing: HIP graph of 32 layers x 4 launches (distinct weights, 3.2 GB), per layer and per phase (a graph of 128 launches of
// that phase).  Reference lines: the plain streaming kernel of chain_probe (no decode, no x, no output).
//
// Build: hipcc --offload-arch=gfx950 -O3 -I exllamav2_amd/csrc tools/probes/lean_probe.hip -o tools/probes/lean_probe
#include "qlayout.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct LeanArgs
{
    const u32* w;           // [tile][F][256 words]
    const f16* x;           // [K = 128 F]
    const f16* sc;          // [tile][F][16]
    f16* out;               // [16 n_tiles]
    float* part;            // [tile][S][16]
    u32* tick;              // [tile]
    int F, S, n_tiles, pad;     // pad: runtime item count of the GUARD variants (== NI)
};

template <int N> DEV void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int WAVES, int NI, int XMODE, int COMB, bool GUARD = false>
KERNEL void __launch_bounds__(WAVES * 64) lean_kernel(const LeanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wv = uniform(wave_id());
    const int gw = bid_x() * WAVES + wv;
    const int S = a.S;
    const int tile = gw / S, r = gw - tile * S;
    if (tile >= a.n_tiles) return;                                     // (grids are exact multiples in this probe)
    const int c = lane & 15, j = lane >> 4;
    const int i0 = r * NI;
    const int n_rt = GUARD ? a.pad : NI;                               // GUARD: the count is a run-time value, every item sits behind a branch

    // x: issued first so that it is the oldest outstanding request
    f16* xl;
    if constexpr (XMODE == 0)
    {
        xl = (f16*)(smem + (size_t)wv * (NI * 256));
        #pragma unroll
        for (int b = 0; b < NI * 16; b += 64)
            if (b + lane < NI * 16) dma_to_lds16(a.x + (size_t)i0 * 128 + (size_t)(b + lane) * 8, (u8*)xl + b * 16);
    }
    else if constexpr (XMODE == 3)
    {
        // plain loads + ds_write into a wave-private area: no LDS-DMA, so the compiler's own vmcnt bookkeeping stays exact
        // (an LDS-DMA in flight makes every compiler-generated vmcnt wait a vmcnt(0): "pending FLAT" in SIInsertWaitcnts)
        xl = (f16*)(smem + (size_t)wv * (NI * 256));
        #pragma unroll
        for (int b = 0; b < NI * 16; b += 64)
            if (b + lane < NI * 16)
            {
                const f16x8 v = *(const f16x8*)(a.x + (size_t)i0 * 128 + (size_t)(b + lane) * 8);
                *(f16x8*)((u8*)xl + (size_t)(b + lane) * 16) = v;
            }
    }
    else if constexpr (XMODE == 1)
    {
        xl = (f16*)smem;
        const int units = a.F * 16;                                    // 16-byte units of the row
        for (int b = wv * 64; b < units; b += WAVES * 64)
            if (b + lane < units) dma_to_lds16(a.x + (size_t)(b + lane) * 8, smem + (size_t)b * 16);
    }
    // scales of my items (one fp16 per item per column)
    f16 s[NI];
    #pragma unroll
    for (int i = 0; i < NI; i++) s[i] = a.sc[((size_t)tile * a.F + i0 + i) * 16 + c];
    // weights: everything in flight
    LaneWords<4> w[NI];
    const u32* wp = a.w + ((size_t)tile * a.F + i0) * 256;
    #pragma unroll
    for (int i = 0; i < NI; i++) if (!GUARD || i < n_rt) load_lane_words<4>(wp + (size_t)i * 256, lane, w[i]);

    if constexpr (XMODE == 1)
    {
        // the row copy is older than the NI + NI loads issued behind it
        if constexpr (2 * NI <= 30) vm_wait<2 * NI>(); else vm_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const ZC zc = make_zc((f16)8.0f);
    const ZC z4[4] = {zc, zc, zc, zc};
    if constexpr (GUARD) vm_wait<0>();
    #pragma unroll
    for (int i = 0; i < NI; i++)
    {
        if (GUARD && i >= n_rt) continue;
        // wait for item i (loads complete in issue order)
        if constexpr (!GUARD) switch (NI - 1 - i)
        {
            case 0: vm_wait<0>(); break; case 1: vm_wait<1>(); break; case 2: vm_wait<2>(); break; case 3: vm_wait<3>(); break;
            case 4: vm_wait<4>(); break; case 5: vm_wait<5>(); break; case 6: vm_wait<6>(); break; case 7: vm_wait<7>(); break;
            case 8: vm_wait<8>(); break; case 9: vm_wait<9>(); break; case 10: vm_wait<10>(); break; default: vm_wait<11>(); break;
        }
        f16x2 p[16];
        dequant_super<4>(w[i].w, z4, p);
        f32x4 part = {0.0f, 0.0f, 0.0f, 0.0f};
        #pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const f16x8 b = {p[4 * q].x, p[4 * q].y, p[4 * q + 1].x, p[4 * q + 1].y, p[4 * q + 2].x, p[4 * q + 2].y, p[4 * q + 3].x, p[4 * q + 3].y};
            f16x8 av;
            if constexpr (XMODE == 0 || XMODE == 3) av = *(const f16x8*)(xl + i * 128 + q * 32 + 8 * j);
            else if constexpr (XMODE == 1) av = *(const f16x8*)(xl + (size_t)(i0 + i) * 128 + q * 32 + 8 * j);
            else av = *(const f16x8*)(a.x + (size_t)(i0 + i) * 128 + q * 32 + 8 * j);
            part = mfma_16x16x32_f16(av, b, part);
        }
        const float sf = (float)s[i];
        #pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = fmaf(sf, part[e], acc[e]);
        if constexpr (GUARD) __builtin_amdgcn_sched_barrier(0);
    }

    // combine: row 0 of the product sits in lanes 0..15, acc[0]
    if constexpr (COMB == 0)
    {
        float* red = (float*)(smem + ((XMODE == 0 || XMODE == 3) ? WAVES * NI * 256 : (XMODE == 1 ? a.F * 256 : 0)));
        if (lane < 16) red[wv * 16 + c] = acc[0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (r == 0 && lane < 16)
        {
            float v = 0.0f;
            for (int q = 0; q < S; q++) v += red[(wv + q) * 16 + c];
            a.out[tile * 16 + c] = (f16)((float)a.out[tile * 16 + c] + v);          // residual add
        }
    }
    else
    {
        float* slot = a.part + ((size_t)tile * S + r) * 16;
        if (lane < 16) store_agent_f32(slot + c, acc[0]);
        wait_vmcnt0();
        u32 old = 0;
        if (lane == 0) old = ticket_add_agent(a.tick + tile, 1u);
        old = uniform(old);
        if (old == (u32)S - 1)
        {
            if (lane < 16)
            {
                float v = 0.0f;
                const float* base = a.part + (size_t)tile * S * 16 + c;
                for (int q0 = 0; q0 < S; q0 += 8)                       // eight partials in flight, summed in slot order
                {
                    float t[8];
                    #pragma unroll
                    for (int u = 0; u < 8; u++) t[u] = q0 + u < S ? load_agent_f32(base + (q0 + u) * 16) : 0.0f;
                    #pragma unroll
                    for (int u = 0; u < 8; u++) v += t[u];
                }
                a.out[tile * 16 + c] = (f16)((float)a.out[tile * 16 + c] + v);
            }
            if (lane == 0) store_relaxed_agent(a.tick + tile, 0u);
        }
    }
}

// plain streaming reference (chain_probe's kernel): same bytes, nothing else
template <int D>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* base, long long units_per_wave, u32* sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long gw = (long long)blockIdx.x * nw + wv;
    const u32x4* p = base + (size_t)gw * units_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[D];
    #pragma unroll
    for (int u = 0; u < D; u++) ring[u] = __builtin_nontemporal_load(p + (size_t)(u < units_per_wave ? u : units_per_wave - 1) * 64);
    #pragma unroll
    for (int u = 0; u < D; u++) acc ^= ring[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[gw] = acc.x;
}

struct Phase { const char* name; int n_tiles, F; };
static const Phase PH[4] = {{"o", 256, 32}, {"gate|up", 1376, 32}, {"down", 256, 88}, {"q|k|v", 768, 32}};

typedef void (*LeanFn)(const LeanArgs);
struct Variant { const char* name; LeanFn fn; int waves, ni, xmode, comb; int cold_sc = 0; };
#define V(W, N, X, C) {#W "w NI" #N " x" #X " c" #C, lean_kernel<W, N, X, C>, W, N, X, C}
#define VG(W, N, X, C) {#W "w NI" #N " x" #X " c" #C " guard", lean_kernel<W, N, X, C, true>, W, N, X, C}
#define VC(W, N, X, C) {#W "w NI" #N " x" #X " c" #C " coldsc", lean_kernel<W, N, X, C>, W, N, X, C, 1}
#define VGC(W, N, X, C) {#W "w NI" #N " x" #X " c" #C " guard coldsc", lean_kernel<W, N, X, C, true>, W, N, X, C, 1}

struct Bufs { char* w; f16* x; f16* sc; char* sc_big; f16* out; float* part; u32* tick; u32* sink; size_t w_bytes; };

static size_t lds_bytes(const Variant& v, int F)
{
    const size_t xb = (v.xmode == 0 || v.xmode == 3) ? (size_t)v.waves * v.ni * 256 : (v.xmode == 1 ? (size_t)F * 256 : 0);
    return xb + (v.comb == 0 ? (size_t)v.waves * 64 : 0);
}

static bool fits(const Variant& v, const Phase& p)
{
    if (p.F % v.ni) return false;
    const int S = p.F / v.ni;
    if (v.comb == 0 && (S > v.waves || v.waves % S)) return false;
    if (((long long)p.n_tiles * S) % v.waves) return false;
    return true;
}

static void launch_phase(const Variant& v, const Phase& p, const Bufs& b, size_t w_off, hipStream_t st)
{
    LeanArgs a;
    // cold_sc: the scale table of every launch lives somewhere else (as in the real model: one table per matrix), 4 MB apart
    static size_t sc_rot = 0;
    const f16* sc = b.sc;
    if (v.cold_sc) { sc = (const f16*)((const char*)b.sc_big + (sc_rot % 200) * (4u << 20)); sc_rot++; }
    a.w = (const u32*)(b.w + w_off); a.x = b.x; a.sc = sc; a.out = b.out; a.part = b.part; a.tick = b.tick;
    a.F = p.F; a.S = p.F / v.ni; a.n_tiles = p.n_tiles; a.pad = v.ni;
    const int wgs = (int)((long long)p.n_tiles * a.S / v.waves);
    hipLaunchKernelGGL(v.fn, dim3(wgs), dim3(v.waves * 64), lds_bytes(v, p.F), st, a);
}

static float time_graph(hipGraphExec_t exec, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int reps)
{
    for (int w = 0; w < 2; w++) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main()
{
    const int layers = 32;
    Bufs b;
    size_t layer_bytes = 0;
    for (int i = 0; i < 4; i++) layer_bytes += (size_t)PH[i].n_tiles * PH[i].F * 1024;
    b.w_bytes = layer_bytes * layers;
    CK(hipMalloc(&b.w, b.w_bytes + (64 << 20))); CK(hipMemset(b.w, 0x37, b.w_bytes));
    CK(hipMalloc(&b.x, 16384 * 2)); CK(hipMemset(b.x, 0, 16384 * 2));
    CK(hipMalloc(&b.sc, (size_t)1376 * 88 * 32)); CK(hipMemset(b.sc, 0, (size_t)1376 * 88 * 32));
    CK(hipMalloc(&b.sc_big, (size_t)204 * (4u << 20))); CK(hipMemset(b.sc_big, 0, (size_t)204 * (4u << 20)));
    CK(hipMalloc(&b.out, 1376 * 32)); CK(hipMemset(b.out, 0, 1376 * 32));
    CK(hipMalloc(&b.part, (size_t)1376 * 64 * 64)); CK(hipMalloc(&b.tick, 1376 * 4)); CK(hipMemset(b.tick, 0, 1376 * 4));
    CK(hipMalloc(&b.sink, 1 << 22));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    std::vector<Variant> vars = {
        V(8, 4, 0, 0), VG(8, 4, 0, 0), VC(8, 4, 0, 0), VGC(8, 4, 0, 0),
        V(8, 8, 0, 0), VG(8, 8, 0, 0), V(8, 11, 0, 0), VG(8, 11, 0, 0), VGC(8, 11, 0, 0),
        V(16, 2, 0, 0), VG(16, 2, 0, 0), V(16, 4, 0, 0), VG(16, 4, 0, 0), VGC(16, 4, 0, 0),
    };

    for (const Variant& v : vars) CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));

    printf("per-phase time (graph of 128 launches of the phase over distinct weights), us per launch incl. the boundary\n");
    printf("%-18s", "variant");
    for (int i = 0; i < 4; i++) printf(" %9s", PH[i].name);
    printf("\n");
    // plain stream reference per phase
    {
        printf("%-18s", "stream 4w (ref)");
        for (int i = 0; i < 4; i++)
        {
            const long long items = (long long)PH[i].n_tiles * PH[i].F;
            const int upw = 8; const long long waves = items / upw;
            hipGraph_t g; hipGraphExec_t ex;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            size_t off = 0;
            for (int l = 0; l < 128; l++)
            {
                if (off + (size_t)items * 1024 > b.w_bytes) off = 0;
                hipLaunchKernelGGL(stream_kernel<8>, dim3((unsigned)(waves / 4)), dim3(256), 0, st, (const u32x4*)(b.w + off), (long long)upw, b.sink);
                off += (size_t)items * 1024;
            }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            printf(" %9.2f", time_graph(ex, st, e0, e1, 4) / 128);
            CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
        }
        printf("\n");
    }
    std::vector<std::vector<float>> best(4);
    std::vector<float> tmat(vars.size() * 4, -1.0f);
    for (size_t vi = 0; vi < vars.size(); vi++)
    {
        const Variant& v = vars[vi];
        printf("%-18s", v.name);
        for (int i = 0; i < 4; i++)
        {
            if (!fits(v, PH[i])) { printf(" %9s", "-"); continue; }
            const size_t pbytes = (size_t)PH[i].n_tiles * PH[i].F * 1024;
            hipGraph_t g; hipGraphExec_t ex;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            size_t off = 0;
            for (int l = 0; l < 128; l++)
            {
                if (off + pbytes > b.w_bytes) off = 0;
                launch_phase(v, PH[i], b, off, st);
                off += pbytes;
            }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            const float us = time_graph(ex, st, e0, e1, 4) / 128;
            tmat[vi * 4 + i] = us;
            printf(" %9.2f", us);
            CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
        }
        printf("\n");
        fflush(stdout);
    }
    // best variant per phase -> whole-layer graph (4 launches per layer, 32 layers)
    int pick[4];
    for (int i = 0; i < 4; i++)
    {
        pick[i] = -1;
        for (size_t vi = 0; vi < vars.size(); vi++)
            if (tmat[vi * 4 + i] > 0 && (pick[i] < 0 || tmat[vi * 4 + i] < tmat[pick[i] * 4 + i])) pick[i] = (int)vi;
        printf("best for %-8s: %-18s %.2f us (%.1f MB -> %.2f TB/s incl. boundary)\n", PH[i].name, vars[pick[i]].name, tmat[pick[i] * 4 + i],
               PH[i].n_tiles * PH[i].F * 1024 / 1e6, PH[i].n_tiles * PH[i].F * 1024 / 1e6 / tmat[pick[i] * 4 + i]);
    }
    {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        size_t off = 0;
        for (int l = 0; l < layers; l++)
            for (int i = 0; i < 4; i++)
            {
                launch_phase(vars[pick[i]], PH[i], b, off, st);
                off += (size_t)PH[i].n_tiles * PH[i].F * 1024;
            }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const float us = time_graph(ex, st, e0, e1, 8) / layers;
        printf("layer of best variants: %.2f us per layer (%.1f MB) = %.2f TB/s = %.3f of 8 TB/s\n", us, layer_bytes / 1e6, layer_bytes / 1e6 / us, layer_bytes / 1e6 / us / 8.0);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    return 0;
}
