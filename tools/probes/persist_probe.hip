// persist_probe.hip -- upper bound for the round-3 plan of DESIGN.md section 8: ONE persistent launch walks the GEMV phases of
// the decode step (o 8.4 MB, gate|up 45 MB, down 22.5 MB, q|k|v 25.2 MB per Llama-2-7B layer, 32 layers = 128 phases) with a
// grid-wide hand-off between phases, and a workgroup that has finished phase p copies its share of phase p + 1 into LDS by
// LDS-DMA BEFORE it waits (weight addresses depend on nothing), so the HBM stream overlaps the hand-off.  Load-only: after
// the hand-off the share is read back from LDS (ds_read_b128, xor-reduced) and whatever did not fit in LDS is streamed from
// global memory.  No decode, no activations: this is the ceiling of the structure, to be compared with
//   * the same bytes as four plain streaming launches per layer in a HIP graph (tools/probes/chain_probe: 34.5 us per layer),
//   * this kernel without the prefetch (hand-off first, then everything from global).
// Hand-off = the protocol measured in round 2 (csrc/chain_sync.h): arrivals add to a counter nobody polls, the last arrival
// publishes the phase number in 8 copies of a "go" word (one 128-byte line each), ONE lane per workgroup polls its copy.
// Every spin is bounded: a launch that cannot make progress (fewer CUs than workgroups) gives up and says so.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/persist_probe.hip -o tools/probes/persist_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define MAX_PHASES 160
#define LDS_BYTES (144 * 1024)
#define SPIN_LIMIT (1 << 21)

struct Args
{
    const char* base[MAX_PHASES];       // phase's region; workgroup b owns [b * wg_bytes, (b + 1) * wg_bytes)
    u32 wg_bytes[MAX_PHASES];           // multiple of 16 KB (16 waves x 1 KB)
    int n_phases, n_wg;
    u32* sync;                          // [0]: arrivals (monotonic), [32 * (1 + c)]: go word copy c, [32 * 10]: gave-up flag
    u32* sink;
};

__device__ inline void lds_dma16(const void* g_lane_ptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane_ptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// phase `done` (1-based count of finished phases) is complete for this workgroup: arrive, wait until every workgroup has
__device__ inline bool hand_off(u32* sync, u32 done, int n_wg)
{
    __builtin_amdgcn_s_barrier();                                      // the workgroup's own loads of the phase are consumed
    __shared__ int ok;
    if (threadIdx.x == 0)
    {
        const u32 old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == done * (u32)n_wg)
            for (int c = 0; c < 8; c++) __hip_atomic_store(sync + 32 * (1 + c), done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32* go = sync + 32 * (1 + (blockIdx.x & 7));
        int spins = 0;
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < done && spins < SPIN_LIMIT) { __builtin_amdgcn_s_sleep(1); spins++; }
        ok = spins < SPIN_LIMIT;
        if (!ok) __hip_atomic_store(sync + 32 * 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_barrier();
    return ok != 0;
}

template <bool PREFETCH>
__global__ void __launch_bounds__(1024) persist_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    u32 in_lds = 0;                                                     // bytes of the CURRENT phase already copied to LDS
    for (int ph = 0; ph < a.n_phases; ph++)
    {
        const char* mine = a.base[ph] + (size_t)blockIdx.x * a.wg_bytes[ph];
        const u32 total = a.wg_bytes[ph];
        // 1. what sits in LDS (copied before the hand-off): wait for the copies, read it back
        if (in_lds)
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (u32 off = (u32)wv * 1024; off < in_lds; off += 16 * 1024)
                acc ^= *(const u32x4*)(lds + off + lane * 16);
        }
        // 2. the rest from global memory: wave-contiguous 1 KB loads, 8 in flight
        {
            const u32 rest = total - in_lds;
            const u32 per_wave = rest / 16;                              // multiple of 1 KB
            const u32x4* p = (const u32x4*)(mine + in_lds + (size_t)wv * per_wave) + lane;
            const u32 n = per_wave / 1024;
            u32 i = 0;
            for (; i + 8 <= n; i += 8)
            {
                u32x4 v[8];
                #pragma unroll
                for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(p + (size_t)(i + u) * 64);
                #pragma unroll
                for (int u = 0; u < 8; u++) acc ^= v[u];
            }
            for (; i < n; i++) acc ^= __builtin_nontemporal_load(p + (size_t)i * 64);
        }
        in_lds = 0;
        if (ph + 1 == a.n_phases) break;
        // 3. before waiting for the others: this workgroup's share of the NEXT phase into LDS (as much as fits)
        if (PREFETCH)
        {
            __builtin_amdgcn_s_barrier();                               // everyone has read the previous LDS contents
            const char* next = a.base[ph + 1] + (size_t)blockIdx.x * a.wg_bytes[ph + 1];
            const u32 fit = a.wg_bytes[ph + 1] < LDS_BYTES ? a.wg_bytes[ph + 1] : LDS_BYTES;
            for (u32 off = (u32)wv * 1024; off < fit; off += 16 * 1024)
                lds_dma16(next + off + lane * 16, lds + off);
            in_lds = fit;
        }
        // 4. hand-off
        if (!hand_off(a.sync, (u32)(ph + 1), a.n_wg)) return;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) a.sink[blockIdx.x] = acc.x;
}

int main()
{
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int n_wg = prop.multiProcessorCount;
    const int layers = 32;
    const double mb[4] = {8.4, 45.0, 22.5, 25.2};                       // o, gate|up, down, q|k|v
    Args a; a.n_phases = layers * 4; a.n_wg = n_wg;
    size_t total = 0;
    for (int p = 0; p < a.n_phases; p++)
    {
        u32 wgb = (u32)(mb[p & 3] * 1e6 / n_wg / 16384 + 0.5) * 16384;
        a.wg_bytes[p] = wgb; total += (size_t)wgb * n_wg;
    }
    char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    size_t off = 0;
    for (int p = 0; p < a.n_phases; p++) { a.base[p] = buf + off; off += (size_t)a.wg_bytes[p] * n_wg; }
    CK(hipMalloc(&a.sync, 4096)); CK(hipMalloc(&a.sink, 4096));
    CK(hipFuncSetAttribute((const void*)persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d workgroups x 1024 threads, %d phases, %.2f GB per launch (bytes per layer %.1f MB)\n", n_wg, a.n_phases, total / 1e9, total / 1e6 / layers);
    for (int variant = 0; variant < 2; variant++)
    {
        float best = 1e30f; u32 gave_up = 0;
        for (int rep = 0; rep < 6; rep++)
        {
            CK(hipMemset(a.sync, 0, 4096));
            CK(hipEventRecord(e0, 0));
            if (variant) hipLaunchKernelGGL(persist_kernel<true>, dim3(n_wg), dim3(1024), LDS_BYTES, 0, a);
            else         hipLaunchKernelGGL(persist_kernel<false>, dim3(n_wg), dim3(1024), LDS_BYTES, 0, a);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&gave_up, a.sync + 32 * 10, 4, hipMemcpyDeviceToHost));
            if (gave_up) break;
            if (rep) best = ms < best ? ms : best;
        }
        if (gave_up) { printf("%s: a hand-off gave up (workgroups not co-resident?)\n", variant ? "prefetch into LDS before the hand-off" : "hand-off first, no prefetch"); continue; }
        printf("%-42s %8.2f us per layer (4 phases)  %.2f TB/s   %.3f ms per 32-layer pass\n",
               variant ? "prefetch into LDS before the hand-off" : "hand-off first, no prefetch", best * 1e3 / layers, total / (best * 1e-3) / 1e12, best);
    }
    return 0;
}
