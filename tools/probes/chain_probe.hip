// Upper-bound probe for a persistent "layer chain" kernel: 256 workgroups x 16 waves stream the weight bytes of the four
// GEMV phases of a Llama-7B layer (o, gate+up, down, qkv) with a grid barrier between phases, weights of the next phase
// prefetched into a register ring before the barrier.  Load-only (xor-reduce so the loads are kept).  Compare with the
// same bytes streamed by four separate launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Phase { const u32x4* base; long long units_per_wave; };   // unit = 1 KB per wave-load (64 lanes x 16 B)
struct Args { Phase ph[8]; int n_phases; u32* sync; u32* sink; int n_wg; int spin_sleep; };

__device__ inline void grid_barrier(u32* counter, u32 target)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0)
    {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_s_barrier();
}
// same but the caller's outstanding loads stay in flight
__device__ inline void grid_barrier_keep(u32* counter, u32 target)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 0)
    {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("s_barrier" ::: "memory");
}

template <int D, bool PREFETCH>
__global__ void __launch_bounds__(1024) chain_kernel(const Args a)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * 16 + wv;                       // global wave index
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[D];
    const u32x4* p = a.ph[0].base + (size_t)gw * a.ph[0].units_per_wave * 64 + lane;
    long long n = a.ph[0].units_per_wave;
    #pragma unroll
    for (int u = 0; u < D; u++) ring[u] = __builtin_nontemporal_load(p + (size_t)(u < n - 1 ? u : n - 1) * 64);
    for (int ph = 0; ph < a.n_phases; ph++)
    {
        long long i = 0;
        while (i + 2 * D <= n)
        {
            #pragma unroll
            for (int u = 0; u < D; u++) { acc ^= ring[u]; ring[u] = __builtin_nontemporal_load(p + (size_t)(i + u + D) * 64); }
            i += D;
        }
        if (i + D < n)
        {
            #pragma unroll
            for (int u = 0; u < D; u++) { acc ^= ring[u]; long long nx = i + u + D; ring[u] = __builtin_nontemporal_load(p + (size_t)(nx < n - 1 ? nx : n - 1) * 64); }
            i += D;
        }
        #pragma unroll
        for (int u = 0; u < D; u++) if (i + u < n) acc ^= ring[u];
        if (ph + 1 < a.n_phases)
        {
            const u32x4* p2 = a.ph[ph + 1].base + (size_t)gw * a.ph[ph + 1].units_per_wave * 64 + lane;
            const long long n2 = a.ph[ph + 1].units_per_wave;
            if (PREFETCH)
            {
                #pragma unroll
                for (int u = 0; u < D; u++) ring[u] = __builtin_nontemporal_load(p2 + (size_t)(u < n2 - 1 ? u : n2 - 1) * 64);
                grid_barrier_keep(a.sync, (u32)(ph + 1) * a.n_wg);
            }
            else
            {
                grid_barrier(a.sync, (u32)(ph + 1) * a.n_wg);
                #pragma unroll
                for (int u = 0; u < D; u++) ring[u] = __builtin_nontemporal_load(p2 + (size_t)(u < n2 - 1 ? u : n2 - 1) * 64);
            }
            p = p2; n = n2;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.sink[gw] = acc.x;
    // reset the barrier counter for the next launch (last workgroup out)
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0)
    {
        const u32 t = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (u32)a.n_wg - 1) { __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

template <int D, bool NT, bool INTERLEAVE>
__global__ void __launch_bounds__(1024) stream_kernel(const u32x4* base, long long units_per_wave, u32* sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long gw = (long long)blockIdx.x * nw + wv;
    // INTERLEAVE: consecutive 1-KB units of a workgroup's region go to consecutive waves; else each wave owns a contiguous run
    const u32x4* p = INTERLEAVE ? base + ((size_t)blockIdx.x * nw * units_per_wave + wv) * 64 + lane
                                : base + (size_t)gw * units_per_wave * 64 + lane;
    const size_t step = INTERLEAVE ? (size_t)nw * 64 : 64;
    const long long n = units_per_wave;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[D];
    #pragma unroll
    for (int u = 0; u < D; u++)
    {
        const u32x4* q = p + (size_t)(u < n - 1 ? u : n - 1) * step;
        ring[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    long long i = 0;
    while (i + 2 * D <= n)
    {
        #pragma unroll
        for (int u = 0; u < D; u++) { acc ^= ring[u]; const u32x4* q = p + (size_t)(i + u + D) * step; ring[u] = NT ? __builtin_nontemporal_load(q) : *q; }
        i += D;
    }
    if (i + D < n)
    {
        #pragma unroll
        for (int u = 0; u < D; u++) { acc ^= ring[u]; long long nx = i + u + D; const u32x4* q = p + (size_t)(nx < n - 1 ? nx : n - 1) * step; ring[u] = NT ? __builtin_nontemporal_load(q) : *q; }
        i += D;
    }
    #pragma unroll
    for (int u = 0; u < D; u++) if (i + u < n) acc ^= ring[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[gw] = acc.x;
}

typedef void (*KernelFn)(const u32x4*, long long, u32*);

int main(int argc, char** argv)
{
    const int layers = 32;
    const double mb[4] = {8.4, 45.0, 22.5, 25.2};          // o, gate+up, down, qkv (Llama-2-7B 4.0bpw)
    const size_t total = (size_t)(101.2e6 * layers) + (64 << 20);
    char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    u32* sink; CK(hipMalloc(&sink, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { const char* name; KernelFn fn; int n_wg; int waves; };
    std::vector<Cfg> cfgs = {
        {"D8 nt contiguous 256x16", stream_kernel<8, true, false>, 256, 16},
        {"D8 nt contiguous 512x8", stream_kernel<8, true, false>, 512, 8},
        {"D8 nt contiguous 1024x4", stream_kernel<8, true, false>, 1024, 4},
        {"D8 nt contiguous 256x8", stream_kernel<8, true, false>, 256, 8},
        {"D8 nt contiguous 256x4", stream_kernel<8, true, false>, 256, 4},
        {"D8 nt interleaved 256x16", stream_kernel<8, true, true>, 256, 16},
        {"D8 nt interleaved 512x8", stream_kernel<8, true, true>, 512, 8},
        {"D8 plain contiguous 256x16", stream_kernel<8, false, false>, 256, 16},
        {"D8 plain interleaved 256x16", stream_kernel<8, false, true>, 256, 16},
        {"D4 nt contiguous 256x16", stream_kernel<4, true, false>, 256, 16},
        {"D4 nt interleaved 256x16", stream_kernel<4, true, true>, 256, 16},
        {"D16 nt interleaved 256x16", stream_kernel<16, true, true>, 256, 16},
        {"D16 nt interleaved 256x8", stream_kernel<16, true, true>, 256, 8},
        {"D8 nt interleaved 2048x2", stream_kernel<8, true, true>, 2048, 2},
        {"D8 nt interleaved 768x4", stream_kernel<8, true, true>, 768, 4},
    };
    for (const Cfg& c : cfgs)
    {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        size_t off = 0, bytes_layer = 0;
        for (int l = 0; l < layers; l++)
            for (int i = 0; i < 4; i++)
            {
                const long long upw = (long long)(mb[i] * 1e6 / ((double)c.n_wg * c.waves) / 1024 + 0.5);
                const size_t b = (size_t)upw * 1024 * c.n_wg * c.waves;
                hipLaunchKernelGGL(c.fn, dim3(c.n_wg), dim3(c.waves * 64), 0, st, (const u32x4*)(buf + off), upw, sink);
                off += b; if (l == 0) bytes_layer += b;
            }
        CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int w = 0; w < 2; w++) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 8;
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_layer = ms * 1e3 / reps / layers;
        printf("%-30s %7.2f us per layer (4 launches, %.1f MB)  %.2f TB/s\n", c.name, us_layer, bytes_layer / 1e6, bytes_layer / us_layer / 1e6);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    // one big launch for reference (whole 3.2 GB)
    {
        const long long upw = (long long)(3.2e9 / (256.0 * 16) / 1024);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((stream_kernel<8, true, true>), dim3(256), dim3(1024), 0, st, (const u32x4*)buf, upw, sink);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("one launch of %.2f GB: %.3f ms  %.2f TB/s\n", upw * 1024.0 * 4096 / 1e9, ms, upw * 1024.0 * 4096 / ms / 1e9);
    }
    return 0;
}
