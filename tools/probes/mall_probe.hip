// mall_probe.hip -- does the 256 MB memory-side cache (Infinity Cache / MALL) of an MI355X help a decode launch if its weight
// stream was READ a few microseconds earlier by someone else?  ("tail prefetch": the waves of launch N that
// are done would read the first part of launch N+1's weights and throw the data away.)
//
// For region sizes like the per-launch weight bytes of the 7B model (8 / 25 / 45 / 100 MB) and beyond the cache (400 MB):
//   cold    : evict (stream a 1.5 GB buffer), then time ONE streaming launch over the region
//   warm    : read the region with the prefetch kernel, then time the same streaming launch
//   warm+X  : read the region, stream X MB of OTHER data (what the launch in between does), then time the launch
// Streaming launch = 256 workgroups x 1024 threads, every wave walks its contiguous share in 1 KB wave-loads, 8 in flight,
// with plain or non-temporal loads (the product kernels stream weights non-temporally).  The prefetch kernel exists in two
// forms: data consumed (xor-reduced) and data discarded (loads issued, never waited for before the wave ends).
// Prints one line per (size, load flavour); time = best of 15 repetitions, HIP events around the single launch.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mall_probe.hip -o tools/probes/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(1024) stream_kernel(const u32x4* base, long long units_per_wave, u32* sink)
{
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 16 + (threadIdx.x >> 6);
    const u32x4* p = base + gw * units_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    long long i = 0;
    for (; i + 8 <= units_per_wave; i += 8)
    {
        u32x4 v[8];
        #pragma unroll
        for (int u = 0; u < 8; u++) v[u] = NT ? __builtin_nontemporal_load(p + (i + u) * 64) : p[(i + u) * 64];
        #pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    for (; i < units_per_wave; i++) acc ^= NT ? __builtin_nontemporal_load(p + i * 64) : p[i * 64];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = 1;          // keeps the loads
}

// loads issued and never consumed: the wave ends with them in flight
__global__ void __launch_bounds__(1024) discard_kernel(const u32x4* base, long long units_per_wave)
{
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 16 + (threadIdx.x >> 6);
    const u32x4* p = base + gw * units_per_wave * 64 + lane;
    for (long long i = 0; i < units_per_wave; i++)
    {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p + i * 64) : "memory");
    }
}

static float time_launch(void (*launch)(void*), void* ctx, hipEvent_t a, hipEvent_t b)
{
    CK(hipEventRecord(a, 0));
    launch(ctx);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f;
}

struct Ctx { const u32x4* base; long long upw; u32* sink; bool nt; };
static void launch_stream(void* c_)
{
    Ctx* c = (Ctx*)c_;
    if (c->nt) hipLaunchKernelGGL(stream_kernel<true>, dim3(256), dim3(1024), 0, 0, c->base, c->upw, c->sink);
    else       hipLaunchKernelGGL(stream_kernel<false>, dim3(256), dim3(1024), 0, 0, c->base, c->upw, c->sink);
}

int main()
{
    const size_t big = (size_t)1536 << 20, other = (size_t)768 << 20, region_max = (size_t)448 << 20;
    char *evict, *oth, *reg; u32* sink;
    CK(hipMalloc(&evict, big)); CK(hipMalloc(&oth, other)); CK(hipMalloc(&reg, region_max)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(evict, 1, big)); CK(hipMemset(oth, 2, other)); CK(hipMemset(reg, 3, region_max)); CK(hipMemset(sink, 0, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const long long wave_bytes = 1024, waves = 256 * 16;
    auto upw = [&](size_t bytes) { return (long long)(bytes / (wave_bytes * waves)); };
    Ctx ev = {(const u32x4*)evict, upw(big), sink, false};
    const int mb[] = {8, 25, 45, 100, 200, 400};
    const int interf[] = {0, 32, 128, 320};
    printf("size_MB loads | cold us (GB/s) | warm: prefetch consumed, then +0 / +32 / +128 / +320 MB of other traffic: us | prefetch discarded, +0: us\n");
    for (int nt = 0; nt < 2; nt++)
    for (int s = 0; s < 6; s++)
    {
        const size_t bytes = (size_t)upw((size_t)mb[s] << 20) * wave_bytes * waves;
        Ctx rc = {(const u32x4*)reg, upw(bytes), sink, nt != 0};
        Ctx pf = {(const u32x4*)reg, upw(bytes), sink, false};
        float cold = 1e30f, warm[4] = {1e30f, 1e30f, 1e30f, 1e30f}, disc = 1e30f;
        for (int rep = 0; rep < 15; rep++)
        {
            launch_stream(&ev); CK(hipDeviceSynchronize());
            cold = std::min(cold, time_launch(launch_stream, &rc, a, b));
            for (int k = 0; k < 4; k++)
            {
                launch_stream(&ev);                                   // start from an evicted cache every time
                launch_stream(&pf);                                   // someone reads the region ...
                if (interf[k]) { Ctx oc = {(const u32x4*)oth, upw((size_t)interf[k] << 20), sink, true}; launch_stream(&oc); }
                CK(hipDeviceSynchronize());
                warm[k] = std::min(warm[k], time_launch(launch_stream, &rc, a, b));
            }
            launch_stream(&ev);
            hipLaunchKernelGGL(discard_kernel, dim3(256), dim3(1024), 0, 0, (const u32x4*)reg, upw(bytes));
            CK(hipDeviceSynchronize());
            disc = std::min(disc, time_launch(launch_stream, &rc, a, b));
        }
        printf("%4d %s | %8.1f (%6.0f) | %8.1f %8.1f %8.1f %8.1f | %8.1f\n", mb[s], nt ? "nt   " : "plain", cold, bytes / cold * 1e-3,
               warm[0], warm[1], warm[2], warm[3], disc);
        fflush(stdout);
    }
    return 0;
}
