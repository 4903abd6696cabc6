// What does COLD CODE cost a short launch on gfx950?  (round 5; output: profiles/history/r05_ifetch_probe.txt)
//
// The chained decode kernel (csrc/qgemv_lean.hip) is straight-line code by design: 140-190 KB per instantiation, of which one
// wave walks ~6-8 KB once.  Its in-kernel timeline shows ~10 cycles per instruction and wave where the instruction mix explains
// 1.5-2.  This probe measures the one candidate the earlier rounds never isolated: instruction fetch.  A wave executes KB
// kilobytes of filler code, either as ONE straight run (every 64-byte line is fetched once, cold) or as a 1 KB body looped KB
// times (one cold kilobyte, then hits); the region is timed per wave with s_memrealtime (100 MHz) and per launch with HIP
// events.  Regimes: the same kernel back to back (is the instruction cache kept across a kernel boundary?), alternating with a
// kernel that streams 64 MB (the L2s lose the code), and with the probe's own waves holding loads in flight (do instruction
// fetches queue behind a CU's data requests?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 stamp() { return __builtin_amdgcn_s_memrealtime(); }

// FILL = 0: s_nop 0 (4 bytes, ~1 cycle); 1: v_add3_u32 (8 bytes, 4 cycles on 64 lanes): 2 bytes of code per cycle, like the decode
template <int KB, bool LOOP, int FILL>
__device__ __forceinline__ void filler(u32& v)
{
    if constexpr (FILL == 0)
    {
        if constexpr (LOOP)
            asm volatile("s_mov_b32 s30, %0\n1:\n .rept 256\n s_nop 0\n .endr\n s_sub_u32 s30, s30, 1\n s_cmp_lg_u32 s30, 0\n s_cbranch_scc1 1b\n" :: "n"(KB) : "s30", "scc");
        else
            asm volatile(".rept %0\n s_nop 0\n .endr\n" :: "n"(KB * 256));
    }
    else
    {
        if constexpr (LOOP)
            asm volatile("s_mov_b32 s30, %1\n1:\n .rept 128\n v_add3_u32 %0, %0, 1, 2\n .endr\n s_sub_u32 s30, s30, 1\n s_cmp_lg_u32 s30, 0\n s_cbranch_scc1 1b\n" : "+v"(v) : "n"(KB) : "s30", "scc");
        else
            asm volatile(".rept %1\n v_add3_u32 %0, %0, 1, 2\n .endr\n" : "+v"(v) : "n"(KB * 128));
    }
}

// grid = wgs workgroups of 256 threads; LOADS 16-byte loads per lane are issued in front of the region and consumed behind it
template <int KB, bool LOOP, int FILL, int LOADS>
__global__ void __launch_bounds__(256) probe_kernel(const u32x4* __restrict__ src, size_t stride16, u64* __restrict__ times, u32* __restrict__ sink)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    u32x4 d[LOADS > 0 ? LOADS : 1];
    if constexpr (LOADS > 0)
    {
        #pragma unroll
        for (int i = 0; i < LOADS; i++) d[i] = __builtin_nontemporal_load(src + (gw * LOADS + i) * 64 + lane + stride16 * 0);
    }
    u32 v = lane;
    const u64 t0 = stamp();
    filler<KB, LOOP, FILL>(v);
    const u64 t1 = stamp();
    u32 acc = v;
    if constexpr (LOADS > 0)
    {
        #pragma unroll
        for (int i = 0; i < LOADS; i++) acc ^= d[i].x ^ d[i].y ^ d[i].z ^ d[i].w;
    }
    if (lane == 0) times[gw] = t1 - t0;
    if (acc == 0x12345677u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) thrash_kernel(const uint4* __restrict__ src, size_t n16, u32* __restrict__ sink)
{
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 q = src[i]; acc ^= q.x ^ q.y ^ q.z ^ q.w; }
    if (acc == 0x12345677u) sink[0] = acc;
}

struct Res { double med, p90, launch_us; };

template <int KB, bool LOOP, int FILL, int LOADS>
static Res run(int wgs, int regime, const uint4* src, size_t src16, u64* times, u32* sink, hipStream_t st)
{
    // regime 0: back to back; 1: a 64 MB streaming kernel between two probe launches
    const int reps = 40;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> med, p90; double total = 0.0;
    std::vector<u64> h((size_t)wgs * 4);
    for (int r = 0; r < reps + 4; r++)
    {
        if (regime == 1) hipLaunchKernelGGL(thrash_kernel, dim3(1024), dim3(256), 0, st, src, (size_t)(64u << 20) / 16, sink);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((probe_kernel<KB, LOOP, FILL, LOADS>), dim3(wgs), dim3(256), 0, st, (const u32x4*)(src + (size_t)(r & 7) * (8u << 20) / 16 * 8), (size_t)0, times, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        if (r < 4) continue;
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms * 1000.0;
        CK(hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        med.push_back(h[h.size() / 2] * 0.01); p90.push_back(h[h.size() * 9 / 10] * 0.01);
    }
    std::sort(med.begin(), med.end()); std::sort(p90.begin(), p90.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return {med[med.size() / 2], p90[p90.size() / 2], total / reps};
}

template <int KB, int FILL, int LOADS>
static void row(int wgs, const uint4* src, size_t src16, u64* times, u32* sink, hipStream_t st)
{
    const Res a = run<KB, false, FILL, LOADS>(wgs, 0, src, src16, times, sink, st);
    const Res b = run<KB, true, FILL, LOADS>(wgs, 0, src, src16, times, sink, st);
    const Res c = run<KB, false, FILL, LOADS>(wgs, 1, src, src16, times, sink, st);
    const Res d = run<KB, true, FILL, LOADS>(wgs, 1, src, src16, times, sink, st);
    printf("%-6s %3d KB  wgs %5d  loads/lane %d | back-to-back: straight %6.2f / %6.2f us  looped %6.2f / %6.2f | after a 64 MB stream: straight %6.2f / %6.2f  looped %6.2f / %6.2f   (region median / p90 per wave) | launch us (events): %6.2f %6.2f %6.2f %6.2f\n",
           FILL ? "v_add3" : "s_nop", KB, wgs, LOADS, a.med, a.p90, b.med, b.p90, c.med, c.p90, d.med, d.p90, a.launch_us, b.launch_us, c.launch_us, d.launch_us);
    fflush(stdout);
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t bytes = (size_t)1 << 30;
    uint4* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    u64* times; CK(hipMalloc(&times, 8 * 4 * 8192)); u32* sink; CK(hipMalloc(&sink, 64));
    printf("ifetch_probe: per-wave time of a code region of N KB, executed as one straight run (cold lines) or as a looped 1 KB body\n");
    // one wave per SIMD (256 workgroups of 4 waves), no loads: pure fetch behaviour
    printf("-- 256 workgroups (one wave per SIMD), no loads in flight\n");
    row<2, 0, 0>(256, src, bytes / 16, times, sink, st);
    row<8, 0, 0>(256, src, bytes / 16, times, sink, st);
    row<32, 0, 0>(256, src, bytes / 16, times, sink, st);
    row<2, 1, 0>(256, src, bytes / 16, times, sink, st);
    row<8, 1, 0>(256, src, bytes / 16, times, sink, st);
    row<32, 1, 0>(256, src, bytes / 16, times, sink, st);
    // six waves per SIMD like the decode kernel (1536 workgroups), 0 and 4 KB per wave in flight (4 loads of 16 B per lane = 24 MB per launch)
    printf("-- 1536 workgroups (six waves per SIMD)\n");
    row<8, 1, 0>(1536, src, bytes / 16, times, sink, st);
    row<8, 1, 4>(1536, src, bytes / 16, times, sink, st);
    row<8, 0, 4>(1536, src, bytes / 16, times, sink, st);
    row<2, 1, 4>(1536, src, bytes / 16, times, sink, st);
    return 0;
}
