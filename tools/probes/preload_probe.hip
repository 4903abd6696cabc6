// Kernel-argument preload (-mllvm -amdgpu-kernarg-preload-count=N): does it work on this stack (gfx950, ROCm 7.2), and
// what does it take off "entry -> first dependent memory access"?  The same source is built twice (with / without the
// flag); each binary runs a 256-workgroup kernel whose first action depends on its arguments (a store through a pointer
// argument at an offset argument), stamps s_memrealtime at entry and right after that store has been issued, and reports
// (a) correctness of every argument as seen on the device, (b) the median entry->issued time, (c) the time per kernel of a
// 200-kernel chain in a captured graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k(int* out, u64* stamps, int a0, int a1, int a2, int a3, u64 a4, u64 a5, int a6, int a7)
{
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const int b = blockIdx.x;
    int v = a0 + 3 * a1 + 5 * a2 + 7 * a3 + (int)(a4 & 0xffff) + (int)(a5 >> 40) + 11 * a6 + 13 * a7;
    if (threadIdx.x == 0) out[b + a6] = v;
    asm volatile("" ::: "memory");
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[b * 2] = t0; stamps[b * 2 + 1] = t1; }
}

int main()
{
    const int WGS = 256;
    int* out; u64* stamps; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&stamps, WGS * 16)); CK(hipMemset(out, 0, 4096));
    const int a0 = 1, a1 = 2, a2 = 3, a3 = 4, a6 = 5, a7 = 6; const u64 a4 = 0x1234, a5 = 0x77ull << 40;
    const int want = a0 + 3 * a1 + 5 * a2 + 7 * a3 + 0x1234 + 0x77 + 11 * a6 + 13 * a7;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<double> med;
    for (int rep = 0; rep < 20; rep++)
    {
        hipLaunchKernelGGL(k, dim3(WGS), dim3(1024), 0, st, out, stamps, a0, a1, a2, a3, a4, a5, a6, a7);
        CK(hipStreamSynchronize(st));
        std::vector<u64> h(WGS * 2); CK(hipMemcpy(h.data(), stamps, WGS * 16, hipMemcpyDeviceToHost));
        std::vector<double> d; for (int b = 0; b < WGS; b++) d.push_back((double)(h[b * 2 + 1] - h[b * 2]) / 100.0);
        std::sort(d.begin(), d.end()); med.push_back(d[WGS / 2]);
    }
    int h[WGS + 8]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    bool ok = true; for (int b = 0; b < WGS; b++) ok = ok && h[b + a6] == want;
    std::sort(med.begin(), med.end());
    printf("arguments on the device: %s;  entry -> dependent store issued: median %.2f us (min %.2f, max %.2f over 20 launches)\n",
           ok ? "ok" : "WRONG", med[10], med[0], med[19]);
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k, dim3(WGS), dim3(1024), 0, st, out, stamps, a0, a1, a2, a3, a4, a5, a6, a7);
    CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 5; r++) CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("200-kernel chain in a graph: %.2f us per kernel\n", ms * 1e3 / 1000.0);
    return 0;
}
