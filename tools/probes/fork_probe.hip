// Do the two branches of a forked, captured HIP graph run concurrently on MI355X, and what does a hand-off through a
// counter in memory cost compared with a kernel boundary?  (Background: csrc/chain_sync.h -- the overlapped decode chain.)
//
// Chain of N kernels K0..K(N-1), 256 workgroups x 1024 threads, ~128 VGPRs (one workgroup fills a CU, like the decode
// q_gemm).  Each kernel: [optional wait for its predecessor's counter] -> busy ~WORK us -> signal.  Three arrangements:
//   serial   : one stream, no counters (kernel boundaries only)
//   forked   : even kernels on stream A, odd on stream B (captured fork/join), dependencies through counters
//   forked+g : the same plus a one-wave gate ahead of K1 that waits until all workgroups of K0 have arrived
// Every workgroup stamps s_memrealtime (100 MHz) at entry, after its wait, and at exit; the host prints per-kernel
// [first entry, last entry, first pass, last exit] relative to K0's first entry, and the number of waits that gave up.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define STRIDE 320          // u32 per kernel: counter line + 8 'go' lines (+ pad)
#define SPIN_LIMIT (1 << 15)

struct Args { const u32* wait; u32 target; u32* signal; u32* arrive; u64* stamps; int k; int work_ticks; int reset_waited; u32 my_total; int go_mode; int poll_sleep; int do_inv; int do_wbl2; };

__global__ void __launch_bounds__(1024) link_kernel(const Args a)
{
    __shared__ float sink[1024];
    const int b = blockIdx.x;
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    if (a.arrive && threadIdx.x == 0) __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // hold ~128 VGPRs so that one workgroup owns the CU
    float r[96];
    #pragma unroll
    for (int i = 0; i < 96; i++) r[i] = (float)(threadIdx.x + i);
    int gave_up = 0;
    if (a.wait)
    {
        if (a.go_mode)
        {
            // one poller per workgroup, on the copy of the "go" word of its residue class; everybody else sleeps at the barrier
            if (threadIdx.x < 64)
            {
                const u32* go = a.go_mode == 2 ? a.wait : a.wait + 32 * (1 + (b & 7));              // 8 copies, one 128-byte line each
                const int need = a.go_mode == 2 ? (int)a.target : 1;
                int spins = 0;
                while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need && ++spins < SPIN_LIMIT)
                    for (int z = 0; z < a.poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
                gave_up = spins >= SPIN_LIMIT;
            }
            __syncthreads();
        }
        else
        {
            int spins = 0;
            while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (int)a.target && ++spins < SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(2);
            gave_up = spins >= SPIN_LIMIT;
        }
        if (a.do_inv) asm volatile("buffer_inv sc1" ::: "memory");
    }
    u64 t1 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t1 < (u64)a.work_ticks)
    {
        #pragma unroll
        for (int i = 0; i < 96; i++) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0; 
    #pragma unroll
    for (int i = 0; i < 96; i++) s += r[i];
    sink[threadIdx.x] = s;
    __syncthreads();
    u64 t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
    {
        u64* st = a.stamps + ((size_t)a.k * 256 + b) * 4;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = (u64)gave_up + (sink[5] == 12345.0f ? 1 : 0) * 0;
        if (a.do_wbl2) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.signal)
        {
            const u32 old = __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.go_mode == 1 && old + 1 == a.my_total)
            {
                // last workgroup out: publish "go" (8 copies) for the consumer, zero this counter and the go words this kernel waited on
                for (int c = 0; c < 8; c++) __hip_atomic_store(a.signal + 32 * (1 + c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.signal, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.wait) for (int c = 0; c < 8; c++) __hip_atomic_store((u32*)a.wait + 32 * (1 + c), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // two-graph mode: the last workgroup of this kernel knows every workgroup has passed the wait -> the waited
            // counter can be zeroed for the next replay (nobody else reads it)
            if (a.go_mode != 1 && a.reset_waited && a.wait && old + 1 == a.my_total) __hip_atomic_store((u32*)a.wait, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64) gate_kernel(u32* arrived, u32 target, int reset)
{
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (int)target && ++spins < SPIN_LIMIT)
        __builtin_amdgcn_s_sleep(2);
    if (reset && threadIdx.x == 0) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv)
{
    const int N = 8, WGS = 256;
    const int work_ticks = argc > 1 ? atoi(argv[1]) : 800;           // 8 us
    u32* flags; CK(hipMalloc(&flags, (N + 2) * STRIDE * 4));
    u64* stamps; CK(hipMalloc(&stamps, (size_t)N * 256 * 4 * 8));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t e0, e1; CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    std::vector<u64> h((size_t)N * 256 * 4);
    for (int mode = 0; mode < 3; mode++)
    {
        const char* names[3] = {"serial (one stream, kernel boundaries)", "forked (two streams, counters)", "forked + gate"};
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeRelaxed));
        CK(hipMemsetAsync(flags, 0, (N + 2) * STRIDE * 4, sa));
        if (mode) { CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0)); }
        for (int k = 0; k < N; k++)
        {
            Args a; memset(&a, 0, sizeof(a));
            a.k = k; a.work_ticks = work_ticks; a.stamps = stamps; a.do_inv = 1; a.do_wbl2 = 1;
            hipStream_t st = sa;
            if (mode)
            {
                a.wait = k ? flags + (k - 1) * STRIDE : nullptr; a.target = WGS; a.signal = flags + k * STRIDE;
                a.arrive = (mode == 2 && k == 0) ? flags + (N + 1) * STRIDE : nullptr;
                st = (k & 1) ? sb : sa;
                if (mode == 2 && k == 1) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, sb, flags + (N + 1) * STRIDE, (u32)WGS, 0);
            }
            hipLaunchKernelGGL(link_kernel, dim3(WGS), dim3(1024), 0, st, a);
        }
        if (mode) { CK(hipEventRecord(e1, sb)); CK(hipStreamWaitEvent(sa, e1, 0)); }
        CK(hipStreamEndCapture(sa, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) { CK(hipGraphLaunch(exec, sa)); CK(hipStreamSynchronize(sa)); }
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        u64 base = ~0ull;
        for (int b = 0; b < WGS; b++) if (h[(size_t)b * 4] < base) base = h[(size_t)b * 4];
        printf("== %s, work %.1f us per kernel\n", names[mode], work_ticks / 100.0);
        int total_gave_up = 0;
        for (int k = 0; k < N; k++)
        {
            u64 e_first = ~0ull, e_last = 0, p_first = ~0ull, x_last = 0; int gu = 0;
            for (int b = 0; b < WGS; b++)
            {
                const u64* s = &h[((size_t)k * 256 + b) * 4];
                if (s[0] < e_first) e_first = s[0]; if (s[0] > e_last) e_last = s[0];
                if (s[1] < p_first) p_first = s[1]; if (s[2] > x_last) x_last = s[2];
                gu += (int)s[3];
            }
            total_gave_up += gu;
            printf("  K%d: entry %7.2f .. %7.2f  first past wait %7.2f  last exit %7.2f us  (waits given up: %d)\n", k,
                   (double)(long long)(e_first - base) / 100.0, (double)(long long)(e_last - base) / 100.0,
                   (double)(long long)(p_first - base) / 100.0, (double)(long long)(x_last - base) / 100.0, gu);
        }
        printf("  total waits given up: %d\n", total_gave_up);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    // ---- two graphs, one per stream, launched side by side (no fork inside a graph); counters zeroed by their consumers --------
    const int cfgs[6][5] = {{1, 1, 2, 1, 1}, {1, 1, 2, 0, 1}, {1, 1, 2, 0, 0}, {1, 1, 1, 0, 0}, {1, 2, 2, 0, 0}, {1, 2, 2, 0, 1}};
    for (int ci = 0; ci < 6; ci++)
    {
        const int gate = cfgs[ci][0], go_mode = cfgs[ci][1], poll_sleep = cfgs[ci][2], do_inv = cfgs[ci][3], do_wbl2 = cfgs[ci][4];
        CK(hipMemset(flags, 0, (N + 2) * STRIDE * 4));
        hipGraph_t ga, gb; hipGraphExec_t xa, xb;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeRelaxed));
        CK(hipStreamBeginCapture(sb, hipStreamCaptureModeRelaxed));
        for (int k = 0; k < N; k++)
        {
            Args a; memset(&a, 0, sizeof(a));
            a.k = k; a.work_ticks = work_ticks; a.stamps = stamps; a.reset_waited = 1; a.my_total = WGS; a.go_mode = go_mode; a.poll_sleep = poll_sleep; a.do_inv = do_inv; a.do_wbl2 = do_wbl2;
            a.wait = k ? flags + (k - 1) * STRIDE : nullptr; a.target = WGS; a.signal = flags + k * STRIDE;
            a.arrive = (gate && k == 0) ? flags + (N + 1) * STRIDE : nullptr;
            if (gate && k == 1) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, sb, flags + (N + 1) * STRIDE, (u32)WGS, 1);
            hipLaunchKernelGGL(link_kernel, dim3(WGS), dim3(1024), 0, (k & 1) ? sb : sa, a);
        }
        CK(hipStreamEndCapture(sa, &ga)); CK(hipStreamEndCapture(sb, &gb));
        CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++)
        {
            CK(hipGraphLaunch(xa, sa)); CK(hipGraphLaunch(xb, sb));
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            // the last kernel's own counter has no consumer: zero it here
            CK(hipMemset(flags + (N - 1) * STRIDE, 0, STRIDE * 4));
        }
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        u64 base = ~0ull;
        for (int b = 0; b < WGS; b++) if (h[(size_t)b * 4] < base) base = h[(size_t)b * 4];
        printf("== two graphs + gate, %s, poll sleep %d, buffer_inv %d, buffer_wbl2 %d, work %.1f us\n", go_mode == 1 ? "go-words (1 poller / workgroup)" : go_mode == 2 ? "counter polled by 1 wave / workgroup" : "every wave polls the counter", poll_sleep, do_inv, do_wbl2, work_ticks / 100.0);
        int total_gave_up = 0;
        for (int k = 0; k < N; k++)
        {
            u64 e_first = ~0ull, e_last = 0, p_first = ~0ull, p_last = 0, x_last = 0; int gu = 0;
            for (int b = 0; b < WGS; b++)
            {
                const u64* s = &h[((size_t)k * 256 + b) * 4];
                if (s[0] < e_first) e_first = s[0]; if (s[0] > e_last) e_last = s[0];
                if (s[1] < p_first) p_first = s[1]; if (s[1] > p_last) p_last = s[1]; if (s[2] > x_last) x_last = s[2];
                gu += (int)s[3];
            }
            total_gave_up += gu;
            printf("  K%d: entry %7.2f .. %7.2f  past wait %7.2f .. %7.2f  last exit %7.2f us  (waits given up: %d)\n", k,
                   (double)(long long)(e_first - base) / 100.0, (double)(long long)(e_last - base) / 100.0,
                   (double)(long long)(p_first - base) / 100.0, (double)(long long)(p_last - base) / 100.0,
                   (double)(long long)(x_last - base) / 100.0, gu);
        }
        printf("  total waits given up: %d\n", total_gave_up);
    }
    return 0;
}
