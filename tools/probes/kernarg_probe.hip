// Does a kernel with more than 4 KiB of by-value arguments launch and read them correctly on this stack (gfx950, ROCm 7.2)?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int* out) { if (threadIdx.x == 0) { long s = 0; for (int i = 0; i < N; i++) s += b.v[i]; out[blockIdx.x] = (int)s; } }
template <int N> int run()
{
    Big<N> b; long want = 0; for (int i = 0; i < N; i++) { b.v[i] = i * 7 + 1; want += b.v[i]; }
    int* d; if (hipMalloc(&d, 64) != hipSuccess) return 2;
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k<N>, dim3(4), dim3(64), 0, 0, b, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    int h[4] = {0, 0, 0, 0}; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("kernarg %6zu bytes: launch %s sync %s result %s\n", sizeof(b) + 8, hipGetErrorString(e), hipGetErrorString(e2), (h[0] == (int)want && h[3] == (int)want) ? "ok" : "WRONG");
    hipFree(d); return 0;
}
int main() { run<900>(); run<1500>(); run<4000>(); run<16000>(); return 0; }
