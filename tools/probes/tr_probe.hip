// ds_read_b64_tr_b16 (gfx950): which lane receives what?  Every lane of a wave supplies the address of 4 consecutive f16 of
// a row-major [16 rows][64] image whose element (r, c) holds the value 64 r + c; lane l = 16 g + i supplies row 4 g + i / 4,
// columns 4 (i % 4) .. + 3.  Prints what each lane gets back.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 hx4 __attribute__((__vector_size__(8)));
__global__ void k(float* out)
{
    __shared__ __attribute__((aligned(16))) f16 lds[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (f16)(float)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    hx4 hv = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hx4*)(lds + (4 * g + (i >> 2)) * 64 + 4 * (i & 3)));
    f16x4 v = __builtin_bit_cast(f16x4, hv);
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (float)v[j];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++)
    {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; j++) printf("  (r %2d, c %2d)", (int)h[l * 4 + j] / 64, (int)h[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
