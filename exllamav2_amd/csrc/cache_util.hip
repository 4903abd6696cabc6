// cache_util.hip -- KV-cache utilities of the dynamic generator's data path for gfx950, + their C ABI:
//   * FP8 cache codec (ext_cache.cpp:14-78 -> cuda/cache.cu:20-142): FP8 = the upper byte of each fp16 (E5M2 by
//     truncation), converted over a token range of a [batch, seq, kv_heads, head_dim] cache;
//   * cache_rotate (cuda/cache.cu:499-576, called by the defragmenter, generator/dynamic.py:1350-1471): cyclic move of
//     whole cache pages  temp <- page[o0]; page[o_i] <- page[o_i+1]; page[o_last] <- temp;
//   * count_match (ext_cache.cpp:285-302): length of the common prefix of two int64 token rows (host).
// All HBM-bound byte movers: 16-byte accesses, >> 256 workgroups.  The rotation keeps each thread's 16 bytes of page
// o0 in registers, so the reference's round trip through the `temp` page in HBM (2 x page bytes) is not made.
#include "hw.h"
#include "errors.h"

// ---- FP8 <-> FP16 ---------------------------------------------------------------------------------------------------

DEV u32 fp8_pack2(u32 v) { return ((v & 0xff000000u) >> 16) | ((v & 0x0000ff00u) >> 8); }     // two halves -> two bytes
DEV u32 fp8_unpack2(u32 v) { return ((v & 0xff00u) << 16) | ((v & 0x00ffu) << 8); }           // two bytes -> two halves

// one thread = 8 consecutive elements of row y; x0/x1 are multiples of 8 (host)
KERNEL void __launch_bounds__(256) fp16_to_fp8_kernel(const f16* in, u8* out, long long stride, int x0, int x1)
{
    const int x = x0 + (bid_x() * 256 + tid()) * 8;
    if (x >= x1) return;
    const size_t o = (size_t)bid_y() * (size_t)stride + (size_t)x;
    const u32x4 v = *(const u32x4*)(in + o);
    u32x2 r;
    r[0] = fp8_pack2(v[0]) | (fp8_pack2(v[1]) << 16);
    r[1] = fp8_pack2(v[2]) | (fp8_pack2(v[3]) << 16);
    *(u32x2*)(out + o) = r;
}

KERNEL void __launch_bounds__(256) fp8_to_fp16_kernel(const u8* in, f16* out, long long stride, int x0, int x1)
{
    const int x = x0 + (bid_x() * 256 + tid()) * 8;
    if (x >= x1) return;
    const size_t o = (size_t)bid_y() * (size_t)stride + (size_t)x;
    const u32x2 v = *(const u32x2*)(in + o);
    u32x4 r;
    r[0] = fp8_unpack2(v[0]); r[1] = fp8_unpack2(v[0] >> 16);
    r[2] = fp8_unpack2(v[1]); r[3] = fp8_unpack2(v[1] >> 16);
    *(u32x4*)(out + o) = r;
}

// ---- page rotation --------------------------------------------------------------------------------------------------

#define ROT_DEPTH 4     // pages requested ahead of the stores

// Thread t owns the 16-byte units {t, t + T, ...} of every page (T = threads of the grid).  For its unit it walks the
// rotation; a page is always read (step i-1 or earlier) before the same thread overwrites it (step i), and no other
// thread touches those bytes, so no synchronisation is needed and up to ROT_DEPTH page reads are in flight per thread.
KERNEL void __launch_bounds__(256) cache_rotate_kernel(u8* cache, const int* order, long long page_bytes, int n)
{
    const long long units = page_bytes >> 4;
    const long long T = (long long)gdim_x() * 256;
    for (long long u = (long long)bid_x() * 256 + tid(); u < units; u += T)
    {
        const size_t off = (size_t)u * 16;
        const u32x4 first = *(const u32x4*)(cache + (size_t)page_bytes * (size_t)order[0] + off);
        for (int i = 0; i < n - 1; i += ROT_DEPTH)
        {
            u32x4 v[ROT_DEPTH];
            #pragma unroll
            for (int j = 0; j < ROT_DEPTH; j++)
                if (i + j < n - 1) v[j] = *(const u32x4*)(cache + (size_t)page_bytes * (size_t)order[i + j + 1] + off);
            #pragma unroll
            for (int j = 0; j < ROT_DEPTH; j++)
                if (i + j < n - 1) *(u32x4*)(cache + (size_t)page_bytes * (size_t)order[i + j] + off) = v[j];
        }
        *(u32x4*)(cache + (size_t)page_bytes * (size_t)order[n - 1] + off) = first;
    }
}

extern "C" {

// in/out: [batch, seq, kv_heads, head_dim]; row_stride = seq * kv_heads * head_dim elements, token_size = kv_heads *
// head_dim; converts tokens [offset, offset + width) of the first `batch_size` rows.  Range rounding to 8 elements as in
// array_fp16_to_fp8_cuda (cache.cu:86-100).
static int fp8_range(long long row_stride, int token_size, int offset, int width, int* x0, int* x1)
{
    const long long mn = (long long)offset * token_size, mx = mn + (long long)width * token_size;
    const long long a = mn / 8 * 8, b = a + (mx - a + 7) / 8 * 8;
    if (b > row_stride || b > 0x7fffffffLL) return 0;
    *x0 = (int)a; *x1 = (int)b;
    return 1;
}

int exl2_fp16_to_fp8(const void* in, void* out, int batch_size, long long row_stride, int token_size, int offset, int width,
                     void* stream)
{
    EXL2_REQUIRE(in && out, "fp16_to_fp8: null argument");
    EXL2_REQUIRE(batch_size >= 0 && token_size > 0 && offset >= 0 && width >= 0 && row_stride % 8 == 0,
                 "fp16_to_fp8: bad shape (row stride %lld, token size %d, offset %d, width %d)", row_stride, token_size, offset, width);
    int x0, x1;
    EXL2_REQUIRE(fp8_range(row_stride, token_size, offset, width, &x0, &x1), "fp16_to_fp8: range exceeds the cache row");
    if (batch_size == 0 || x1 <= x0) return EXL2_OK;
    LAUNCH(fp16_to_fp8_kernel, dim3((unsigned)(((x1 - x0) / 8 + 255) / 256), (unsigned)batch_size), dim3(256), 0, stream,
           (const f16*)in, (u8*)out, row_stride, x0, x1);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_fp8_to_fp16(const void* in, void* out, int batch_size, long long row_stride, int token_size, int offset, int width,
                     void* stream)
{
    EXL2_REQUIRE(in && out, "fp8_to_fp16: null argument");
    EXL2_REQUIRE(batch_size >= 0 && token_size > 0 && offset >= 0 && width >= 0 && row_stride % 8 == 0,
                 "fp8_to_fp16: bad shape (row stride %lld, token size %d, offset %d, width %d)", row_stride, token_size, offset, width);
    int x0, x1;
    EXL2_REQUIRE(fp8_range(row_stride, token_size, offset, width, &x0, &x1), "fp8_to_fp16: range exceeds the cache row");
    if (batch_size == 0 || x1 <= x0) return EXL2_OK;
    LAUNCH(fp8_to_fp16_kernel, dim3((unsigned)(((x1 - x0) / 8 + 255) / 256), (unsigned)batch_size), dim3(256), 0, stream,
           (const u8*)in, (f16*)out, row_stride, x0, x1);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// cache: num_pages pages of page_bytes each (any dtype, contiguous); order: int32[n] on the device, distinct page ids.
// The reference's `temp` page is not needed (see the kernel); its size check is the host mirror's.
int exl2_cache_rotate(void* cache, const int* order, long long page_bytes, int n, void* stream)
{
    EXL2_REQUIRE(cache && order, "cache_rotate: null argument");
    EXL2_REQUIRE(page_bytes > 0 && page_bytes % 16 == 0, "cache_rotate: page size %lld bytes must be a multiple of 16", page_bytes);
    if (n <= 1) return EXL2_OK;
    const long long wgs = (page_bytes / 16 + 255) / 256;
    LAUNCH(cache_rotate_kernel, dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, stream, (u8*)cache, order, page_bytes, n);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// host: number of leading positions at which two int64 rows agree, at most min(max_a, len_b)
int exl2_count_match(const long long* a, const long long* b, int max_a, int len_b, int* match)
{
    EXL2_REQUIRE(a && b && match, "count_match: null argument");
    const int m = max_a < len_b ? max_a : len_b;
    int i = 0;
    while (i < m && a[i] == b[i]) i++;
    *match = i;
    return EXL2_OK;
}

}  // extern "C"
