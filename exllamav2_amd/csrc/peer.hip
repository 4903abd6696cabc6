// peer.hip -- copies between the devices of a single-process tensor-parallel split, + their C ABI.
// The reference's tp_gather / tp_broadcast (ext_tp.cpp:129-287) move every exchange through a pinned host buffer: one
// cudaMemcpyAsync down per source device, one up per target device.  On an MI355X node the devices reach each other over
// xGMI, so the device targets are written directly, slice by slice, by a 2-D copy on the TARGET device's stream (row r of a
// slice goes to row r of the gathered matrix at the slice's column offset: a strided destination).  The host buffer is
// still filled when the caller asks for the gathered matrix on the host (broadcast_type_target < 0: logits for the sampler).
// Pure byte movers; no kernels.
#include "hw.h"
#include "errors.h"

extern "C" {

// `height` rows of `width_bytes`: row r from src + r * spitch to dst + r * dpitch, asynchronously on `stream` (a stream of
// the device the caller is on).  Either side may be pinned host memory, memory of this device or of a peer device.
int exl2_memcpy_2d_async(void* dst, long long dpitch, const void* src, long long spitch, long long width_bytes,
                         long long height, void* stream)
{
    EXL2_REQUIRE(dst && src && dpitch >= width_bytes && spitch >= width_bytes && width_bytes >= 0 && height >= 0,
                 "memcpy_2d_async: bad argument");
    if (width_bytes == 0 || height == 0) return EXL2_OK;
    if (dpitch == width_bytes && spitch == width_bytes)
        HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(width_bytes * height), hipMemcpyDefault, (hipStream_t)stream));
    else
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width_bytes, (size_t)height, hipMemcpyDefault,
                                 (hipStream_t)stream));
    return EXL2_OK;
}

// Enables peer access between every ordered pair of `devices` (make_tp_context time).  Returns the number of ordered pairs
// that cannot reach each other directly (their copies are staged by the runtime); < 0 on error.
int exl2_enable_peer_access(const int* devices, int n)
{
    EXL2_REQUIRE(n >= 0 && (n == 0 || devices), "enable_peer_access: bad argument");
    int prev = 0, unreachable = 0;
    HIP_TRY(hipGetDevice(&prev));
    for (int i = 0; i < n; i++)
    {
        for (int j = 0; j < n; j++)
        {
            if (i == j || devices[i] == devices[j]) continue;
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, devices[i], devices[j]));
            if (!can) { unreachable++; continue; }
            HIP_TRY(hipSetDevice(devices[i]));
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipSetDevice(prev); HIP_TRY(e); }
            (void)hipGetLastError();
        }
    }
    HIP_TRY(hipSetDevice(prev));
    return unreachable;
}

}
