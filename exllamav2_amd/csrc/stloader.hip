// stloader.hip -- the load path of the C ABI: bytes of a safetensors file -> host or device memory,
// and the two CPU column re-orderings the reference applies to freshly loaded tensors.  Host code only (no kernels).
//
// Replaces ext_stloader.cpp:11-157 (stloader_read), :160-184 (tensor_remap), :186-219 (tensor_remap_4bit).
//
// The reference reads a device-bound tensor into a pageable malloc(size) bounce buffer (8 threads, interleaved 1 MiB blocks)
// and issues cudaMemcpyAsync from pageable memory block by block, which the runtime stages through its own pinned buffer
// synchronously.  Here:
//   * host target: the readers pread straight into the tensor, each a CONTIGUOUS share of the range (sequential read-ahead
//     per thread instead of eight interleaved streams);
//   * device target: a ring of PINNED slots per device, allocated once and kept (ST_SLOTS x ST_CHUNK = 64 MiB).  Reader r takes
//     chunks r, r + R, ...; chunk c lives in slot c % ST_SLOTS.  The calling thread issues one hipMemcpyAsync per chunk, in
//     file order, on the caller's stream, and records an event per slot; a slot is handed back to the readers when its
//     event has completed.  The copies are true DMA from pinned memory and overlap the reads of the following chunks; peak
//     host memory is the ring, not the tensor.  The call returns when the last copy has completed (the reference ends with
//     cudaDeviceSynchronize; a stream synchronize is what the contract needs).
//   * a 288 GB device takes whole 70B checkpoints per GPU: the ring is sized for the PCIe link (8 readers x 4 MiB in flight),
//     not for the tensor.
#include "hw.h"
#include "errors.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <atomic>

#define ST_CHUNK   ((size_t)4 << 20)
#define ST_READERS 8
#define ST_SLOTS   16                    // 2 x readers: see the hand-back rule in exl2_stloader_read

// pread until done (short reads are legal); false on error / EOF inside the range
static bool read_fully(int fd, uint8_t* dst, size_t n, uint64_t off)
{
    while (n)
    {
        const ssize_t r = pread(fd, dst, n, (off_t)off);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;
        dst += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

struct PinnedRing { uint8_t* slot[ST_SLOTS]; hipEvent_t done[ST_SLOTS]; bool ready; };
static PinnedRing g_ring[EXL2_MAX_DEVICES];
static std::mutex g_ring_mutex[EXL2_MAX_DEVICES];        // one load per device at a time (the ring is shared)

static int ring_for_device(int dev, PinnedRing** out)
{
    PinnedRing& R = g_ring[dev];
    if (!R.ready)
    {
        for (int i = 0; i < ST_SLOTS; i++)
        {
            HIP_TRY(hipHostMalloc((void**)&R.slot[i], ST_CHUNK, 0));
            HIP_TRY(hipEventCreateWithFlags(&R.done[i], hipEventDisableTiming));
        }
        R.ready = true;
    }
    *out = &R;
    return EXL2_OK;
}

extern "C" int exl2_stloader_read(const char* filename, uint64_t offset, uint64_t size, void* target, int target_device,
                                  void* stream)
{
    EXL2_REQUIRE(filename && (target || !size), "stloader_read: null argument");
    if (!size) return EXL2_OK;
    EXL2_REQUIRE(target_device < EXL2_MAX_DEVICES, "stloader_read: device index %d", target_device);
    const int fd = open(filename, O_RDONLY);
    if (fd < 0) EXL2_FAIL(EXL2_E_INVALID, "stloader_read: cannot open %s: %s", filename, strerror(errno));

    if (target_device < 0)
    {
        // ---- host target: contiguous shares, straight into the tensor -------------------------------------------------------
        const size_t n_blocks = (size + ST_CHUNK - 1) / ST_CHUNK;
        const int readers = (int)(n_blocks < ST_READERS ? n_blocks : ST_READERS);
        std::atomic<bool> failed{false};
        auto work = [&](int r)
        {
            const size_t b0 = n_blocks * r / readers, b1 = n_blocks * (r + 1) / readers;
            const size_t a = b0 * ST_CHUNK, b = b1 * ST_CHUNK < size ? b1 * ST_CHUNK : size;
            if (a < b && !read_fully(fd, (uint8_t*)target + a, b - a, offset + a)) failed = true;
        };
        std::vector<std::thread> th;
        for (int r = 1; r < readers; r++) th.emplace_back(work, r);
        work(0);
        for (auto& t : th) t.join();
        close(fd);
        if (failed) EXL2_FAIL(EXL2_E_INVALID, "stloader_read: I/O error reading tensor (%s, %llu bytes at %llu)", filename,
                              (unsigned long long)size, (unsigned long long)offset);
        return EXL2_OK;
    }

    // ---- device target: pinned ring, copies in file order on the caller's stream ---------------------------------------------
    DeviceGuard guard(target_device);
    std::lock_guard<std::mutex> one_load(g_ring_mutex[target_device]);
    PinnedRing* R = nullptr;
    { const int rc = ring_for_device(target_device, &R); if (rc) { close(fd); return rc; } }

    const size_t n_chunks = (size + ST_CHUNK - 1) / ST_CHUNK;
    const int readers = (int)(n_chunks < ST_READERS ? n_chunks : ST_READERS);
    std::mutex mtx;
    std::condition_variable cv;
    std::vector<uint8_t> filled(n_chunks, 0);
    size_t retired = 0;                          // chunks [0, retired) have left their slots (copy completed)
    bool failed = false;

    auto reader = [&](int r)
    {
        for (size_t c = (size_t)r; c < n_chunks; c += (size_t)readers)
        {
            {
                std::unique_lock<std::mutex> lk(mtx);
                cv.wait(lk, [&] { return failed || c < retired + ST_SLOTS; });      // slot c % ST_SLOTS is free again
                if (failed) return;
            }
            const size_t a = c * ST_CHUNK, n = (a + ST_CHUNK <= size) ? ST_CHUNK : size - a;
            const bool ok = read_fully(fd, R->slot[c % ST_SLOTS], n, offset + a);
            std::lock_guard<std::mutex> lk(mtx);
            if (!ok) failed = true; else filled[c] = 1;
            cv.notify_all();
            if (!ok) return;
        }
    };
    std::vector<std::thread> th;
    for (int r = 0; r < readers; r++) th.emplace_back(reader, r);

    // Hand-back rule: while this thread waits for chunk c the readers may be working on chunks up to c + readers - 1, which
    // need chunks <= c + readers - 1 - ST_SLOTS retired; retiring everything below c - (ST_SLOTS - readers) before the wait
    // guarantees it (no deadlock), and with ST_SLOTS = 2 x readers those copies were issued >= `readers` chunks ago.
    int rc = EXL2_OK;
    hipError_t herr = hipSuccess;
    size_t issued = 0;
    auto retire_to = [&](size_t upto)
    {
        while (herr == hipSuccess && retired < upto)
        {
            herr = hipEventSynchronize(R->done[retired % ST_SLOTS]);
            std::lock_guard<std::mutex> lk(mtx);
            retired++;
            cv.notify_all();
        }
    };
    for (size_t c = 0; c < n_chunks && herr == hipSuccess; c++)
    {
        if (c + (size_t)readers > ST_SLOTS) retire_to(c + (size_t)readers - ST_SLOTS);
        {
            std::unique_lock<std::mutex> lk(mtx);
            cv.wait(lk, [&] { return failed || filled[c]; });
            if (failed) break;
        }
        const size_t a = c * ST_CHUNK, n = (a + ST_CHUNK <= size) ? ST_CHUNK : size - a;
        herr = hipMemcpyAsync((uint8_t*)target + a, R->slot[c % ST_SLOTS], n, hipMemcpyHostToDevice, (hipStream_t)stream);
        if (herr == hipSuccess) herr = hipEventRecord(R->done[c % ST_SLOTS], (hipStream_t)stream);
        if (herr == hipSuccess) issued = c + 1;
    }
    if (herr != hipSuccess)
    {
        std::lock_guard<std::mutex> lk(mtx);
        failed = true;
        cv.notify_all();
    }
    for (auto& t : th) t.join();
    close(fd);
    // every issued copy must have left its slot before the ring serves the next call
    const hipError_t serr = hipStreamSynchronize((hipStream_t)stream);
    (void)issued;
    if (herr != hipSuccess || serr != hipSuccess)
    {
        exl2_set_error("stloader_read: host-to-device copy failed: %s", hipGetErrorString(herr != hipSuccess ? herr : serr));
        rc = EXL2_E_HIP;
    }
    else if (failed)
    {
        exl2_set_error("stloader_read: I/O error reading tensor (%s, %llu bytes at %llu)", filename, (unsigned long long)size,
                       (unsigned long long)offset);
        rc = EXL2_E_INVALID;
    }
    return rc;
}

// tensor_remap (ext_stloader.cpp:160-184): in place, new[r][c] = old[r][index[c]]  (int32 [rows, cols], index int32 [cols])
extern "C" int exl2_tensor_remap(int32_t* tensor, int rows, int cols, const int32_t* index)
{
    EXL2_REQUIRE(rows >= 0 && cols >= 0 && (tensor || !rows || !cols) && (index || !cols), "tensor_remap: null argument");
    for (int c = 0; c < cols; c++)
        EXL2_REQUIRE(index[c] >= 0 && index[c] < cols, "tensor_remap: index[%d] = %d outside [0, %d)", c, index[c], cols);
    std::vector<int32_t> row((size_t)cols);
    for (int r = 0; r < rows; r++)
    {
        int32_t* a = tensor + (size_t)r * cols;
        memcpy(row.data(), a, sizeof(int32_t) * (size_t)cols);
        for (int c = 0; c < cols; c++) a[c] = row[(size_t)index[c]];
    }
    return EXL2_OK;
}

// tensor_remap_4bit (ext_stloader.cpp:186-219): the same on 4-bit values packed eight to an int32 along the columns
// (q_scale): tensor int32 [rows, cols / 8], index int32 [cols]
extern "C" int exl2_tensor_remap_4bit(int32_t* tensor, int rows, int cols, const int32_t* index)
{
    EXL2_REQUIRE(rows >= 0 && cols >= 0 && !(cols & 7) && (tensor || !rows || !cols) && (index || !cols),
                 "tensor_remap_4bit: bad argument (cols = %d must be a multiple of 8)", cols);
    for (int c = 0; c < cols; c++)
        EXL2_REQUIRE(index[c] >= 0 && index[c] < cols, "tensor_remap_4bit: index[%d] = %d outside [0, %d)", c, index[c], cols);
    const int words = cols >> 3;
    std::vector<uint8_t> nib((size_t)cols);
    for (int r = 0; r < rows; r++)
    {
        uint32_t* a = (uint32_t*)tensor + (size_t)r * words;
        for (int w = 0; w < words; w++)
            for (int b = 0; b < 8; b++) nib[(size_t)w * 8 + b] = (uint8_t)((a[w] >> (4 * b)) & 0xFu);
        for (int w = 0; w < words; w++)
        {
            uint32_t v = 0;
            for (int b = 0; b < 8; b++) v |= (uint32_t)nib[(size_t)index[w * 8 + b]] << (4 * b);
            a[w] = v;
        }
    }
    return EXL2_OK;
}
