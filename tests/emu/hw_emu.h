// hw_emu.h -- CPU emulation of the primitives in exllamav2_amd/csrc/hw.h plus the handful of HIP runtime calls the
// library's host code makes.  TEST INFRASTRUCTURE ONLY: it lets the GPU-less `-m "not gpu"` tests drive the very same
// kernel and host sources (force-included ahead of hw.h, whose include guard it pre-empts) on tiny shapes, so indexing /
// layout / host-logic bugs surface before any GPU time is spent.  It is never built into, nor loadable by, the product
// package (exllamav2_amd/_lib.py loads only the gfx950 library and raises if it is missing).
//
// Model: one workgroup runs at a time; its blockDim.x threads are fibers (user-level contexts) scheduled round-robin on
// the launching OS thread; __syncthreads is a barrier over the live threads of the block; wave64 cross-lane operations
// go through a per-wave exchange buffer with a per-wave barrier.
#ifndef EXL2_HW_H
#define EXL2_HW_H
#define EXL2_EMU 1

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <mutex>
#include <vector>
#include <functional>
#include <algorithm>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t  i32;

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef u32      u32x2 __attribute__((ext_vector_type(2)));
typedef u32      u32x4 __attribute__((ext_vector_type(4)));

#define DEV  inline
#define HD   inline
#define KERNEL static
#define NOINLINE_DEV inline
#define __launch_bounds__(...)
#define ASM_MARK(id) do { } while (0)      // (hw.h: a numbered comment in the gfx950 code)
#define WAVE 64

using std::min;
using std::max;

// ---- runtime model ---------------------------------------------------------------------------------------------------
// Cooperative barrier: every thread of a workgroup is a FIBER on the launching OS thread (emu_runtime.cpp), so there is
// nothing to lock.  A fiber that has to wait records (barrier, generation) and yields; the scheduler does not resume it
// until the generation has moved -- one context switch per waiting thread per barrier, no kernel involvement.
struct EmuBarrier;
void emu_block_on(EmuBarrier* bar, unsigned gen);
struct EmuBarrier
{
    int live = 0, arrived = 0;
    unsigned gen = 0;
    void reset(int n) { live = n; arrived = 0; }
    void wait()
    {
        const unsigned g = gen;
        if (++arrived >= live) { arrived = 0; gen++; }
        else emu_block_on(this, g);
    }
    void leave()
    {
        live--;
        if (live > 0 && arrived >= live) { arrived = 0; gen++; }
    }
};

struct EmuWave
{
    EmuBarrier bar;
    alignas(16) unsigned char slot[64][64];
};

struct EmuDim { unsigned x, y, z; };

struct EmuCtx
{
    EmuBarrier block_bar;
    EmuWave wave[16];
    unsigned char* dyn_smem;
    EmuDim grid, block;
};

struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern int emu_tid_;
extern EmuDim emu_bid_;
extern EmuCtx* emu_ctx_;
void emu_launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
// every workgroup of a small grid alive at once (kernels whose workgroups wait for each other inside one launch); a thread
// polling another workgroup's writes calls emu_spin_yield() in its loop.  See emu_runtime.cpp.
void emu_launch_coop(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void emu_spin_yield();

DEV int lane_id() { return emu_tid_ & 63; }
DEV int wave_id() { return emu_tid_ >> 6; }
DEV int tid()     { return emu_tid_; }
DEV int nthreads(){ return (int)emu_ctx_->block.x; }
DEV int bid_x()   { return (int)emu_bid_.x; }
DEV int bid_y()   { return (int)emu_bid_.y; }
DEV int bid_z()   { return (int)emu_bid_.z; }
DEV int gdim_x()  { return (int)emu_ctx_->grid.x; }
DEV int gdim_y()  { return (int)emu_ctx_->grid.y; }

DEV void block_sync() { emu_ctx_->block_bar.wait(); }
DEV u32 uniform(u32 x) { return x; }
DEV int uniform(int x) { return x; }

// ---- bit casts / math ------------------------------------------------------------------------------------------------
DEV f16x2 as_h2(u32 x)   { return __builtin_bit_cast(f16x2, x); }
DEV u32   as_u32(f16x2 x) { return __builtin_bit_cast(u32, x); }
DEV f16   as_h(u16 x)    { return __builtin_bit_cast(f16, x); }
DEV u16   as_u16(f16 x)  { return __builtin_bit_cast(u16, x); }
DEV float as_f32(u32 x)  { return __builtin_bit_cast(float, x); }
DEV u32   f32_bits(float x) { return __builtin_bit_cast(u32, x); }

DEV f16 h_fma(f16 a, f16 b, f16 c) { return (f16)((double)a * (double)b + (double)c); }
DEV f16x2 h2_fma(f16x2 a, f16x2 b, f16x2 c) { return (f16x2){h_fma(a.x, b.x, c.x), h_fma(a.y, b.y, c.y)}; }
DEV f16 h_div_rn(f16 a, f16 b) { return (f16)((float)a / (float)b); }
DEV f16x2 h2_div_rn(f16x2 a, f16x2 b) { return (f16x2){h_div_rn(a.x, b.x), h_div_rn(a.y, b.y)}; }
DEV f16x2 h2_dup(f16 x) { return (f16x2){x, x}; }

DEV float fast_exp(float x) { return expf(x); }
DEV float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }
DEV float fast_rcp(float x) { return 1.0f / x; }

template <typename T> DEV void pin_scalar(T&) { }
template <typename T> DEV void pin_vector(T&) { }

// ---- cross-lane ------------------------------------------------------------------------------------------------------
template <typename T> DEV T emu_wave_read(T mine, int src_lane)
{
    static_assert(sizeof(T) <= 64, "exchange slot too small");
    EmuWave& w = emu_ctx_->wave[wave_id()];
    memcpy(w.slot[lane_id()], &mine, sizeof(T));
    w.bar.wait();
    T r;
    memcpy(&r, w.slot[src_lane & 63], sizeof(T));
    w.bar.wait();
    return r;
}
DEV u32 shfl_xor_u32(u32 v, int mask) { return emu_wave_read(v, lane_id() ^ mask); }
DEV float shfl_xor_f32(float v, int mask) { return emu_wave_read(v, lane_id() ^ mask); }
DEV u32 shfl_idx_u32(u32 v, int src) { return emu_wave_read(v, src); }
DEV void wave_sync() { EmuWave& w = emu_ctx_->wave[wave_id()]; w.bar.wait(); }
DEV u64 wave_ballot(bool pred)
{
    EmuWave& w = emu_ctx_->wave[wave_id()];
    const u32 mine = pred ? 1u : 0u;
    memcpy(w.slot[lane_id()], &mine, sizeof(mine));
    w.bar.wait();
    u64 m = 0;
    for (int l = 0; l < 64; l++) { u32 v; memcpy(&v, w.slot[l], sizeof(v)); if (v) m |= 1ull << l; }
    w.bar.wait();
    return m;
}
DEV float shfl_idx_f32(float v, int src) { return emu_wave_read(v, src); }

template <int MASK> DEV u32 swz_xor_u32(u32 v) { return emu_wave_read(v, lane_id() ^ MASK); }

DEV int emu_half_mirror(int l) { return (l & ~7) | (7 - (l & 7)); }
DEV int emu_row_mirror(int l)  { return (l & ~15) | (15 - (l & 15)); }
DEV float row16_allreduce_add(float v)
{
    const int l = lane_id();
    v += emu_wave_read(v, l ^ 1);
    v += emu_wave_read(v, l ^ 2);
    v += emu_wave_read(v, emu_half_mirror(l));
    v += emu_wave_read(v, emu_row_mirror(l));
    return v;
}
DEV float quad_allreduce_add(float v)
{
    const int l = lane_id();
    v += emu_wave_read(v, l ^ 1);
    v += emu_wave_read(v, l ^ 2);
    return v;
}
DEV float row8_allreduce_add(float v)
{
    const int l = lane_id();
    v += emu_wave_read(v, l ^ 1);
    v += emu_wave_read(v, l ^ 2);
    v += emu_wave_read(v, emu_half_mirror(l));
    return v;
}
DEV float row16_allreduce_max(float v)
{
    const int l = lane_id();
    v = fmaxf(v, emu_wave_read(v, l ^ 1));
    v = fmaxf(v, emu_wave_read(v, l ^ 2));
    v = fmaxf(v, emu_wave_read(v, emu_half_mirror(l)));
    v = fmaxf(v, emu_wave_read(v, emu_row_mirror(l)));
    return v;
}
DEV float wave_allreduce_add(float v) { v = row16_allreduce_add(v); v += shfl_xor_f32(v, 16); v += shfl_xor_f32(v, 32); return v; }
DEV float wave_allreduce_max(float v)
{
    v = row16_allreduce_max(v); v = fmaxf(v, shfl_xor_f32(v, 16)); v = fmaxf(v, shfl_xor_f32(v, 32)); return v;
}

// ---- matrix core: v_mfma_f32_16x16x32_f16 ----------------------------------------------------------------------------
DEV f32x4 mfma_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c)
{
    EmuWave& w = emu_ctx_->wave[wave_id()];
    const int l = lane_id();
    memcpy(w.slot[l], &a, 16);
    memcpy(w.slot[l] + 16, &b, 16);
    w.bar.wait();
    const int col = l & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; r++)
    {
        const int row = (l >> 4) * 4 + r;
        float acc = d[r];
        for (int j = 0; j < 4; j++)
        {
            f16x8 av, bv;
            memcpy(&av, w.slot[row + 16 * j], 16);
            memcpy(&bv, w.slot[col + 16 * j] + 16, 16);
            for (int e = 0; e < 8; e++) acc += (float)av[e] * (float)bv[e];
        }
        d[r] = acc;
    }
    w.bar.wait();
    return d;
}

DEV float dot2_f32_f16(f16x2 a, f16x2 b, float c) { return c + (float)a.x * (float)b.x + (float)a.y * (float)b.y; }

// ds_read_b64_tr_b16 (hw.h): lane 16 g + i gets element (i % 4) of what lanes 16 g + 4 j + i / 4 (j = 0 .. 3) pointed at
DEV f16x4 lds_read_tr16_b64(const f16* p)
{
    EmuWave& w = emu_ctx_->wave[wave_id()];
    const int l = lane_id(), i = l & 15, g = l >> 4;
    memcpy(w.slot[l], p, 8);
    w.bar.wait();
    f16x4 r;
    for (int j = 0; j < 4; j++)
    {
        f16 e[4];
        memcpy(e, w.slot[16 * g + 4 * j + (i >> 2)], 8);
        r[j] = e[i & 3];
    }
    w.bar.wait();
    return r;
}

DEV u64 cycle_stamp() { return 0; }
DEV void sched_fence() { }
DEV u32 fence_load(const u32* p) { return *p; }
DEV void fence_load_use(u32) { }

// ---- memory ----------------------------------------------------------------------------------------------------------
DEV u64 realtime_stamp() { return 0; }
DEV void dma_to_lds16(const void* g_lane_ptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + lane_id() * 16, g_lane_ptr, 16); }
DEV void dma_buf_to_lds16(const void* base, u32 voffset_bytes, void* lds_wave_base) { memcpy((char*)lds_wave_base + lane_id() * 16, (const char*)base + voffset_bytes, 16); }
DEV void dma_buf_to_lds16_agent(const void* base, u32 voffset_bytes, void* lds_wave_base) { dma_buf_to_lds16(base, voffset_bytes, lds_wave_base); }
DEV void dma_buf_to_lds16_so(const void* base, u32 voffset_bytes, u32 soffset_bytes, void* lds_wave_base) { dma_buf_to_lds16(base, voffset_bytes + soffset_bytes, lds_wave_base); }
DEV void dma_to_lds16_nt(const void* g_lane_ptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + lane_id() * 16, g_lane_ptr, 16); }
DEV void wait_lds_reads() { }
DEV void wave_converge() { emu_ctx_->wave[wave_id()].bar.wait(); }
DEV void dma_to_lds4(const void* g_lane_ptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + lane_id() * 4, g_lane_ptr, 4); }
template <int N> DEV void wait_vmcnt_le() { }
DEV void wait_vmcnt_builtin0() { }
DEV void block_sync_lds() { block_sync(); }
DEV u32 byte_perm(u32 hi, u32 lo, u32 sel)
{
    const unsigned long long pool = ((unsigned long long)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++)
    {
        const u32 s = (sel >> (8 * i)) & 0xFF;
        const u32 b = s < 8 ? (u32)((pool >> (8 * s)) & 0xFF) : (s == 0x0C ? 0u : 0xFFu);
        r |= b << (8 * i);
    }
    return r;
}
DEV u32 and_or(u32 x, u32 mask, u32 magic) { return (x & mask) | magic; }
template <typename T> DEV T ld_nt(const T* p) { return *p; }
template <typename T> DEV void st_nt(T* p, T v) { *p = v; }
DEV float atomic_add_f32(float* p, float v)
{
    static std::mutex m; std::lock_guard<std::mutex> g(m); const float o = *p; *p = o + v; return o;
}
DEV u32 atomic_add_u32(u32* p, u32 v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

DEV void wait_vmcnt0() { }
DEV void fence_release_agent() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
DEV void fence_acquire_agent() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
DEV u32 ticket_add_agent(u32* p, u32 v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
DEV float load_agent_f32(const float* p) { return *(const volatile float*)p; }
DEV void store_agent_f32(float* p, float v) { *(volatile float*)p = v; }
DEV void store_agent_f32x4(f32x4* p, f32x4 v) { *p = v; }
DEV f32x4 load_agent_f32x4(const f32x4* p) { return *p; }
DEV void store_relaxed_agent(u32* p, u32 v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
DEV void store_agent_f16(f16* p, f16 v) { *p = v; }
DEV u32 load_agent_u32(const u32* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
// hand-off between overlapped launches (hw.h): launches run to completion one after the other here, so a consumer finds its
// producer's "go" already published; a missing one is a bug in the host's bookkeeping -- reported, not waited for
#define SYNC_BLOCK_WORDS 1024
#define SYNC_GO_COPIES 8
#define SYNC_SHARDS 8
DEV u32* sync_fin_shard(u32* block, u32 c) { return block + 320 + 32 * (c & (SYNC_SHARDS - 1)); }
DEV u32* sync_entry_shard(u32* block, u32 c) { return block + 576 + 32 * (c & (SYNC_SHARDS - 1)); }
DEV const u32* sync_go_word(const u32* block, int cls) { return block + 32 * (1 + (cls & (SYNC_GO_COPIES - 1))); }
DEV void sync_wait_go(const u32* producer_block, int cls)
{
    if (__atomic_load_n(sync_go_word(producer_block, cls), __ATOMIC_SEQ_CST) == 0)
    {
        fprintf(stderr, "emu: overlapped launch would wait forever (no go word)\n");
        abort();
    }
}
DEV void sync_arrive_publish(u32* own_block, u32 total, const u32* waited_block)
{
    if (lane_id() != 0) return;
    const u32 old = __atomic_fetch_add(own_block, 1u, __ATOMIC_SEQ_CST);
    if (old + 1 != total) return;
    for (int c = 0; c < SYNC_GO_COPIES; c++) *(u32*)sync_go_word(own_block, c) = 1u;
    *own_block = 0u;
    if (waited_block) for (int c = 0; c < SYNC_GO_COPIES; c++) *(u32*)sync_go_word(waited_block, c) = 0u;
}
DEV void sync_arrive_publish_sharded(u32* own_block, u32 lin, u32 per_wg, u32 total_wgs, const u32* waited_block)
{
    if (lane_id() != 0) return;
    const u32 c = lin & (SYNC_SHARDS - 1);
    const u32 in_shard = ((total_wgs - c + SYNC_SHARDS - 1) / SYNC_SHARDS) * per_wg;
    const u32 n_shards = total_wgs < SYNC_SHARDS ? total_wgs : SYNC_SHARDS;
    u32 old = __atomic_fetch_add(sync_fin_shard(own_block, c), 1u, __ATOMIC_SEQ_CST);
    if (old + 1 > in_shard) { fprintf(stderr, "emu: more arrivals than the shard holds (%u of %u)\n", old + 1, in_shard); abort(); }
    if (old + 1 != in_shard) return;
    *sync_fin_shard(own_block, c) = 0u;
    old = __atomic_fetch_add(own_block, 1u, __ATOMIC_SEQ_CST);
    if (old + 1 != n_shards) return;
    for (int k = 0; k < SYNC_GO_COPIES; k++) *(u32*)sync_go_word(own_block, k) = 1u;
    *own_block = 0u;
    if (waited_block) for (int k = 0; k < SYNC_GO_COPIES; k++) *(u32*)sync_go_word(waited_block, k) = 0u;
}
DEV void sync_report_entry(u32* own_block, u32 lin) { (void)__atomic_fetch_add(sync_entry_shard(own_block, lin), 1u, __ATOMIC_SEQ_CST); }
DEV void sync_gate_wait(u32* producer_block, u32 target)
{
    if (lane_id() != 0) return;                  // (lanes run one after the other here: only the lane that zeroes may look)
    u32 sum = 0;
    for (u32 c = 0; c < SYNC_SHARDS; c++) sum += __atomic_load_n(sync_entry_shard(producer_block, c), __ATOMIC_SEQ_CST);
    if (sum != target) { fprintf(stderr, "emu: gate would wait forever / over-count (%u entries of %u)\n", sum, target); abort(); }
    for (u32 c = 0; c < SYNC_SHARDS; c++) *sync_entry_shard(producer_block, c) = 0u;
}
DEV f16 load_agent_f16(const f16* p) { return *p; }
DEV f16x8 load_agent_f16x8(const f16* p) { return *(const f16x8*)p; }
DEV void dma_to_lds16_agent(const void* g_lane_ptr, void* lds_wave_base) { memcpy((char*)lds_wave_base + lane_id() * 16, g_lane_ptr, 16); }

DEV void* global_ptr_of(u32 lo, u32 hi) { return (void*)(((u64)hi << 32) | lo); }

#define DYN_SMEM(name) unsigned char* name = emu_ctx_->dyn_smem
#define SHARED static

#define LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu_launch(grid, block, smem, [&]() { kernel(__VA_ARGS__); })
#define LAUNCH_COOP(kernel, grid, block, smem, stream, ...) \
    emu_launch_coop(grid, block, smem, [&]() { kernel(__VA_ARGS__); })

// ---- HIP runtime shims used by host code -----------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToHost = 0, hipMemcpyDefault = 4 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; };

DEV hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
DEV hipError_t hipFree(void* p) { free(p); return hipSuccess; }
DEV hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
DEV hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
DEV hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
DEV hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
DEV hipError_t hipDeviceSynchronize() { return hipSuccess; }
DEV hipError_t hipGetLastError() { return hipSuccess; }
DEV hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
static const hipError_t hipErrorPeerAccessAlreadyEnabled = (hipError_t)704;
DEV hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
DEV hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
DEV hipError_t hipSetDevice(int) { return hipSuccess; }
DEV hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
DEV hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
DEV const char* hipGetErrorString(hipError_t) { return "emu"; }
DEV hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 4; return hipSuccess; }
DEV hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

// events: launches complete before they return here, so ordering between "streams" is program order
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
DEV hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
DEV hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
DEV hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
DEV hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
DEV hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(4096, (n + 4095) / 4096 * 4096); return *p ? hipSuccess : hipErrorOutOfMemory; }

// graphs cannot be emulated: capture is refused, callers fall back to eager launches in the emu tests
typedef void* hipGraph_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
DEV hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
typedef void* hipGraphExec_t;
enum { hipStreamCaptureModeRelaxed = 2 };
DEV hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorInvalidValue; }
DEV hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorInvalidValue; }
DEV hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorInvalidValue; }
DEV hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
DEV hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
DEV hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

#endif  // EXL2_HW_H
