// simt_host.h -- the few SIMT builtins the reference's cache_q.cuh uses (warp shuffles over 32 lanes, __syncthreads,
// __shared__), executed on the host: the logical threads of ONE block are fibers run round-robin (simt_host.cpp).
// TEST INFRASTRUCTURE ONLY (see cuda_shim.h).
#pragma once
#include <string.h>
#include <functional>

namespace simt
{
void run_block(int nthreads, const std::function<void(int)>& body);    // body(t), t = 0 .. nthreads-1
int tid();
void barrier(int group);                                               // group = 0: whole block; 32: the caller's warp
extern unsigned char xchg[1024][16];
}

template <typename T> static inline T simt_exchange(T v, int src_lane)
{
    static_assert(sizeof(T) <= 16, "exchange slot too small");
    const int t = simt::tid();
    memcpy(simt::xchg[t], &v, sizeof(T));
    simt::barrier(32);
    T r;
    memcpy(&r, simt::xchg[(t & ~31) | (src_lane & 31)], sizeof(T));
    simt::barrier(32);
    return r;
}
// every lane executes the instruction in the reference code; the mask only names the lanes whose result is used
template <typename T> static inline T __shfl_xor_sync(unsigned, T v, int lane_mask) { return simt_exchange(v, (simt::tid() & 31) ^ lane_mask); }
template <typename T> static inline T __shfl_down_sync(unsigned, T v, int delta)
{
    const int lane = simt::tid() & 31;
    return simt_exchange(v, lane + delta < 32 ? lane + delta : lane);
}
static inline void __syncthreads() { simt::barrier(0); }

// kernel coordinates for __global__ functions run through simt::run_grid
namespace simt
{
struct Dim { unsigned x, y, z; };
extern Dim block_idx, grid_dim, block_dim;
// blocks are launched with a linear thread count; block_dim.x (set by run_grid: the whole count unless block_dim_x is
// given) folds it into (x, y) for kernels written for 2-D blocks
static inline Dim thread_idx() { Dim d = {(unsigned)tid() % block_dim.x, (unsigned)tid() / block_dim.x, 0u}; return d; }
// every block of the grid, one after the other, `nthreads` logical threads each
void run_grid(unsigned gx, unsigned gy, int nthreads, const std::function<void()>& kernel_call, unsigned gz = 1,
              unsigned block_dim_x = 0);
}
#define threadIdx (simt::thread_idx())
#define blockIdx  (simt::block_idx)
#define gridDim   (simt::grid_dim)
#define blockDim  (simt::block_dim)
#define __global__ static
#define __shared__ static
#define __restrict__
