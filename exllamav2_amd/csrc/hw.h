// hw.h -- the gfx950 (CDNA4, wave64) primitives every kernel in this library is written against.
//
// Kernels use clang-native vector types (`_Float16` ext vectors) instead of hip_fp16.h so that packed
// fp16 math lowers directly to v_pk_{add,mul,fma}_f16 and the MFMA builtin takes the fragments as they are.
// tests/emu/hw.h provides the same names for the CPU emulation build used by the GPU-less host-logic tests;
// the product library is only ever built from THIS header (no dual paths in kernel sources).
#ifndef EXL2_HW_H
#define EXL2_HW_H
#include <hip/hip_runtime.h>
#include <stdint.h>

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

#define DEV  __device__ __forceinline__
#define HD   __host__ __device__ __forceinline__
#define KERNEL __global__
#define NOINLINE_DEV __device__ __attribute__((noinline))

#define WAVE 64

// ---- bit casts ------------------------------------------------------------------------------------------------------
DEV f16x2 as_h2(u32 x)   { return __builtin_bit_cast(f16x2, x); }
DEV u32   as_u32(f16x2 x) { return __builtin_bit_cast(u32, x); }
DEV f16   as_h(u16 x)    { return __builtin_bit_cast(f16, x); }
DEV u16   as_u16(f16 x)  { return __builtin_bit_cast(u16, x); }
DEV float as_f32(u32 x)  { return __builtin_bit_cast(float, x); }
DEV u32   f32_bits(float x) { return __builtin_bit_cast(u32, x); }

// ---- packed fp16 math (single instructions on gfx950) ---------------------------------------------------------------
DEV f16x2 h2_fma(f16x2 a, f16x2 b, f16x2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV f16x2 h2_dup(f16 x) { return (f16x2){x, x}; }
DEV f16   h_fma(f16 a, f16 b, f16 c) { return __builtin_fmaf16(a, b, c); }
// correctly rounded fp16 quotient (== __hdiv / __h2div of the reference): the fp32 quotient of two halves is correctly
// rounded (IEEE divide, -fhip-fp32-correctly-rounded-divide-sqrt is the compiler default) and rounding it once more to
// 11 bits is innocuous (24 >= 2*11 + 2), so the result is the exactly rounded a / b.  A plain `a / b` on _Float16 lowers
// to v_rcp_f32 * a, which lands one code away on ties.
DEV f16   h_div_rn(f16 a, f16 b) { return (f16)__fdiv_rn((float)a, (float)b); }
DEV f16x2 h2_div_rn(f16x2 a, f16x2 b) { return (f16x2){h_div_rn(a.x, b.x), h_div_rn(a.y, b.y)}; }

// ---- thread / wave identity (all kernels use 1-D blocks whose size is a multiple of 64) -----------------------------
DEV int lane_id() { return threadIdx.x & 63; }
DEV int wave_id() { return threadIdx.x >> 6; }
DEV int tid()     { return threadIdx.x; }
DEV int nthreads(){ return blockDim.x; }
DEV int bid_x()   { return blockIdx.x; }
DEV int bid_y()   { return blockIdx.y; }
DEV int bid_z()   { return blockIdx.z; }
DEV int gdim_x()  { return gridDim.x; }
DEV int gdim_y()  { return gridDim.y; }

DEV void block_sync() { __syncthreads(); }

// value of lane 0's copy, provably wave-uniform to the compiler (lets it live in SGPRs)
DEV u32 uniform(u32 x) { return __builtin_amdgcn_readfirstlane(x); }
DEV int uniform(int x) { return (int)__builtin_amdgcn_readfirstlane((u32)x); }

// Pin a wave-uniform value into a scalar register HERE: kernel-argument loads feeding it cannot be sunk past this point, so
// a run of pins at the top of a kernel turns scattered single-dword argument loads into one batch of wide scalar loads.
template <typename T> DEV void pin_scalar(T& x) { asm volatile("" : "+s"(x)); }
template <typename T> DEV void pin_vector(T& x) { asm volatile("" : "+v"(x)); }        // the value is made HERE, in a vector register

// ---- cross-lane ----------------------------------------------------------------------------------------------------
// butterfly exchange within a wave: value held by lane (lane ^ mask).  Masks 1,2 map to DPP quad_perm, 4/8 to
// row_half_mirror / row_mirror AFTER the lower steps of an all-reduce made the halves uniform; the generic form
// goes through ds_bpermute (LDS crossbar, no LDS memory).
DEV u32 shfl_xor_u32(u32 v, int mask) {
    return (u32)__builtin_amdgcn_ds_bpermute(((lane_id() ^ mask) << 2), (int)v);
}
DEV float shfl_xor_f32(float v, int mask) { return as_f32(shfl_xor_u32(f32_bits(v), mask)); }
DEV u32 shfl_idx_u32(u32 v, int src_lane) { return (u32)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }
// the lanes of a wave meet (LDS operations of one wave execute in order: what a lane wrote before is visible to every lane behind
// this point; no instruction -- the compiler must not move LDS accesses across it)
DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// bit l = the predicate of lane l (every lane of the wave calls)
DEV u64 wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
DEV float shfl_idx_f32(float v, int src_lane) { return as_f32(shfl_idx_u32(f32_bits(v), src_lane)); }

// exchange with lane ^ MASK for MASK < 32 (stays inside each 32-lane half): ds_swizzle bit-mask mode, no LDS memory
template <int MASK> DEV u32 swz_xor_u32(u32 v) { return (u32)__builtin_amdgcn_ds_swizzle((int)v, (MASK << 10) | 0x1F); }

// all-reduce (sum) over each aligned group of 16 lanes, 4 DPP adds, result in every lane of the group
DEV float row16_allreduce_add(float v) {
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
// all-reduce (sum) over each aligned group of 4 lanes
DEV float quad_allreduce_add(float v) {
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0xB1, 0xF, 0xF, true));
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x4E, 0xF, 0xF, true));
    return v;
}
// all-reduce (sum) over each aligned group of 8 lanes
DEV float row8_allreduce_add(float v) {
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0xB1, 0xF, 0xF, true));
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x4E, 0xF, 0xF, true));
    v += as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x141, 0xF, 0xF, true));
    return v;
}
DEV float row16_allreduce_max(float v) {
    v = fmaxf(v, as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((u32)__builtin_amdgcn_update_dpp(0, (int)f32_bits(v), 0x140, 0xF, 0xF, true)));
    return v;
}
// all-reduce over the whole wave (64 lanes)
DEV float wave_allreduce_add(float v) {
    v = row16_allreduce_add(v);
    v += shfl_xor_f32(v, 16);
    v += shfl_xor_f32(v, 32);
    return v;
}
DEV float wave_allreduce_max(float v) {
    v = row16_allreduce_max(v);
    v = fmaxf(v, shfl_xor_f32(v, 16));
    v = fmaxf(v, shfl_xor_f32(v, 32));
    return v;
}

// ---- matrix core ---------------------------------------------------------------------------------------------------
// D[16x16] += A[16x32] * B[32x16], fp16 in / fp32 accumulate.  Lane l holds A[i = l&15][k-slot (l>>4, e=0..7)],
// B[k-slot (l>>4, e)][j = l&15]; D: column j = l&15, rows (l>>4)*4 + r, r = 0..3.
DEV f32x4 mfma_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// fp32 += dot(half2, half2): v_dot2_f32_f16
DEV float dot2_f32_f16(f16x2 a, f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }

// ds_read_b64_tr_b16 (gfx950): LDS read with a 4 x 16 transpose inside every 16-lane group.  Lane 16 g + i supplies the
// (8-byte aligned) address of 4 consecutive f16; taking the group's 16 x 4 elements as a row-major [4][16] block (lane i ->
// row i / 4, columns 4 (i % 4) .. + 3; the row stride is whatever the addresses say), lane i receives COLUMN i: rows 0 .. 3.
// (Checked on the MI355X: tools/probes/tr_probe.hip.)  This is what turns a row-major [key][feature] image of V into the
// A operand of O^T = V^T P^T without a transposing write.
DEV f16x4 lds_read_tr16_b64(const f16* p)
{
    typedef __fp16 hx4 __attribute__((__vector_size__(8)));
    const hx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hx4*)p);
    return __builtin_bit_cast(f16x4, v);
}

// ---- memory --------------------------------------------------------------------------------------------------------
// streamed-once data (packed weights, KV pages): non-temporal so it does not displace the activation vector / tables
// ---- asynchronous global -> LDS copies (no VGPR round trip; cdna_hip_programming.md "LDS DMA") --------------------------
// Every active lane moves 16 (4) bytes from ITS global address to lds_wave_base + lane * 16 (4); lds_wave_base must be
// wave-uniform.  Completion is counted by vmcnt like any vector load.
DEV void dma_to_lds16(const void* g_lane_ptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane_ptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same for data streamed once (packed weights): non-temporal policy (aux = 2: MI355X_MICROARCH.md "nt-weights")
DEV void dma_to_lds16_nt(const void* g_lane_ptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane_ptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
DEV void dma_to_lds4(const void* g_lane_ptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane_ptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// vmcnt(0) the COMPILER sees (it forgets its pending LDS-DMAs here; the inline-asm waits below are invisible to it): gfx9
// encoding vmcnt = 0, expcnt = 7, lgkmcnt = 15
DEV void wait_vmcnt_builtin0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// wait until at most N vector-memory operations of this wave are still in flight (they complete in issue order)
template <int N> DEV void wait_vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// The same copy in its BUFFER form (buffer_load_dwordx4 ... lds): lane l moves the 16 bytes at base + voffset_bytes(l) to
// lds_wave_base + l * 16; `base` must be wave-uniform.  Why a second form: the compiler models a pending global_load_lds like a
// flat access that may return out of order and answers EVERY later wait on a loaded register with vmcnt(0) -- which drains all the
// requests a wave wanted to keep in flight.  The buffer form is counted like any other vector load, so the compiler's own counted
// waits stay exact with copies in flight (the hardware completes a wave's vector-memory operations in issue order).  What it
// still does not do is order a ds_read behind the copy: fence_load below.
DEV void dma_buf_to_lds16(const void* base, u32 voffset_bytes, void* lds_wave_base)
{
    // raw buffer descriptor: base, stride 0, 2 GB of records, gfx9 data format word (out-of-range lanes would read zeros)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voffset_bytes, 0, 0, 0);
}
// the same with the wave-uniform part of the address in a SCALAR register (soffset): a copy whose per-lane offsets never change --
// a GEMM's stage fill -- then costs no vector instruction and no vector register beyond the one offset it keeps for the whole loop
// (qgemm_mfma.hip: the global_load_lds form made a 64-bit per-lane address per copy, `v_lshl_add_u64` x 11 per K step)
DEV void dma_buf_to_lds16_so(const void* base, u32 voffset_bytes, u32 soffset_bytes, void* lds_wave_base)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voffset_bytes, (int)soffset_bytes, 0, 0);
}
// the same at agent scope (sc1: served by the L2 / memory side, never by this CU's L1): what a producer that may still be running
// wrote with agent-scope stores (overlapped chain, chain_sync.h) -- and what a previous launch wrote, at the price of an L1 bypass
DEV void dma_buf_to_lds16_agent(const void* base, u32 voffset_bytes, void* lds_wave_base)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voffset_bytes, 0, 0, 16);
}
// A "fence load": an ordinary 4-byte vector load the COMPILER tracks, issued behind LDS-DMA copies.  Vector-memory operations
// of a wave complete in issue order, so when this load's value is available the copies issued before it have landed in LDS;
// fence_load_use() makes the compiler wait for exactly this load -- it inserts `s_waitcnt vmcnt(n)` with n = the number of
// vector-memory INSTRUCTIONS it scheduled behind the load (whatever their widths), which a hand-written count could get wrong.
// Loads issued behind the fence load therefore stay in flight across the wait.  (The compiler does not order a ds_read behind
// a pending LDS-DMA copy by itself on this toolchain.)  Use with dma_buf_to_lds16: with a global_load_lds pending the wait
// degenerates to vmcnt(0).
DEV u32 fence_load(const u32* p)
{
    asm volatile("" ::: "memory");                    // the LDS-DMA builtins above stay above
    const u32 v = *p;
    return v;
}
DEV void fence_load_use(u32 v) { asm volatile("" :: "v"(v) : "memory"); }
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for every
// prefetched weight load of the wave before letting anybody pass
DEV void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// all LDS reads of this wave have returned (before an LDS-DMA may overwrite what they read: the compiler does not order
// a global_load_lds behind a pending ds_read)
DEV void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// wave-private LDS hand-over between the lanes of ONE wave (lane A wrote / copied, lane B reads): the lanes execute in
// lockstep and a wave's LDS operations complete in order, so nothing is needed here; the CPU emulation (tests/emu), whose
// lanes are fibers that run one after the other, meets at a per-wave barrier
DEV void wave_converge() { }

// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {hi:lo} (selector 0-3 -> lo, 4-7 -> hi, 0x0C -> 0x00)
DEV u32 byte_perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// (x & mask) | magic in ONE VALU op.  gfx950 VOP3 takes no 32-bit literal, so the compiler splits it into v_and + v_or
// with literals; spelling it out keeps the mask in a scalar register and the magic in a vector register.
DEV u32 and_or(u32 x, u32 mask, u32 magic)
{
    u32 r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
    return r;
}
template <typename T> DEV T ld_nt(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T> DEV void st_nt(T* p, T v) { __builtin_nontemporal_store(v, p); }

DEV float atomic_add_f32(float* p, float v) { return atomicAdd(p, v); }
DEV u32 atomic_add_u32(u32* p, u32 v) { return atomicAdd(p, v); }

// ---- inter-workgroup hand-off inside one launch (cdna_hip_programming.md G16: agent scope, placement independent) ------
DEV void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
DEV void fence_release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
DEV void fence_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
DEV u32 ticket_add_agent(u32* p, u32 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV float load_agent_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// write-through store at agent scope (sc1): visible to the other XCDs without flushing the whole L2 (buffer_wbl2)
DEV void store_agent_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a lane's four accumulators as two 8-byte agent-scope accesses (16-byte atomics do not exist; the halves are independent values)
DEV void store_agent_f32x4(f32x4* p, f32x4 v)
{
    struct Pair { u64 a, b; } w = __builtin_bit_cast(Pair, v);
    __hip_atomic_store((u64*)p, w.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((u64*)p + 1, w.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV f32x4 load_agent_f32x4(const f32x4* p)
{
    struct Pair { u64 a, b; } w;
    w.a = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.b = __hip_atomic_load((const u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(f32x4, w);
}
DEV void store_relaxed_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void store_agent_f16(f16* p, f16 v) { __hip_atomic_store((u16*)p, as_u16(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV u32 load_agent_u32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- hand-off BETWEEN launches that overlap (two streams; the dependency is carried by words in memory instead of the
// kernel boundary; chain_sync.h).  Measured on the MI355X (tools/probes/fork_probe.hip, profiles/r02_fork_probe.txt):
//   * thousands of waves polling the counter the producers add to starve those adds (25-45 us per hand-off): producers add
//     to a counter nobody polls, the LAST one publishes "go" words, ONE wave per workgroup polls the copy of its class;
//   * an agent-scope acquire fence (buffer_inv sc1) per wave costs ~25 us per kernel: none is issued -- what the consumer
//     reads from its producer is read with agent-scope loads, what the producer writes is written with agent-scope stores;
//   * then a hand-off costs ~3.2 us after the producer's last workgroup, about what a kernel boundary costs (3.1 us) -- the
//     gain is the consumer's start-up (arguments, tables, weight ring) running during it.
// Block of one launch: word 0 = arrivals counter, words 32 (1 + c) = copy c of "go" (own 128-byte lines), word 2 = waits
// that gave up (debug).  The spin is bounded: a missing producer must show up as wrong numbers in a test, never as a hung GPU.
#define SYNC_BLOCK_WORDS 1024
#define SYNC_GO_COPIES 8
// round 4: the counters are sharded 8 ways (own 128-byte lines: words 320 + 32 c = workgroups finished, 576 + 32 c = workgroups
// entered), workgroup `lin` uses shard lin % 8 -- a launch of 700-2000 small workgroups adding to ONE word takes ~12 ns per add,
// i.e. longer than the launch itself.  The last arrival of a shard forwards one arrival to word 0.
#define SYNC_SHARDS 8
DEV u32* sync_fin_shard(u32* block, u32 c) { return block + 320 + 32 * (c & (SYNC_SHARDS - 1)); }
DEV u32* sync_entry_shard(u32* block, u32 c) { return block + 576 + 32 * (c & (SYNC_SHARDS - 1)); }
#ifndef SYNC_POLL_SLEEP
#define SYNC_POLL_SLEEP 16                     // s_sleep units (64 cycles each) between two polls: pollers next to a weight stream cost it
#endif                                         // bandwidth (MI355X_MICROARCH.md "polling-cost"); 1 in round 2
#define FLAG_SPIN_LIMIT (1 << 15)              // ~ 30 ms at the default sleep
DEV const u32* sync_go_word(const u32* block, int cls) { return block + 32 * (1 + (cls & (SYNC_GO_COPIES - 1))); }
// consumer: the calling WAVE polls (one wave per workgroup; the others wait at the workgroup barrier behind it)
DEV void sync_wait_go(const u32* producer_block, int cls)
{
    const u32* go = sync_go_word(producer_block, cls);
    int spins = 0;
    while (uniform((int)load_agent_u32(go)) == 0 && ++spins < FLAG_SPIN_LIMIT) __builtin_amdgcn_s_sleep(SYNC_POLL_SLEEP);
    if (spins >= FLAG_SPIN_LIMIT && lane_id() == 0) (void)ticket_add_agent((u32*)producer_block + 2, 1u);
}
// producer: the calling wave's outputs were agent-scope stores; when they have completed it arrives; the last of `total`
// arrivals publishes "go", zeroes its own counter and the "go" words this launch waited on (every consumer of those has
// passed: this launch is finished) -- the words are back to zero for the next replay without any memset.
DEV void sync_arrive_publish(u32* own_block, u32 total, const u32* waited_block)
{
    wait_vmcnt0();
    if (lane_id() != 0) return;
    const u32 old = ticket_add_agent(own_block, 1u);
    if (old + 1 != total) return;
    for (int c = 0; c < SYNC_GO_COPIES; c++) store_relaxed_agent((u32*)sync_go_word(own_block, c), 1u);
    store_relaxed_agent(own_block, 0u);
    if (waited_block) for (int c = 0; c < SYNC_GO_COPIES; c++) store_relaxed_agent((u32*)sync_go_word(waited_block, c), 0u);
}
// sharded form of the same (round 4): workgroup `lin` of `total_wgs`, `per_wg` arrivals per workgroup (its finalising waves)
DEV void sync_arrive_publish_sharded(u32* own_block, u32 lin, u32 per_wg, u32 total_wgs, const u32* waited_block)
{
    wait_vmcnt0();
    if (lane_id() != 0) return;
    const u32 c = lin & (SYNC_SHARDS - 1);
    const u32 in_shard = ((total_wgs - c + SYNC_SHARDS - 1) / SYNC_SHARDS) * per_wg;        // arrivals of the workgroups with lin % 8 == c
    const u32 n_shards = total_wgs < SYNC_SHARDS ? total_wgs : SYNC_SHARDS;
    u32 old = ticket_add_agent(sync_fin_shard(own_block, c), 1u);
    if (old + 1 != in_shard) return;
    store_relaxed_agent(sync_fin_shard(own_block, c), 0u);
    old = ticket_add_agent(own_block, 1u);
    if (old + 1 != n_shards) return;
    for (int k = 0; k < SYNC_GO_COPIES; k++) store_relaxed_agent((u32*)sync_go_word(own_block, k), 1u);
    store_relaxed_agent(own_block, 0u);
    if (waited_block) for (int k = 0; k < SYNC_GO_COPIES; k++) store_relaxed_agent((u32*)sync_go_word(waited_block, k), 0u);
}
// "this workgroup holds its slot": one lane per workgroup, on entry (the gate below counts them)
DEV void sync_report_entry(u32* own_block, u32 lin) { (void)ticket_add_agent(sync_entry_shard(own_block, lin), 1u); }
// the gate ahead of every launch of an overlapped chain (chain_sync.h): one wave waits until all `target` workgroups of the
// launch's PRODUCER have entered (the consumer's workgroups spin while they hold their slots: they must not take slots the
// producer still needs), then zeroes the counters for the next replay
DEV void sync_gate_wait(u32* producer_block, u32 target)
{
    const int lane = lane_id();
    int spins = 0;
    while (true)
    {
        u32 v = lane < SYNC_SHARDS ? load_agent_u32(sync_entry_shard(producer_block, (u32)lane)) : 0u;
        #pragma unroll
        for (int m = 1; m < SYNC_SHARDS; m <<= 1) v += shfl_xor_u32(v, m);
        if (uniform((int)v) >= (int)target) break;
        if (++spins >= FLAG_SPIN_LIMIT) { if (lane == 0) (void)ticket_add_agent(producer_block + 2, 1u); break; }
        __builtin_amdgcn_s_sleep(SYNC_POLL_SLEEP);
    }
    if (lane < SYNC_SHARDS) store_relaxed_agent(sync_entry_shard(producer_block, (u32)lane), 0u);
}
// agent-scope loads of what a still-running producer has written (no acquire fence: see above)
DEV f16 load_agent_f16(const f16* p) { return __builtin_bit_cast(f16, (u16)__hip_atomic_load((const u16*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
DEV f16x8 load_agent_f16x8(const f16* p)
{
    struct Pair { u64 a, b; } v;
    v.a = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.b = __hip_atomic_load((const u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(f16x8, v);
}
// LDS-DMA at agent scope (sc1)
DEV void dma_to_lds16_agent(const void* g_lane_ptr, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane_ptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 16);
}

// a pointer to GLOBAL memory from its two halves (kernel-argument words fetched as raw dwords): the integer goes through the
// global address space so that the accesses stay global_load / global_store (a bare integer -> pointer cast would make
// them flat_* instructions, which also tick the LDS counter)
DEV void* global_ptr_of(u32 lo, u32 hi)
{
    typedef __attribute__((address_space(1))) char* GP;
    return (void*)(GP)(((u64)hi << 32) | lo);
}

// dynamic LDS (16-byte aligned base, guide G17)
#define DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define SHARED __shared__

// fast math
DEV float fast_exp(float x) { return __expf(x); }
DEV float fast_rsqrt(float x) { return rsqrtf(x); }
DEV float fast_rcp(float x) { return __frcp_rn(x); }

// shader-clock timestamp (s_memtime): used only by the EXL2_TRACE profiling build
DEV u64 cycle_stamp() { return __builtin_amdgcn_s_memtime(); }
// constant 100 MHz counter shared by all XCDs (10 ns ticks): timelines across workgroups in the EXL2_TRACE build
DEV u64 realtime_stamp() { return __builtin_amdgcn_s_memrealtime(); }

// a numbered comment in the generated code: keeps identical code sequences of two template instantiations apart (tail merging would
// join their pending-register sets, qgemv_lean.hip) and gives tests/test_lean_isa.py its landmarks
#define ASM_MARK(id) asm volatile("; lean mark %0" :: "n"(id))

// scheduling fence: keeps the compiler from interleaving the decode of consecutive super-chunks (register pressure)
DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// ---- launch --------------------------------------------------------------------------------------------------------
#define LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#endif  // EXL2_HW_H
