// emu_runtime.cpp -- workgroup scheduler of the CPU emulation (see hw_emu.h).  TEST INFRASTRUCTURE ONLY.
//
// The threads of a workgroup are fibers on the calling OS thread: a hand-rolled x86-64 stack switch (callee-saved
// registers + stack pointer; ucontext elsewhere), one 256 KB lazily committed stack per thread slot, round-robin
// scheduling that skips fibers whose barrier generation has not moved.  A 1024-thread workgroup costs one user-level
// switch per thread per barrier; the first version used one OS thread per lane with mutex / condition-variable barriers
// and spent 80 % of the test suite's time in futex calls.
#include "hw_emu.h"
#include <sys/mman.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

int emu_tid_ = 0;
EmuDim emu_bid_ = {0, 0, 0};
EmuCtx* emu_ctx_ = nullptr;
static std::mutex g_launch_mutex;

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;

struct Fiber
{
#if defined(__x86_64__)
    void* sp = nullptr;
#else
    ucontext_t uc;
#endif
    EmuBarrier* wait_bar = nullptr;
    unsigned wait_gen = 0;
    bool done = true;
    bool spinning = false;                          // gave its slice up inside a spin loop (emu_spin_yield)
};

Fiber g_fiber_one[MAX_THREADS];                     // one workgroup at a time (emu_launch)
Fiber* g_fiber = g_fiber_one;                       // emu_launch_coop swaps in its own array of grid x block fibers
char* g_stacks = nullptr;
char* g_stacks_one = nullptr;
const std::function<void()>* g_body = nullptr;
int g_cur = -1;

#if defined(__x86_64__)
void* g_sched_sp = nullptr;
extern "C" void emu_switch_(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch_
    .type emu_switch_,@function
emu_switch_:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch_, .-emu_switch_
)");
inline void to_scheduler() { emu_switch_(&g_fiber[g_cur].sp, g_sched_sp); }
inline void to_fiber(int i) { emu_switch_(&g_sched_sp, g_fiber[i].sp); }
#else
ucontext_t g_sched_uc;
inline void to_scheduler() { swapcontext(&g_fiber[g_cur].uc, &g_sched_uc); }
inline void to_fiber(int i) { swapcontext(&g_sched_uc, &g_fiber[i].uc); }
#endif

// first frame of every fiber: run the kernel body for the current block, drop out of the barriers, hand back
void fiber_main()
{
    Fiber& f = g_fiber[g_cur];
    (*g_body)();
    emu_ctx_->wave[emu_tid_ >> 6].bar.leave();
    emu_ctx_->block_bar.leave();
    f.done = true;
    to_scheduler();
    abort();                                        // a finished fiber is never resumed
}

void prepare(int i)
{
    Fiber& f = g_fiber[i];
    f.wait_bar = nullptr;
    f.done = false;
    char* top = g_stacks + (size_t)(i + 1) * STACK_BYTES;
#if defined(__x86_64__)
    // what emu_switch_ pops: r15 r14 r13 r12 rbx rbp, then `ret` into fiber_main with rsp = 8 (mod 16) as after a call
    void** sp = (void**)(top - 64);                 // 16-byte aligned slot of the return address
    sp[0] = (void*)&fiber_main;
    sp[1] = nullptr;                                // fiber_main's (never used) return address
    sp -= 6;
    for (int k = 0; k < 6; k++) sp[k] = nullptr;
    f.sp = sp;
#else
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = top - STACK_BYTES;
    f.uc.uc_stack.ss_size = STACK_BYTES;
    f.uc.uc_link = nullptr;
    makecontext(&f.uc, fiber_main, 0);
#endif
}

}  // namespace

void emu_block_on(EmuBarrier* bar, unsigned gen)
{
    Fiber& f = g_fiber[g_cur];
    f.wait_bar = bar; f.wait_gen = gen;
    to_scheduler();
    f.wait_bar = nullptr;
}

void emu_launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body)
{
    std::lock_guard<std::mutex> guard(g_launch_mutex);
    if (block.y != 1 || block.z != 1 || block.x % 64 != 0 || block.x > MAX_THREADS)
    {
        fprintf(stderr, "emu_launch: unsupported block shape %u,%u,%u\n", block.x, block.y, block.z);
        abort();
    }
    if (g_cur >= 0)
    {
        fprintf(stderr, "emu_launch: launch from inside a kernel\n");
        abort();
    }
    if (!g_stacks_one)
    {
        g_stacks_one = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks_one == (char*)MAP_FAILED) { perror("emu_launch: mmap"); abort(); }
    }
    g_stacks = g_stacks_one; g_fiber = g_fiber_one;
    EmuCtx* ctx = new EmuCtx();
    ctx->grid = {grid.x, grid.y, grid.z};
    ctx->block = {block.x, block.y, block.z};
    ctx->dyn_smem = (unsigned char*)aligned_alloc(256, (smem + 511) / 256 * 256);
    const int nt = (int)block.x;
    emu_ctx_ = ctx;
    g_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++)
    {
        ctx->block_bar.reset(nt);
        for (int w = 0; w < nt / 64; w++) ctx->wave[w].bar.reset(64);
        emu_bid_ = {bx, by, bz};
        for (int i = 0; i < nt; i++) prepare(i);
        int remaining = nt;
        while (remaining > 0)
        {
            bool progressed = false;
            for (int i = 0; i < nt; i++)
            {
                Fiber& f = g_fiber[i];
                if (f.done || (f.wait_bar && f.wait_bar->gen == f.wait_gen)) continue;
                g_cur = i; emu_tid_ = i;
                to_fiber(i);
                progressed = true;
                if (f.done) remaining--;
            }
            if (!progressed)
            {
                fprintf(stderr, "emu_launch: deadlock -- every live thread of block (%u,%u,%u) waits on a barrier\n", bx, by, bz);
                abort();
            }
        }
    }
    g_cur = -1;
    g_body = nullptr;
    free(ctx->dyn_smem);
    delete ctx;
    emu_ctx_ = nullptr;
}


// ---- co-resident grids ---------------------------------------------------------------------------------------------------
// emu_launch runs one workgroup to completion before the next starts: enough for every kernel whose workgroups are
// independent, a deadlock for a kernel whose workgroups WAIT FOR EACH OTHER inside one launch (persistent multi-phase
// kernels with grid-wide hand-offs).  emu_launch_coop keeps every workgroup of a (small) grid alive at once: grid x block
// fibers, one EmuCtx (barriers, wave exchange buffers, dynamic LDS) per workgroup, round-robin over all of them.  A thread
// that polls memory written by another workgroup calls emu_spin_yield() inside its loop; a launch in which, for many rounds
// in a row, every runnable thread only spins is reported as a livelock.  Kernels launched this way must keep their
// workgroup-local storage in dynamic LDS (DYN_SMEM): `SHARED` is a function-local static here, i.e. ONE copy for all
// workgroups.
void emu_spin_yield()
{
    if (g_cur < 0) return;
    g_fiber[g_cur].spinning = true;
    to_scheduler();
}

void emu_launch_coop(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body)
{
    std::lock_guard<std::mutex> guard(g_launch_mutex);
    const unsigned n_wg = grid.x * grid.y * grid.z;
    if (block.y != 1 || block.z != 1 || block.x % 64 != 0 || block.x > MAX_THREADS || n_wg == 0 || n_wg > 32)
    {
        fprintf(stderr, "emu_launch_coop: unsupported shape (grid %u workgroups, block %u)\n", n_wg, block.x);
        abort();
    }
    if (g_cur >= 0) { fprintf(stderr, "emu_launch_coop: launch from inside a kernel\n"); abort(); }
    const int nt = (int)block.x;
    const size_t total = (size_t)n_wg * nt;
    char* stacks = (char*)mmap(nullptr, STACK_BYTES * total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char*)MAP_FAILED) { perror("emu_launch_coop: mmap"); abort(); }
    Fiber* fibers = new Fiber[total];
    EmuCtx* ctxs = new EmuCtx[n_wg];
    for (unsigned w = 0; w < n_wg; w++)
    {
        ctxs[w].grid = {grid.x, grid.y, grid.z};
        ctxs[w].block = {block.x, block.y, block.z};
        ctxs[w].dyn_smem = (unsigned char*)aligned_alloc(256, (smem + 511) / 256 * 256);
        ctxs[w].block_bar.reset(nt);
        for (int v = 0; v < nt / 64; v++) ctxs[w].wave[v].bar.reset(64);
    }
    g_fiber = fibers; g_stacks = stacks; g_body = &body;
    for (size_t i = 0; i < total; i++) prepare((int)i);
    size_t remaining = total;
    int idle_rounds = 0;
    while (remaining > 0)
    {
        bool progressed = false, only_spins = true;
        for (unsigned w = 0; w < n_wg; w++)
            for (int t = 0; t < nt; t++)
            {
                const int idx = (int)(w * nt + t);
                Fiber& f = g_fiber[idx];
                if (f.done || (f.wait_bar && f.wait_bar->gen == f.wait_gen)) continue;
                emu_ctx_ = &ctxs[w];
                emu_bid_ = {w % grid.x, (w / grid.x) % grid.y, w / (grid.x * grid.y)};
                emu_tid_ = t; g_cur = idx;
                f.spinning = false;
                to_fiber(idx);
                progressed = true;
                if (!f.spinning) only_spins = false;
                if (f.done) remaining--;
            }
        if (!progressed) { fprintf(stderr, "emu_launch_coop: deadlock -- every live thread waits on a barrier\n"); abort(); }
        idle_rounds = only_spins ? idle_rounds + 1 : 0;
        if (idle_rounds > 100000) { fprintf(stderr, "emu_launch_coop: livelock -- every runnable thread only spins\n"); abort(); }
    }
    g_cur = -1; g_body = nullptr; emu_ctx_ = nullptr;
    g_fiber = g_fiber_one; g_stacks = g_stacks_one;
    for (unsigned w = 0; w < n_wg; w++) free(ctxs[w].dyn_smem);
    delete[] ctxs; delete[] fibers;
    munmap(stacks, STACK_BYTES * total);
}

// ---- self-test of the co-resident mode (tests/test_abi.py): a persistent multi-phase kernel with grid-wide hand-offs -------
// Every workgroup writes a value per phase, arrives at a counter, the last arrival publishes the phase number, everybody polls
// it, then reads what the OTHER workgroups wrote in that phase.  Returns the number of wrong reads (0 = pass).
namespace {
struct CoopTest { unsigned* sync; unsigned* data; unsigned* errors; int n_wg, n_phases; };
void coop_kernel(CoopTest a)
{
    DYN_SMEM(smem);
    unsigned* part = (unsigned*)smem;                                    // workgroup-local: one slot per wave
    for (int ph = 0; ph < a.n_phases; ph++)
    {
        // a value that needs the whole workgroup: sum over waves of (wave + 1), times a phase / workgroup tag
        if (lane_id() == 0) part[wave_id()] = (unsigned)wave_id() + 1;
        block_sync();
        if (tid() == 0)
        {
            unsigned s = 0;
            for (int w = 0; w < nthreads() / 64; w++) s += part[w];
            a.data[ph * a.n_wg + bid_x()] = s * 1000u + (unsigned)ph * 37u + (unsigned)bid_x();
            const unsigned old = __atomic_fetch_add(a.sync, 1u, __ATOMIC_SEQ_CST);
            if (old + 1 == (unsigned)(ph + 1) * (unsigned)a.n_wg) __atomic_store_n(a.sync + 32, (unsigned)(ph + 1), __ATOMIC_SEQ_CST);
            while (__atomic_load_n(a.sync + 32, __ATOMIC_SEQ_CST) < (unsigned)(ph + 1)) emu_spin_yield();
        }
        block_sync();
        if (tid() < a.n_wg)
        {
            const unsigned want = (unsigned)(nthreads() / 64) * (unsigned)(nthreads() / 64 + 1) / 2 * 1000u + (unsigned)ph * 37u + (unsigned)tid();
            if (a.data[ph * a.n_wg + tid()] != want) __atomic_fetch_add(a.errors, 1u, __ATOMIC_SEQ_CST);
        }
        block_sync();
    }
}
}  // namespace

extern "C" int emu_selftest_coop(int n_wg, int n_phases, int block)
{
    std::vector<unsigned> sync(64, 0), data((size_t)n_wg * n_phases, 0), errors(1, 0);
    CoopTest a = {sync.data(), data.data(), errors.data(), n_wg, n_phases};
    emu_launch_coop(dim3((unsigned)n_wg), dim3((unsigned)block), 1024, [&]() { coop_kernel(a); });
    return (int)errors[0];
}
