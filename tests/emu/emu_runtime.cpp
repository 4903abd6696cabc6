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
};

Fiber g_fiber[MAX_THREADS];
char* g_stacks = nullptr;
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
    emu_ctx_->wave[g_cur >> 6].bar.leave();
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
    if (!g_stacks)
    {
        g_stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("emu_launch: mmap"); abort(); }
    }
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
