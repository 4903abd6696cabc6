// simt_host.cpp -- fiber scheduler behind simt_host.h.  TEST INFRASTRUCTURE ONLY.  Same technique as tests/emu: one
// lazily committed stack per logical thread, a hand-rolled x86-64 stack switch (ucontext elsewhere), round-robin that
// skips threads whose barrier generation has not moved.
#include "simt_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

namespace simt
{
unsigned char xchg[1024][16];

namespace
{
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;
struct Barrier { int live = 0, arrived = 0; unsigned gen = 0; };
struct Fiber
{
#if defined(__x86_64__)
    void* sp = nullptr;
#else
    ucontext_t uc;
#endif
    Barrier* wait_bar = nullptr; unsigned wait_gen = 0; bool done = true;
};
Fiber fibers[MAX_THREADS];
Barrier block_bar, warp_bar[MAX_THREADS / 32];
char* stacks = nullptr;
const std::function<void(int)>* body_ = nullptr;
int cur = -1;

#if defined(__x86_64__)
void* sched_sp = nullptr;
extern "C" void simt_switch_(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl simt_switch_
    .type simt_switch_,@function
simt_switch_:
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
    .size simt_switch_, .-simt_switch_
)");
inline void to_scheduler() { simt_switch_(&fibers[cur].sp, sched_sp); }
inline void to_fiber(int i) { simt_switch_(&sched_sp, fibers[i].sp); }
#else
ucontext_t sched_uc;
inline void to_scheduler() { swapcontext(&fibers[cur].uc, &sched_uc); }
inline void to_fiber(int i) { swapcontext(&sched_uc, &fibers[i].uc); }
#endif

void leave(Barrier& b) { b.live--; if (b.live > 0 && b.arrived >= b.live) { b.arrived = 0; b.gen++; } }

void fiber_main()
{
    (*body_)(cur);
    leave(warp_bar[cur >> 5]);
    leave(block_bar);
    fibers[cur].done = true;
    to_scheduler();
    abort();
}
}  // namespace

Dim block_idx = {0, 0, 0}, grid_dim = {1, 1, 1}, block_dim = {1, 1, 1};

void run_grid(unsigned gx, unsigned gy, int nthreads, const std::function<void()>& kernel_call, unsigned gz, unsigned block_dim_x)
{
    grid_dim = {gx, gy, gz};
    if (block_dim_x) block_dim = {block_dim_x, (unsigned)nthreads / block_dim_x, 1};
    else block_dim = {(unsigned)nthreads, 1, 1};
    const int padded = (nthreads + 31) / 32 * 32;          // whole warps; the extra lanes do not call the kernel
    for (unsigned bz = 0; bz < gz; bz++)                    // z outermost: slice z = 0 (which clears c) runs first
        for (unsigned by = 0; by < gy; by++)
            for (unsigned bx = 0; bx < gx; bx++)
            {
                block_idx = {bx, by, bz};
                run_block(padded, [&](int t) { if (t < nthreads) kernel_call(); });
            }
}

int tid() { return cur; }

void barrier(int group)
{
    Barrier& b = group == 0 ? block_bar : warp_bar[cur >> 5];
    const unsigned g = b.gen;
    if (++b.arrived >= b.live) { b.arrived = 0; b.gen++; return; }
    Fiber& f = fibers[cur];
    f.wait_bar = &b; f.wait_gen = g;
    to_scheduler();
    f.wait_bar = nullptr;
}

void run_block(int nthreads, const std::function<void(int)>& body)
{
    if (nthreads <= 0 || nthreads > MAX_THREADS || nthreads % 32) { fprintf(stderr, "simt::run_block: bad thread count %d\n", nthreads); abort(); }
    if (!stacks)
    {
        stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (char*)MAP_FAILED) { perror("simt::run_block: mmap"); abort(); }
    }
    body_ = &body;
    block_bar = Barrier(); block_bar.live = nthreads;
    for (int w = 0; w < nthreads / 32; w++) { warp_bar[w] = Barrier(); warp_bar[w].live = 32; }
    for (int i = 0; i < nthreads; i++)
    {
        Fiber& f = fibers[i];
        f.wait_bar = nullptr; f.done = false;
        char* top = stacks + (size_t)(i + 1) * STACK_BYTES;
#if defined(__x86_64__)
        void** sp = (void**)(top - 64);
        sp[0] = (void*)&fiber_main; sp[1] = nullptr;
        sp -= 6;
        for (int k = 0; k < 6; k++) sp[k] = nullptr;
        f.sp = sp;
#else
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = top - STACK_BYTES; f.uc.uc_stack.ss_size = STACK_BYTES; f.uc.uc_link = nullptr;
        makecontext(&f.uc, fiber_main, 0);
#endif
    }
    int remaining = nthreads;
    while (remaining > 0)
    {
        bool progressed = false;
        for (int i = 0; i < nthreads; i++)
        {
            Fiber& f = fibers[i];
            if (f.done || (f.wait_bar && f.wait_bar->gen == f.wait_gen)) continue;
            cur = i;
            to_fiber(i);
            progressed = true;
            if (f.done) remaining--;
        }
        if (!progressed) { fprintf(stderr, "simt::run_block: deadlock\n"); abort(); }
    }
    cur = -1; body_ = nullptr;
}
}  // namespace simt
