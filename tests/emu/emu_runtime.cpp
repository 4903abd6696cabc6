// emu_runtime.cpp -- workgroup scheduler of the CPU emulation (see hw_emu.h).  TEST INFRASTRUCTURE ONLY.
#include "hw_emu.h"

thread_local int emu_tid_ = 0;
thread_local EmuDim emu_bid_ = {0, 0, 0};
EmuCtx* emu_ctx_ = nullptr;
static std::mutex g_launch_mutex;

void emu_launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body)
{
    std::lock_guard<std::mutex> guard(g_launch_mutex);
    if (block.y != 1 || block.z != 1 || block.x % 64 != 0 || block.x > 1024)
    {
        fprintf(stderr, "emu_launch: unsupported block shape %u,%u,%u\n", block.x, block.y, block.z);
        abort();
    }
    EmuCtx* ctx = new EmuCtx();
    ctx->grid = {grid.x, grid.y, grid.z};
    ctx->block = {block.x, block.y, block.z};
    ctx->dyn_smem = (unsigned char*)aligned_alloc(256, (smem + 511) / 256 * 256);
    const int nt = (int)block.x;
    ctx->done_bar.reset(nt);
    emu_ctx_ = ctx;
    std::vector<std::thread> threads;
    threads.reserve(nt);
    for (int i = 0; i < nt; i++)
    {
        threads.emplace_back([=, &body]() {
            emu_tid_ = i;
            for (unsigned bz = 0; bz < grid.z; bz++)
            for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
            {
                ctx->done_bar.wait();
                if (i == 0)
                {
                    ctx->block_bar.reset(nt);
                    for (int w = 0; w < nt / 64; w++) ctx->wave[w].bar.reset(64);
                }
                ctx->done_bar.wait();
                emu_bid_ = {bx, by, bz};
                body();
                ctx->wave[i >> 6].bar.leave();
                ctx->block_bar.leave();
            }
        });
    }
    for (auto& t : threads) t.join();
    free(ctx->dyn_smem);
    delete ctx;
    emu_ctx_ = nullptr;
}
