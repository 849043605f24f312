// hipemu.cpp — scheduler of the test-only SIMT simulator (see hipemu.h).
#include "hipemu.h"

namespace hipemu {

State g;
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

void trampoline() {
    g.body();
    g.fibers[g.cur].done = true;
    for (;;) hipemu_switch(&g.fibers[g.cur].sp, g.sched_sp);   // never returns
}

static void run_block(dim3 block) {
    const int n = (int)(block.x * block.y * block.z);
    g.nthreads = n;
    if ((int)g.fibers.size() < n) g.fibers.resize(n);
    g.bar_count = 0;
    for (int w = 0; w < 64; ++w) g.wave_count[w] = 0;
    for (int i = 0; i < n; ++i) {
        Fiber& f = g.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(kStack);
        f.done = false;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        // initial frame: six callee-saved registers, then the entry address that `ret` jumps to
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)(top - 16);
        sp[0] = (void*)trampoline;
        sp[1] = nullptr;
        sp -= 6;
        for (int r = 0; r < 6; ++r) sp[r] = nullptr;
        f.sp = (void*)sp;
    }
    int remaining = n;
    long idle_passes = 0;
    while (remaining) {
        long before = g.progress;
        int finished = 0;
        for (int i = 0; i < n; ++i) {
            Fiber& f = g.fibers[i];
            if (f.done) continue;
            g.cur = i;
            g_threadIdx = f.tid;
            hipemu_switch(&g.sched_sp, f.sp);
            if (f.done) { --remaining; ++finished; }
        }
        if (g.progress == before && finished == 0) {
            if (++idle_passes > 1000) {
                fprintf(stderr, "hipemu: deadlock (divergent barrier/shuffle?) block=(%u,%u,%u)\n", g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
                abort();
            }
        } else {
            idle_passes = 0;
        }
    }
}

void run_grid(dim3 grid, dim3 block, size_t shmem) {
    if (shmem > g.dyn_cap) {
        free(g.dyn_smem);
        g.dyn_cap = shmem + 64;
        g.dyn_smem = (unsigned char*)aligned_alloc(64, (g.dyn_cap + 63) / 64 * 64);
    }
    g_blockDim = block;
    g_gridDim = grid;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_blockIdx = dim3(x, y, z);
                run_block(block);
            }
}

}  // namespace hipemu
