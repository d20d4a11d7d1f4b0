// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE: fiber scheduler that runs a HIP-style launch on
// the CPU.  One OS thread per block (up to NSR_EMU_THREADS at a time), one fiber per GPU thread.
#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

#include "nsr_dev.h"

extern "C" void nsr_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl nsr_emu_switch
.type nsr_emu_switch,@function
nsr_emu_switch:
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
.size nsr_emu_switch,.-nsr_emu_switch
)");

namespace nsr {
namespace emu {

thread_local Block *B = nullptr;

static constexpr size_t kStack = 512 * 1024;

void yield_to_scheduler() {
    Block *b = B;
    Fiber *f = b->cur;
    nsr_emu_switch(&f->sp, b->sched_sp);
}

static void fiber_main() {
    Block *b = B;
    Fiber *f = b->cur;
    (*b->body)();
    f->done = true;
    nsr_emu_switch(&f->sp, b->sched_sp);
    std::abort();   // a finished fiber is never resumed
}

static void run_block(Block *b, const std::function<void()> &body, std::vector<char *> &stacks) {
    B = b;
    b->body = &body;
    b->block_arrived = 0;
    b->block_gen = 0;
    for (int w = 0; w < 32; ++w) { b->wave_arrived[w] = 0; b->wave_gen[w] = 0; }
    for (int t = 0; t < b->nthreads; ++t) {
        Fiber &f = b->fibers[t];
        f = Fiber();
        f.tid = t;
        f.stack = stacks[t];
        // initial frame: six callee-saved registers, then the return address = fiber_main.
        // After `ret` the stack pointer must be 8 mod 16 (as right after a call).
        uintptr_t top = reinterpret_cast<uintptr_t>(f.stack + kStack);
        top &= ~uintptr_t(15);
        void **sp = reinterpret_cast<void **>(top);
        *--sp = nullptr;                                   // fake caller return slot (alignment)
        *--sp = reinterpret_cast<void *>(&fiber_main);     // popped by `ret`
        for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    int remaining = b->nthreads;
    while (remaining > 0) {
        bool progressed = false;
        for (int t = 0; t < b->nthreads; ++t) {
            Fiber &f = b->fibers[t];
            if (f.done) continue;
            if (f.wait == 1 && b->wave_gen[t >> 6] == f.wait_gen) continue;
            if (f.wait == 2 && b->block_gen == f.wait_gen) continue;
            f.wait = 0;
            b->cur = &f;
            nsr_emu_switch(&b->sched_sp, f.sp);
            progressed = true;
            if (f.done) --remaining;
        }
        if (!progressed) {
            std::fprintf(stderr, "nsr emu: deadlock in block (%u,%u): %d threads stuck (divergent collective?)\n",
                         b->bid.x, b->bid.y, remaining);
            std::abort();
        }
    }
}

void launch(dim3 grid, dim3 block, int lds_bytes, const std::function<void()> &body) {
    const int nthreads = (int)block.x;
    const long long nblocks = (long long)grid.x * grid.y;
    int nworkers = 8;
    if (const char *e = std::getenv("NSR_EMU_THREADS")) nworkers = std::atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks) nworkers = (int)nblocks;
    std::atomic<long long> next{0};
    auto worker = [&]() {
        Block *b = new Block();
        b->nthreads = nthreads;
        b->gdim = grid;
        b->fibers = new Fiber[nthreads];
        std::vector<char *> stacks(nthreads);
        for (int t = 0; t < nthreads; ++t) {
            void *m = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (m == MAP_FAILED) { std::perror("mmap"); std::abort(); }
            stacks[t] = static_cast<char *>(m);
        }
        char *lds = static_cast<char *>(aligned_alloc(64, (size_t)(lds_bytes > 0 ? lds_bytes : 64) + 64));
        for (;;) {
            const long long id = next.fetch_add(1);
            if (id >= nblocks) break;
            b->bid = dim3((unsigned)(id % grid.x), (unsigned)(id / grid.x), 0);
            std::memset(lds, 0xFF, (size_t)(lds_bytes > 0 ? lds_bytes : 64));   // poison: LDS is uninitialised on a GPU
            b->lds = lds;
            run_block(b, body, stacks);
        }
        for (int t = 0; t < nthreads; ++t) munmap(stacks[t], kStack);
        free(lds);
        delete[] b->fibers;
        delete b;
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < nworkers; ++i) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
}

}  // namespace emu
}  // namespace nsr
