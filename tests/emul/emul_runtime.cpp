// Emulator runtime: thread-local HIP index variables, host stable sort standing in
// for the rocPRIM translation unit (sort_pairs.hip).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "host_util.h"

#include <sys/mman.h>

thread_local hipemul_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local hipemul::Block *hipemul::cur_block = nullptr;

namespace hipemul {
namespace {
constexpr size_t kStackBytes = 1024 * 1024;  // (+ a guard page below: an overflow faults instead of corrupting a neighbour)
// stacks are kept for the life of the thread (a launch needs blockDim.x of them)
thread_local std::vector<void *> stack_pool;

void *stack_of(unsigned t) {
    while (stack_pool.size() <= t) {
        void *p = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            perror("hipemul: mmap(fiber stack)");
            abort();
        }
        mprotect(p, 4096, PROT_NONE);
        stack_pool.push_back(p);
    }
    return stack_pool[t];
}

#if !HIPEMUL_UCONTEXT
// switch(from, to): push the callee-saved registers and the floating-point control words, store the
// stack pointer in *from, load the one in *to, pop, return - into whoever switched away from `to`
// (or into fiber_start for a fresh lane, see prepare_fiber)
extern "C" void hipemul_switch(FiberCtx *from, FiberCtx *to);
asm(R"(
    .text
    .globl hipemul_switch
    .type hipemul_switch,@function
hipemul_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemul_switch,.-hipemul_switch
)");
#endif

void switch_ctx(FiberCtx *from, FiberCtx *to) {
#if HIPEMUL_UCONTEXT
    swapcontext(from, to);
#else
    hipemul_switch(from, to);
#endif
}

void fiber_entry() {
    Block *b = cur_block;
    (*b->fn)();
    b->fibers[b->cur].done = true;
    ++b->progress;
    // (ucontext: uc_link returns to the scheduler)
}

#if !HIPEMUL_UCONTEXT
extern "C" void hipemul_fiber_start() {
    fiber_entry();
    Block *b = cur_block;
    hipemul_switch(&b->fibers[b->cur].ctx, &b->sched);   // a finished lane is never resumed
    abort();
}
#endif

void prepare_fiber(Block &blk, Fiber &f, unsigned t) {
#if HIPEMUL_UCONTEXT
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stack_of(t);
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = &blk.sched;
    makecontext(&f.ctx, fiber_entry, 0);
#else
    (void)blk;
    // the frame hipemul_switch pops: control words | r15 r14 r13 r12 rbx rbp | return address.  The return
    // address sits at a 16-byte boundary, so the started function sees the stack as after a call.
    uintptr_t top = (reinterpret_cast<uintptr_t>(stack_of(t)) + kStackBytes) & ~uintptr_t(15);
    uint64_t *ret = reinterpret_cast<uint64_t *>(top - 32);
    ret[0] = reinterpret_cast<uint64_t>(&hipemul_fiber_start);
    ret[1] = 0;   // (a fake caller frame: never returned into)
    uint64_t *sp = ret - 7;
    for (int i = 1; i < 7; ++i) sp[i] = 0;
    uint32_t *cw = reinterpret_cast<uint32_t *>(sp);
    cw[0] = 0x1f80;   // MXCSR: all exceptions masked, round to nearest
    cw[1] = 0x037f;   // x87 control word: the ABI's default
    f.ctx.sp = sp;
#endif
}
}  // namespace

void fiber_yield() {
    Block *b = cur_block;
    switch_ctx(&b->fibers[b->cur].ctx, &b->sched);
}
void note_progress() { ++cur_block->progress; }
const char *&fiber_waiting_at() { return cur_block->fibers[cur_block->cur].waiting_at; }

void run_block(Block &blk) {
    const unsigned nt = blk.block.x;
    blk.bar.init(nt);
    for (auto &w : blk.waves) w.bar.init(64);
    Block *outer = cur_block;
    const hipemul_idx o_t = threadIdx, o_b = blockIdx, o_bd = blockDim, o_gd = gridDim;
    cur_block = &blk;
    blockIdx = {blk.bx, blk.by, 0};
    blockDim = {nt, 1, 1};
    gridDim = {blk.grid.x, blk.grid.y, 1};
    for (unsigned t = 0; t < nt; ++t) {
        Fiber &f = blk.fibers[t];
        f.done = false;
        f.waiting_at = "";
        prepare_fiber(blk, f, t);
    }
    unsigned live = nt;
    while (live > 0) {
        const unsigned long before = blk.progress;
        for (unsigned t = 0; t < nt; ++t) {
            Fiber &f = blk.fibers[t];
            if (f.done) continue;
            blk.cur = t;
            threadIdx = {t, 0, 0};
            switch_ctx(&blk.sched, &f.ctx);
            if (f.done) --live;
        }
        if (live > 0 && blk.progress == before) {
            for (unsigned t = 0; t < nt; ++t)
                if (!blk.fibers[t].done) {
                    fprintf(stderr,
                            "hipemul: DEADLOCK at %s (block %u thread %u): lanes diverged around a "
                            "wave/block collective\n",
                            blk.fibers[t].waiting_at, blk.bx, t);
                    break;
                }
            abort();
        }
    }
    cur_block = outer;
    threadIdx = o_t;
    blockIdx = o_b;
    blockDim = o_bd;
    gridDim = o_gd;
}
}  // namespace hipemul

namespace gnntrk {
size_t sort_pairs_temp_bytes(int64_t) { return 16; }
int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, int64_t n, int, void *, size_t, hipStream_t) {
    std::vector<int64_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return keys_in[a] < keys_in[b]; });
    std::vector<uint32_t> k(n), v(n);
    for (int64_t i = 0; i < n; ++i) {
        k[i] = keys_in[o[i]];
        v[i] = vals_in[o[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), vals_out);
    return 0;
}
size_t sort_pairs_u64_temp_bytes(int64_t) { return 16; }
int sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, int64_t n, void *, size_t,
                   hipStream_t) {
    std::vector<int64_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return keys_in[a] < keys_in[b]; });
    std::vector<unsigned long long> k(n);
    std::vector<uint32_t> v(n);
    for (int64_t i = 0; i < n; ++i) {
        k[i] = keys_in[o[i]];
        v[i] = vals_in[o[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), vals_out);
    return 0;
}
// (only the bits below end_bit order the keys, as in the device library's radix passes)
int sort_pairs_u64_bits(const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                        uint32_t *vals_out, int64_t n, int end_bit, void *, size_t, hipStream_t) {
    const unsigned long long m = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1);
    std::vector<int64_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return (keys_in[a] & m) < (keys_in[b] & m); });
    std::vector<unsigned long long> k(n);
    std::vector<uint32_t> v(n);
    for (int64_t i = 0; i < n; ++i) {
        k[i] = keys_in[o[i]];
        v[i] = vals_in[o[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), vals_out);
    return 0;
}
}  // namespace gnntrk
