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

void fiber_entry() {
    Block *b = cur_block;
    (*b->fn)();
    b->fibers[b->cur].done = true;
    ++b->progress;
    // (uc_link returns to the scheduler)
}
}  // namespace

void fiber_yield() {
    Block *b = cur_block;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
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
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stack_of(t);
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = &blk.sched;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    unsigned live = nt;
    while (live > 0) {
        const unsigned long before = blk.progress;
        for (unsigned t = 0; t < nt; ++t) {
            Fiber &f = blk.fibers[t];
            if (f.done) continue;
            blk.cur = t;
            threadIdx = {t, 0, 0};
            swapcontext(&blk.sched, &f.ctx);
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
}  // namespace gnntrk
