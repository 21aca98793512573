// Forward launchers of the bf16-storage fused MLP kernels (mlp_bf16_kernels.h).  Own translation
// unit: the forward kernels are compiled with -amdgpu-sched-strategy=max-ilp (_build.py), which
// suits their load-heavy tile loop (edge-weight head 1.58 -> 1.39 ms) but not the backward.
#include <atomic>
#include "mlp_bf16_kernels.h"

namespace gnntrk {

// The persistent grid of an instantiation = the workgroups of it that are RESIDENT at once (asked of the runtime once
// per instantiation: registers and LDS decide), not a fixed five per CU: a larger grid runs in rounds whose last one
// leaves CUs idle (round 5: the hot instantiations hold 164-230 registers = two, not five, workgroups per CU).
#define GNNTRK_FWD16_GRID(kfn_)  /* (occupancy cached per DEVICE ordinal, atomically: launch threads race, devices differ) */ \
    {                                                                                   \
        static std::atomic<int> occ_dev_[16];                                           \
        int dev_ = 0;                                                                   \
        (void)hipGetDevice(&dev_);                                                      \
        std::atomic<int>& slot_ = occ_dev_[dev_ & 15];                                  \
        int occ_ = slot_.load(std::memory_order_relaxed);                               \
        if (occ_ <= 0) {                                                                \
            int o_ = 0;                                                                 \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_, kfn_, kBlock, 0) != hipSuccess || o_ < 1) \
                o_ = kFwd16BlocksPerCu;                                                 \
            occ_ = o_ > 8 ? 8 : o_;                                                     \
            slot_.store(occ_, std::memory_order_relaxed);                               \
        }                                                                               \
        grid = grid16(a->n_rows, occ_, kWaves);                                         \
        if (grid > kFwdMaxBlocks) grid = kFwdMaxBlocks - kFwdMaxBlocks % 8;             \
    }
#define GNNTRK_FWD16_LAUNCH(KI_, HT_, T_, S_, R_)                                       \
    {                                                                                   \
        if (wide) {                                                                     \
            auto kfn = mlp16_fwd_kernel<KI_, HT_, T_, S_, R_, true>;                    \
            GNNTRK_FWD16_GRID(kfn)                                                      \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);           \
        } else {                                                                        \
            auto kfn = mlp16_fwd_kernel<KI_, HT_, T_, S_, R_, false>;                   \
            GNNTRK_FWD16_GRID(kfn)                                                      \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);           \
        }                                                                               \
    }
// the I/O skeleton of the two large forward shapes (three hidden tiles, shared output tile): debug_flags & 4096
#define GNNTRK_FWD16_SKEL(S_, W_)                                                       \
    if (!launched && (a->debug_flags & 4096) && P.KI == 1 && P.HT == 3 && three && share && sig == S_ && wide == W_) { \
        auto kfn = mlp16_fwd_skel_kernel<1, 3, true, S_, 4, W_>;                        \
        GNNTRK_FWD16_GRID(kfn)                                                          \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);               \
        launched = true;                                                                \
    }
#define GNNTRK_FWD16_CASE(KI_, HT_)                                                     \
    if (!launched && P.KI == KI_ && P.HT == HT_) {                                      \
        if (three && sig && share) GNNTRK_FWD16_LAUNCH(KI_, HT_, true, true, 4)         \
        else if (three && sig) GNNTRK_FWD16_LAUNCH(KI_, HT_, true, true, 1)             \
        else if (three && share) GNNTRK_FWD16_LAUNCH(KI_, HT_, true, false, 4)          \
        else if (three) GNNTRK_FWD16_LAUNCH(KI_, HT_, true, false, 1)                   \
        else if (sig && share) GNNTRK_FWD16_LAUNCH(KI_, HT_, false, true, 4)            \
        else if (sig) GNNTRK_FWD16_LAUNCH(KI_, HT_, false, true, 1)                     \
        else if (share) GNNTRK_FWD16_LAUNCH(KI_, HT_, false, false, 4)                  \
        else GNNTRK_FWD16_LAUNCH(KI_, HT_, false, false, 1)                             \
        launched = true;                                                                \
    }

// hidden widths 64 .. 95 (five / six hidden tiles): the plain forms only (own output tile per tile,
// 8-byte loads) - four instantiations per shape instead of sixteen
#define GNNTRK_FWD16_CASE_PLAIN(KI_, HT_)                                                   \
    if (P.KI == KI_ && P.HT == HT_) {                                                       \
        if (three && sig) { auto kfn = mlp16_fwd_kernel<KI_, HT_, true, true, 1, false>;    \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else if (three) { auto kfn = mlp16_fwd_kernel<KI_, HT_, true, false, 1, false>;     \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else if (sig) { auto kfn = mlp16_fwd_kernel<KI_, HT_, false, true, 1, false>;       \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else { auto kfn = mlp16_fwd_kernel<KI_, HT_, false, false, 1, false>;               \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        launched = true;                                                                    \
    }

// hidden width 64 with biases (SlotPlan::bias_init): plain forms of the accumulator-initialised kernels
#define GNNTRK_FWD16_CASE_BI(KI_, HT_)                                                      \
    if (P.KI == KI_ && P.HT == HT_) {                                                       \
        if (three && sig) { auto kfn = mlp16_fwd_bi_kernel<KI_, HT_, true, true>;           \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else if (three) { auto kfn = mlp16_fwd_bi_kernel<KI_, HT_, true, false>;            \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else if (sig) { auto kfn = mlp16_fwd_bi_kernel<KI_, HT_, false, true>;              \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else { auto kfn = mlp16_fwd_bi_kernel<KI_, HT_, false, false>;                      \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        launched = true;                                                                    \
    }

// outputs of 17 .. 48 features / inputs of 65 .. 128 slots (three hidden tiles): the plain output-tile kernels
#define GNNTRK_FWD16_CASE_OT(KI_, OT_)                                                      \
    if (P.KI == KI_ && ot == OT_) {                                                         \
        if (three) { auto kfn = mlp16_fwd_ot_kernel<KI_, 3, OT_, true>;                     \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        else { auto kfn = mlp16_fwd_ot_kernel<KI_, 3, OT_, false>;                          \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a); }             \
        launched = true;                                                                    \
    }

// exact forward instantiation
int mlp16_fwd_kernel_name(const gnntrk_mlp_fwd_args *a, char *buf, size_t len) {
    if (!a || !buf || len == 0) return fail(GNNTRK_EINVAL, "mlp_kernel_name: bad argument");
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, nullptr);
    const bool plain = P.HT >= 5;
    if (a->mlp.out_dim > 16 || P.KI > 2) {
        snprintf(buf, len, "mlp16_fwd_ot_kernel<%d, %d, %d, %s>", P.KI, P.HT, (a->mlp.out_dim + 15) / 16,
                 a->mlp.n_layers == 3 ? "true" : "false");
        return GNNTRK_OK;
    }
    if (P.bias_init) {
        snprintf(buf, len, "mlp16_fwd_bi_kernel<%d, %d, %s, %s>", P.KI, P.HT, a->mlp.n_layers == 3 ? "true" : "false",
                 a->epilogue == GNNTRK_EPI_SIGMOID ? "true" : "false");
        return GNNTRK_OK;
    }
    snprintf(buf, len, "mlp16_fwd_kernel<%d, %d, %s, %s, %d, %s>", P.KI, P.HT, a->mlp.n_layers == 3 ? "true" : "false",
             a->epilogue == GNNTRK_EPI_SIGMOID ? "true" : "false", (a->mlp.out_dim <= 4 && !plain) ? 4 : 1,
             (!plain && wide_ok(P, a->seg, a->n_rows)) ? "true" : "false");
    return GNNTRK_OK;
}

int mlp_forward_bf16_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: NULL args");
    int rc = check_bf16_mlp(a->mlp, a->n_seg, a->seg, "mlp_forward_bf16", a->n_rows);
    if (rc) return rc;
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad epilogue");
    if (a->n_rows == 0) return GNNTRK_OK;  // nothing to write: NULL row pointers are fine
    const int out_pad = (a->mlp.out_dim + 3) / 4 * 4;
    if (a->epilogue == GNNTRK_EPI_SIGMOID) {
        if (!a->out || a->out_stride < a->mlp.out_dim)
            return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad (fp32) output");
    } else if (!a->out || a->out_stride < out_pad || a->out_stride % 4 != 0 || ((uintptr_t)a->out & 7) != 0) {
        return fail(GNNTRK_EINVAL,
                    "mlp_forward_bf16: output rows must be 8-byte aligned bf16, stride a multiple of 4 >= "
                    "out_dim rounded up to 4");
    }
    if (a->epilogue == GNNTRK_EPI_RESIDUAL &&
        (!a->res || a->res_stride < out_pad || a->res_stride % 4 != 0 || ((uintptr_t)a->res & 7) != 0))
        return fail(GNNTRK_EINVAL, "mlp_forward_bf16: residual epilogue needs padded bf16 res rows");
    if (a->n_rows < 0 || a->n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad n_rows");
    if (a->n_rows == 0) return GNNTRK_OK;
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, nullptr);
    if (!P.ok || P.KI > kMaxChunks16 / 8)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: shape outside the instantiations (include/gnntrk.h)");
    const bool three = a->mlp.n_layers == 3, sig = a->epilogue == GNNTRK_EPI_SIGMOID;
    const bool share = a->mlp.out_dim <= 4;  // four tiles share one output tile and one store
    const bool wide = wide_ok(P, a->seg, a->n_rows);  // one 16-byte load per lane and k-step
    // (five / six hidden tiles: 230 .. 330 registers per lane - two workgroups per CU are resident with one
    //  k-step, one with two; the grid of the persistent tile schedule matches what is resident)
    int grid = grid16(a->n_rows, P.HT >= 5 ? ((P.KI == 1 && P.HT <= 6) ? 2 : 1) : kFwd16BlocksPerCu, kWaves);
    if (grid > kFwdMaxBlocks) grid = kFwdMaxBlocks - kFwdMaxBlocks % 8;
    bool launched = false;
    if (a->mlp.out_dim > 16 || P.KI > 2) {
        if (sig) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: SIGMOID epilogue with a wide input / output");
        const int ot = (a->mlp.out_dim + 15) / 16;
        grid = grid16(a->n_rows, P.KI > 2 ? 1 : 2, kWaves);
        GNNTRK_FWD16_CASE_OT(1, 2) GNNTRK_FWD16_CASE_OT(1, 3)
        GNNTRK_FWD16_CASE_OT(2, 2) GNNTRK_FWD16_CASE_OT(2, 3)
        GNNTRK_FWD16_CASE_OT(3, 1) GNNTRK_FWD16_CASE_OT(3, 2) GNNTRK_FWD16_CASE_OT(3, 3)
        GNNTRK_FWD16_CASE_OT(4, 1) GNNTRK_FWD16_CASE_OT(4, 2) GNNTRK_FWD16_CASE_OT(4, 3)
        if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: no instantiation (output tiles)");
        return check_launch("mlp_forward_bf16");
    }
    if (P.bias_init && P.HT == 8) return launch_fwd16_bi8(a, P, grid16(a->n_rows, 1, kWaves), stream);
    if (P.bias_init) {
        grid = grid16(a->n_rows, 3, kWaves);
        GNNTRK_FWD16_CASE_BI(1, 4)
        GNNTRK_FWD16_CASE_BI(2, 4)
        if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: no instantiation (bias_init)");
        return check_launch("mlp_forward_bf16");
    }
    GNNTRK_FWD16_SKEL(false, true)    // relational / object-shaped: bf16 output, 16-byte loads
    GNNTRK_FWD16_SKEL(true, false)    // the edge-weight head: fp32 sigmoid output, 8-byte loads
    GNNTRK_FWD16_CASE(1, 1)
    GNNTRK_FWD16_CASE(1, 2)
    GNNTRK_FWD16_CASE(1, 3)
    GNNTRK_FWD16_CASE(1, 4)
    GNNTRK_FWD16_CASE(2, 1)
    GNNTRK_FWD16_CASE(2, 2)
    GNNTRK_FWD16_CASE(2, 3)
    GNNTRK_FWD16_CASE(2, 4)
    GNNTRK_FWD16_CASE_PLAIN(1, 5)
    GNNTRK_FWD16_CASE_PLAIN(1, 6)
    GNNTRK_FWD16_CASE_PLAIN(2, 5)
    GNNTRK_FWD16_CASE_PLAIN(2, 6)
    GNNTRK_FWD16_CASE_PLAIN(1, 7)
    GNNTRK_FWD16_CASE_PLAIN(1, 8)
    if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: no instantiation");
    return check_launch("mlp_forward_bf16");
}

}  // namespace gnntrk
