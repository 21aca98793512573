// Hidden width 128 with biases in bf16 storage: eight hidden tiles with accumulator-initialised biases
// (SlotPlan::bias_init; with a constant-one row it would be nine tiles - 81 weight-gradient tiles for the
// middle layer alone).  Own translation unit because it is compiled WITHOUT -amdgpu-mfma-vgpr-form (see
// _build.py): with the MFMA results forced into VGPRs the compiler's AGPR-copy rewrite crashes on the
// eight-tile backward with two gradient tiles (ROCm 7.2).  The three-layer backward instantiations spill
// 117-133 registers (388 accumulators + the tile state in 512 registers) - still two orders of magnitude
// faster than the library-GEMM path these models took before.
#include "mlp_bf16_kernels.h"

namespace gnntrk {

int launch_fwd16_bi8(const gnntrk_mlp_fwd_args *a, const SlotPlan &P, int grid, hipStream_t stream) {
    if (P.KI != 1 || P.HT != 8) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: no instantiation (hidden 128)");
    const bool three = a->mlp.n_layers == 3, sig = a->epilogue == GNNTRK_EPI_SIGMOID;
#define GNNTRK_LAUNCH(T_, S_)                                                        \
    {                                                                                \
        auto kfn = mlp16_fwd_bi_kernel<1, 8, T_, S_>;                                \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);            \
    }
    if (three && sig) GNNTRK_LAUNCH(true, true)
    else if (three) GNNTRK_LAUNCH(true, false)
    else if (sig) GNNTRK_LAUNCH(false, true)
    else GNNTRK_LAUNCH(false, false)
#undef GNNTRK_LAUNCH
    return check_launch("mlp_forward_bf16");
}

int launch_bwd16_bi8(const gnntrk_mlp_bwd_args *a, const SlotPlan &P, int GT, int g32, int grid, float *part,
                     uint8_t *trash, hipStream_t stream) {
    if (P.KI != 1 || P.HT != 8 || GT < 0 || GT > 2)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: no instantiation (hidden 128)");
    const bool three = a->mlp.n_layers == 3;
    BufPlan B;
    make_buf_plan(B, P, a, GT);
#define GNNTRK_LAUNCH(GT_, T_, G_)                                                               \
    {                                                                                            \
        auto kfn = mlp16_bwd_bi_kernel<1, 8, GT_, T_, G_>;                                       \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B);        \
    }
#define GNNTRK_GT(GT_)                                                                           \
    if (GT == GT_) {                                                                             \
        if (three && g32) GNNTRK_LAUNCH(GT_, true, true)                                         \
        else if (three) GNNTRK_LAUNCH(GT_, true, false)                                          \
        else if (g32) GNNTRK_LAUNCH(GT_, false, true)                                            \
        else GNNTRK_LAUNCH(GT_, false, false)                                                    \
    }
    GNNTRK_GT(0) GNNTRK_GT(1) GNNTRK_GT(2)
#undef GNNTRK_GT
#undef GNNTRK_LAUNCH
    return check_launch("mlp_backward_bf16");
}

}  // namespace gnntrk
