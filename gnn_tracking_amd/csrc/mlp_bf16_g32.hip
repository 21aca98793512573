// Backward instantiations of the bf16-storage fused MLP with an fp32 upstream gradient
// (GNNTRK_EPI_SIGMOID: the edge-weight head feeds the fp32 BCE loss).  Separate
// translation unit only to halve the build time of mlp_bf16.hip.
#include "mlp_bf16_kernels.h"

namespace gnntrk {

int launch_bwd16_g32(const gnntrk_mlp_bwd_args *a, const SlotPlan &P, int GT, int grid, int grid_buf, int *used,
                     float *part, uint8_t *trash, hipStream_t stream) {
    return launch_bwd16<true>(a, P, GT, grid, grid_buf, used, part, trash, stream);
}

}  // namespace gnntrk
