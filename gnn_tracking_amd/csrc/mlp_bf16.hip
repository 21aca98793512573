// Backward launchers of the bf16-storage fused MLP kernels (mlp_bf16_kernels.h).  The
// instantiations with an fp32 upstream gradient (EPI_SIGMOID: the edge-weight head) are
// compiled in mlp_bf16_g32.hip, the forward kernels in mlp_bf16_fwd.hip (own scheduling
// strategy, see _build.py), so the three parts also build in parallel.
#include "mlp_bf16_kernels.h"

namespace gnntrk {

int launch_bwd16_g32(const gnntrk_mlp_bwd_args *a, const SlotPlan &P, int GT, int grid, int grid_buf, int *used, float *part,
                     uint8_t *trash, hipStream_t stream);  // mlp_bf16_g32.hip


int mlp16_kernel_name(const gnntrk_mlp *m, int n_seg, const gnntrk_seg *seg, int backward, char *buf,
                      size_t len) {
    if (!m || !seg || !buf || len == 0) return fail(GNNTRK_EINVAL, "mlp_kernel_name: bad argument");
    SlotPlan P;
    make_slot_plan(P, *m, n_seg, seg, nullptr);
    // (the backward instantiation also depends on how many input gradients are wanted:
    // GT = 1 or 2 KI gradient tiles; the name reports the k-step and hidden-tile counts)
    snprintf(buf, len, "mlp16_%s_kernel<%d, %d, %s>", backward ? "bwd" : "fwd", P.KI, P.HT,
             m->n_layers == 3 ? "true" : "false");  // (forward: + the sigmoid flag, see ops_bf16.py)
    return GNNTRK_OK;
}

// exact backward instantiation (needs the gradient slices and the epilogue)
int mlp16_bwd_kernel_name(const gnntrk_mlp_bwd_args *a, char *buf, size_t len) {
    if (!a || !buf || len == 0) return fail(GNNTRK_EINVAL, "mlp_kernel_name: bad argument");
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, a->gseg);
    const int GT = (P.GT == 0 && P.KI == 1) ? 0 : (P.GT <= 1) ? 1 : 2 * P.KI;  // 0: no input gradient wanted
    const int D = (P.KI == 1 && P.HT <= 3 && !(a->debug_flags & 64)) ? 2 : 1;  // as launch_bwd16 dispatches
    const bool g32 = a->epilogue == GNNTRK_EPI_SIGMOID;
    if (a->mlp.out_dim > 16 || P.KI > 2) {
        snprintf(buf, len, "mlp16_bwd_ot_kernel<%d, %d, %d, %s>", P.KI, P.HT, (a->mlp.out_dim + 15) / 16,
                 a->mlp.n_layers == 3 ? "true" : "false");
        return GNNTRK_OK;
    }
    if (P.bias_init) {
        snprintf(buf, len, "mlp16_bwd_bi_kernel<%d, %d, %d, %s, %s>", P.KI, P.HT, GT,
                 a->mlp.n_layers == 3 ? "true" : "false", g32 ? "true" : "false");
        return GNNTRK_OK;
    }
    BufPlan B;
    make_buf_plan(B, P, a, GT);
    const char *io = buf_io_name(B, P.KI, P.HT, GT, a->mlp.n_layers == 3, g32, a->debug_flags, a->epilogue);
    // (as rocprofv3 prints the instantiation: a template argument that itself ends in '>' is followed by a space)
    // (as rocprofv3 prints the class: the fold flag is a template argument, the launch tables use comma-free aliases)
    const char *ion = !io[0] ? "IoNone"
                      : !strcmp(io, "IoRelational<2>") ? "IoRelational<2, false>"
                      : !strcmp(io, "IoRelational<3>") ? "IoRelational<3, false>"
                      : !strcmp(io, "IoRelationalF2") ? "IoRelational<2, true>"
                      : !strcmp(io, "IoRelationalF3") ? "IoRelational<3, true>"
                      : !strcmp(io, "IoHead") ? "IoHeadT<false>"
                      : !strcmp(io, "IoHeadF") ? "IoHeadT<true>"
                      : io;
    snprintf(buf, len, "mlp16_bwd_kernel<%d, %d, %d, %s, %s, %d, %s%s>", P.KI, P.HT, GT,
             a->mlp.n_layers == 3 ? "true" : "false", g32 ? "true" : "false", D, ion,
             ion[strlen(ion) - 1] == '>' ? " " : "");
    return GNNTRK_OK;
}

// the number of upstream terms (2 or 3) the launch described by `a` can take: 3 where it runs
// buffer-addressed with a further term on the tile's rows
int mlp_backward_bf16_max_terms(const gnntrk_mlp_bwd_args *a) {
    if (!a || a->epilogue == GNNTRK_EPI_SIGMOID || a->n_rows <= 0 || a->n_gout < 1 || a->n_gout > 3) return 2;
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, a->gseg);
    if (!P.ok || P.KI != 1) return 2;
    const int GT = (P.GT == 0) ? 0 : (P.GT <= 1) ? 1 : 2;
    gnntrk_mlp_bwd_args b = *a;
    while (b.n_gout < 3) {   // probe with stand-in terms: rows of the tile, sized like the first term
        b.gout[b.n_gout] = b.gout[0];
        b.gout[b.n_gout].idx = nullptr;
        b.gout[b.n_gout].rows = (int32_t)(a->n_rows < 0x7fffffff ? a->n_rows : 0x7fffffff);
        b.n_gout += 1;
    }
    BufPlan B;
    make_buf_plan(B, P, &b, GT);
    const char *io = buf_io_name(B, P.KI, P.HT, GT, a->mlp.n_layers == 3, false, a->debug_flags, a->epilogue);
    return (io[0] && B.n_gout == 3) ? 3 : 2;
}

// 1 if the launch described by `a` (fold block filled in) runs on an instantiation that folds inside the kernel
int mlp_backward_bf16_can_fold(const gnntrk_mlp_bwd_args *a) {
    if (!a || !a->fold.ids || a->fold.seg < 0 || a->fold.seg >= a->n_seg || a->n_rows <= 0 || a->n_rows > 0x7fffffff ||
        a->n_gout < 1 || a->n_gout > 3 || (a->debug_flags & (64 | 128)))
        return 0;
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, a->gseg);
    if (!P.ok || P.KI != 1 || P.bias_init || a->mlp.out_dim > 16) return 0;
    const int GT = (P.GT == 0) ? 0 : (P.GT <= 1) ? 1 : 2;
    BufPlan B;
    make_buf_plan(B, P, a, GT);
    const char *io = buf_io_name(B, P.KI, P.HT, GT, a->mlp.n_layers == 3, a->epilogue == GNNTRK_EPI_SIGMOID, a->debug_flags,
                                 a->epilogue);
    return (io[0] && B.fold_on) ? 1 : 0;
}

// workspace = one partial block per wave | one 8-byte trash slot per lane
constexpr int kBwd16MaxWaves = kBwd16BufWaves > kWaves ? kBwd16BufWaves : kWaves;
static size_t bwd16_partial_bytes(const gnntrk_mlp *m) {
    return align_up((size_t)cu_count() * kBwd16BlocksPerCuMax * kBwd16MaxWaves * (size_t)part_total(*m) * sizeof(float), 256);
}
size_t mlp_backward_bf16_ws_bytes(const gnntrk_mlp *m) {
    if (!m) return 0;
    return bwd16_partial_bytes(m) + (size_t)cu_count() * kBwd16BlocksPerCuMax * kBwd16MaxWaves * 64 * 8;
}

int mlp_backward_bf16_launch(const gnntrk_mlp_bwd_args *a, void *ws, size_t ws_bytes, hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_backward_bf16: NULL args");
    int rc = check_bf16_mlp(a->mlp, a->n_seg, a->seg, "mlp_backward_bf16", a->n_rows);
    if (rc) return rc;
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_backward_bf16: bad epilogue");
    const bool empty = a->n_rows == 0;  // no rows: only the parameter gradients are written (zeros)
    if (a->n_gout < 1 || a->n_gout > 3)
        return fail(GNNTRK_EINVAL, "mlp_backward_bf16: bad upstream gradient terms");
    for (int t = 0; t < a->n_gout && !empty; ++t)
        if (!a->gout[t].ptr) return fail(GNNTRK_EINVAL, "mlp_backward_bf16: bad upstream gradient terms");
    const int out_pad = (a->mlp.out_dim + 3) / 4 * 4;
    if (a->epilogue == GNNTRK_EPI_SIGMOID) {
        if (a->n_gout != 1 || a->gout[0].stride < a->mlp.out_dim)
            return fail(GNNTRK_EINVAL, "mlp_backward_bf16: SIGMOID takes one fp32 upstream gradient term");
    } else {
        for (int t = 0; t < a->n_gout; ++t)
            if (a->gout[t].stride < out_pad || a->gout[t].stride % 4 != 0 || ((uintptr_t)a->gout[t].ptr & 7) != 0)
                return fail(GNNTRK_EINVAL, "mlp_backward_bf16: upstream gradient rows must be padded bf16 rows");
    }
    for (int j = 0; j < a->n_seg; ++j) {
        const gnntrk_gseg &gs = a->gseg[j];
        if (!gs.ptr) continue;
        if (gs.accumulate) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: gseg.accumulate is reserved");
        if (gs.stride < (a->seg[j].dim + 3) / 4 * 4 || gs.stride % 4 != 0 || ((uintptr_t)gs.ptr & 7) != 0)
            return fail(GNNTRK_EINVAL, "mlp_backward_bf16: gradient slices must be padded bf16 rows");
    }
    if (a->n_rows < 0 || a->n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "mlp_backward_bf16: bad n_rows");
    if (a->fold.ids && !empty) {
        const gnntrk_gfold &f = a->fold;
        if (f.seg < 0 || f.seg >= a->n_seg || !a->gseg[f.seg].ptr || a->gseg[f.seg].idx || a->seg[f.seg].idx != f.ids ||
            f.n_nodes <= 0 || a->gseg[f.seg].stride != 8 || ((uintptr_t)a->gseg[f.seg].ptr & 15) != 0 ||
            false)
            return fail(GNNTRK_EINVAL, "mlp_backward_bf16: bad fold block (include/gnntrk.h: gnntrk_gfold)");
        if (!mlp_backward_bf16_can_fold(a))
            return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: this launch does not take a fold (gnntrk_mlp_backward_bf16_can_fold)");
    }
    if (a->n_gout == 3 && a->n_rows > 0 && mlp_backward_bf16_max_terms(a) < 3)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: three upstream terms only on the buffer-addressed shapes "
                                         "(gnntrk_mlp_backward_bf16_max_terms)");
    const bool want_dw = a->gW[0] != nullptr;
    if (want_dw)
        for (int i = 0; i < a->mlp.n_layers; ++i)
            if (!a->gW[i]) return fail(GNNTRK_EINVAL, "mlp_backward_bf16: gW must be all set or all NULL");
    if (!ws || ws_bytes < mlp_backward_bf16_ws_bytes(&a->mlp))
        return fail(GNNTRK_EINVAL, "mlp_backward_bf16: workspace too small (always required)");
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, a->gseg);
    if (!P.ok || P.KI > kMaxChunks16 / 8)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: shape outside the instantiations (include/gnntrk.h)");
    const bool three = a->mlp.n_layers == 3;
    // gradient M tiles: 1 or the maximum of the k-step count (keeps the instantiation list short)
    const bool wide_io = a->mlp.out_dim > 16 || P.KI > 2;   // output tiles / wide inputs: every gradient tile
    const int GT = wide_io ? 2 * P.KI : (P.GT == 0 && P.KI == 1) ? 0 : (P.GT <= 1) ? 1 : 2 * P.KI;  // 0: no input gradient wanted
    int grid = 0, waves = kWaves, used[2] = {0, kWaves};
    if (a->n_rows > 0) {
        // (five / six hidden tiles: one workgroup per CU is resident - its share of the rows is simply larger)
        const int per_cu = (P.HT >= 5 || wide_io || (P.bias_init && P.KI >= 2)) ? 1 : (GT == 0 && P.HT <= 3 && !(a->debug_flags & 1024)) ? kBwd16BlocksPerCuLight : kBwd16BlocksPerCu;
        grid = grid16(a->n_rows, per_cu, kWaves);
        // (the buffer-addressed kernels: workgroups of kBwd16BufWaves waves, two resident per CU)
        const int grid_buf = grid16(a->n_rows, (kBwd16BufD == 1 && per_cu == kBwd16BlocksPerCu) ? 3 : (kBwd16BufWaves == kWaves || per_cu < kBwd16BlocksPerCu) ? per_cu : kBwd16BlocksPerCu, kBwd16BufWaves);
        float *part = reinterpret_cast<float *>(ws);
        uint8_t *trash = reinterpret_cast<uint8_t *>(ws) + bwd16_partial_bytes(&a->mlp);
        rc = (a->epilogue == GNNTRK_EPI_SIGMOID)
                 ? launch_bwd16_g32(a, P, GT, grid, grid_buf, used, part, trash, stream)
                 : launch_bwd16<false>(a, P, GT, grid, grid_buf, used, part, trash, stream);
        if (rc) return rc;
        grid = used[0];
        waves = used[1];
    }
    if (want_dw) {
        // one partial block per workgroup when the parameters fit the kernel's LDS image
        const bool via_lds = part_total(a->mlp) <= bwd16_img_dwords(P.KI, P.HT, GT, three) && P.HT <= 4 && !(a->debug_flags & 2048);
        rc = reduce_partials_launch(reinterpret_cast<const float *>(ws), via_lds ? grid : grid * waves, &a->mlp,
                                    a->gW, a->gb, a->accumulate_params, stream);
    }
    return rc;
}

}  // namespace gnntrk
