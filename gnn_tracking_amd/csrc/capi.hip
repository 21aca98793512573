// extern "C" surface of libgnntrk.so (include/gnntrk.h): argument plumbing, error
// strings, device queries.  No kernels here.
#include <stdio.h>
#include <string.h>

#include "host_util.h"

// ABI layout guards: gnn_tracking_amd/_capi.py mirrors these structs with ctypes
static_assert(sizeof(gnntrk_seg) == 32, "gnntrk_seg layout");
static_assert(sizeof(gnntrk_mlp) == 64, "gnntrk_mlp layout");
static_assert(sizeof(gnntrk_mlp_fwd_args) == 448, "gnntrk_mlp_fwd_args layout");
static_assert(sizeof(gnntrk_mlp_bwd_args) == 808, "gnntrk_mlp_bwd_args layout");
static_assert(sizeof(gnntrk_graph_index) == 72, "gnntrk_graph_index layout");
static_assert(sizeof(gnntrk_graph_index_carry) == 48, "gnntrk_graph_index_carry layout");
static_assert(sizeof(gnntrk_resfcnn) == 8 * (5 + 2 * GNNTRK_RESFCNN_MAX_HIDDEN) + 32, "gnntrk_resfcnn layout");
static_assert(sizeof(gnntrk_resfcnn_grads) == 8 * (5 + 2 * GNNTRK_RESFCNN_MAX_HIDDEN), "gnntrk_resfcnn_grads layout");
static_assert(sizeof(gnntrk_hinge_args) == 56, "gnntrk_hinge_args layout");

namespace gnntrk {

static thread_local char g_err[512] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return GNNTRK_OK;
    const char *s = hipGetErrorString(e);
    const bool oom = s && (strstr(s, "out of memory") || strstr(s, "OutOfMemory"));
    // utils/oom.py:12-18 of the reference looks for "out of memory" in the message
    snprintf(g_err, sizeof(g_err), "%s: HIP error: %s%s", what, s ? s : "?",
             oom ? " (HIP out of memory)" : "");
    return oom ? GNNTRK_ENOMEM : GNNTRK_EHIP;
}

int check_launch(const char *what) { return check_hip(hipGetLastError(), what); }

int cu_count() {
    static thread_local int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
        n = 256;
    cached = n;
    return n;
}

// segment.hip
int segment_sum_launch(const float *, int, int, const int32_t *, const int32_t *, int64_t, float *,
                       int, int, hipStream_t);
int permute_rows_launch(const float *, int, int, const int32_t *, int64_t, float *, int, int,
                        hipStream_t);
int axpby_launch(float, const float *, float, const float *, const float *, float *, int64_t,
                 hipStream_t);
size_t bce_ws_bytes(int64_t);
int bce_forward_launch(const float *, const float *, const int64_t *, const float *, float, int64_t,
                       float *, void *, size_t, hipStream_t);
int bce_backward_launch(const float *, const float *, const int64_t *, const float *, float,
                        int64_t, const float *, float *, hipStream_t);
int edge_targets_csr_launch(const void *, int, const int32_t *, const int32_t *, const float *, float, int64_t, float *,
                            hipStream_t);
int bce_csr_launch(const float *, const uint8_t *, const int32_t *, const float *, float, int64_t, float *, float *, void *,
                   size_t, hipStream_t);
int focal_forward_launch(const float *, const float *, const int64_t *, const float *, float, float, float, float,
                         int, int64_t, float *, void *, size_t, hipStream_t);
int focal_backward_launch(const float *, const float *, const int64_t *, const float *, float, float, float, float,
                          int, int64_t, const float *, float *, hipStream_t);
// graph_index.hip
size_t graph_index_ws_bytes(int64_t, int64_t, int);
int graph_index_place(const gnntrk_graph_index *, int64_t, int64_t, const gnntrk_graph_index *, const uint8_t *, uint8_t *,
                      const uint16_t *, uint16_t *, const int32_t *, int32_t *, const int32_t *, int32_t *, hipStream_t);
size_t node_order_ws_bytes(int64_t);
int node_order(const float *, int64_t, const int64_t *, int64_t, int64_t, int32_t *, int32_t *, void *, size_t, hipStream_t);
int graph_index_build(const int64_t *, const gnntrk_graph_index *, const gnntrk_graph_index_carry *, void *, size_t, int,
                      hipStream_t);

// knn.hip
int knn_search_launch(const float *, int64_t, int, int, int, float, const int64_t *, int, int32_t *, int32_t *,
                      hipStream_t);
int knn_emit_launch(const int32_t *, const int32_t *, int64_t, int, int, int64_t *, int64_t *, int64_t,
                    hipStream_t);
size_t knn_workspace_bytes(int64_t, int, int);
int knn_search_ws_launch(const float *, int64_t, int, int, int, float, const int64_t *, int, int32_t *, int32_t *,
                         void *, size_t, int, hipStream_t);
int edge_features_launch(const float *, int, int, const int64_t *, int64_t, float *, hipStream_t);
int edge_labels_launch(const int64_t *, const int64_t *, int64_t, int64_t *, hipStream_t);

// oc.hip
int good_node_mask_launch(const float *, const int64_t *, const float *, const float *, int64_t, float,
                          float, uint8_t *, hipStream_t);
size_t oc_select_ws_bytes(int64_t);
int oc_select_launch(const float *, const int64_t *, const uint8_t *, int64_t, int, int32_t *,
                     int32_t *, int32_t *, void *, size_t, hipStream_t);
size_t oc_forward_ws_bytes(int64_t);
int oc_forward_launch(const gnntrk_oc_args *, float *, void *, size_t, hipStream_t);
size_t oc_spatial_ws_bytes(int64_t, int);
int oc_forward_spatial_launch(const gnntrk_oc_args *, float *, void *, size_t, hipStream_t);
int oc_backward_spatial_launch(const gnntrk_oc_args *, const float *, const float *, float *, float *, int64_t, void *,
                               size_t, hipStream_t);
size_t oc_backward_ws_bytes(int64_t, int);
int oc_backward_launch(const gnntrk_oc_args *, const float *, const float *, float *, float *, int64_t, void *,
                       size_t, hipStream_t);

}  // namespace gnntrk

using namespace gnntrk;

extern "C" {

int gnntrk_version(void) { return GNNTRK_VERSION; }
const char *gnntrk_last_error(void) { return g_err; }
int gnntrk_device_cu_count(void) { return cu_count(); }

size_t gnntrk_graph_index_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    return graph_index_ws_bytes(n_nodes, n_edges, 0);
}
int gnntrk_graph_index_build(const int64_t *edge_index, const gnntrk_graph_index *out,
                             void *workspace, size_t workspace_bytes, void *stream) {
    return graph_index_build(edge_index, out, nullptr, workspace, workspace_bytes, 0, (hipStream_t)stream);
}
int gnntrk_graph_index_build_ex(const int64_t *edge_index, const gnntrk_graph_index *out, void *workspace,
                                size_t workspace_bytes, int32_t flags, void *stream) {
    return graph_index_build(edge_index, out, nullptr, workspace, workspace_bytes, flags, (hipStream_t)stream);
}
size_t gnntrk_graph_index_workspace_bytes_carry(int64_t n_nodes, int64_t n_edges, int32_t carry_rows) {
    return graph_index_ws_bytes(n_nodes, n_edges, carry_rows);
}
int gnntrk_graph_index_build_carry(const int64_t *edge_index, const gnntrk_graph_index *out,
                                   const gnntrk_graph_index_carry *carry, void *workspace, size_t workspace_bytes,
                                   int32_t flags, void *stream) {
    return graph_index_build(edge_index, out, carry, workspace, workspace_bytes, flags, (hipStream_t)stream);
}
int gnntrk_graph_index_place(const gnntrk_graph_index *part, int64_t node_offset, int64_t edge_offset,
                             const gnntrk_graph_index *batch, const uint8_t *label_part, uint8_t *label_batch,
                             const uint16_t *rows_part, uint16_t *rows_batch, const int32_t *node_perm_part,
                             int32_t *node_perm_batch, const int32_t *node_rank_part, int32_t *node_rank_batch, void *stream) {
    return graph_index_place(part, node_offset, edge_offset, batch, label_part, label_batch, rows_part, rows_batch,
                             node_perm_part, node_perm_batch, node_rank_part, node_rank_batch, (hipStream_t)stream);
}
size_t gnntrk_node_order_workspace_bytes(int64_t n_nodes) { return node_order_ws_bytes(n_nodes); }
int gnntrk_node_order(const float *key, int64_t key_stride, const int64_t *batch, int64_t n_events, int64_t n_nodes,
                      int32_t *perm, int32_t *rank, void *workspace, size_t workspace_bytes, void *stream) {
    return node_order(key, key_stride, batch, n_events, n_nodes, perm, rank, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

int gnntrk_mlp_forward(const gnntrk_mlp_fwd_args *args, void *stream) {
    return mlp_forward_launch(args, (hipStream_t)stream);
}
int gnntrk_rows_to_bf16(const float *in, int32_t dim, int32_t in_stride, const int32_t *idx, int64_t n_rows,
                        uint16_t *out, int32_t out_stride, void *stream) {
    return rows_to_bf16_launch(in, dim, in_stride, idx, n_rows, out, out_stride, (hipStream_t)stream);
}
int gnntrk_segment_sum_bf16(const uint16_t *rows, int32_t dim, int32_t row_stride, const int32_t *rowptr,
                            const int32_t *pos, int64_t n_segments, uint16_t *out, int32_t out_stride,
                            void *stream) {
    return segment_sum_bf16_launch(rows, dim, row_stride, rowptr, pos, n_segments, out, out_stride, nullptr, 0,
                                   (hipStream_t)stream);
}
int gnntrk_segment_sum_bf16_add(const uint16_t *rows, int32_t dim, int32_t row_stride, const int32_t *rowptr,
                                const int32_t *pos, int64_t n_segments, const uint16_t *addend,
                                int32_t addend_stride, uint16_t *out, int32_t out_stride, void *stream) {
    return segment_sum_bf16_launch(rows, dim, row_stride, rowptr, pos, n_segments, out, out_stride, addend,
                                   addend_stride, (hipStream_t)stream);
}
int gnntrk_permute_rows_bf16(const uint16_t *in, int32_t dim, int32_t in_stride, const int32_t *idx,
                             int64_t n_rows, uint16_t *out, int32_t out_stride, int32_t scatter, void *stream) {
    return permute_rows_bf16_launch(in, dim, in_stride, idx, n_rows, out, out_stride, scatter,
                                    (hipStream_t)stream);
}
int gnntrk_mlp_forward_bf16(const gnntrk_mlp_fwd_args *args, void *stream) {
    return mlp_forward_bf16_launch(args, (hipStream_t)stream);
}
size_t gnntrk_mlp_backward_bf16_workspace_bytes(const gnntrk_mlp *mlp) {
    return mlp_backward_bf16_ws_bytes(mlp);
}
int gnntrk_mlp_backward_bf16_max_terms(const gnntrk_mlp_bwd_args *args) { return mlp_backward_bf16_max_terms(args); }
int gnntrk_mlp_backward_bf16_can_fold(const gnntrk_mlp_bwd_args *args) { return mlp_backward_bf16_can_fold(args); }
int gnntrk_fold_finish_bf16(uint16_t *out, int32_t out_stride, int64_t n_nodes, const int32_t *rowptr,
                            int64_t n_units, const uint16_t *x, int32_t x_stride, void *stream) {
    return fold_finish_bf16_launch(out, out_stride, n_nodes, rowptr, out + n_nodes * out_stride, n_units, x, x_stride,
                                   (hipStream_t)stream);
}
int gnntrk_mlp_backward_bf16(const gnntrk_mlp_bwd_args *args, void *workspace, size_t workspace_bytes,
                             void *stream) {
    return mlp_backward_bf16_launch(args, workspace, workspace_bytes, (hipStream_t)stream);
}
int gnntrk_mlp_forward_bf16_kernel_name(const gnntrk_mlp_fwd_args *args, char *buf, size_t len) {
    return mlp16_fwd_kernel_name(args, buf, len);
}
int gnntrk_mlp_backward_bf16_kernel_name(const gnntrk_mlp_bwd_args *args, char *buf, size_t len) {
    return mlp16_bwd_kernel_name(args, buf, len);
}
int gnntrk_mlp_kernel_name(const gnntrk_mlp *mlp, int32_t n_seg, const gnntrk_seg *seg,
                           int32_t backward, char *buf, size_t len) {
    if (backward & 2) return mlp16_kernel_name(mlp, n_seg, seg, backward & 1, buf, len);
    return mlp_kernel_name(mlp, n_seg, seg, backward, buf, len);
}
size_t gnntrk_mlp_backward_workspace_bytes(const gnntrk_mlp *mlp) {
    return mlp_backward_ws_bytes(mlp);
}
int gnntrk_mlp_backward(const gnntrk_mlp_bwd_args *args, void *workspace, size_t workspace_bytes,
                        void *stream) {
    return mlp_backward_launch(args, workspace, workspace_bytes, (hipStream_t)stream);
}

int gnntrk_segment_sum(const float *rows, int32_t dim, int32_t row_stride, const int32_t *rowptr,
                       const int32_t *pos, int64_t n_segments, float *out, int32_t out_stride,
                       int32_t accumulate, void *stream) {
    return segment_sum_launch(rows, dim, row_stride, rowptr, pos, n_segments, out, out_stride,
                              accumulate, (hipStream_t)stream);
}
int gnntrk_permute_rows(const float *in, int32_t dim, int32_t in_stride, const int32_t *idx,
                        int64_t n_rows, float *out, int32_t out_stride, int32_t scatter,
                        void *stream) {
    return permute_rows_launch(in, dim, in_stride, idx, n_rows, out, out_stride, scatter,
                               (hipStream_t)stream);
}
int gnntrk_axpby(float a, const float *x, float b, const float *y, const float *relu_mask,
                 float *out, int64_t n, void *stream) {
    return axpby_launch(a, x, b, y, relu_mask, out, n, (hipStream_t)stream);
}

size_t gnntrk_bce_workspace_bytes(int64_t n) { return bce_ws_bytes(n); }
int gnntrk_bce_forward(const float *w, const float *y, const int64_t *src_node, const float *pt,
                       float pt_thld, int64_t n, float *loss_out, void *workspace,
                       size_t workspace_bytes, void *stream) {
    return bce_forward_launch(w, y, src_node, pt, pt_thld, n, loss_out, workspace, workspace_bytes,
                              (hipStream_t)stream);
}
int gnntrk_bce_backward(const float *w, const float *y, const int64_t *src_node, const float *pt,
                        float pt_thld, int64_t n, const float *gscale, float *gw, void *stream) {
    return bce_backward_launch(w, y, src_node, pt, pt_thld, n, gscale, gw, (hipStream_t)stream);
}

int gnntrk_bce_csr(const float *w, const uint8_t *label_csr, const int32_t *src_csr, const float *pt, float pt_thld,
                   int64_t n, float *loss, float *gw_unit, void *workspace, size_t workspace_bytes, void *stream) {
    return bce_csr_launch(w, label_csr, src_csr, pt, pt_thld, n, loss, gw_unit, workspace, workspace_bytes,
                          (hipStream_t)stream);
}
int gnntrk_edge_targets_csr(const void *y, int32_t y_is_u8, const int32_t *perm, const int32_t *src_csr,
                            const float *pt, float pt_thld, int64_t n, float *out, void *stream) {
    return edge_targets_csr_launch(y, y_is_u8, perm, src_csr, pt, pt_thld, n, out, (hipStream_t)stream);
}

int gnntrk_focal_forward(const float *w, const float *y, const int64_t *src_node, const float *pt, float pt_thld,
                         float alpha, float gamma, float pos_weight, int32_t haughty, int64_t n, float *loss_out,
                         void *workspace, size_t workspace_bytes, void *stream) {
    return focal_forward_launch(w, y, src_node, pt, pt_thld, alpha, gamma, pos_weight, haughty, n, loss_out,
                                workspace, workspace_bytes, (hipStream_t)stream);
}
int gnntrk_focal_backward(const float *w, const float *y, const int64_t *src_node, const float *pt, float pt_thld,
                          float alpha, float gamma, float pos_weight, int32_t haughty, int64_t n,
                          const float *gscale, float *gw, void *stream) {
    return focal_backward_launch(w, y, src_node, pt, pt_thld, alpha, gamma, pos_weight, haughty, n, gscale, gw,
                                 (hipStream_t)stream);
}
int gnntrk_knn_search(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k,
                      float max_radius, int32_t *nbr, int32_t *cnt, void *stream) {
    return knn_search_launch(x, n, dim, x_stride, k, max_radius, nullptr, 0, nbr, cnt, (hipStream_t)stream);
}
int gnntrk_knn_search_batched(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k,
                              float max_radius, const int64_t *seg_ptr, int32_t n_seg, int32_t *nbr,
                              int32_t *cnt, void *stream) {
    return knn_search_launch(x, n, dim, x_stride, k, max_radius, seg_ptr, n_seg, nbr, cnt, (hipStream_t)stream);
}
size_t gnntrk_knn_workspace_bytes(int64_t n, int32_t dim, int32_t k) { return knn_workspace_bytes(n, dim, k); }
int gnntrk_knn_search_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k, float max_radius,
                         const int64_t *seg_ptr, int32_t n_seg, int32_t *nbr, int32_t *cnt, void *workspace,
                         size_t workspace_bytes, int32_t flags, void *stream) {
    return knn_search_ws_launch(x, n, dim, x_stride, k, max_radius, seg_ptr, n_seg, nbr, cnt, workspace,
                                workspace_bytes, flags, (hipStream_t)stream);
}
int gnntrk_knn_emit(const int32_t *nbr, const int32_t *cnt, int64_t n, int32_t k, int64_t *offsets,
                    int64_t *edge_index, int64_t n_edges, void *stream) {
    return knn_emit_launch(nbr, cnt, n, k, k, offsets, edge_index, n_edges, (hipStream_t)stream);
}
int gnntrk_knn_emit_prefix(const int32_t *nbr, const int32_t *cnt, int64_t n, int32_t k_stride, int32_t k_take,
                           int64_t *offsets, int64_t *edge_index, int64_t n_edges, void *stream) {
    return knn_emit_launch(nbr, cnt, n, k_stride, k_take, offsets, edge_index, n_edges, (hipStream_t)stream);
}
int gnntrk_edge_labels(const int64_t *particle_id, const int64_t *edge_index, int64_t n_edges,
                       int64_t *y, void *stream) {
    return edge_labels_launch(particle_id, edge_index, n_edges, y, (hipStream_t)stream);
}
int gnntrk_edge_features(const float *x, int32_t dim, int32_t x_stride, const int64_t *edge_index,
                         int64_t n_edges, float *out, void *stream) {
    return edge_features_launch(x, dim, x_stride, edge_index, n_edges, out, (hipStream_t)stream);
}

size_t gnntrk_compact_workspace_bytes(int64_t n) { return compact_ws_bytes(n); }
int gnntrk_threshold_compact(const float *w, int64_t n, float threshold, uint8_t *mask, int32_t *idx,
                             int64_t *n_out, void *workspace, size_t workspace_bytes, void *stream) {
    return threshold_compact_launch(w, n, threshold, mask, idx, n_out, workspace, workspace_bytes,
                                    (hipStream_t)stream);
}
int gnntrk_connected_nodes(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, uint8_t *hit,
                           int32_t *node_idx, int32_t *newid, int64_t *n_out, int64_t *edge_index_out,
                           void *workspace, size_t workspace_bytes, void *stream) {
    return connected_nodes_launch(edge_index, n_edges, n_nodes, hit, node_idx, newid, n_out, edge_index_out,
                                  workspace, workspace_bytes, (hipStream_t)stream);
}

size_t gnntrk_radius_points_workspace_bytes(int64_t n, int32_t dim) { return radius_points_ws_bytes(n, dim); }
size_t gnntrk_radius_edges_workspace_bytes(int64_t m_edges) { return radius_edges_ws_bytes(m_edges); }
int gnntrk_radius_count_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius, int32_t *cnt,
                           int64_t *offsets, void *ws_points, size_t ws_points_bytes, int32_t flags, void *stream) {
    return radius_count_ws_launch(x, n, dim, x_stride, radius, cnt, offsets, ws_points, ws_points_bytes, flags,
                                  (hipStream_t)stream);
}
int gnntrk_radius_fill_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius,
                          const int64_t *offsets, int64_t m_edges, int32_t *nbr, double *dist, void *ws_points,
                          size_t ws_points_bytes, void *ws_edges, size_t ws_edges_bytes, int32_t flags, void *stream) {
    return radius_fill_ws_launch(x, n, dim, x_stride, radius, offsets, m_edges, nbr, dist, ws_points, ws_points_bytes,
                                 ws_edges, ws_edges_bytes, flags, (hipStream_t)stream);
}
int gnntrk_radius_count(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius, int32_t *cnt,
                        int64_t *offsets, void *stream) {
    return radius_count_launch(x, n, dim, x_stride, radius, cnt, offsets, (hipStream_t)stream);
}
int gnntrk_radius_fill(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius,
                       const int64_t *offsets, int32_t *nbr, double *dist, void *stream) {
    return radius_fill_launch(x, n, dim, x_stride, radius, offsets, nbr, dist, (hipStream_t)stream);
}
int gnntrk_dbscan_init(const int64_t *offsets, const double *dist, int64_t n, double eps, int32_t min_pts,
                       uint8_t *core, int32_t *root, void *stream) {
    return dbscan_init_launch(offsets, dist, n, eps, min_pts, core, root, (hipStream_t)stream);
}
int gnntrk_dbscan_propagate(const int64_t *offsets, const int32_t *nbr, const double *dist, int64_t n, double eps,
                            const uint8_t *core, int32_t *root, int32_t rounds, int32_t *changed, void *stream) {
    return dbscan_propagate_launch(offsets, nbr, dist, n, eps, core, root, rounds, changed, (hipStream_t)stream);
}
size_t gnntrk_dbscan_workspace_bytes(int64_t n) { return dbscan_ws_bytes(n); }
int gnntrk_dbscan_labels(const int64_t *offsets, const int32_t *nbr, const double *dist, int64_t n, double eps,
                         const uint8_t *core, const int32_t *root, int64_t *labels, int64_t *n_clusters,
                         void *workspace, size_t workspace_bytes, void *stream) {
    return dbscan_labels_launch(offsets, nbr, dist, n, eps, core, root, labels, n_clusters, workspace,
                                workspace_bytes, (hipStream_t)stream);
}

int gnntrk_good_node_mask(const float *pt, const int64_t *particle_id, const float *reconstructable,
                          const float *eta, int64_t n, float pt_thld, float max_eta, uint8_t *mask,
                          void *stream) {
    return good_node_mask_launch(pt, particle_id, reconstructable, eta, n, pt_thld, max_eta, mask,
                                 (hipStream_t)stream);
}
size_t gnntrk_oc_select_workspace_bytes(int64_t n) { return oc_select_ws_bytes(n); }
int gnntrk_oc_select_cps(const float *score, const int64_t *particle_id, const uint8_t *mask,
                         int64_t n, int32_t mode, int32_t *alphas, int32_t *gid, int32_t *n_cp,
                         void *workspace, size_t workspace_bytes, void *stream) {
    return oc_select_launch(score, particle_id, mask, n, mode, alphas, gid, n_cp, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
size_t gnntrk_oc_forward_workspace_bytes(int64_t n) { return oc_forward_ws_bytes(n); }
int gnntrk_oc_forward(const gnntrk_oc_args *args, float *out, void *workspace, size_t workspace_bytes,
                      void *stream) {
    return oc_forward_launch(args, out, workspace, workspace_bytes, (hipStream_t)stream);
}
size_t gnntrk_oc_spatial_workspace_bytes(int64_t n, int32_t dim) { return oc_spatial_ws_bytes(n, dim); }
int gnntrk_oc_forward_spatial(const gnntrk_oc_args *args, float *out, void *spatial, size_t spatial_bytes,
                              void *stream) {
    return oc_forward_spatial_launch(args, out, spatial, spatial_bytes, (hipStream_t)stream);
}
int gnntrk_oc_backward_spatial(const gnntrk_oc_args *args, const float *g, const float *fwd, float *gx,
                               float *gbeta, int64_t max_cps, void *spatial, size_t spatial_bytes, void *stream) {
    return oc_backward_spatial_launch(args, g, fwd, gx, gbeta, max_cps, spatial, spatial_bytes, (hipStream_t)stream);
}
size_t gnntrk_oc_backward_workspace_bytes(int64_t n, int32_t dim) { return oc_backward_ws_bytes(n, dim); }
int gnntrk_oc_backward(const gnntrk_oc_args *args, const float *g, const float *fwd, float *gx,
                       float *gbeta, int64_t max_cps, void *workspace, size_t workspace_bytes, void *stream) {
    return oc_backward_launch(args, g, fwd, gx, gbeta, max_cps, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
