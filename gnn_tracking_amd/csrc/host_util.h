// Host-side helpers shared by the C-ABI translation units: thread-local last-error
// string, launch checking, device queries.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gnntrk.h"

namespace gnntrk {

// records msg as the thread's last error and returns code
int fail(int code, const char *msg);
// hipGetLastError() -> GNNTRK_OK / GNNTRK_EHIP / GNNTRK_ENOMEM (+ message)
int check_launch(const char *what);
int check_hip(hipError_t e, const char *what);
int cu_count();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// stable LSD radix sort of (key,value) pairs on the device (sort_pairs.hip: rocPRIM)
size_t sort_pairs_temp_bytes(int64_t n);
int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, int64_t n, int end_bit, void *temp, size_t temp_bytes,
                   hipStream_t stream);
size_t sort_pairs_u64_temp_bytes(int64_t n);
int sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, int64_t n, void *temp, size_t temp_bytes, hipStream_t stream);
// (keys whose bits from `end_bit` up are all zero: the passes over them are skipped)
int sort_pairs_u64_bits(const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                        uint32_t *vals_out, int64_t n, int end_bit, void *temp, size_t temp_bytes, hipStream_t stream);

// knn.hip: points sorted by (event, Morton code) in chunks of 64 with bounding boxes
void scan_counts_launch(const int32_t *cnt, int k_take, int64_t n, int64_t *off, hipStream_t stream);
int spatial_dp(int dim);
int spatial_n_chunks(int64_t n);
size_t spatial_scratch_bytes(int64_t n);
int spatial_chunks_build(const float *x, int64_t n, int dim, int stride, const int64_t *seg_ptr, int n_seg,
                         float *xs, int32_t *sidx, float *box, void *scratch, size_t scratch_bytes,
                         hipStream_t stream);

// mlp.hip
int mlp_forward_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream);
size_t mlp_backward_ws_bytes(const gnntrk_mlp *m);
int mlp_kernel_name(const gnntrk_mlp *m, int n_seg, const gnntrk_seg *seg, int backward, char *buf,
                    size_t len);
int mlp_backward_launch(const gnntrk_mlp_bwd_args *a, void *ws, size_t ws_bytes,
                        hipStream_t stream);
int reduce_partials_launch(const float *part, int n_part, const gnntrk_mlp *mlp, float *const gW[3],
                           float *const gb[3], int accumulate, hipStream_t stream);


// mlp_bf16.hip
int mlp16_kernel_name(const gnntrk_mlp *m, int n_seg, const gnntrk_seg *seg, int backward, char *buf,
                      size_t len);
int mlp16_fwd_kernel_name(const gnntrk_mlp_fwd_args *a, char *buf, size_t len);
int mlp16_bwd_kernel_name(const gnntrk_mlp_bwd_args *a, char *buf, size_t len);
int mlp_forward_bf16_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream);
// hidden width 128 with biases (eight hidden tiles, accumulator-initialised biases): mlp_bf16_bi8.hip
struct SlotPlan;
int launch_fwd16_bi8(const gnntrk_mlp_fwd_args *a, const SlotPlan &P, int grid, hipStream_t stream);
int launch_bwd16_bi8(const gnntrk_mlp_bwd_args *a, const SlotPlan &P, int GT, int g32, int grid, float *part,
                     uint8_t *trash, hipStream_t stream);
size_t mlp_backward_bf16_ws_bytes(const gnntrk_mlp *m);
int mlp_backward_bf16_launch(const gnntrk_mlp_bwd_args *a, void *ws, size_t ws_bytes, hipStream_t stream);
int mlp_backward_bf16_max_terms(const gnntrk_mlp_bwd_args *a);
int mlp_backward_bf16_can_fold(const gnntrk_mlp_bwd_args *a);


// compact.hip
size_t compact_ws_bytes(int64_t n);
int threshold_compact_launch(const float *w, int64_t n, float threshold, uint8_t *mask, int32_t *idx,
                             int64_t *n_out, void *ws, size_t ws_bytes, hipStream_t stream);
int connected_nodes_launch(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, uint8_t *hit,
                           int32_t *node_idx, int32_t *newid, int64_t *n_out, int64_t *edge_index_out,
                           void *ws, size_t ws_bytes, hipStream_t stream);

int compact_bytes_launch(const uint8_t *flags, int64_t n, int32_t *idx, int32_t *newid, int64_t *n_out, void *ws,
                         size_t ws_bytes, hipStream_t stream);

// dbscan.hip
int radius_count_launch(const float *x, int64_t n, int dim, int stride, double radius, int32_t *cnt,
                        int64_t *offsets, hipStream_t stream);
int radius_fill_launch(const float *x, int64_t n, int dim, int stride, double radius, const int64_t *off,
                       int32_t *nbr, double *dist, hipStream_t stream);
size_t radius_points_ws_bytes(int64_t n, int dim);
size_t radius_edges_ws_bytes(int64_t m_edges);
int radius_count_ws_launch(const float *x, int64_t n, int dim, int stride, double radius, int32_t *cnt,
                           int64_t *offsets, void *ws_points, size_t ws_bytes, int flags, hipStream_t stream);
int radius_fill_ws_launch(const float *x, int64_t n, int dim, int stride, double radius, const int64_t *off,
                          int64_t m_edges, int32_t *nbr, double *dist, void *ws_points, size_t ws_bytes,
                          void *ws_edges, size_t ws_edges_bytes, int flags, hipStream_t stream);
int dbscan_init_launch(const int64_t *off, const double *dist, int64_t n, double eps, int min_pts, uint8_t *core,
                       int32_t *root, hipStream_t stream);
int dbscan_propagate_launch(const int64_t *off, const int32_t *nbr, const double *dist, int64_t n, double eps,
                            const uint8_t *core, int32_t *root, int rounds, int32_t *changed, hipStream_t stream);
size_t dbscan_ws_bytes(int64_t n);
int dbscan_labels_launch(const int64_t *off, const int32_t *nbr, const double *dist, int64_t n, double eps,
                         const uint8_t *core, const int32_t *root, int64_t *labels, int64_t *n_clusters, void *ws,
                         size_t ws_bytes, hipStream_t stream);

// rows_bf16.hip
int rows_to_bf16_launch(const float *in, int dim, int in_stride, const int32_t *idx, int64_t n_rows,
                        uint16_t *out, int out_stride, hipStream_t stream);
int segment_sum_bf16_launch(const uint16_t *rows, int dim, int row_stride, const int32_t *rowptr,
                            const int32_t *pos, int64_t n_seg, uint16_t *out, int out_stride,
                            const uint16_t *addend, int addend_stride, hipStream_t stream);
int permute_rows_bf16_launch(const uint16_t *in, int dim, int in_stride, const int32_t *idx, int64_t n_rows,
                             uint16_t *out, int out_stride, int scatter, hipStream_t stream);
int fold_finish_bf16_launch(uint16_t *out, int out_stride, int64_t n_nodes, const int32_t *rowptr, const uint16_t *carry,
                            int64_t n_units, const uint16_t *x, int x_stride, hipStream_t stream);

}  // namespace gnntrk
