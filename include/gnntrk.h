/*
 * gnntrk.h - C ABI of libgnntrk.so: the MI355X (gfx950) native hot path of
 * gnn_tracking's Interaction-Network edge classification, kNN graph construction
 * and object-condensation loss reductions.
 *
 * This is the drop-in boundary.  The reference is pure Python; the "FFI" its hot
 * path reaches is ATen / PyG / torch_cluster.  Every entry
 * point below replaces one group of those calls and cites the reference call site
 * (paths relative to /root/reference/src/gnn_tracking).  The Python side
 * (gnn_tracking_amd/_capi.py) binds these with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless noted;
 *    the library never allocates or frees caller memory: scratch is a caller
 *    workspace whose size comes from the matching *_workspace_bytes() query.
 *  - every call is asynchronous on the given HIP stream (`stream` is a
 *    hipStream_t passed as void*; NULL = default stream) and stateless/re-entrant.
 *  - return value: 0 ok, 1 bad argument, 2 HIP runtime error, 3 out of memory
 *    (message contains "out of memory" so utils/oom.py:12-18 keeps working),
 *    4 unsupported size.  gnntrk_last_error() returns the thread-local message.
 *  - floating point data is fp32 row-major; indices produced by the library are
 *    int32 (graphs up to 2^31-1 nodes/edges), indices consumed from the reference
 *    surface (`edge_index`, `particle_id`) are int64 exactly as PyG stores them.
 */
#ifndef GNNTRK_H
#define GNNTRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNTRK_VERSION 600 /* 0.6.0: + the gfold block of the bf16 backward arguments: target-side gradient fold inside the kernel, mlp_backward_bf16_can_fold, fold_finish_bf16; 0.5.0: + node_order, graph_index_carry.node_rank (node renumbering inside the index build); 0.4.0: + resfcnn_*, hinge_* (metric-learning stage), mlp_*_wide (fp32 in <= 128 / hidden <= 128 / out <= 48); 0.3.0: + graph_index_build_ex / _carry (own counting sort), bce_csr; 0.2.3: + radius_*_ws; 0.2.2: oc_*_spatial; 0.2.1: knn_search_ws / knn_workspace_bytes (0.2.0: edge_targets_csr, knn_search_batched, oc_backward workspace, oc_args.rep_keep_prob/rep_seed) */
#define GNNTRK_MAX_SEGS 10 /* concat segments of one fused MLP input           */
#define GNNTRK_MAX_IN 48   /* max concatenated input width of a fused MLP      */
#define GNNTRK_MAX_HIDDEN 64
#define GNNTRK_MAX_OUT 16

enum {
    GNNTRK_OK = 0,
    GNNTRK_EINVAL = 1,
    GNNTRK_EHIP = 2,
    GNNTRK_ENOMEM = 3,
    GNNTRK_EUNSUPPORTED = 4
};

int gnntrk_version(void);
const char *gnntrk_last_error(void);
/* number of compute units of the current device (grid sizing; 256 on MI355X) */
int gnntrk_device_cu_count(void);

/* ------------------------------------------------------------------ graph index
 * Replaces the bookkeeping PyG's MessagePassing.propagate does per call
 * (models/interaction_network.py:67: x_j = x[edge_index[0]], x_i = x[edge_index[1]],
 * scatter-add onto edge_index[1]) by a one-off index of the COO edge list:
 *
 *   perm[k]      original edge id of the k-th edge in target-sorted ("CSR") order;
 *                the sort is STABLE, so edges of one target keep their COO order
 *   tgt[k],src[k] endpoints of that edge (int32)
 *   rowptr_t[n]  first CSR position whose target is n          (N+1 entries)
 *   rowptr_s[n]  same for the source-sorted order              (N+1 entries)
 *   spos[m]      CSR position of the m-th edge in source-sorted order (stable)
 *
 * With it, aggregation and all gradient scatters become deterministic segment
 * sums (no atomics).  edge_index is int64 [2,E] row-major, unsorted.
 */
typedef struct gnntrk_graph_index {
    int64_t n_nodes;
    int64_t n_edges;
    int32_t *perm;     /* [E]   */
    int32_t *tgt;      /* [E]   */
    int32_t *src;      /* [E]   */
    int32_t *rowptr_t; /* [N+1] */
    int32_t *rowptr_s; /* [N+1] */
    int32_t *spos;     /* [E]   */
    int32_t *spos_inv; /* [E] or NULL: inverse of spos (CSR position -> position in the
                          source-sorted order): lets a backward kernel write per-edge source
                          gradients already source-sorted, so their fold streams */
} gnntrk_graph_index;

size_t gnntrk_graph_index_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int gnntrk_graph_index_build(const int64_t *edge_index, const gnntrk_graph_index *out,
                             void *workspace, size_t workspace_bytes, void *stream);
/* The same with `flags` (tests, measurements; both forms give identical arrays):
 *   bit 0: the library radix-sort form (two stable rocPRIM sorts + gather / boundary passes)
 *   bit 1: the own form also where the library form would be chosen for speed (dense graphs)
 *   default: the library's own two-level counting sort (buckets of 256 nodes ranked in LDS; made
 *   for collated batches - events with disjoint id ranges and contiguous edge ranges -, correct for
 *   any edge list); shapes outside it (more than 2^24 nodes, average degree in the thousands) take
 *   the library form by themselves.
 * The first int32 of the workspace holds the number of node ids outside [0, N) afterwards. */
int gnntrk_graph_index_build_ex(const int64_t *edge_index, const gnntrk_graph_index *out, void *workspace,
                                size_t workspace_bytes, int32_t flags, void *stream);

/* Per-edge inputs of the caller that ride along into CSR order inside the build (instead of a random
 * gather through `perm` afterwards; models/edge_classifier.py:97 reads `data.edge_attr`, training/
 * ec.py:43 `data.y` - both in edge_index order, both needed in CSR order by this library):
 *   edge_label  1-byte labels (the dataset's bool `y`)  ->  label_csr[k] = edge_label[perm[k]] != 0
 *   edge_rows   fp32 [E, 4] rows (`edge_attr`)          ->  rows_csr_bf16[k] = bf16(edge_rows[perm[k]]), RNE
 * Either input may be NULL.  The label travels in bit 31 of the edge id, the row as a second 8-byte
 * record: sequential reads, run-wise writes.  Identical to the gathers (which the library form of the
 * build runs instead). */
typedef struct gnntrk_graph_index_carry {
    const uint8_t *edge_label; /* [E] or NULL */
    uint8_t *label_csr;        /* [E] out */
    const float *edge_rows;    /* [E, 4] fp32, 16-byte aligned, or NULL */
    uint16_t *rows_csr_bf16;   /* [E, out_stride] bf16 out, 8-byte aligned */
    int32_t rows_stride;       /* floats per input row (multiple of 4) */
    int32_t out_stride;        /* bf16 per output row (multiple of 4) */
    const int32_t *node_rank;  /* [N] or NULL: renumbering old node id -> new node id (a permutation of [0, N),
                                  gnntrk_node_order.rank) applied to both endpoints of every edge as they are
                                  read: tgt / src / rowptr_* come out in the NEW numbering, perm / spos still
                                  refer to the caller's edge order.  The caller gathers its node rows through
                                  gnntrk_node_order.perm and hands node results back through rank. */
} gnntrk_graph_index_carry;
/* carry_rows: bit 0 = edge_rows are carried, bit 1 = node_rank is given (room for the translated int32 edge list) */
size_t gnntrk_graph_index_workspace_bytes_carry(int64_t n_nodes, int64_t n_edges, int32_t carry_rows);
int gnntrk_graph_index_build_carry(const int64_t *edge_index, const gnntrk_graph_index *out,
                                   const gnntrk_graph_index_carry *carry, void *workspace, size_t workspace_bytes,
                                   int32_t flags, void *stream);

/* A cached per-event index placed into the arrays of a collated batch.  The reference's datasets are static across
 * epochs (utils/loading.py:97-100) and its DataLoader collates disjoint graphs with shifted ids (utils/loading.py:
 * 233-239, PyG Batch): the batch's stable target sort, source sort and inverses are the events' own, shifted by the
 * event's node / edge offset - so an index built ONCE per event is copied, not sorted again:
 *   batch.perm[eo + k] = part.perm[k] + eo, .tgt / .src + no, .spos / .spos_inv + eo, .rowptr_*[no + n] = part + eo
 *   (n = 0 .. part.n_nodes), carried label_csr / 8-byte rows_csr rows copied (pairs of NULL: not carried),
 *   node_perm / node_rank + no (gnntrk_node_order of the event; NULL: not renumbered).
 * One call per event; the calls of a batch write disjoint ranges (entry n_nodes of an event's row pointers equals
 * entry 0 of the next event's).  Identical to gnntrk_graph_index_build_carry on the collated edge list. */
int gnntrk_graph_index_place(const gnntrk_graph_index *part, int64_t node_offset, int64_t edge_offset,
                             const gnntrk_graph_index *batch, const uint8_t *label_part, uint8_t *label_batch,
                             const uint16_t *rows_part, uint16_t *rows_batch, const int32_t *node_perm_part,
                             int32_t *node_perm_batch, const int32_t *node_rank_part, int32_t *node_rank_batch, void *stream);

/* Node renumbering for gather locality.  The reference keeps the hits of an event in file order
 * (graph_construction/graph_builder.py:396-455: node order = order of the hit table; utils/loading.py:17-113
 * hands the graphs on unchanged), which is unrelated to the geometry; the message passing then gathers
 * x[edge_index[0]] (models/interaction_network.py:67) from random rows.  gnntrk_node_order sorts the nodes
 * of every event by a caller-supplied key (one float per node, row stride key_stride floats; for tracking
 * graphs the azimuth column of data.x - edges join hits of neighbouring azimuth); ties keep the old order:
 *   perm[new] = old, rank[old] = new, events (batch[i] in [0, n_events), non-decreasing, or NULL = one event) keep
 *   their id ranges.  Locality is all the key has to buy (any order gives the same results up to summation order), so
 *   with the event count stated (1 <= n_events <= 64; batch == NULL counts as one event) the key is
 *   QUANTISED to 65 536 levels, q = trunc((key - lo) * (65535 / (hi - lo))) in fp32 with lo / hi the smallest / largest
 *   key of the node's own EVENT (so an event is ordered the same alone and inside a batch; NaN keys: the last level), and the nodes are put in (event, q) order by this library's own
 *   counting sort (the two-level sort of the graph index, records = nodes): nodes of one event that share a level
 *   keep their old order - deterministic, no library sort.  Otherwise (n_events <= 0: not stated - all 32 bits of
 *   batch[i] are sorted; more events; events beyond a million nodes) a stable radix sort of N pairs: with b = ceil(log2(n_events)) <= 8
 *   one 32-bit key {event : b bits, top 32 - b bits of the key's order-preserving integer image} - keys that agree in
 *   those bits keep their old order -, otherwise 32 + b bits of a 64-bit key (the full key).
 * workspace: gnntrk_node_order_workspace_bytes(n_nodes). */
size_t gnntrk_node_order_workspace_bytes(int64_t n_nodes);
int gnntrk_node_order(const float *key, int64_t key_stride, const int64_t *batch, int64_t n_events, int64_t n_nodes,
                      int32_t *perm, int32_t *rank, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------- fused gather-MLP
 * One kernel family replaces, for every MLP site of the path
 * (models/mlp.py:59-62 = addmm + clamp_min chain):
 *   - the gathers feeding it (index_select via PyG _lift, interaction_network.py:67;
 *     h_ec[edge_index[k]], edge_classifier.py:112-113),
 *   - the torch.cat that builds its input (interaction_network.py:86, :102;
 *     edge_classifier.py:110,114),
 *   - the ReLU applied to its inputs by the residual stack (resin.py:103-104),
 *   - its epilogue (residual combination resin.py:17-26; ReLU of the encoders
 *     edge_classifier.py:102-103; clamped sigmoid edge_classifier.py:115-116).
 *
 * Input row m of the MLP is the concatenation over segments j of
 *     act_j( seg[j].ptr[ (seg[j].idx ? seg[j].idx[m] : m) * seg[j].stride + 0..dim ) )
 * Weights are nn.Linear layout [out,in] row-major fp32; bias pointers may be NULL
 * (bias=False encoders, edge_classifier.py:62-67).  n_layers = 2 or 3
 * (Linear-ReLU-Linear[-ReLU-Linear]); hidden <= 64, in <= 48, out <= 16.
 */
typedef struct gnntrk_seg {
    const float *ptr;   /* [rows, stride]                                        */
    const int32_t *idx; /* [M] row gather index, or NULL for identity            */
    int32_t dim;        /* number of features taken from each row                */
    int32_t stride;     /* row stride in floats                                  */
    int32_t relu;       /* apply ReLU to the loaded values                       */
    int32_t rows;       /* rows of the tensor behind ptr, 0 = not stated.  Only read by
                           gnntrk_mlp_backward_bf16: with the sizes of every tensor stated
                           (and below 2 GB each) it addresses them through buffer descriptors
                           - hardware range checks instead of per-lane address arithmetic   */
} gnntrk_seg;

typedef struct gnntrk_mlp {
    int32_t n_layers; /* 2 or 3 */
    int32_t in_dim;   /* must equal the sum of segment dims */
    int32_t hidden;
    int32_t out_dim;
    const float *W[3];
    const float *b[3]; /* NULL = no bias */
} gnntrk_mlp;

enum {
    GNNTRK_EPI_NONE = 0,     /* y                                                 */
    GNNTRK_EPI_RELU = 1,     /* relu(y)                                           */
    GNNTRK_EPI_RESIDUAL = 2, /* ca * res[m] + cb * y        (resin.py:26)         */
    GNNTRK_EPI_SIGMOID = 3   /* ca + cb * sigmoid(y)        (edge_classifier:116) */
};

typedef struct gnntrk_mlp_fwd_args {
    gnntrk_mlp mlp;
    int32_t n_seg;
    int32_t epilogue;
    gnntrk_seg seg[GNNTRK_MAX_SEGS];
    int64_t n_rows; /* M */
    float ca, cb;
    const float *res; /* [M, res_stride] residue rows (EPI_RESIDUAL)              */
    int32_t res_stride;
    int32_t out_stride;
    float *out;             /* [*, out_stride]; row m is written at out_idx[m] or m */
    const int32_t *out_idx; /* optional row scatter (a permutation)               */
    int32_t debug_flags;    /* 0 in production; 1 skip stores, 2 skip input loads  */
    int32_t _pad;
} gnntrk_mlp_fwd_args;

int gnntrk_mlp_forward(const gnntrk_mlp_fwd_args *args, void *stream);

/* bf16-storage variant (BASELINE configs 3/4: "bf16 storage for x, e, e~, aggr and the
 * MFMA inputs, fp32 accumulate"; the reference reaches it through Lightning's
 * precision="bf16-mixed" autocast around the same modules).  Same argument block, read
 * with these changes:
 *   - seg[j].ptr, res and out address bf16 elements (uint16_t storage; the `float *`
 *     field types are reinterpreted), strides are in ELEMENTS, must be multiples of 4 and
 *     >= the feature count rounded up to 4, row starts 8-byte aligned.  Padding elements
 *     of input rows are ignored (forced to 0), padding elements of output rows up to the
 *     next multiple of 4 are written as 0;
 *   - weights/biases are fp32 in memory and rounded to bf16 when loaded; every layer
 *     accumulates in fp32; hidden activations and the output are rounded to bf16 (RNE);
 *   - GNNTRK_EPI_SIGMOID writes an fp32 output [*, out_stride floats] (the edge weights
 *     feed the fp32 loss); GNNTRK_EPI_RESIDUAL reads bf16 res rows;
 *   - limits: <= 16 four-feature input chunks; hidden + the bias row (present when a layer after the
 *     first has a bias; hidden 64 and - with at most eight input chunks - 128 do without it) <= 96, or
 *     <= 128 when the inputs fit eight chunks; above 64 one 16-row tile per iteration and one
 *     workgroup per CU; out <= 16; with hidden + bias row in 33 .. 48 (three hidden tiles) also up to
 *     32 input chunks and out <= 48 (epilogues NONE / RESIDUAL; the forward alone also RELU); n_rows < 2^31. */
int gnntrk_mlp_forward_bf16(const gnntrk_mlp_fwd_args *args, void *stream);

/* Backward of the same fused op with full recompute (nothing but the op inputs is
 * saved): replaces autograd's index_add_ / mm / threshold_backward chain
 * (training/base.py:114-116).
 *
 * Upstream gradient of output row m, feature f:
 *     g[m][f] = sum_t gout[t].ptr[(gout[t].idx ? gout[t].idx[m] : m)*gout[t].stride + f]
 * (two terms let the relational model add "direct" and "through aggr" gradients
 * without materialising their sum).  The epilogue is differentiated inside.
 *
 * Per-row input gradients are written row-aligned: for segment j, if
 * gseg[j].ptr != NULL, row m of the segment's gradient slice goes to
 *     gseg[j].ptr[m * gseg[j].stride + 0..dim)      (ReLU masks of the segment applied;
 * gseg[j].idx and gseg[j].accumulate are reserved and must be NULL / 0 in the fp32 entry
 * point; gnntrk_mlp_backward_bf16 accepts gseg[j].idx (int32[M], a permutation): row m is
 * then written at gseg[j].ptr[gseg[j].idx[m] * stride + ...]).
 * Gathered segments are reduced onto their source rows afterwards with
 * gnntrk_segment_sum (deterministic).  gres (EPI_RESIDUAL) is NOT produced here:
 * it is ca * g, an elementwise op of the caller.
 *
 * Weight/bias gradients are accumulated per wave in registers, written as partials
 * into the workspace and reduced in a fixed order into gW[i]/gb[i]
 * (+= when accumulate_params, else =): bit-reproducible run to run.
 */
typedef struct gnntrk_gterm {
    const float *ptr;
    const int32_t *idx;
    int32_t stride;
    int32_t rows; /* rows of the tensor behind ptr, 0 = not stated (see gnntrk_seg.rows) */
} gnntrk_gterm;

typedef struct gnntrk_gseg {
    float *ptr;
    const int32_t *idx;
    int32_t stride;
    int32_t accumulate;
} gnntrk_gseg;

/* Target-side fold of ONE gathered segment's gradient inside gnntrk_mlp_backward_bf16 (round 6).
 *
 * The rows of an interaction network's relational model / of the edge-weight head are in CSR order, i.e. sorted by
 * target node (models/interaction_network.py:67,75-89; models/edge_classifier.py:108-116), so the gradient of the
 * target-gathered node rows can leave the kernel already summed per node instead of as one row per edge that a
 * segment sum re-reads (16 B written + 16 B read per edge).  With `ids != NULL`:
 *   - gseg[seg].ptr addresses the FOLDED gradient: [n_nodes] padded bf16 rows of gseg[seg].stride = 8 elements
 *     (+ the carry rows, below), ZERO-FILLED by the caller (nodes without an edge are not written); gseg[seg].idx must be NULL;
 *   - ids = the sorted id stream the segment is gathered through (== seg[seg].idx: int32[n_rows], non-decreasing);
 *   - a unit of the kernel is 32 consecutive rows: every run of equal ids that STARTS in a unit is written to its
 *     node's row (its part inside the unit, NOT yet gated by the segment's ReLU); the part of a run that began in
 *     an earlier unit goes to the unit's carry row (8 bf16 = 16 bytes per unit: row n_nodes + unit of the same
 *     allocation - gseg[seg].ptr addresses n_nodes + (n_rows + 31) / 32 rows).  gnntrk_fold_finish_bf16 then
 *     adds, per node, the carries of the units its run continues into (unit order: deterministic, one thread per
 *     node) and applies the relu' gate of the segment - call it on the same stream after the backward;
 *   - arithmetic: the hidden-layer gradient (bf16, as the per-row form uses it) is summed per node in fp32 by an
 *     MFMA against a 0/1 segment matrix, rounded to bf16 once, and goes through W1^T like a row's: one rounding
 *     per node and unit instead of one per edge.
 * Only the launches gnntrk_mlp_backward_bf16_can_fold says 1 for (the buffer-addressed relational / head shapes
 * with two 16-row halves per unit, every gradient segment under the same ReLU flag); everything else rejects a
 * fold with GNNTRK_EUNSUPPORTED. */
typedef struct gnntrk_gfold {
    const int32_t *ids; /* int32[n_rows], non-decreasing; NULL: no fold */
    int64_t n_nodes;    /* node rows behind gseg[seg].ptr; (n_rows + 31) / 32 carry rows follow them */
    int32_t seg;        /* folded segment */
    int32_t _pad;
} gnntrk_gfold;

typedef struct gnntrk_mlp_bwd_args {
    gnntrk_mlp mlp;
    int32_t n_seg;
    int32_t epilogue;
    gnntrk_seg seg[GNNTRK_MAX_SEGS];
    int64_t n_rows;
    float ca, cb;
    int32_t n_gout; /* 1 or 2; gnntrk_mlp_backward_bf16: 3 where gnntrk_mlp_backward_bf16_max_terms says so */
    int32_t accumulate_params;
    gnntrk_gterm gout[3];
    gnntrk_gseg gseg[GNNTRK_MAX_SEGS];
    float *gW[3]; /* may be NULL: skip parameter gradients of that layer */
    float *gb[3];
    int32_t debug_flags; /* 0 in production. Ablation switches for tools/ablate_mlp.py:
                            1 skip input-gradient stores, 2 skip input loads (zeros),
                            8 skip the LDS transposes (weight grads become garbage)      */
    int32_t _pad;
    gnntrk_gfold fold;   /* ids == NULL: off (a zero-filled block is a valid "no fold") */
} gnntrk_mlp_bwd_args;

/* 1 if the launch described by `args` (filled as for the launch, fold included) takes the in-kernel fold. */
int gnntrk_mlp_backward_bf16_can_fold(const gnntrk_mlp_bwd_args *args);
/* Completes a fold: for every node n with rows [rowptr[n], rowptr[n + 1]) (the CSR row pointers of the sorted ids)
 * out[n] = gate(bf16(out[n] + sum of the carry rows out[n_nodes + u] of the units u after the first one its rows reach)), fp32 adds in
 * unit order, one rounding; gate: x != NULL (the folded segment's own input rows, read through a ReLU by the
 * launch) zeroes the features whose input is not positive - the relu' of the segment, applied once per node.
 * out: the folded rows of the launch (gseg[fold.seg].ptr; stride in elements, 8). */
int gnntrk_fold_finish_bf16(uint16_t *out, int32_t out_stride, int64_t n_nodes, const int32_t *rowptr,
                            int64_t n_units, const uint16_t *x, int32_t x_stride, void *stream);

/* How many upstream-gradient terms gnntrk_mlp_backward_bf16 takes for this launch (args filled as for
 * the launch, n_gout ignored): 3 for the shapes that run on buffer descriptors (an edge embedding read by
 * the next interaction network AND by the edge-weight head hands both gradients over without a sum pass in
 * between), else 2. */
int gnntrk_mlp_backward_bf16_max_terms(const gnntrk_mlp_bwd_args *args);
size_t gnntrk_mlp_backward_workspace_bytes(const gnntrk_mlp *mlp);
int gnntrk_mlp_backward(const gnntrk_mlp_bwd_args *args, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Backward of gnntrk_mlp_forward_bf16 (same argument block as gnntrk_mlp_backward):
 *   - seg[j].ptr, gout[t].ptr and gseg[j].ptr address padded bf16 rows (rules of
 *     gnntrk_mlp_forward_bf16); with GNNTRK_EPI_SIGMOID the single upstream term is fp32;
 *   - the forward is recomputed with the forward's rounding; the upstream gradient (after
 *     the epilogue derivative) and the gradient at every hidden layer are rounded to bf16
 *     before they feed the next MFMA; input gradients are written as bf16 (padding
 *     elements as 0); weight / bias gradients are fp32, accumulated in fp32 per wave and
 *     reduced in a fixed order (bit-reproducible). */
size_t gnntrk_mlp_backward_bf16_workspace_bytes(const gnntrk_mlp *mlp);
int gnntrk_mlp_backward_bf16(const gnntrk_mlp_bwd_args *args, void *workspace,
                             size_t workspace_bytes, void *stream);

/* Helpers of the bf16-storage path.  All bf16 row tensors follow the padding rules of
 * gnntrk_mlp_forward_bf16 (uint16_t storage, stride in elements, multiple of 4).
 *  rows_to_bf16:      out[m] = bf16(in[idx ? idx[m] : m]) (RNE), padding written as 0: how the
 *                     dataset's fp32 x / edge_attr (graph_builder.py:440-455) enter the stack,
 *                     edge_attr permuted into CSR order in the same pass.
 *  segment_sum_bf16:  gnntrk_segment_sum over bf16 rows, fp32 accumulation in CSR order, one
 *                     rounding at the end (aggregation and node-gradient folds).
 *  segment_sum_bf16_add: the same with one more bf16 term per segment, out[n] = bf16(addend[n] + sum):
 *                     the two folds of a node embedding that an interaction network gathers by
 *                     target AND by source (interaction_network.py:75-89) leave as one tensor,
 *                     with one rounding, instead of autograd adding two (addend and out must not overlap).
 *  permute_rows_bf16: gnntrk_permute_rows for padded bf16 rows. */
int gnntrk_rows_to_bf16(const float *in, int32_t dim, int32_t in_stride, const int32_t *idx,
                        int64_t n_rows, uint16_t *out, int32_t out_stride, void *stream);
int gnntrk_segment_sum_bf16(const uint16_t *rows, int32_t dim, int32_t row_stride,
                            const int32_t *rowptr, const int32_t *pos, int64_t n_segments,
                            uint16_t *out, int32_t out_stride, void *stream);
int gnntrk_segment_sum_bf16_add(const uint16_t *rows, int32_t dim, int32_t row_stride,
                                const int32_t *rowptr, const int32_t *pos, int64_t n_segments,
                                const uint16_t *addend, int32_t addend_stride, uint16_t *out,
                                int32_t out_stride, void *stream);
int gnntrk_permute_rows_bf16(const uint16_t *in, int32_t dim, int32_t in_stride, const int32_t *idx,
                             int64_t n_rows, uint16_t *out, int32_t out_stride, int32_t scatter,
                             void *stream);

/* Name (as rocprofv3 prints it) of the kernel instantiation gnntrk_mlp_forward /
 * gnntrk_mlp_backward dispatch to for this MLP and input-segment list; written
 * NUL-terminated into buf[len].  backward: 0 forward, 1 backward, +2 for the bf16 entry
 * points.  For matching host-side timings with profiles. */
int gnntrk_mlp_kernel_name(const gnntrk_mlp *mlp, int32_t n_seg, const gnntrk_seg *seg,
                           int32_t backward, char *buf, size_t len);
/* same for the bf16 entry points, whose instantiations also depend on the epilogue, the
 * segment layout (16-byte loads when every chunk pair allows them) and the gradient slices */
int gnntrk_mlp_forward_bf16_kernel_name(const gnntrk_mlp_fwd_args *args, char *buf, size_t len);
int gnntrk_mlp_backward_bf16_kernel_name(const gnntrk_mlp_bwd_args *args, char *buf, size_t len);

/* --------------------------------------------------------------- segment sums
 * out[n][0..dim) (=|+=) sum_{k in [rowptr[n], rowptr[n+1])} rows[(pos ? pos[k] : k)][0..dim)
 * Replaces ATen scatter_add_ (PyG SumAggregation, interaction_network.py:36,67) in
 * the forward, and index_add_ in the backward of the gathers.  Summation order is
 * the CSR order (= COO order per target): deterministic.
 */
int gnntrk_segment_sum(const float *rows, int32_t dim, int32_t row_stride,
                       const int32_t *rowptr, const int32_t *pos, int64_t n_segments,
                       float *out, int32_t out_stride, int32_t accumulate, void *stream);

/* out[m][0..dim) = in[idx[m]][0..dim)  (gather, scatter=0)  or
 * out[idx[m]][0..dim) = in[m][0..dim)  (scatter=1; idx must be a permutation).      */
int gnntrk_permute_rows(const float *in, int32_t dim, int32_t in_stride, const int32_t *idx,
                        int64_t n_rows, float *out, int32_t out_stride, int32_t scatter,
                        void *stream);

/* out = a*x + b*y elementwise (n floats); y may be NULL (then out = a*x).
 * relu_mask (optional, n floats): out is zeroed where relu_mask <= 0.               */
int gnntrk_axpby(float a, const float *x, float b, const float *y, const float *relu_mask,
                 float *out, int64_t n, void *stream);

/* ------------------------------------------------------------------ BCE loss
 * metrics/losses/ec.py:71-121: mean binary cross entropy of edge weights w in
 * (0,1) against y (float 0/1) with torch's log clamp at -100; optional
 * falsify_low_pt_edges: y' = y && pt[src_node[e]] > pt_thld (pt_thld <= 0: off).
 * loss_out: 1 float (device).  workspace >= gnntrk_bce_workspace_bytes().
 * backward: gw[e] = gscale[0] * d/dw mean-BCE  (gscale: 1 device float, the
 * upstream gradient of the scalar loss).
 */
size_t gnntrk_bce_workspace_bytes(int64_t n);
int gnntrk_bce_forward(const float *w, const float *y, const int64_t *src_node,
                       const float *pt, float pt_thld, int64_t n, float *loss_out,
                       void *workspace, size_t workspace_bytes, void *stream);
int gnntrk_bce_backward(const float *w, const float *y, const int64_t *src_node,
                        const float *pt, float pt_thld, int64_t n, const float *gscale,
                        float *gw, void *stream);

/* EdgeWeightBCELoss (metrics/losses/ec.py:95-121) on CSR-ordered edge weights against the 1-byte
 * CSR-ordered labels a graph-index build carried along (gnntrk_graph_index_carry.label_csr), forward
 * and the gradient for a unit upstream value in ONE pass over the edges:
 *   t = label_csr[k] != 0 [&& pt[src_csr[k]] > pt_thld]      (falsify_low_pt_edges, ec.py:71-92)
 *   loss = mean of the per-edge BCE (torch's log clamp at -100);
 *   gw_unit[k] = d loss / d w[k] (NULL: not wanted) - the caller scales it by the upstream gradient.
 * workspace: gnntrk_bce_workspace_bytes(n). */
int gnntrk_bce_csr(const float *w, const uint8_t *label_csr, const int32_t *src_csr, const float *pt, float pt_thld,
                   int64_t n, float *loss, float *gw_unit, void *workspace, size_t workspace_bytes, void *stream);

/* Edge labels in the CSR order of a graph index, pt-falsified (metrics/losses/ec.py:71-92):
 *   out[k] = y[perm[k]]                                           pt_thld <= 0
 *   out[k] = (y[perm[k]] != 0 && pt[src_csr[k]] > pt_thld) ? 1:0  otherwise
 * y: float labels, or (y_is_u8) the dataset's 1-byte bool labels: a quarter of the array, so
 * the random gather mostly hits the Infinity Cache.  perm / src_csr: gnntrk_graph_index.perm /
 * .src.  One gather per batch; the losses then run
 * with src_node = NULL, pt_thld = 0 on the CSR-ordered edge weights the classification head
 * writes (models/edge_classifier.py:108-116 without the scatter back to edge_index order). */
int gnntrk_edge_targets_csr(const void *y, int32_t y_is_u8, const int32_t *perm, const int32_t *src_csr,
                            const float *pt, float pt_thld, int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------ focal loss
 * metrics/losses/ec.py:13-68 (binary_focal_loss), :124-150 (EdgeWeightFocalLoss), :153-183
 * (HaughtyFocalLoss): mean over edges of
 *   -alpha*pw*(1-w)^gamma*t*log(w) - (1-alpha)*w^gamma*(1-t)*log(1-w)        (no log clamp)
 * haughty = 0: t = falsify_low_pt_edges(y), pw = pos_weight (scalar);
 * haughty = 1: t = y, pw = falsify_low_pt_edges(y) per edge.  Workspace as for BCE.      */
int gnntrk_focal_forward(const float *w, const float *y, const int64_t *src_node, const float *pt, float pt_thld,
                         float alpha, float gamma, float pos_weight, int32_t haughty, int64_t n, float *loss_out,
                         void *workspace, size_t workspace_bytes, void *stream);
int gnntrk_focal_backward(const float *w, const float *y, const int64_t *src_node, const float *pt, float pt_thld,
                          float alpha, float gamma, float pos_weight, int32_t haughty, int64_t n,
                          const float *gscale, float *gw, void *stream);

/* ------------------------------------------------------------- kNN graph build
 * models/graph_construction.py:222-237 knn_with_max_radius(x, k, max_radius) =
 * torch_cluster.knn_graph(x, k) (no batch, loop=False, flow source_to_target) followed by
 * the strict L2 filter ||x_j - x_i|| < max_radius.  Arithmetic contract (bit-exact against
 * oracle/knn_ref.c): d2 = fmaf chain over dimensions in order; the k smallest (d2, index)
 * pairs per query, ties -> lower index, self excluded by index; filter sqrtf(d2) < r.
 *
 *  gnntrk_knn_search  : nbr[q*k + i], i < cnt[q]: neighbours of q, ascending distance
 *                       (max_radius <= 0: no radius).  dim <= 32, k <= 448.
 *  gnntrk_knn_emit    : edge_index == NULL: offsets[0..n] = exclusive scan of cnt
 *                       (offsets[n] = number of edges M; read it back to size the output);
 *                       else writes int64 edge_index[2, M]: row 0 = neighbour (source j),
 *                       row 1 = query (target i), grouped by query ascending.
 */
int gnntrk_knn_search(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k,
                      float max_radius, int32_t *nbr, int32_t *cnt, void *stream);
/* The `batch` argument of torch_cluster's knn_graph / radius_graph (metrics/losses/
 * metric_learning.py:97, :232): rows [seg_ptr[s], seg_ptr[s+1]) are event s of a collated batch
 * (seg_ptr: n_seg + 1 ascending int64 offsets on the device, seg_ptr[0] = 0, seg_ptr[n_seg] = n);
 * neighbours are searched inside the query's own event only - all events in ONE launch. */
int gnntrk_knn_search_batched(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k,
                              float max_radius, const int64_t *seg_ptr, int32_t n_seg, int32_t *nbr,
                              int32_t *cnt, void *stream);
/* The same search with a caller-owned workspace (gnntrk_knn_workspace_bytes; 0 = this shape is
 * not covered, pass NULL): for dim <= 16 and at least 8192 rows the points are sorted
 * by (event, Morton code), cut into chunks of 64 with bounding boxes, and a query only streams
 * the chunks whose box can hold a neighbour closer than its current threshold.  The bound is
 * evaluated in the search's own arithmetic (monotone fp32 rounding), so nbr / cnt are
 * bit-identical to gnntrk_knn_search(_batched) - only faster (200 k clustered hits, dim 8: see
 * DESIGN.md 4.6).  seg_ptr may be NULL (one event).  flags: bit 0 = use the pruned search
 * below 8192 rows too, bit 1 = brute force regardless (both for tests / measurements). */
size_t gnntrk_knn_workspace_bytes(int64_t n, int32_t dim, int32_t k);
int gnntrk_knn_search_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, int32_t k, float max_radius,
                         const int64_t *seg_ptr, int32_t n_seg, int32_t *nbr, int32_t *cnt, void *workspace,
                         size_t workspace_bytes, int32_t flags, void *stream);
int gnntrk_knn_emit(const int32_t *nbr, const int32_t *cnt, int64_t n, int32_t k, int64_t *offsets,
                    int64_t *edge_index, int64_t n_edges, void *stream);
/* The k_take <= k_stride nearest neighbours out of a search done with k = k_stride: the same
 * edge list a search with k = k_take returns (neighbours are sorted by the key (d2, index) and
 * the radius filter keeps a prefix).  GraphConstructionKNNScanner scans k = 1..9 with one
 * search per k (graph_construction/k_scanner.py:203-285, :267-269); here ONE search at max(ks)
 * feeds every k. */
int gnntrk_knn_emit_prefix(const int32_t *nbr, const int32_t *cnt, int64_t n, int32_t k_stride,
                           int32_t k_take, int64_t *offsets, int64_t *edge_index, int64_t n_edges,
                           void *stream);

/* MLGraphConstruction.forward, models/graph_construction.py:365-367 and :386-393:
 *   y[e]        = (pid[e0] == pid[e1]) && pid[e0] > 0        (int64 compare, int64 0/1 out)
 *   feat[e]     = [x[e0] - x[e1], x[e0] + x[e1]]             ([M, 2*dim])
 * with e0 = edge_index[0][e], e1 = edge_index[1][e] (int64 [2, M]).                      */
int gnntrk_edge_labels(const int64_t *particle_id, const int64_t *edge_index, int64_t n_edges,
                       int64_t *y, void *stream);
int gnntrk_edge_features(const float *x, int32_t dim, int32_t x_stride, const int64_t *edge_index,
                         int64_t n_edges, float *out, void *stream);

/* ------------------------------------------------------ ModularGraphTCN glue
 * Threshold cut and orphan-node masking between the edge classifier and the track condenser
 * (models/track_condensation_networks.py:251-262).  Deterministic three-pass stream
 * compaction: outputs in ascending index order, exactly what boolean indexing /
 * `unique` return.  Counts are written to device memory (`n_out`); the caller reads them
 * back to size its tensors (the reference's `x[mask]` synchronises in the same place).
 *
 * threshold_compact  replaces `mask = W > ec_threshold; data.edge_subgraph(mask)`:
 *   mask[i] = w[i] > threshold (NaN -> 0), idx[0..n_out[0]) = ascending i with mask[i].
 *   Edge-level attributes are then gathered with idx.
 * connected_nodes    replaces `edge_index.flatten().unique()`, `index_to_mask` and the
 *   relabelling of `data.subgraph(connected)`:
 *   hit[v] = 1 iff node v is an endpoint of an edge; node_idx[0..n_out[0]) = ascending
 *   connected nodes; newid[v] = rank of v among them or -1; edge_index_out = newid[edge_index]
 *   ([2, n_edges] int64); n_out[1] != 0 if an id was outside [0, n_nodes).              */
size_t gnntrk_compact_workspace_bytes(int64_t n);
int gnntrk_threshold_compact(const float *w, int64_t n, float threshold, uint8_t *mask, int32_t *idx,
                             int64_t *n_out, void *workspace, size_t workspace_bytes, void *stream);
int gnntrk_connected_nodes(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, uint8_t *hit,
                           int32_t *node_idx, int32_t *newid, int64_t *n_out, int64_t *edge_index_out,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------ DBSCAN post-processing
 * postprocessing/fastrescanner.py:6-66 (`DBSCANFastRescan`): the radius-neighbourhood graph
 * at max_eps (sklearn `NearestNeighbors.radius_neighbors`), then for any eps <= max_eps and
 * min_pts the labels of sklearn's `dbscan_inner` on the edges with dist <= eps.
 *
 * Arithmetic = sklearn's kd-tree path (its choice for <= 15 features): fp64 coordinates,
 * d2 = sequential sum of squared differences, member iff d2 <= radius*radius, dist = sqrt(d2);
 * every point is its own neighbour (min_pts counts it).
 *
 *  radius_count : cnt[q] = neighbourhood size, offsets[0..n] = exclusive scan (offsets[n] = M:
 *                 read it back to size nbr / dist).  dim <= 32.
 *  radius_fill  : nbr / dist [M]: the neighbourhoods, CSR by query, ascending neighbour index.
 *  dbscan_init  : core[i] = |{e: dist[e] <= eps}| >= min_pts; root[i] = i (core) or -1.
 *  dbscan_propagate : `rounds` rounds of min-label propagation with pointer jumping over the
 *                 core-core edges; changed[0] != 0 iff the last round still moved a label -
 *                 call again until it is 0 (the fixpoint is unique: root = lowest core index
 *                 of the component).
 *  dbscan_labels: labels[i] (int64) = cluster number (clusters numbered by ascending lowest
 *                 core index, as dbscan_inner discovers them); border points take the lowest
 *                 cluster number among their core neighbours; noise = -1.  n_clusters[0].  */
int gnntrk_radius_count(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius, int32_t *cnt,
                        int64_t *offsets, void *stream);
int gnntrk_radius_fill(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius,
                       const int64_t *offsets, int32_t *nbr, double *dist, void *stream);
/* The same graph with caller-owned workspaces (dim <= 16, from 4096 points on; otherwise these fall
 * back to the two entries above): points sorted into chunks of 64 with bounding boxes; every query
 * walks only the candidate chunks whose box its own point can reach - per-query point-to-box bound
 * `sum_d max(lo_d - q_d, q_d - hi_d, 0)^2 <= r^2 (1 + 1e-5)` evaluated in fp32: the relative margin
 * of 1e-5 covers the fp32 rounding of the bound ((D + 2) ulp) against the graph's fp64 distances,
 * so no neighbour can be lost; the survivors are decided by the graph's own fp64 arithmetic -; the
 * lists are then put into ascending neighbour order.  cnt / offsets / nbr / dist are identical to radius_count / radius_fill.
 * ws_points (gnntrk_radius_points_workspace_bytes; 0 = not covered, pass NULL) is filled by the
 * count pass and read by the fill pass of the same points; ws_edges
 * (gnntrk_radius_edges_workspace_bytes(M)) stages the unordered lists.  flags: bit 0 = pruned form
 * below its size threshold too, bit 1 = brute force regardless. */
size_t gnntrk_radius_points_workspace_bytes(int64_t n, int32_t dim);
size_t gnntrk_radius_edges_workspace_bytes(int64_t m_edges);
int gnntrk_radius_count_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius, int32_t *cnt,
                           int64_t *offsets, void *ws_points, size_t ws_points_bytes, int32_t flags, void *stream);
int gnntrk_radius_fill_ws(const float *x, int64_t n, int32_t dim, int32_t x_stride, double radius,
                          const int64_t *offsets, int64_t m_edges, int32_t *nbr, double *dist, void *ws_points,
                          size_t ws_points_bytes, void *ws_edges, size_t ws_edges_bytes, int32_t flags, void *stream);
int gnntrk_dbscan_init(const int64_t *offsets, const double *dist, int64_t n, double eps, int32_t min_pts,
                       uint8_t *core, int32_t *root, void *stream);
int gnntrk_dbscan_propagate(const int64_t *offsets, const int32_t *nbr, const double *dist, int64_t n, double eps,
                            const uint8_t *core, int32_t *root, int32_t rounds, int32_t *changed, void *stream);
size_t gnntrk_dbscan_workspace_bytes(int64_t n);
int gnntrk_dbscan_labels(const int64_t *offsets, const int32_t *nbr, const double *dist, int64_t n, double eps,
                         const uint8_t *core, const int32_t *root, int64_t *labels, int64_t *n_clusters,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------ condensation losses
 * utils/graph_masks.py:19-28: mask = pt > thld && pid > 0 && reconstructable > 0 && |eta| < max_eta */
int gnntrk_good_node_mask(const float *pt, const int64_t *particle_id, const float *reconstructable,
                          const float *eta, int64_t n, float pt_thld, float max_eta, uint8_t *mask,
                          void *stream);

/* Condensation-point selection (metrics/losses/oc.py:16-43 and :279-292): for every particle
 * id that has at least one masked hit, the hit with the largest score (beta); mode 0 (RG):
 * among its masked hits, mode 1 (Tiger): among all its hits; ties -> lowest hit index.
 * alphas[0..K): CP hit per particle of interest, ascending particle id; gid[h] = k of hit h's
 * particle or -1; n_cp[0] = K (device).  Index work: one stable 64-bit radix sort + scans. */
size_t gnntrk_oc_select_workspace_bytes(int64_t n);
int gnntrk_oc_select_cps(const float *score, const int64_t *particle_id, const uint8_t *mask,
                         int64_t n, int32_t mode, int32_t *alphas, int32_t *gid, int32_t *n_cp,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Potential / background loss terms (metrics/losses/oc.py:46-161 RG with mode 0,
 * :251-347 Tiger with mode 1; q = atanh(beta)^2 + q_min):
 *   out[0] attractive = sum q_j q_k |x_j-x_k|^2 / (1e-9 + n_oi - K)
 *   out[1] repulsive  = sum_{pid_j != pid_k, |x_j-x_k| < radius} q_j q_k (radius - sqrt(eps_sqrt + d2))
 *                       / (1e-9 + (K-1) n)
 *   out[2] coward     = mean(1 - beta[alphas]);  out[3] noise = mean(beta[noise hits])
 *   out[4..8]         = norm_att, norm_rep, K, number of repulsive pairs, number of noise hits
 * backward: gx[n,dim], gbeta[n] = sum_t g[t] * d out[t] / d(x, beta); fwd = forward's out.   */
typedef struct gnntrk_oc_args {
    const float *x;    /* [n, stride] latent coordinates */
    const float *beta; /* [n] */
    const int64_t *particle_id;
    const uint8_t *mask;
    const int32_t *gid;
    const int32_t *alphas;
    const int32_t *n_cp;
    int64_t n;
    int32_t dim, stride;
    float q_min, radius, eps_sqrt;
    int32_t mode;
    /* condensation_loss_tiger's max_n_rep (oc.py:322-328): keep a repulsive pair (hit j, condensation
     * point k) with probability rep_keep_prob, decided by a hash of (rep_seed, j, k) - the same
     * pairs in the forward and both backward passes - and scale norm_rep by it.  >= 1: all pairs. */
    float rep_keep_prob;
    int32_t _pad;
    uint64_t rep_seed;
    /* CondensationLossRG's radius_graph(max_num_neighbors) cap, nearest first (oc.py:115-117): per hit
     * the index of its max_num_neighbors-th nearest hit inside the radius (last neighbour of a
     * gnntrk_knn_search with k = max_num_neighbors), -1 where the hit has fewer.  A condensation point
     * only repels a hit if it is among that hit's nearest max_num_neighbors.  NULL: no cap. */
    const int32_t *cap_nbr;
} gnntrk_oc_args;

size_t gnntrk_oc_forward_workspace_bytes(int64_t n);
int gnntrk_oc_forward(const gnntrk_oc_args *args, float *out /*[9]*/, void *workspace,
                      size_t workspace_bytes, void *stream);
/* backward: hit pass (thread = hit) + condensation-point pass (thread = CP, the hits cut into
 * slices whose partials - the workspace - are added in slice order: deterministic).  max_cps:
 * host-side upper bound of the condensation-point count (n always works). */
size_t gnntrk_oc_backward_workspace_bytes(int64_t n, int32_t dim);
int gnntrk_oc_backward(const gnntrk_oc_args *args, const float *g /*[4]*/, const float *fwd /*[9]*/,
                       float *gx, float *gbeta, int64_t max_cps, void *workspace,
                       size_t workspace_bytes, void *stream);

/* The same loss terms and gradients without the N x K walk (dim <= 16; workspace_bytes == 0: not
 * covered, use the entry points above).  A repulsive pair needs |x_j - x_k| < radius: the hits are
 * sorted into chunks of 64 with bounding boxes and a (chunk, condensation point) pair is only
 * looked at when the box reaches into the radius (conservative bound; the exact per-pair test is
 * unchanged); the attractive term is summed over (hit, its own condensation point) directly.  Same
 * pairs and per-pair arithmetic as gnntrk_oc_forward / gnntrk_oc_backward, fixed summation order.
 * `spatial` is ONE caller-owned buffer: the forward fills it (sorted hits, boxes, per-hit and
 * per-point records), the backward of the same inputs reads it and uses its scratch part. */
size_t gnntrk_oc_spatial_workspace_bytes(int64_t n, int32_t dim);
int gnntrk_oc_forward_spatial(const gnntrk_oc_args *args, float *out /*[9]*/, void *spatial, size_t spatial_bytes,
                              void *stream);
int gnntrk_oc_backward_spatial(const gnntrk_oc_args *args, const float *g /*[4]*/, const float *fwd /*[9]*/,
                               float *gx, float *gbeta, int64_t max_cps, void *spatial, size_t spatial_bytes,
                               void *stream);

/* ------------------------------------------------------------------ wide fp32 fused MLP
 * The operator of gnntrk_mlp_forward / gnntrk_mlp_backward (same argument structs, same semantics of
 * segments, epilogues NONE / RELU / RESIDUAL, upstream terms, gradient slices, accumulate_params) for the
 * shapes beyond the register-resident fp32 kernels: in_dim <= 128, hidden <= 128, out_dim <= 48 - e.g. the
 * reference-default GraphConstructionResIN(hidden_dim=40), models/graph_construction.py:136-219, whose
 * relational model is 120 -> 40 -> 40 -> 40, in the reference's own precision.  One forward and one
 * backward launch (+ a fragment-packing and a reduction launch); one layer's weights at a time in LDS,
 * activations in registers (csrc/mlp_wide.hip).  The forward leaves the hidden layers' pre-activations in
 * `acts` ([n_layers - 1][n_rows][gnntrk_mlp_wide_hidden_pad(hidden)] floats, 16-byte aligned; NULL when no
 * backward follows) - the backward walks the layers last to first with one layer's weight-gradient tiles
 * in registers at a time.  out_idx and gseg.idx are not supported (plain per-row slices).
 */
int32_t gnntrk_mlp_wide_hidden_pad(int32_t hidden);
size_t gnntrk_mlp_wide_forward_workspace_bytes(const gnntrk_mlp *mlp);
int gnntrk_mlp_forward_wide(const gnntrk_mlp_fwd_args *args, float *acts, void *workspace, size_t workspace_bytes,
                            void *stream);
size_t gnntrk_mlp_wide_backward_workspace_bytes(const gnntrk_mlp *mlp, int64_t n_rows);
/* `out`: the forward's output (read for EPI_RELU only) */
int gnntrk_mlp_backward_wide(const gnntrk_mlp_bwd_args *args, const float *acts, const float *out, int32_t out_stride,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ residual FCNN
 * ResFCNN (models/mlp.py:65-120) and the embedding networks built on it
 * (models/graph_construction.py:25-132) as ONE forward and ONE backward launch:
 *
 *   x  <- x / max(||x||_2, 1e-12)                      (normalize != 0; mlp.py:116)
 *   h  <- W_enc x + b_enc                              (mlp.py:117)
 *   h  <- sqrt(alpha) h + sqrt(1 - alpha) (W_l relu(h) + b_l),  l = 1 .. n_hidden   (mlp.py:118-119)
 *   y  <- W_dec relu(h) + b_dec                        (mlp.py:120)
 *   y  <- y * out_scale[0]   (out_scale != NULL: GraphConstructionFCNN._latent_normalization,
 *                             graph_construction.py:52)
 *   y  <- relu(y)            (out_relu != 0: the caller's relu(encoder(x)), graph_construction.py:123)
 *
 * fp32 throughout (v_mfma_f32_16x16x4_f32, bit-exact fmaf chains).  A wave keeps the activations
 * of its rows in registers through all layers; the weights of one layer at a time are staged in LDS
 * as MFMA A-operand fragments (packed once per call into the workspace).  Limits: in_dim <= 64,
 * hidden <= 128, out_dim <= 32, n_hidden <= GNNTRK_RESFCNN_MAX_HIDDEN.
 *
 * Backward: the forward optionally leaves the pre-activation residual stream of every layer in
 * `acts` ([n_hidden + 1][n_rows][gnntrk_resfcnn_hidden_pad(hidden)] floats); the backward walks the
 * layers from the decoder down, every wave over its own rows, with the weight gradients of ONE
 * layer at a time in registers (hidden 128: 256 accumulator registers), the gradient of the residual
 * stream in a workspace buffer between layers, per-block partial sums reduced in a fixed order
 * (deterministic, no atomics).  Weight / bias pointers of gnntrk_resfcnn_grads may be NULL (no
 * gradient wanted); accumulate != 0 adds into them.
 */
#define GNNTRK_RESFCNN_MAX_HIDDEN 16 /* hidden (residual) layers = depth - 1 */
#define GNNTRK_RESFCNN_MAX_IN 64
#define GNNTRK_RESFCNN_MAX_WIDTH 128
#define GNNTRK_RESFCNN_MAX_OUT 32

typedef struct gnntrk_resfcnn {
    const float *W_enc, *b_enc;                        /* [hidden, in_dim], [hidden] or NULL     */
    const float *W_hid[GNNTRK_RESFCNN_MAX_HIDDEN];     /* [hidden, hidden]                       */
    const float *b_hid[GNNTRK_RESFCNN_MAX_HIDDEN];     /* [hidden] or NULL                       */
    const float *W_dec, *b_dec;                        /* [out_dim, hidden], [out_dim] or NULL   */
    const float *out_scale;                            /* device scalar or NULL                  */
    int32_t in_dim, hidden, out_dim, n_hidden;
    float alpha;
    int32_t normalize, out_relu, _pad;
} gnntrk_resfcnn;

typedef struct gnntrk_resfcnn_grads {
    float *W_enc, *b_enc;
    float *W_hid[GNNTRK_RESFCNN_MAX_HIDDEN];
    float *b_hid[GNNTRK_RESFCNN_MAX_HIDDEN];
    float *W_dec, *b_dec;
    float *out_scale;
} gnntrk_resfcnn_grads;

int32_t gnntrk_resfcnn_hidden_pad(int32_t hidden); /* floats per row of `acts` (multiple of 16) */
size_t gnntrk_resfcnn_forward_workspace_bytes(const gnntrk_resfcnn *m);
int gnntrk_resfcnn_forward(const gnntrk_resfcnn *m, const float *x, int32_t x_stride, int64_t n_rows, float *out,
                           int32_t out_stride, float *acts /* or NULL */, void *workspace, size_t workspace_bytes,
                           void *stream);
size_t gnntrk_resfcnn_backward_workspace_bytes(const gnntrk_resfcnn *m, int64_t n_rows);
/* `out` is the forward's output (read for out_relu only), `gx` ([n_rows, gx_stride], NULL: not wanted)
 * the gradient w.r.t. the un-normalised input rows */
int gnntrk_resfcnn_backward(const gnntrk_resfcnn *m, const float *x, int32_t x_stride, int64_t n_rows,
                            const float *acts, const float *out, int32_t out_stride, const float *gout,
                            int32_t gout_stride, float *gx, int32_t gx_stride, const gnntrk_resfcnn_grads *grads,
                            int32_t accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ hinge embedding loss
 * The two edge-list reductions of GraphConstructionHingeEmbeddingLoss
 * (metrics/losses/metric_learning.py:14-55, `_hinge_loss_components`; edge selection :88-110):
 * per edge (a, b) = (edges[0][e], edges[1][e]) with d = ||x[a] - x[b]||_2
 *     attractive (repulsive = 0):  d^p                    (:30-32)
 *     repulsive  (repulsive = 1):  relu(r_emb - d^p)      (:50-53)
 * summed over the edges that pass the reference's selection, applied INSIDE the kernels instead of
 * compacting the edge list first (boolean indexing = a host round trip per mask):
 *     node_mask   != NULL: only edges with node_mask[a] != 0      (`mask[edges[0]]`, :104-106, :109)
 *     particle_id != NULL: only edges with pid[a] != pid[b]       (:107)
 * out[0] = sum / denom, out[1] = number of selected edges, out[2] = denom, with
 * denom = (norm ? norm[0] : out[1]) + 1e-9 (the reference's three normalisations: its own edge count, the
 * hits of interest, the attractive edge count - the latter two handed in as device scalars).
 * Deterministic two-stage sums (fp64 partials).
 *
 * Backward: node-centric, no per-edge intermediate and no atomics.  With the graph index of the SAME
 * edge list (gnntrk_graph_index_build: CSR by edges[1], source-sorted view of edges[0]) one thread owns a
 * node and walks both of its segments, recomputing every incident edge's term:
 *     gx[n] (+)= g[0] / denom * ( sum_{e: a = n} c_e (x[a] - x[b])  -  sum_{e: b = n} c_e (x[a] - x[b]) ),
 *     c_e = s p d^(p-2)  (s = 1 attractive; s = -1 where r_emb - d^p > 0, else 0; c_e = 0 at d = 0, as
 *     torch's norm backward)
 */
typedef struct gnntrk_hinge_args {
    const float *x;             /* [n_nodes, x_stride] embedding, dim <= 32                        */
    int32_t dim, x_stride;
    int64_t n_nodes;
    const uint8_t *node_mask;   /* [n_nodes] or NULL                                               */
    const int64_t *particle_id; /* [n_nodes] or NULL                                               */
    float r_emb, p;
    int32_t repulsive, _pad;
} gnntrk_hinge_args;
size_t gnntrk_hinge_workspace_bytes(int64_t n_edges);
int gnntrk_hinge_forward(const gnntrk_hinge_args *args, const int64_t *edges /* [2, n_edges], rows edge_stride apart */,
                         int64_t n_edges, int64_t edge_stride, const float *norm /* device scalar or NULL */,
                         float *out /* [3] */, void *workspace, size_t workspace_bytes, void *stream);
int gnntrk_hinge_backward(const gnntrk_hinge_args *args, const gnntrk_graph_index *index, const float *g,
                          const float *denom, float *gx, int32_t gx_stride, int32_t accumulate, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNTRK_H */
