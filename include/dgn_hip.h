/*
 * dgn_hip.h -- C ABI of libdgn_hip.so: the MI355X (gfx950) implementation of the DGN
 * directional-aggregation hot path.
 *
 * The reference (Saro00/DGN) is pure Python and has no FFI; what this library replaces is
 * the span between `g.apply_edges(...)` and the end of `reduce_func` inside every DGN layer
 * variant, i.e. (paths relative to the reference tree)
 *
 *   realworld_benchmark/nets/dgn_layer.py:183-186  (DGNLayerSimple:  apply_edges + update_all)
 *   realworld_benchmark/nets/dgn_layer.py:112-115  (DGNLayerComplex: apply_edges + update_all)
 *   realworld_benchmark/nets/dgn_layer.py:261-264  (DGNTower:        apply_edges + update_all)
 *   realworld_benchmark/nets/dgn_layer.py:161-173  (reduce_func: aggregator concat, scaler concat)
 *   realworld_benchmark/nets/aggregators.py:8-71   (the aggregators)
 *   realworld_benchmark/nets/scalers.py:7-18       (the degree scalers)
 *
 * plus the autograd backward of all of the above.  The reference-side binding a maintainer
 * would add (a ctypes stub called from a torch.autograd.Function) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C; every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *   - sizes/strides are int64_t, strides ("ld") are in ELEMENTS (floats), not bytes;
 *   - every call enqueues on the caller's hipStream_t (passed as void*), never synchronises,
 *     keeps no global mutable state, and is re-entrant from several host threads;
 *   - return value: 0 = ok, < 0 = error (DGN_ERR_*); dgn_last_error() gives the thread-local text;
 *   - the graph is CSR by DESTINATION: slot range [indptr[i], indptr[i+1]) holds the in-edges of
 *     node i in ascending original edge id (the DGL mailbox order the oracle encodes).
 */
#ifndef DGN_HIP_H
#define DGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGN_ABI_VERSION 28

#define DGN_MAX_AGG 16     /* aggregators per launch (the host splits longer lists)            */
#define DGN_MAX_CH 4       /* edge-weight channels per launch                                   */
#define DGN_MAX_SCALERS 4  /* applied scalers                                                   */

enum {
    DGN_OK = 0,
    DGN_ERR_INVALID = -1,   /* bad argument / unsupported shape                                 */
    DGN_ERR_WORKSPACE = -2, /* workspace too small                                              */
    DGN_ERR_HIP = -3        /* a HIP runtime call failed                                        */
};

/* aggregator op codes; `ch` = edge-weight channel the op reads (ignored by the first six)      */
enum {
    DGN_AGG_MEAN = 0,        /* aggregators.py:8   (1/d) sum_j m_j                              */
    DGN_AGG_SUM = 1,         /* aggregators.py:31  sum_j m_j                                    */
    DGN_AGG_MAX = 2,         /* aggregators.py:12                                               */
    DGN_AGG_MIN = 3,         /* aggregators.py:16                                               */
    DGN_AGG_STD = 4,         /* aggregators.py:20  sqrt(var + eps)                              */
    DGN_AGG_VAR = 5,         /* aggregators.py:24  relu(mean(m^2) - mean(m)^2)                  */
    DGN_AGG_DIR_AV = 6,      /* aggregators.py:35  sum_j |w_j| m_j            (ABSNORM channel) */
    DGN_AGG_DIR_WSUM = 7,    /* aggregators.py:42  sum_j w_j m_j              (SOFTMAX channel) */
    DGN_AGG_DIR_DX = 8,      /* aggregators.py:48/:62  |sum_j w_j m_j - (sum_j w_j) x_i|        */
    DGN_AGG_DIR_DX_NO_ABS = 9, /* aggregators.py:55  sum_j w_j m_j - (sum_j w_j) x_i            */
    DGN_AGG_X_IN = 10        /* not an aggregator: copies x_in (h_in) into the output row, so that
                                posttrans([h || agg]) of dgn_layer.py:116-119 is ONE GEMM on the sweep's output */
};

/* edge-weight channel kinds: how delta_j = eig[src_j,k] - eig[dst,k] becomes a per-edge weight */
enum {
    DGN_W_ABSNORM = 0,  /* delta_j / (sum_j |delta_j| + eps)                 aggregators.py:49  */
    DGN_W_BALANCED = 1, /* (relu(d)/(sum relu(d)+eps) + relu(-d)/(sum relu(-d)+eps)) / 2   :63-69 */
    DGN_W_SOFTMAX = 2   /* softmax_j(alpha * |delta_j|)                      aggregators.py:43  */
};

/* degree scalers, scalers.py:7-18; D = in-degree, avg = avg_d['log']                           */
enum {
    DGN_SCALE_IDENTITY = 0,
    DGN_SCALE_AMPLIFICATION = 1, /* x * (log(D+1) / avg)                                        */
    DGN_SCALE_ATTENUATION = 2    /* x * (avg / log(D+1))                                        */
};

typedef struct DgnGraph {
    int64_t n_nodes;
    int64_t n_edges;
    const int32_t* indptr;  /* [n_nodes+1]                                                       */
    const int32_t* src;     /* [n_edges] source node of each CSR slot                            */
    /* Rows with more than hub_threshold in-edges are "hub rows": the row kernels skip them and
     * they are processed in slices of hub_chunk edges (partials in the workspace, then a combine).
     * n_hub == 0 disables the mechanism.                                                        */
    int64_t n_hub;
    const int32_t* hub_rows;      /* [n_hub] node ids, ascending                                 */
    const int32_t* hub_chunk_ptr; /* [n_hub+1] first chunk of every hub row                      */
    int64_t n_chunks;
    const int32_t* chunk_hub;     /* [n_chunks] index into hub_rows                              */
    int32_t hub_threshold;
    int32_t hub_chunk;
    /* Optional transposed view (NULL = absent), used by dgn_agg_backward to avoid atomics: csc_pos[j] is the
     * rank of CSR slot j when the slots are ordered by (source node, slot) and csc_ptr[u] the first such rank
     * of source u.  With it (and a large enough workspace) the backward writes every per-edge gradient row
     * to its csc position and a second kernel sums each source's contiguous rows: deterministic, no atomics. */
    const int32_t* csc_ptr;  /* [n_src+1]   */
    const int32_t* csc_pos;  /* [n_edges]   */
    /* Hint: largest in-degree, 0 = unknown.  Lets launches that only concern long rows be skipped.       */
    int32_t max_in_degree;
    /* Number of SOURCE nodes when it differs from n_nodes (0 = same): a bipartite CSR whose rows are e.g. the
     * graphs of a batch and whose sources are its nodes -- the readouts (dgl.mean_nodes / sum_nodes / max_nodes,
     * nets/.../dgn_net.py:71-86) are this sweep over such a CSR.  x_src / g_src have n_src rows, csc_ptr
     * n_src + 1 entries; x_dst, x_in, log_deg, out are per destination row.                               */
    int64_t n_src;
    /* Destination-range shard of a larger graph (one giant graph split across GPUs, dgn_amd/dist.py): row i of
     * this CSR is node row_base + i of the source node set.  Only dgn_edge_weights uses it (the destination side of
     * eig is read at row row_base + i); the per-row arrays of the sweep (x_dst, x_in, log_deg, out) are the shard's. */
    int64_t row_base;
    /* Optional (NULL = absent): block description of a batch of small graphs, built by dgn_graph_build_cuts.  blk_cut[i], i in
     * [0, n_nodes], is the last CLOSED cut <= i -- a cut c is closed when no edge joins a node below c to a node at or above it (the
     * graph boundaries of a dgl.batch, data/molecules.py:229); blk_gap is the largest distance between consecutive closed cuts (the
     * largest graph).  With it dgn_agg_backward (define mode, hot aggregator lists, one feature tile, no hub rows, no edge term, at most
     * 3 edges per node) lets ONE WAVE own a run of whole graphs and accumulate d x_src in its LDS rows: one kernel, no [E, F] staging
     * buffer, the adds in ascending (source, slot) order as in the staged path (run-to-run reproducible).                       */
    const int32_t* blk_cut;  /* [n_nodes+1] */
    int32_t blk_gap;
    /* Optional (NULL = absent): batches of graphs too large for that (k-NN superpixel graphs, data/superpixels.py:139-145; SBM graphs).
     * gblk_desc [n_gblk][4] int32, 16-byte aligned: first row, end row, first CSR slot, end CSR slot of blocks of WHOLE graphs that
     * partition [0, n_nodes) in order, gblk_rows = the largest block; csc_order[k] = the CSR slot with (source, slot) rank k (the
     * inverse of csc_pos), dst_csr[j] = the row of CSR slot j.  With them (and csc_ptr) dgn_agg_backward (define mode, baked-in lists
     * without max / min / std / var, no edge term, the aux table where the list has a dx aggregator) runs ONE kernel, a workgroup per
     * block: the destination rows' coefficient vectors in LDS, every source row gathering its out-edges -- no [E, F] staging buffer.  */
    const int32_t* gblk_desc; int64_t n_gblk; int32_t gblk_rows;
    const int32_t* csc_order;  /* [n_edges] */
    const int32_t* dst_csr;    /* [n_edges] */
} DgnGraph;

typedef struct DgnChannel {
    int32_t kind;     /* DGN_W_*                                                                 */
    int32_t eig_col;  /* column k of eig                                                         */
    float alpha;      /* SOFTMAX only                                                            */
    float eps;        /* 1e-8 on the DGL path (aggregators.py:5)                                 */
} DgnChannel;

typedef struct DgnAggSpec {
    int32_t n_agg;
    int32_t agg_op[DGN_MAX_AGG];
    int32_t agg_ch[DGN_MAX_AGG];
    int32_t n_ch;                      /* channels referenced by agg_ch (0..DGN_MAX_CH)          */
    int32_t n_scalers;                 /* scalers APPLIED; a lone scaler of any name is the
                                          identity (dgn_layer.py:170)                            */
    int32_t scaler[DGN_MAX_SCALERS];
    float avg_log;                     /* avg_d['log']                                           */
    float eps;                         /* EPS of aggregators.py:5, used by STD                   */
    int32_t n_towers;                  /* T >= 1: output columns are laid out [T][S][A][F/T];
                                          T == 1 is the reference order [S][A][F]                */
    /* A launch may compute a slice of a longer aggregator list: the columns it writes are those of
     * aggregators [agg_offset, agg_offset + n_agg) out of agg_total (0 = n_agg, offset 0).       */
    int32_t agg_total;
    int32_t agg_offset;
    /* Distance (in elements) between the column blocks of consecutive towers; 0 = n_scalers*agg_total*(F/T),
     * i.e. the towers of a node are contiguous inside its row.  With tower_stride = n_nodes*ld_out and
     * ld_out = n_scalers*agg_total*(F/T) the output is tower-major [T][N][S*A*F/T]: the batched per-tower GEMMs
     * that follow then read (and their backward writes) contiguous matrices.  Applies to out and g_out.   */
    int64_t tower_stride;
} DgnAggSpec;

/* The message of CSR slot j into node i is  m_j = x_src[src_j] + x_dst[i] + m_edge[j];
 * any of the three terms may be NULL (at least one must be given).
 *   simple layer:            x_src = h                                   (dgn_layer.py:154-155)
 *   complex/towers, 1-layer pretrans:  x_src = h W_s^T, x_dst = h W_d^T + b (+ m_edge = ef W_e^T)
 *   any other pretrans:      m_edge = materialised messages in CSR slot order
 * x_in is h_in of reduce_func (dgn_layer.py:162), needed by the dx aggregators.                 */
typedef struct DgnMsg {
    int64_t F;
    const float* x_src;  int64_t ld_src;   /* [n_nodes, ld_src]                                  */
    const float* x_dst;  int64_t ld_dst;   /* [n_nodes, ld_dst]                                  */
    const float* m_edge; int64_t ld_edge;  /* [n_edges, ld_edge], CSR slot order                 */
    const float* x_in;   int64_t ld_in;    /* [n_nodes, ld_in]                                   */
    /* Edge-TYPE table (optional; needs x_src): with edge_type != NULL, m_edge is a table [n_edge_types, ld_edge] and slot j
     * adds row edge_type[j] (values in 0 .. n_edge_types-1, CSR slot order) -- the edge-feature term of the pretrans Linear when
     * the edge features are an embedding lookup (nets/molecules_graph_regression/dgn_net.py:53,75: e = embedding_e(bond type);
     * dgn_layer.py:159-160: pretrans(cat[h_src, h_dst, e])): the caller passes table = embedding W_e^T instead of materialising
     * [n_edges, F].  n_edge_types * F <= DGN_MAX_EDGE_TABLE floats.  The backward then needs the two-phase scatter (g->csc_*
     * and a workspace that includes dgn_agg_edge_table_workspace_bytes()); DgnMsgGrad.g_edge is the TABLE's gradient.       */
    const int32_t* edge_type; int32_t n_edge_types;
    /* f_valid (0: every row holds F columns): the rows of x_src and x_in hold only f_valid = F - 1 columns (F even; dense rows of an ODD
     * width at their own strides, 4-byte aligned) and the sweep reads the last column pair as (x[F - 2], 0) -- what a zero-padded copy of
     * the rows would give, without the copy (the simple layer at hidden 75 / 65: configs ZINC simple, CIFAR10).  Only with x_src (and
     * x_in) alone -- no x_dst, no m_edge, one tower -- and only for the aggregator lists dgn_agg_f_valid_supported() accepts.          */
    int32_t f_valid;
} DgnMsg;
int dgn_agg_f_valid_supported(const DgnAggSpec* spec);      /* 1: this list has kernels for rows of an odd width (DgnMsg.f_valid) */
#define DGN_MAX_EDGE_TABLE 8192

/* Gradient sinks of dgn_agg_backward; NULL = not wanted.  g_edge is always overwritten.  g_src / g_dst / g_in:
 *   accumulate != 0: the call ADDS into caller-initialised buffers (a plan split into several launches, or a
 *                    caller summing several graphs' gradients);
 *   accumulate == 0: the call DEFINES them, the buffers may arrive uninitialised (with the two-phase scatter
 *                    every row is written exactly once: no zero-fill and no read-modify-write traffic; the
 *                    atomic scatter zero-fills them itself).
 * g_in may alias g_src (the simple layer's x_in IS x_src: both gradients land in one buffer).          */
typedef struct DgnMsgGrad {
    float* g_src;  int64_t ld_src;
    float* g_dst;  int64_t ld_dst;
    float* g_edge; int64_t ld_edge;
    float* g_in;   int64_t ld_in;
    int32_t accumulate;
} DgnMsgGrad;

int dgn_abi_version(void);
size_t dgn_sizeof(const char* struct_name);      /* sizeof of a struct of this header as the library was compiled ("DgnGraph", ...); 0: unknown */
/* Process-wide library options (experiments and tests; every default is what the benchmarks run).  Each option takes its initial value
 * from an environment variable ONCE, when the library first looks; afterwards only dgn_set_option changes it -- no entry point reads the
 * environment on a launch path.  Names: "blk_lds_kb" (DGN_BLK_LDS_KB, 13), "blk_min_nodes" (DGN_BLK_MIN_NODES, 131072),
 * "bwd_rows_per_wave" (DGN_BWD_ROWS_PER_WAVE, 4), "tile_gemm" / "tile_wgrad" (DGN_TILE_GEMM / DGN_TILE_WGRAD, -1 = by shape),
 * "no_zmask" (DGN_NO_ZMASK set, 0), "linear_small_min_waves" (8), "graph_bwd_tiles" (DGN_GRAPH_BWD_TILES, 0 = by batch size: feature tiles of
 * the graph backward of the sweep), "odd_direct" (DGN_ODD_DIRECT, 1: DgnMsg.f_valid in the simple whole-layer call instead of a padded copy).  dgn_set_option: DGN_ERR_INVALID for an unknown name;
 * dgn_get_option: INT64_MIN for an unknown name.                                                                            */
int dgn_set_option(const char* name, int64_t value);
int64_t dgn_get_option(const char* name);
const char* dgn_last_error(void);

/* Per-edge directional weights, once per (graph, eig): w[c*ld_w + j] for channel c, CSR slot j.
 * Either eig [n_nodes, ld_eig] (gathered through src / the row id), or -- for the reference's
 * mailbox-level function API (aggregators.py:74-93) -- eig_s_edge / eig_d_edge [n_edges, ld_eig]
 * given per slot.  Replaces the eig_s/eig_d materialisation of dgn_layer.py:155 and the
 * per-aggregator weight chains of aggregators.py:35-71.
 * ws: workspace of at least dgn_edge_weights_workspace_bytes() bytes (hub rows only).          */
size_t dgn_edge_weights_workspace_bytes(const DgnGraph* g, int32_t n_ch);
int dgn_edge_weights(const DgnGraph* g, const float* eig, const float* eig_s_edge, const float* eig_d_edge,
                     int64_t ld_eig, int32_t n_ch, const DgnChannel* ch, float* w, int64_t ld_w,
                     void* ws, size_t ws_bytes, void* stream);

/* Fused gather -> directional weighting -> multi-aggregator reduce -> degree scalers, one CSR
 * sweep.  out [n_nodes, ld_out], columns [T][S][A][F/T]; zero-in-degree rows are zeros.
 * log_deg [n_nodes] = (float)log((double)(in_degree + 1))  (scalers.py:13 evaluates np.log in
 * float64 and casts; the division by avg is done in fp32 like the reference).                  */
size_t dgn_agg_workspace_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F);
int dgn_agg_forward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                    const float* log_deg, float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream);

/* The forward / backward pair with an AUX table between them (training): one byte per (row, feature) in which the forward records
 * what the backward otherwise recomputes from the messages -- the slots of the row's first maximum / minimum and the sign of each
 * dx residual -- so that the backward gathers no source row, reads no x_dst / x_in row and still produces the same bits.
 * dgn_agg_aux_bytes() > 0 says the launch has such a table (molecule-like batches: at most three in-edges per row on average,
 * a baked-in list with max / min / dx but without std / var, at most two weight channels, even F); `aux` NULL = the plain calls.
 * Rows the 4-rows-per-wave kernels hand to the per-row routine (more than four in-edges, ...) are recomputed as before.          */
size_t dgn_agg_aux_bytes(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg);
int dgn_agg_forward_aux(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                        const float* log_deg, float* out, int64_t ld_out, unsigned char* aux, void* ws, size_t ws_bytes, void* stream);
int dgn_agg_backward_aux(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                         const float* log_deg, const float* g_out, int64_t ld_gout, const unsigned char* aux, const DgnMsgGrad* grads,
                         void* ws, size_t ws_bytes, void* stream);

/* Backward of dgn_agg_forward for upstream gradient g_out [n_nodes, ld_gout].
 * Workspace: dgn_agg_backward_workspace_bytes(); with `deterministic` != 0 (needs g->csc_*) it includes the
 * [n_edges, F] staging buffer of the two-phase scatter; a smaller workspace silently selects the atomic path. */
size_t dgn_agg_backward_workspace_bytes(const DgnGraph* g, const DgnAggSpec* spec, int64_t F, int32_t deterministic);
/* extra workspace bytes of dgn_agg_backward when DgnMsg.edge_type is set (appended to the deterministic workspace) */
size_t dgn_agg_edge_table_workspace_bytes(int64_t F, int32_t n_edge_types);
int dgn_agg_backward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                     const float* log_deg, const float* g_out, int64_t ld_gout, const DgnMsgGrad* grads,
                     void* ws, size_t ws_bytes, void* stream);

/* Degree scalers folded behind the post-aggregation Linear (they are per-row factors):
 *     y[n, t*f_out + o] = row_scale[n] * (bias[t*f_out + o] + sum_s scale[n, s] * z[t][n][s*f_out + o])
 * z is tower-major [T][N][S*f_out] (the batched GEMM output), y node-major [N, ld_y].  Replaces the scaler concat
 * of reduce_func (dgn_layer.py:170-171) + the Linear's bias + the graph-norm multiply `h * snorm_n`
 * (dgn_layer.py:121-122, :192-193, :270-271).  scale == NULL means S == 1 with factor 1; bias / row_scale may be NULL.
 * The backward writes g_z (same layout as z) from g_y and, if g_bias != NULL, ADDS the bias gradient
 * sum_n row_scale[n] * g_y[n, :] to g_bias [T*f_out] (caller zero-initialises).  The sum is formed from
 * per-workgroup partials in `ws` (dgn_scale_combine_backward_workspace_bytes(); needed only with g_bias) in a
 * fixed order -- no atomics, bitwise reproducible.  n_towers * f_out <= 4096.                               */
int dgn_scale_combine_forward(int64_t n_nodes, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* z,
                              const float* scale, const float* bias, const float* row_scale, float* y, int64_t ld_y,
                              void* stream);
size_t dgn_scale_combine_backward_workspace_bytes(int64_t n_nodes, int32_t n_towers, int32_t f_out);
/* Optional fusion with the BatchNorm tail that follows the combine (dgn_bn_tail_*): when `bn` is given, g_y is not
 * read; the upstream gradient of the combine is formed on the fly from the tail's inputs,
 *     g_y = gamma * invstd * (g' - sums[c]/N - xhat * sums[F+c]/N),   g' = g_out masked by the ReLU,  xhat = (y - mean) * invstd
 * with sums / g_gamma / g_beta from dgn_bn_tail_backward(..., g_x = NULL, sums): the [N, F] gradient between the two
 * steps is never written or re-read.                                                                        */
typedef struct DgnBnGrad {
    const float* g_out;    /* [N, ld] gradient of the tail's output                                          */
    const float* y;        /* [N, ld] the combine's output = the tail's input                                */
    int64_t ld;
    const float* gamma;    /* [F] or NULL                                                                    */
    const float* beta;     /* [F] or NULL                                                                    */
    const float* mean;     /* [F] save_mean of the forward                                                   */
    const float* invstd;   /* [F] save_invstd                                                                */
    const float* sums;     /* [2F] from dgn_bn_tail_backward                                                 */
    int32_t relu;
    const int64_t* n_valid; /* DEVICE scalar or NULL: only rows < *n_valid are the batch; the rest is padding of a batch held at a
                               fixed capacity (shape-bucketed HIP-graph replay): excluded from the statistics, zero gradient  */
} DgnBnGrad;
int dgn_scale_combine_backward(int64_t n_nodes, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* g_y,
                               int64_t ld_gy, const float* scale, const float* row_scale, float* g_z, float* g_bias,
                               void* ws, size_t ws_bytes, const DgnBnGrad* bn, void* stream);

/* Layer tail: BatchNorm1d over the node dimension, optionally followed by ReLU and the residual add
 *     y = [relu]( (x - mean) * invstd * gamma + beta ) [+ residual]          (dgn_layer.py:123-128, :194-199, :272-273)
 * training != 0: batch statistics (biased variance), running_mean / running_var updated in place with `momentum`
 * (running_var with the unbiased variance, like torch.nn.BatchNorm1d); save_mean / save_invstd [F] are written for
 * the backward.  training == 0: running statistics are used.  n_valid (DEVICE int64 scalar or NULL): rows >= *n_valid are
 * padding -- they take no part in the statistics (count = *n_valid) and get a zero input gradient in the backward -- so that a
 * batch can be held at a fixed row capacity (shape-bucketed HIP-graph replay).  1 <= F <= 1024.  All [N, F] tensors share the row
 * stride ld.  ws: dgn_bn_tail_workspace_bytes() of scratch (8-byte aligned; training / backward only) holding the
 * per-workgroup fp64 column partials, which are added in a fixed order (no atomics, bitwise reproducible).   */
size_t dgn_bn_tail_workspace_bytes(int64_t n_rows, int32_t F);
int dgn_bn_tail_forward(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                        int32_t relu, const float* residual, float* y, float* save_mean, float* save_invstd, void* ws,
                        size_t ws_bytes, const int64_t* n_valid, void* stream);
/* Backward of the training-mode tail: g_x [N, F] (written), g_gamma / g_beta [F] (written; may be NULL).
 * The gradient of `residual` is g_y itself (left to the caller).  g_x == NULL skips the apply pass: only the column
 * sums are produced (`sums` [2F], may be NULL when g_x is given) for dgn_scale_combine_backward's fused form.  */
int dgn_bn_tail_backward(int64_t n_rows, int32_t F, const float* g_y, const float* x, int64_t ld, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd, int32_t relu, float* g_x,
                         float* g_gamma, float* g_beta, float* sums, void* ws, size_t ws_bytes, const int64_t* n_valid, void* stream);

/* Tail of an FCLayer (Linear -> activation, nets/layers.py:101-112) on the bias-free GEMM output x [N, F]:
 *     y = act(x + bias) [+ residual]        act: 0 none, 1 ReLU, 2 LeakyReLU(slope)
 * -- the towers' mixing network (LeakyReLU) with the layer's residual add (dgn_layer.py:319-324) in one pass.  Backward:
 * g_x = g_y * act'(x + bias) (written) and g_bias = sum_n g_x[n, :] (written; NULL = not wanted; fp64 per-workgroup
 * partials in `ws` of dgn_bn_tail_workspace_bytes(n_rows, F) bytes, fixed-order sum); the residual's gradient is g_y.
 * All [N, F] tensors share the row stride ld; F <= 1024 for the backward.                                           */
int dgn_bias_act_forward(int64_t n_rows, int32_t F, const float* x, int64_t ld, const float* bias, int32_t act, float slope,
                         const float* residual, float* y, void* stream);
int dgn_bias_act_backward(int64_t n_rows, int32_t F, const float* g_y, const float* x, int64_t ld, const float* bias,
                          int32_t act, float slope, float* g_x, float* g_bias, void* ws, size_t ws_bytes, void* stream);

/* ---- dropout (dgn_bn_tail.hip): F.dropout(h, p, training) at the end of DGNLayerSimple / DGNLayerComplex.forward and inside DGNTower
 * (nets/dgn_layer.py:130, :201, :275; configs HIV / PCBA / CIFAR10 ship dropout 0.3) -------------------------------------------------
 * y = keep ? x / (1 - p) : 0 over a CONTIGUOUS array; keep bits from Philox4x32-10 keyed by *seed (a DEVICE int64 scalar the caller
 * draws from its generator: capturable) and `offset`; mask: dgn_dropout_mask_bytes(n_elems) bytes, bit i of byte g = element 8 g + i,
 * what the backward re-applies: g_x = keep ? g_y / (1 - p) : 0.  The same (seed, offset) gives the same mask.  y may be x and g_x
 * may be g_y (in place).                                                                                                           */
size_t dgn_dropout_mask_bytes(int64_t n_elems);
int dgn_dropout_forward(int64_t n_elems, const float* x, float p, const int64_t* seed, uint64_t offset, float* y, unsigned char* mask,
                        void* stream);
int dgn_dropout_backward(int64_t n_elems, const float* g_y, const unsigned char* mask, float p, float* g_x, void* stream);

/* ---- tall-skinny fp32 Linear (dgn_linear.hip) ---------------------------------------------------------------------
 * The nn.Linear of the reference's pretrans / posttrans MLPs (layers.py:101-112, called from nets/dgn_layer.py:67-75
 * and :116-119) for matrices with ~1e5..1e6 rows (nodes) and k, n <= 160 columns, batched over towers:
 *   dgn_linear_forward   c[t] = a[t] . op(w[t]) (+ bias[t])    a: [n_rows, k], c: [n_rows, n], dense rows (lda == k, ldc == n)
 *                        w_is_kn == 0: w[t] is [n, k] (nn.Linear weight: c = a w^T, the forward)
 *                        w_is_kn == 1: w[t] is [k, n] (c = a w: the input gradient with a = g_out, w = weight)
 *   dgn_linear_wgrad     dw[t] = g[t]^T . x[t]                 g: [n_rows, n], x: [n_rows, k], both dense; dw: [n, k]
 *                        dbias[t] = column sums of g[t] (NULL = not wanted; needs k % 16 != 0: it is computed as the
 *                        product with an extra column of ones in x's tile padding, at no extra pass)
 * Exact fp32 (v_mfma_f32_16x16x4_f32 = an fmaf chain); the weight gradient is summed over per-wave partials in a fixed
 * order (bitwise reproducible).  dgn_linear_supported(k, n) says whether the pair is handled (even, <= 160, and for
 * the weight gradient at most 45 16x16 tiles); callers use a library GEMM otherwise.  stride_* are element offsets
 * between batch entries (towers).  `ws` of dgn_linear_wgrad_workspace_bytes(...) bytes holds the partials.           */
int dgn_linear_supported(int32_t k, int32_t n, int32_t wgrad);
int dgn_linear_forward(int64_t n_rows, int32_t k, int32_t n, int32_t batch, const float* a, int64_t lda, int64_t stride_a,
                       const float* w, int64_t ldw, int64_t stride_w, int32_t w_is_kn, const float* bias,
                       int64_t stride_bias, float* c, int64_t ldc, int64_t stride_c, void* stream);
/* The same two products with the operand a (resp. x) replaced by its BatchNorm -- ((v - mean[c]) * invstd[c]) * gamma[c] + beta[c], the
 * arithmetic of dgn_bn_tail_forward's apply pass in the same order, formed while the strips are staged: the normalised tensor is
 * never written (the towers layer: y1 = BatchNorm(y0) feeds the mixing Linear, nets/dgn_layer.py:272-273 -> :319).  batch = 1, dense
 * rows; gamma / beta may be NULL.  dgn_bn_tail_forward(..., y = NULL, ...) computes the statistics alone.                      */
int dgn_linear_forward_bn(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, int32_t w_is_kn,
                          const float* bias, float* c, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                          const float* bn_beta, void* stream);
int dgn_linear_wgrad_bn(int64_t n_rows, int32_t k, int32_t n, const float* g, const float* x, float* dw, int64_t lddw, float* dbias,
                        const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta, void* ws,
                        size_t ws_bytes, void* stream);
/* c = (g * act'(z + act_bias)) . op(w): the gradient through bias + activation (dgn_bias_act_backward's arithmetic) formed while the
 * strips of g and z ([n_rows, k] dense) are staged, then the Linear's input-gradient product; gz_out (may be NULL) receives the formed
 * operand (the Linear's weight gradient needs it: dgn_linear_wgrad / _bn, whose dbias output is then the bias gradient).
 * Replaces dgn_bias_act_backward + dgn_linear_forward(w_is_kn = 1) for LeakyReLU(Linear(.)) (mixing network, nets/dgn_layer.py:319). */
/* c = (add1 + a . op(w)) + add2: two more [n_rows, n] dense operands added in the product's epilogue (add2 may be NULL) -- the three-way
 * sum of the contributions to d h at the end of the towers layer's backward (residual, h_in of the sweep, the P|Q input gradient). */
int dgn_linear_add_supported(int32_t k, int32_t n);
int dgn_linear_forward_add(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, int32_t w_is_kn,
                           const float* add1, const float* add2, float* c, void* stream);
/* BatchNorm -> Linear -> bias + activation (+ residual) in one pass: z_out = BatchNorm(a) w^T (the pre-activation the backward needs),
 * out = act(z_out + act_bias) + residual (dgn_bn_tail's apply, dgn_linear_forward and dgn_bias_act_forward in their arithmetic and order);
 * widths as dgn_linear_add_supported.  The tail of the towers layer: batchnorm_h -> mixing_network -> residual (nets/dgn_layer.py:272-273,
 * :318-324).                                                                                                                          */
int dgn_linear_forward_bn_act(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, const float* bn_mean,
                              const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const float* act_bias, int32_t act,
                              float slope, const float* residual, float* z_out, float* out, void* stream);
/* The same two products with the pre-activation replaced by a byte mask (1/8 of its bytes): byte i of zmask describes elements 2 i and
 * 2 i + 1 of the dense [n_rows, n] pre-activation, bit 0 / bit 1 = (z + act_bias > 0).  dgn_linear_forward_bn_act_mask writes it (and
 * `out`) instead of z_out; dgn_linear_forward_act_mask takes the activation's derivative from it (1 where set, else slope for
 * LeakyReLU / 0 for ReLU): the values dgn_linear_forward_act computes from z.                                                    */
size_t dgn_linear_act_mask_bytes(int64_t n_rows, int32_t n);
int dgn_linear_forward_bn_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* a, const float* w, int64_t ldw, const float* bn_mean,
                                   const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const float* act_bias, int32_t act,
                                   float slope, const float* residual, unsigned char* zmask_out, float* out, void* stream);
int dgn_linear_forward_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                const float* w, int64_t ldw, int32_t w_is_kn, float* c, float* gz_out, void* stream);
/* Round 6 -- the towers layer's mixing-network backward without the tensors between its steps (autograd through
 * nets/dgn_layer.py:272-273 BatchNorm -> :319 mixing Linear -> LeakyReLU):
 *   dgn_linear_wgrad_bn_act_mask     dw = (g * act'(mask))^T . BatchNorm(x), dbias = sum_m g * act'(mask): dgn_linear_wgrad_bn whose G
 *                                    operand is formed from the layer's output gradient g [n_rows, n] and the forward's byte mask while
 *                                    it is staged (the masked tensor g_z is never written or re-read);
 *   dgn_linear_forward_act_mask_bnb  the input-gradient product g_y1 = (g * act'(mask)) . op(w) of dgn_linear_forward_act_mask with
 *                                    BatchNorm's backward and the graph norm in its epilogue,
 *                                        gz[t][m][o] = row_scale[m] * gamma[c] invstd[c] (g_y1[m][c] - sums[c] / M - xhat[m][c] sums[n + c] / M),
 *                                        c = t f_out + o,  xhat = (y[m][c] - mean[c]) invstd[c]
 *                                    (dgn_scale_combine_backward's fused form, same arithmetic and order), written tower-major
 *                                    [n / f_out][n_rows][f_out] (stride_gz floats between towers): what dgn_linear_combine_backward_*
 *                                    read.  y [n_rows, n] is BatchNorm's input, sums [2 n] = (sum g_y1, sum g_y1 xhat) -- e.g. from the
 *                                    mixing weight gradient (csrc/dgn_towers.hip: mix_bn_finalize); g_y1 itself is never written.
 * Widths: dgn_linear_bnb_supported (k, n <= 80 and the limits of dgn_linear_act_supported / _add_supported); f_out even, n / f_out <= 15.  */
int dgn_linear_bnb_supported(int32_t k, int32_t n);
int dgn_linear_wgrad_bn_act_mask(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                 const float* x, float* dw, int64_t lddw, float* dbias, const float* bn_mean, const float* bn_invstd,
                                 const float* bn_gamma, const float* bn_beta, void* ws, size_t ws_bytes, void* stream);
int dgn_linear_forward_act_mask_bnb(int64_t n_rows, int32_t k, int32_t n, const float* g, const unsigned char* zmask, int32_t act, float slope,
                                    const float* w, int64_t ldw, int32_t w_is_kn, const float* y, const float* bn_mean, const float* bn_invstd,
                                    const float* bn_gamma, const float* sums, const float* row_scale, int32_t f_out, float* gz,
                                    int64_t stride_gz, void* stream);
int dgn_linear_act_supported(int32_t k, int32_t n);      /* (the widest tile shapes are not: two prefetched strips per wave) */
int dgn_linear_forward_act(int64_t n_rows, int32_t k, int32_t n, const float* g, const float* z, const float* act_bias, int32_t act,
                           float slope, const float* w, int64_t ldw, int32_t w_is_kn, float* c, float* gz_out, void* stream);
/* The towers' posttrans Linear with the scale-combine epilogue of dgn_scale_combine_forward in the same pass (the
 * [T, N, S*f_out] product never reaches memory):
 *   y[m, t*f_out + o] = row_scale[m] * (bias[t*f_out + o] + sum_s scale[m, s] * (a[t] w[t]^T)[m, s*f_out + o])
 * a: [T][n_rows][k] dense (stride_a between towers), w[t]: [S*f_out, k]; scale [n_rows, S] (NULL only if S == 1),
 * bias [T*f_out] / row_scale [n_rows] may be NULL.  Same width limits as dgn_linear_forward with n = S*f_out.        */
int dgn_linear_combine_forward(int64_t n_rows, int32_t k, int32_t n_towers, int32_t n_scalers, int32_t f_out, const float* a,
                               int64_t stride_a, const float* w, int64_t ldw, int64_t stride_w, const float* scale,
                               const float* bias, const float* row_scale, float* y, int64_t ld_y, void* stream);
/* Backward of dgn_linear_combine_forward without the [T, N, S*f_out] gradient in memory.  With
 *   G[t][m][s*f_out + o] = scale[m, s] * gy[t][m][o]
 * (gy [T][n_rows][f_out], stride_gy between towers = row_scale * the gradient of the combine's output in tower-major form,
 * i.e. what dgn_scale_combine_backward writes when called with ONE scaler and no scale table) the two products are formed
 * while the strips are staged:
 *   dgn_linear_combine_backward_input    g_a[t] = G[t] . w[t]        [n_rows, k]   (w[t]: [S*f_out, k])
 *   dgn_linear_combine_backward_weight   dw[t]  = G[t]^T . a[t]      [S*f_out, k]  (ws as for dgn_linear_wgrad with n = S*f_out)
 * f_out even, at most 3 scalers, widths as for dgn_linear_forward / dgn_linear_wgrad.                                */
int dgn_linear_combine_backward_input(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k, const float* gy,
                                      int64_t stride_gy, const float* scale, const float* w, int64_t ldw, int64_t stride_w, float* g_a,
                                      int64_t stride_ga, void* stream);
int dgn_linear_combine_backward_weight(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k, const float* gy,
                                       int64_t stride_gy, const float* scale, const float* a, int64_t stride_a, float* dw, int64_t lddw,
                                       int64_t stride_dw, void* ws, size_t ws_bytes, void* stream);
/* ... and with g_sum [T][S*f_out] (may be NULL; needs k % 16 != 0): g_sum[t][s*f_out + o] = sum_m G[t][m][s*f_out + o], the column sums of the
 * expanded gradient from a column of ones in a's padding -- for the identity scaler that is the gradient of posttrans' bias
 * (sum_m row_scale[m] g_y[m, t*f_out + o]), at no cost.                                                                          */
int dgn_linear_combine_backward_weight_bias(int64_t n_rows, int32_t n_towers, int32_t n_scalers, int32_t f_out, int32_t k, const float* gy,
                                            int64_t stride_gy, const float* scale, const float* a, int64_t stride_a, float* dw, int64_t lddw,
                                            int64_t stride_dw, float* g_sum, void* ws, size_t ws_bytes, void* stream);
size_t dgn_linear_wgrad_workspace_bytes(int64_t n_rows, int32_t k, int32_t n, int32_t batch);
int dgn_linear_wgrad(int64_t n_rows, int32_t k, int32_t n, int32_t batch, const float* g, int64_t ldg, int64_t stride_g,
                     const float* x, int64_t ldx, int64_t stride_x, float* dw, int64_t lddw, int64_t stride_dw, float* dbias,
                     int64_t stride_dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- block-diagonal pretrans Linear of the towers layer (dgn_linear_bd.hip) ---------------------------------------------
 * With divide_input every tower's pretrans MLP acts on its own f_in-column slice of h (nets/dgn_layer.py:226-231 via :309-316), so
 * the fused product pq = h [W_s | W_d]^T + [0 | b] has T diagonal blocks of [f_in, f_in] per half: 1/T of the dense
 * [2 T f_in, T f_in] matrix.  These entry points skip the structural zeros (ZINC towers: 40 instead of 180 MFMAs per 16 rows):
 *   dgn_linear_bd_forward          c [n_rows, 2 Fm] = a [n_rows, Fm] . blockdiag(w)^T (+ bias [2 Fm]),   Fm = n_towers * f_in
 *   dgn_linear_bd_backward_input   c [n_rows, Fm]   = (add1 + g [n_rows, 2 Fm] . blockdiag(w)) + add2    (add1 / add2 may be NULL)
 *   dgn_linear_bd_wgrad            dw [2 Fm, lddw]: diagonal blocks of g^T . x, ZERO elsewhere; dbias [2 Fm] = column sums of g (or NULL)
 * w is the dense [2 Fm, ldw] operand of dgn_linear_forward (row j Fm + t f_in + a, column t f_in + b; other entries are not read), so
 * the two routes are interchangeable.  Dense rows, 16-byte aligned operands.  dgn_linear_bd_supported(n_towers, f_in): the
 * instantiated shapes (five towers x even widths 10..30, + (4 | 2) x 14); callers use dgn_linear_forward / _wgrad otherwise.
 * Exact fp32 MFMA; the weight gradient is summed over per-workgroup partials in a fixed order (bitwise reproducible).        */
int dgn_linear_bd_supported(int32_t n_towers, int32_t f_in);
int dgn_linear_bd_forward(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* a, const float* w, int64_t ldw,
                          const float* bias, float* c, void* stream);
int dgn_linear_bd_backward_input(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* w, int64_t ldw,
                                 const float* add1, const float* add2, float* c, void* stream);
size_t dgn_linear_bd_wgrad_workspace_bytes(int64_t n_rows, int32_t n_towers, int32_t f_in);
int dgn_linear_bd_wgrad(int64_t n_rows, int32_t n_towers, int32_t f_in, const float* g, const float* x, float* dw, int64_t lddw,
                        float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- wide tall-skinny fp32 GEMMs (dgn_gemm.hip) -----------------------------------------------------------------------
 * The same nn.Linear (layers.py:101-112) for the widths the simple / complex layers' posttrans has after scaler folding
 * (nets/dgn_layer.py:148,187-190 and :69,116-119): k = aggregators x features up to a few hundred, n = scalers x f_out, any
 * parity (hidden 75 / 65), any row stride >= the row.  Exact fp32 MFMA, shape independent (no library solution selection).
 *   dgn_gemm_forward   c = a . op(w) (+ bias)      a [n_rows, k], c [n_rows, n]; w_is_kn == 0: w [n, k] (forward),
 *                                                   w_is_kn == 1: w [k, n] (input gradient: a = g_out, w = weight)
 *   dgn_gemm_wgrad     dw [n, k] = g^T . x           g [n_rows, n], x [n_rows, k]; dbias [n] = column sums of g (NULL = not wanted:
 *                                                   it rides as a column of ones appended to x, no extra pass); per-workgroup partials
 *                                                   in ws (dgn_gemm_wgrad_workspace_bytes), fixed-order sum: bitwise reproducible.
 *                                                   From 4096 rows on (and always for n > 256 or with dbias): the tile kernel --
 *                                                   v_mfma_f32_32x32x2_f32, a block of up to 256 x 256 outputs held in one workgroup's
 *                                                   registers, every operand element read from memory once per k block */
int dgn_gemm_supported(int32_t k, int32_t n);
int dgn_gemm_forward(int64_t n_rows, int32_t k, int32_t n, const float* a, int64_t lda, const float* w, int64_t ldw, int32_t w_is_kn,
                     const float* bias, float* c, int64_t ldc, void* stream);
size_t dgn_gemm_wgrad_workspace_bytes(int64_t n_rows, int32_t k, int32_t n);
int dgn_gemm_wgrad(int64_t n_rows, int32_t k, int32_t n, const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw,
                   int64_t lddw, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- the posttrans product inside the sweep (dgn_fused.hip) -----------------------------------------------------------
 * dgn_agg_forward + dgn_linear_combine_forward as ONE kernel: the aggregate rows ([T][A][F/T] per node: 1 680 bytes on the
 * ZINC towers config, 0.46 GB per pass) stay in LDS, are multiplied there by the posttrans weights (w[t]: [S*f_out, K],
 * K = agg_total * F/T, exact fp32 MFMA) and leave as
 *     y[i, t*f_out + o] = row_scale[i] * (bias[t*f_out + o] + sum_s scale[i, s] * (agg[t][i] w[t]^T)[s*f_out + o])
 * -- reduce_func's row feeding posttrans (nets/dgn_layer.py:237-249 -> :266-271) without the [N, A*F] tensor in between.
 * Domain (dgn_layer_fused_supported): spec with ONE identity scaler (scalers folded behind posttrans) and an aggregator list of
 * the reference's configs, even F <= 128 and F/T, K <= 96 (a multiple of 4), n_towers * ceil(S*f_out / 16) <= 16, graphs without
 * hub rows whose rows fit one slot batch (batched molecules / superpixel graphs), no bipartite CSR.                   */
int dgn_layer_fused_supported(const DgnGraph* g, const DgnAggSpec* spec, int64_t F, int32_t n_scalers, int32_t f_out);
int dgn_layer_fused_forward(const DgnGraph* g, const DgnAggSpec* spec, const DgnMsg* msg, const float* w, int64_t ld_w,
                            const float* log_deg, const float* weight, int64_t ldw, int64_t stride_w, int32_t n_scalers,
                            int32_t f_out, const float* scale, const float* bias, const float* row_scale, float* y, int64_t ld_y,
                            void* stream);
/* ---- graph batch preparation on the device (dgn_graph_build.hip) ------------------------------------------------------
 * The edge list of a (batched) graph in edge-id order -> the DgnGraph arrays, by a handful of kernels on `stream`, no host
 * synchronisation inside.  Replaces what DGL does per update_all call (degree bucketing, nets/dgn_layer.py:115,186,264) and
 * the ~40 torch ops of the Python build; `dgl.batch` (data/molecules.py:229) is where a caller would invoke it.
 *   dgn_graph_build          indptr [N+1], src_csr / dst_csr [E] (source / destination of every CSR slot; slots of a row in
 *                            ascending edge id), eid [E] (slot -> edge id), log_deg [N], in_degree [N] (int64, may be NULL),
 *                            stats[0] = largest in-degree, stats[1] = rows with more than hub_threshold in-edges
 *   dgn_graph_build_csc      csc_ptr [N+1], csc_pos [E], csc_order [E] (rank -> slot) of DgnGraph's transposed view
 * src / dst are int64 (what torch / DGL hand over); all outputs are caller-allocated device arrays; stats is int32[4] on the
 * device (the caller reads it when it needs the numbers); ws: dgn_graph_build_workspace_bytes() bytes for every call.    */
size_t dgn_graph_build_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int dgn_graph_build(int64_t n_nodes, int64_t n_edges, const int64_t* src, const int64_t* dst, int32_t* indptr, int32_t* src_csr,
                    int32_t* dst_csr, int64_t* eid, float* log_deg, int64_t* in_degree, int32_t* stats, int32_t hub_threshold,
                    void* ws, size_t ws_bytes, void* stream);
/* Closed cuts of a batch (DgnGraph.blk_cut / blk_gap above): blk_cut [n_nodes+1], gap_out a DEVICE int32 (the caller reads it back).
 * dst_csr [n_edges] = destination of every CSR slot (dgn_graph_build's output).  Workspace: dgn_graph_build_workspace_bytes().  */
int dgn_graph_build_cuts(int64_t n_nodes, int64_t n_edges, const int32_t* src_csr, const int32_t* dst_csr, int32_t* blk_cut,
                         int32_t* gap_out, void* ws, size_t ws_bytes, void* stream);
int dgn_graph_build_csc(int64_t n_nodes, int64_t n_edges, const int32_t* src_csr, int32_t* csc_ptr, int32_t* csc_pos,
                        int32_t* csc_order, void* ws, size_t ws_bytes, void* stream);


/* ---- whole towers layer in one call (dgn_towers.hip) ------------------------------------------------------------------
 * DGNLayerTower.forward of the reference (nets/dgn_layer.py:309-325 over DGNTower.forward :254-276) for the fused form the
 * host side already uses (dgn_amd/dgn_layer.py::_fused_towers): single-affine pretrans / posttrans, divide_input, scalers
 * folded behind posttrans, training-mode BatchNorm, mixing network + residual, no edge features, no dropout:
 *     pq   = h [W_s | W_d]^T + [0 | b]                    dgn_linear_forward
 *     aggx = sweep(pq, h)  tower-major [T][N][K]           dgn_agg_forward        (spec: the towers, ONE identity scaler,
 *                                                                                  aggregators incl. the h_in block)
 *     y0   = snorm * (b_post + sum_s scale_s * (aggx W_s^T))   dgn_linear_combine_forward
 *     y1   = BatchNorm(y0)                                  dgn_bn_tail_forward (training; running stats updated in place)
 *     z    = y1 W_mix^T ;  out = LeakyReLU(z + b_mix) [+ h]  dgn_linear_forward, dgn_bias_act_forward
 * One call enqueues all of it on `stream`; dgn_towers_layer_backward enqueues the whole backward.  pq, aggx, y0, y1, z,
 * save_mean, save_invstd are written by the forward and read by the backward (caller-owned: the autograd-saved tensors).
 * Shapes: h [N, T*f_in]; w_sd [2 T f_in, T f_in]; w_post [T][S*f_out][K], K = agg_total * f_in; w_mix [T f_out, T f_out].
 * dgn_towers_layer_supported() says whether the widths fit the streaming Linear kernels.                            */
typedef struct DgnTowersLayer {
    const DgnGraph* graph;
    const DgnAggSpec* spec;
    const float* w;            /* edge weights [n_ch][ld_w] (dgn_edge_weights), or NULL                     */
    int64_t ld_w;
    const float* log_deg;
    int32_t n_towers, f_in, f_out, n_scalers;
    int32_t residual;          /* add h to the output (in_dim == out_dim and the layer's residual flag)     */
    float momentum, eps, slope;
    const float* h;            /* [N, T*f_in]                                                               */
    const float* snorm;        /* [N] graph-norm factor or NULL                                             */
    const float* scale;        /* [N, S] degree-scaler table (NULL iff S == 1)                              */
    const float* w_sd;  const float* bias_sd;
    const float* w_post; const float* b_post;
    const float* bn_gamma; const float* bn_beta;
    float* running_mean; float* running_var;
    const float* w_mix; const float* b_mix;
    float* pq; float* aggx; float* y0; float* save_mean; float* save_invstd; float* y1; float* z;
    float* out;                /* [N, T*f_out]  (forward only)                                              */
    void* ws; size_t ws_bytes; /* scratch: dgn_towers_layer_{forward,backward}_workspace_bytes()            */
    const int64_t* n_valid;    /* DEVICE scalar or NULL: rows >= *n_valid are padding (see dgn_bn_tail_forward)         */
    /* Optional (dgn_towers_layer_zmask_supported): dgn_linear_act_mask_bytes(N, T*f_out) bytes.  When set, the mixing network's
     * pre-activation is never written: the forward leaves the sign mask of (z + b_mix) here, the backward reads the activation's
     * derivative from it (same values bit for bit), and `z` may be NULL.                                                      */
    unsigned char* zmask;
    /* Optional: dgn_agg_aux_bytes(graph, spec, the sweep's message) bytes -- the forward sweep leaves its aux table here, the
     * backward sweep works from it (dgn_agg_forward_aux / dgn_agg_backward_aux).  NULL: the backward recomputes.                */
    unsigned char* agg_aux;
    /* Optional: the towers' F.dropout(h, p, training) between BatchNorm and the mixing network (nets/dgn_layer.py:275).  drop_p > 0
     * needs y1 (the normalised rows are then materialised: the mask is applied to them in place, the mixing Linear and its weight
     * gradient read the dropped rows), drop_seed / drop_offset as dgn_dropout_forward takes them (forward only), and drop_mask of
     * dgn_dropout_mask_bytes(N * T * f_out) bytes (written by the forward, read by the backward); zmask must be NULL.                */
    float drop_p;
    const int64_t* drop_seed;
    uint64_t drop_offset;
    unsigned char* drop_mask;
    /* Round 6 (ABI 26): 1 + the index of the IDENTITY scaler among the n_scalers (its block of w_post carries the h columns,
     * dgn_amd/dgn_layer.py::_assemble); 0 (a zero-initialised struct): unknown -- the separate passes run.  With it the backward takes
     * posttrans' bias gradient from the weight-gradient pass (dgn_linear_combine_backward_weight_bias) and BatchNorm's backward rides in
     * the mixing network's input-gradient product (dgn_linear_forward_act_mask_bnb).                                                */
    int32_t id_slot1;
    /* Round 6 (ABI 28): the BatchNorm modules' num_batches_tracked counters (n_nbt of them, DEVICE memory; torch's
     * `num_batches_tracked += 1` of a training-mode forward, torch/nn/modules/batchnorm.py) -- each is incremented by the forward's
     * statistics kernel instead of by a launch of the caller's.  NULL / 0: not touched.                                            */
    int64_t* num_batches_tracked;
    int32_t n_nbt;
} DgnTowersLayer;
typedef struct DgnTowersGrads {
    const float* g_out;        /* [N, T*f_out]                                                              */
    float* g_h;                /* [N, T*f_in]   written                                                     */
    float* g_w_sd; float* g_bias_sd; float* g_w_post; float* g_b_post; float* g_gamma; float* g_beta; float* g_w_mix; float* g_b_mix;
} DgnTowersGrads;
/* The fused operand buffer from the per-tower parameters (the reference's state_dict layout, nets/dgn_layer.py:205-276 x towers) in ONE
 * launch: out[i] = ((const float*)param_ptrs[map_param[i]])[map_off[i]], or 0 where map_param[i] < 0.  param_ptrs is a DEVICE array of
 * device addresses (int64), the maps are device int32 arrays built once per layer (dgn_amd/dgn_layer.py::_operands).          */
int dgn_assemble_params(int64_t n_out, const int64_t* param_ptrs, const int32_t* map_param, const int32_t* map_off, float* out, void* stream);

/* ---- the simple / complex layers as ONE call per direction (dgn_layers.hip) --------------------------------------------------
 * DGNLayerSimple.forward (nets/dgn_layer.py:178-202) and DGNLayerComplex.forward (:103-132) in the fused form of
 * dgn_amd/dgn_layer.py: single-affine pretrans / posttrans, the degree scalers folded behind the posttrans Linear, training-mode
 * BatchNorm -> ReLU -> residual, no edge features, no dropout:
 *     hp   = h with a zero column appended when f_in is odd (hidden 75 / 65 / 45 / 47: 8-byte lanes + two-phase scatter in the sweep)
 *     pq   = hp [W_s | W_d]^T + [0 | b]                     complex only (pretrans on [h_src || h_dst], decomposed)
 *     agg  = sweep(hp | pq)  [N, K],  K = (n_agg (+ 1: the h_in block, complex)) * f_pad
 *     y    = snorm * (b_post + sum_s scale_s * (agg W_f^T)_s)   W_f = the posttrans weight folded scaler-major, zero columns at the padding
 *     out  = relu(BatchNorm(y)) [+ h]
 * Parameters keep the REFERENCE's layout (w_post [f_out, (complex: f_in +) S * n_agg * f_in], w_pre [f_in, 2 f_in]); the folds, the
 * padding and their adjoints are kernels of the call.  hp, pq, agg, y, wf, wsd, save_mean, save_invstd are written by the forward and
 * read by the backward (caller-owned: the autograd-saved tensors): hp [N, f_pad] (used only when f_in is odd), pq [N, 2 f_pad],
 * agg [N, K], y [N, f_out], wf [2 * S f_out * K] (W_f and its transpose), wsd [4 f_pad^2 + 2 f_pad] (complex: W_sd, its transpose,
 * bias_sd).  spec: the sweep's list (aggregators, + DGN_AGG_X_IN last for the complex layer) with ONE identity scaler, one tower.   */
/* ---- degree classes: the posttrans product with its degree scalers at 1 / S of the folded product's flops -------------------------
 * Every scaler of nets/scalers.py:7-18 multiplies a node's whole aggregate row by a function of its IN-DEGREE, so all nodes of one
 * in-degree d share y = agg (sum_s scale_s(d) W_f[s])^T: one product with f_out columns instead of S f_out (dgn_dc_kernels.hpp).
 * The kernels walk a virtual row space: nodes stably sorted by class = in-degree (0 .. DGN_DC_CLASSES - 1; graphs with larger
 * in-degrees keep the folded route), each class padded to a multiple of DGN_DC_UNIT rows.                                          */
#define DGN_DC_CLASSES 32
#define DGN_DC_UNIT 64
typedef struct DgnDegreeClasses {
    int64_t n_units;              /* units (DGN_DC_UNIT rows) of the virtual row space                                  */
    const int32_t* vperm;         /* [DGN_DC_UNIT * n_units] node of a virtual row, -1: padding                          */
    const int32_t* unit_class;    /* [n_units] class of a unit, -1: empty.  INVARIANT (not checked on the device): every class
                                   * lies in [0, DGN_DC_CLASSES) and the classes ASCEND along the units -- dgn_dc_wgrad keys its partial
                                   * blocks on (workgroup + class) and a 32-bit class mask, which alias otherwise.  DGNGraph.degree_classes()
                                   * (dgn_amd/graph.py) builds it from a stable sort by in-degree and refuses in-degrees >= DGN_DC_CLASSES. */
    const int32_t* present;       /* [DGN_DC_CLASSES] rows per class                                                     */
    const float* scale;           /* [DGN_DC_CLASSES, S] the layer's scaler factors per class (set per layer)            */
} DgnDegreeClasses;
/* Optional: the posttrans weight (and its gradient) in the reference's OWN layout, nn.Linear [f_out, (h_off +) S n_agg f_in], instead of
 * the scaler-major folded matrix wf [S f_out, k]: column kk = a f_pad + f of the product reads W[o][h_off + (s n_agg + a) f_in + f]; the
 * complex layer's h block (a == n_agg; h_off = f_in) reads W[o][f] through the identity scaler's slot; padded columns f >= f_in are 0.   */
typedef struct DgnDcLayout {
    int32_t n_agg, f_pad, f_in, h_off, id_slot;
    int64_t ld;
} DgnDcLayout;
int dgn_dc_supported(int32_t k, int32_t n);
int dgn_dc_wgrad_supported(int32_t k, int32_t n);
/* wc[c][t][o][kk] = sum_s scale[c][s] wf[t][s n + o][kk] and wct[c][t][kk][o] (its transpose), classes present only;
 * wf [towers][S n, k], wc / wct [DGN_DC_CLASSES][towers][n k]                                                                       */
int dgn_dc_fold(const DgnDegreeClasses* d, int32_t S, int32_t n, int32_t k, int32_t towers, const float* wf, const DgnDcLayout* layout /* NULL: wf as above */,
                float* wc, float* wct, void* stream);
/* c[node] = row_scale[node] * (bias + a[node] w_class(node)^T) for every tower t (element offsets t * a_tower / w_tower / c_tower, bias
 * t * n); w: class c at w + c * class_stride, [n, k] rows of stride ldw                                                             */
int dgn_dc_gemm(const DgnDegreeClasses* d, int32_t k, int32_t n, int32_t towers, const float* a, int64_t lda, int64_t a_tower, const float* w,
                int64_t ldw, int64_t class_stride, int64_t w_tower, const float* bias, const float* row_scale, float* c, int64_t ldc,
                int64_t c_tower, int32_t stream_out, void* stream);
/* g_wf[s n + o][kk] = sum_nodes scale[class(node)][s] g[node][o] x[node][kk]  (per-class products, fixed-order finalize)             */
size_t dgn_dc_wgrad_workspace_bytes(int64_t n_units, int32_t k, int32_t n);
int dgn_dc_wgrad(const DgnDegreeClasses* d, int32_t S, int32_t k, int32_t n, const float* g, int64_t ldg, const float* x, int64_t ldx,
                 float* g_wf, int64_t ldw, const DgnDcLayout* layout /* NULL, or: g_wf is the reference-layout gradient [n, layout->ld] */, void* ws,
                 size_t ws_bytes, void* stream);

typedef struct DgnDenseLayer {
    const DgnGraph* graph;
    const DgnAggSpec* spec;
    const float* w;            /* edge weights [n_ch][ld_w] (dgn_edge_weights), or NULL                      */
    int64_t ld_w;
    const float* log_deg;
    int32_t type;              /* 0 = simple, 1 = complex                                                    */
    int32_t f_in, f_out, n_scalers, n_agg;
    int32_t id_slot;           /* complex: position of the identity scaler among the applied ones           */
    int32_t residual;
    float momentum, eps;
    const float* h;            /* [N, f_in]                                                                  */
    const float* snorm;        /* [N] graph-norm factor or NULL                                              */
    const float* scale;        /* [N, S] degree-scaler table (NULL iff S == 1)                               */
    const float* w_pre;  const float* b_pre;     /* complex: pretrans Linear (bias may be NULL)              */
    const float* w_post; const float* b_post;    /* posttrans Linear (bias may be NULL)                      */
    const float* bn_gamma; const float* bn_beta;
    float* running_mean; float* running_var;
    float* hp; float* pq; float* agg; float* y; float* wf; float* wsd; float* save_mean; float* save_invstd;
    float* out;                /* [N, f_out]  (forward only)                                                 */
    void* ws; size_t ws_bytes;
    const int64_t* n_valid;    /* DEVICE scalar or NULL (padded batches, see dgn_bn_tail_forward)            */
    unsigned char* agg_aux;    /* optional: dgn_dense_layer_agg_aux_bytes() bytes, the sweep's aux table (dgn_agg_forward_aux) */
    const DgnDegreeClasses* dc; /* optional (S > 1, f_out <= 128): degree-class posttrans; wf then holds (2 S + 2 DGN_DC_CLASSES) f_out K floats */
    int64_t* num_batches_tracked; /* ABI 28: the BatchNorm module's counter (DEVICE; see DgnTowersLayer), incremented by the forward; NULL: not touched */
} DgnDenseLayer;
typedef struct DgnDenseGrads {
    const float* g_out;        /* [N, f_out]                                                                 */
    float* g_h;                /* [N, f_in]  (written; includes the residual's share)                        */
    float* g_w_pre; float* g_b_pre; float* g_w_post; float* g_b_post; float* g_gamma; float* g_beta;   /* written */
} DgnDenseGrads;
int dgn_dense_layer_supported(int32_t type, int32_t f_in, int32_t f_out, int32_t n_scalers, int32_t n_agg);
size_t dgn_dense_layer_agg_aux_bytes(const DgnDenseLayer* L);      /* graph, spec, type and widths set; 0: no aux table */
size_t dgn_dense_layer_forward_workspace_bytes(const DgnDenseLayer* layer);
int dgn_dense_layer_forward(const DgnDenseLayer* layer, void* stream);
size_t dgn_dense_layer_backward_workspace_bytes(const DgnDenseLayer* layer);
int dgn_dense_layer_backward(const DgnDenseLayer* layer, const DgnDenseGrads* grads, void* stream);

int dgn_towers_layer_supported(int32_t n_towers, int32_t f_in, int32_t f_out, int32_t n_scalers, int32_t n_agg_total);
/* 1 when dgn_towers_layer_forward / _backward can work from DgnTowersLayer.zmask alone (the fused mixing-network kernels exist for
 * T * f_out columns and are not switched off): the caller then need not allocate `z`.                                            */
int dgn_towers_layer_zmask_supported(int32_t n_towers, int32_t f_out);
/* bytes of DgnTowersLayer.agg_aux for this layer (graph, spec and widths set; 0: the sweep has no aux table)                    */
size_t dgn_towers_layer_agg_aux_bytes(const DgnTowersLayer* L);
size_t dgn_towers_layer_forward_workspace_bytes(const DgnTowersLayer* layer);
int dgn_towers_layer_forward(const DgnTowersLayer* layer, void* stream);
size_t dgn_towers_layer_backward_workspace_bytes(const DgnTowersLayer* layer);
int dgn_towers_layer_backward(const DgnTowersLayer* layer, const DgnTowersGrads* grads, void* stream);

/* ---- the layer of a batch at the reference's own batch size as FIVE launches per step (dgn_blk_layer.hip) ---------------------------
 * Every shipped config trains at batch 128 (configs/molecules_graph_regression_DGN_ZINC.json:12, superpixels_graph_classification_DGN_
 * CIFAR10.json:12; the hot loop of main_molecules.py / train/train_molecules_graph_regression.py:13-45): 3 000 - 15 000 rows, where a
 * layer through the streaming kernels above is ~28 launches of a few microseconds each.  A dgl.batch (data/molecules.py:229) is block
 * diagonal: a workgroup that owns whole graphs runs the layer for them out of LDS.  Replaces, for one DGNLayer{Simple,Complex,Tower}
 * .forward + autograd backward (nets/dgn_layer.py:178-202, :103-132, :254-276 + :309-325) in training mode:
 *     forward   blk_forward  (a workgroup per block): edge weights from eig (nets/aggregators.py:35-71) -> pretrans as P[src] + Q[dst]
 *               -> aggregators -> scalers -> posttrans -> graph norm = y0, BatchNorm partial sums
 *               blk_tail_fwd (16 rows per wave): BatchNorm (training statistics, running statistics and num_batches_tracked updated)
 *               -> ReLU -> + h (simple / complex)   or   -> mixing Linear -> LeakyReLU -> + h (towers)
 *     backward  blk_tail_bwd, blk_backward (recomputes the block's forward in LDS: only y0, mean, invstd were saved), blk_reduce
 * Parameters and their gradients keep the REFERENCE's state_dict layout, one tensor per tower (no fold / assembly launches):
 *     w_pre[t] [f_in, 2 f_in], b_pre[t] [f_in]  (complex / towers; no edge features on this route)
 *     w_post[t] [f_out, (complex / towers: f_in +) S * n_agg * f_in], b_post[t] [f_out];  gamma[t], beta[t] [f_out]
 *     w_mix [T f_out, T f_out], b_mix [T f_out]  (towers)
 * simple / complex: n_towers = 1, f_in = the hidden size.  No dropout, no padded batches, at most 3 edge-weight channels.           */
#define DGN_BLK_MAX_TOWERS 8
typedef struct DgnBlockTable {
    int32_t n_blocks;
    int32_t max_rows;        /* largest block: rows ...                                                                       */
    int32_t max_edges;       /* ... and CSR slots                                                                             */
    /* DEVICE [n_blocks][4] int32 (16-byte aligned): first row, end row, first CSR slot, end CSR slot.  The blocks partition [0, N) in
     * order and are CLOSED: every edge of a block's rows starts inside the block (whole graphs of the batch).                      */
    const int32_t* desc;
} DgnBlockTable;
typedef struct DgnBlockLayer {
    const DgnGraph* graph;       /* indptr, src; the backward also csc_ptr / csc_pos                                          */
    const DgnBlockTable* blocks;
    const DgnAggSpec* spec;      /* aggregators (no DGN_AGG_X_IN), n_ch, the APPLIED scalers (dgn_layer.py:170), avg_log, eps       */
    const DgnChannel* channels;  /* HOST array [spec->n_ch]: how eig becomes each weight channel                              */
    const float* eig; int64_t ld_eig; int32_t n_eig_cols;      /* [N, ld_eig] node eigenvectors (g.ndata['eig'])               */
    const float* log_deg;        /* [N] log(in-degree + 1) as the scalers use it (nets/scalers.py:11,16)                       */
    int32_t type;                /* 0 simple, 1 complex, 2 towers (n_towers >= 2, divide_input)                               */
    int32_t n_towers, f_in, f_out;      /* per tower                                                                          */
    int32_t residual;
    float momentum, eps, slope;  /* BatchNorm momentum / eps; LeakyReLU slope of the mixing network                           */
    const float* h;              /* [N, T f_in]                                                                               */
    const float* snorm;          /* [N] graph-norm factors, NULL: graph_norm off                                              */
    const float* const* w_pre; const float* const* b_pre; const float* const* w_post; const float* const* b_post;      /* HOST arrays of T device pointers */
    const float* const* gamma; const float* const* beta;
    const float* w_mix; const float* b_mix;
    float* running_mean; float* running_var;      /* [T f_out] (the towers' statistics behind each other), updated in place     */
    int64_t* num_batches_tracked; int32_t n_nbt;  /* n_nbt counters (one per BatchNorm module), each incremented; may be NULL   */
    float* y0; float* save_mean; float* save_invstd;      /* [N, T f_out], [T f_out] x 2: written by the forward, read by the backward */
    float* out;                  /* [N, T f_out] (forward)                                                                    */
    void* ws; size_t ws_bytes;   /* dgn_block_layer_{forward,backward}_workspace_bytes()                                      */
    /* Padded batches (a captured step replayed over batches of different sizes: the buffers hold N = graph->n_nodes rows of which the
     * first *n_valid are the batch): n_valid = DEVICE int64 scalar, NULL = every row is real.  BatchNorm counts *n_valid rows; out and
     * g_h rows from *n_valid on are written as zeros.  The block table then changes per batch inside static buffers: entries with
     * first row == end row are unused, negative slot fields are read from graph->indptr, and blocks->max_rows / max_edges are the
     * CAPACITY the LDS plan is made for -- a block beyond it is skipped and *overflow (DEVICE int32, may be NULL) set to 1.          */
    const int64_t* n_valid; int32_t* overflow;
    /* eval_mode = 1 (dgn_block_layer_forward only): the layer in evaluation mode -- BatchNorm with the RUNNING statistics, which are
     * read and not updated; num_batches_tracked, save_mean / save_invstd are not touched (may be NULL); y0 is still the hand-over
     * buffer between the two launches.  nn.Module.eval() + torch.no_grad() of the reference's evaluation loops
     * (train/train_molecules_graph_regression.py:47-66).                                                                         */
    int32_t eval_mode;
    float* dbg_agg; float* dbg_gagg;      /* tests only: [N, T n_agg f_in] aggregate rows (forward) / their gradients (backward); NULL */
    int64_t* dbg_time;           /* profiling only: [n_blocks][16] wall-clock stamps (100 MHz) of the block kernel's phases; NULL      */
    /* Round 6 (ABI 27): F.dropout(h, p, training) inside the tail kernels -- towers (type 2): between the towers' BatchNorm and the mixing
     * network (nets/dgn_layer.py:275); simple / complex (types 0 / 1; ABI 28's library, same fields): the layer's LAST op, on the finished
     * output rows (:130, :201; the backward masks the output gradient, the residual's share of d h included).  drop_p in (0, 1): drop_seed (DEVICE int64 key; forward only) / drop_offset as dgn_dropout_forward takes
     * them, drop_mask = dgn_dropout_mask_bytes(N * T * f_out) bytes, written by the forward (the very bits dgn_dropout_forward would
     * draw for the dense [N, T f_out] tensor) and read by the backward.  drop_p = 0: none.  Ignored in eval_mode.                   */
    float drop_p; const int64_t* drop_seed; uint64_t drop_offset; unsigned char* drop_mask;
} DgnBlockLayer;
typedef struct DgnBlockGrads {
    const float* g_out;          /* [N, T f_out]                                                                              */
    float* g_h;                  /* [N, T f_in] written (includes the residual's share)                                       */
    /* dgn_block_layer_param_grad_floats() floats, written: per tower [w_pre | b_pre] (complex / towers) [w_post | b_post], then
     * (towers) [w_mix | b_mix] -- each in the parameter's own layout                                                          */
    float* g_params;
    float* g_gamma; float* g_beta;      /* [T f_out] each                                                                     */
} DgnBlockGrads;
int dgn_block_layer_supported(const DgnBlockLayer* layer);      /* graph, blocks, spec, type and widths set: 1 if every block fits */
int64_t dgn_block_layer_param_grad_floats(const DgnBlockLayer* layer);
size_t dgn_block_layer_forward_workspace_bytes(const DgnBlockLayer* layer);
int dgn_block_layer_forward(const DgnBlockLayer* layer, void* stream);
size_t dgn_block_layer_backward_workspace_bytes(const DgnBlockLayer* layer);
int dgn_block_layer_backward(const DgnBlockLayer* layer, const DgnBlockGrads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGN_HIP_H */
