/*
 * dcr.h — C ABI of libdcr_hip.so: MI355X (gfx950) Balanced Forman curvature,
 * SDRF rewiring primitives and the GCN sparse aggregation.
 *
 * The reference (jakubbober/discrete-curvature-rewiring) is pure Python and has
 * no FFI of its own; its only native launches are numba kernels taking torch
 * CUDA tensors (curvature/bfc_cuda.py:64,157).  This header is the boundary a
 * maintainer binds with ctypes underneath the reference's Python call surface
 * (see INTEGRATION.md).  Each entry point names the reference code it replaces
 * (file:line into the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, a negative DCR_E* code on failure;
 *     dcr_last_error() returns a thread-local message for the last failure;
 *   - plain pointers and sizes only; HOST pointers unless the name says _dev;
 *   - the caller owns every buffer it passes in; buffers returned through
 *     `const T**` are library-owned pinned host memory, valid until the next
 *     call on the same handle;
 *   - a dcr_graph handle is not thread-safe; distinct handles are independent;
 *   - all calls are synchronous on return (the library owns one HIP stream
 *     per handle) unless documented otherwise.
 */
#ifndef DCR_H
#define DCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dcr_graph dcr_graph;

enum {
    DCR_OK = 0,
    DCR_EINVAL = -1,   /* bad argument (self-loop, id out of range, ...)        */
    DCR_ENOMEM = -2,   /* host or device allocation failed                      */
    DCR_EHIP = -3,     /* a HIP runtime call or kernel failed                   */
    DCR_ECAPACITY = -4,/* caller buffer too small / degree beyond kernel limit  */
    DCR_ENOTFOUND = -5,/* edge not present                                      */
    DCR_ESTATE = -6    /* call sequence error (e.g. argext before a pass)       */
};

/* curvature kinds: curvature/bfc_naive.py:7 and curvature/classical_curvatures.py:14-28 */
enum { DCR_CURV_BFC = 0, DCR_CURV_1D = 1, DCR_CURV_AUGMENTED = 2, DCR_CURV_HAANTJES = 3 };

const char *dcr_last_error(void);
int dcr_device_count(int *out);

/* ---- graph container ------------------------------------------------------
 * Replaces the nx.Graph built by torch_geometric.utils.to_networkx(data,
 * to_undirected=True) at rewiring/sdrf_no_cuda.py:20: walks (src[e], dst[e]) in
 * order, keeps pairs with dst <= src, appends each new neighbour at the end of
 * both adjacency rows (insertion order is part of the contract: SURVEY §8 A5).
 * Rows live in HBM as slack-padded int32 arrays in insertion order. */
int dcr_graph_create(int device, int64_t num_nodes, int64_t m_directed, const int64_t *src, const int64_t *dst,
                     dcr_graph **out);
int dcr_graph_destroy(dcr_graph *g);
int dcr_graph_num_nodes(const dcr_graph *g, int64_t *out);
int dcr_graph_num_edges(const dcr_graph *g, int64_t *out_undirected);

/* G.add_edge / G.remove_edge / G.has_edge / G.degree — sdrf_no_cuda.py:35,43,46,51,63 */
int dcr_graph_add_edge(dcr_graph *g, int32_t u, int32_t v);
int dcr_graph_remove_edge(dcr_graph *g, int32_t u, int32_t v);
int dcr_graph_has_edge(dcr_graph *g, int32_t u, int32_t v, int *out);
int dcr_graph_degree(dcr_graph *g, int32_t u, int32_t *out);
/* list(G.neighbors(u)) in insertion order — sdrf_no_cuda.py:29-30 */
int dcr_graph_neighbors(dcr_graph *g, int32_t u, int64_t cap, int32_t *out, int64_t *n_out);

/* G.edges enumeration order — sdrf_no_cuda.py:27,59,61; out_u/out_v sized num_edges */
int dcr_graph_edges(dcr_graph *g, int32_t *out_u, int32_t *out_v);
/* from_networkx(G).edge_index — sdrf_no_cuda.py:68; out is int64 [2][2*num_edges] */
int dcr_graph_export_edge_index(dcr_graph *g, int64_t *out2xM);

/* ---- curvature ------------------------------------------------------------ */
/* One full pass: compute_curvature_graph(G, curv_type), sdrf_no_cuda.py:24
 * (classical_curvatures.py:38-46; 'bfc' = bfc_naive.bfc(G), bfc_naive.py:43-52).
 * Values stay in HBM, keyed by adjacency slot; the stale-read semantics of
 * sdrf_no_cuda.py:57-61 follow from later calls reading that buffer. */
int dcr_curvature_pass(dcr_graph *g, int curv_type);
/* Same result, less work: recompute only the edges whose value can have changed since the previous pass of the same
 * kind, i.e. those with an endpoint in {a,b} ∪ N(a) ∪ N(b) of an edge (a,b) added or removed meanwhile (every edit
 * through this API is tracked).  Falls back to a full pass when there is no complete buffer to build on.  The
 * reference recomputes everything each iteration (sdrf_no_cuda.py:24); this is an optional mode. */
int dcr_curvature_pass_incremental(dcr_graph *g, int curv_type);
/* A pass followed by min(G.edges, key=curvature) — sdrf_no_cuda.py:24 and :27 — with one host synchronisation. */
int dcr_curvature_pass_argmin(dcr_graph *g, int curv_type, int incremental, int32_t *out_u, int32_t *out_v,
                              double *out_val);
/* Copy the last pass out in G.edges order (float64 per undirected edge). */
int dcr_curvature_read(dcr_graph *g, double *out_curv, int32_t *out_u, int32_t *out_v);
/* bfc_edge(G, v1, v2), bfc_naive.py:7-40 / compute_curvature_edge, classical_curvatures.py:6 */
int dcr_curvature_edge(dcr_graph *g, int32_t u, int32_t v, int curv_type, double *out);
/* integer ingredients of bfc_edge: d1,d2,triangles,|sq1|,|sq2|,gamma (diagnostic / tests) */
int dcr_bfc_ingredients(dcr_graph *g, int32_t u, int32_t v, int64_t out6[6]);

/* min(G.edges, key=curv) / max([e for e in G.edges if e != (k,l)], key=curv):
 * first extremum in G.edges order over the buffer of the last pass
 * (sdrf_no_cuda.py:27,59,61).  excl_u < 0 means no exclusion. */
int dcr_argext(dcr_graph *g, int want_max, int32_t excl_u, int32_t excl_v, int32_t *out_u, int32_t *out_v,
               double *out_val);

/* Candidate set and improvements for edge (x, y): sdrf_no_cuda.py:29-46.
 * Candidates are produced in the reference's nested-loop order (duplicates
 * kept) as sorted pairs; improvement[c] = curv(x,y | G + cand[c]) - curv(x,y | G).
 * *n_out is the candidate count; the three arrays are library-owned pinned
 * host buffers (cand arrays are filled only if want_candidates != 0). */
int dcr_improvements(dcr_graph *g, int32_t x, int32_t y, int curv_type, int want_candidates, int64_t *n_out,
                     const double **out_improvement, const int32_t **out_ci, const int32_t **out_cj);
/* first index of the maximum improvement of the last dcr_improvements call
 * (np.argmax, utils/softmax.py:7) and the candidate pair at any index */
int dcr_improvements_argmax(dcr_graph *g, int64_t *out_index);
int dcr_candidate_at(dcr_graph *g, int64_t index, int32_t *out_i, int32_t *out_j);

/* Tail of one SDRF iteration, sdrf_no_cuda.py:51,56-66, fused on the device:
 * add (k,l) if add_k >= 0; then, if do_remove, take the first maximum of the
 * stale buffer (excluding (k,l) when it was added) and remove that edge iff
 * its value > removal_bound.  out_removed = {u, v} or {-1,-1}; out_max_val =
 * the maximum examined. */
int dcr_sdrf_tail(dcr_graph *g, int32_t add_k, int32_t add_l, int do_remove, double removal_bound,
                  int32_t out_removed[2], double *out_max_val);
/* The same with the edge to add given as an index into the candidate list of the last dcr_improvements call (the
 * index np.random.choice returned, sdrf_no_cuda.py:49-51): the pair is looked up on the device, out_added receives it. */
int dcr_sdrf_tail_at(dcr_graph *g, int64_t cand_index, int do_remove, double removal_bound, int32_t out_added[2],
                     int32_t out_removed[2], double *out_max_val);

/* Which implementation ran the last curvature pass: 0 two-hop kernels (csrc/dcr_bfc_h2.hip: full Balanced Forman passes of
 * graphs whose 2-hop neighbourhoods are a small part of the graph — the default there — or always with DCR_PASS=h2 when the
 * graph is created), 2 node-centric kernels (csrc/dcr_bfc_nc.hip: everything else, or DCR_PASS=nc), 1 edge-centric kernels
 * (DCR_PASS=edge); -1 before the first pass.  All three produce the same bits. */
int dcr_pass_engine(dcr_graph *g, int *out);

/* dcr_sdrf_tail_at followed by dcr_curvature_pass_argmin of the NEXT iteration (sdrf_no_cuda.py:51,56-66 then :24,:27) with
 * one host synchronisation instead of two: the pass is enqueued right behind the edit.  Same results as the two calls. */
int dcr_sdrf_tail_at_pass_argmin(dcr_graph *g, int64_t cand_index, int do_remove, double removal_bound, int curv_type,
                                 int incremental, int32_t out_added[2], int32_t out_removed[2], int32_t *out_u, int32_t *out_v,
                                 double *out_val);

/* One whole loop iteration for finite tau or tau = +inf (then the draw is the first arg-max, utils/softmax.py:5-8) (sdrf_no_cuda.py:29-66 for the edge (x, y) the previous call returned, then :24,:27 of
 * the next iteration) with the improvements staying on the device (two host round trips of the 1 KB result block; no
 * transfer of the improvements, no host arithmetic).  The draw np.random.choice(n, p=softmax(improvements, tau)) (:49-50) runs
 * on the device from `uniform`, the one double the caller has taken from numpy's global stream: the index is the first i whose
 * prefix sum of exp(tau * improvement) exceeds uniform * total, accepted only when that comparison is decided by a margin wider
 * than every rounding difference between numpy's arithmetic and the device's.  *out_status: 0 done (outputs as
 * dcr_sdrf_tail_at_pass_argmin); 1 undecided or not finite, 2 no candidates: nothing was edited, no pass was run (out_u /
 * out_v / out_val untouched), and the caller runs this iteration through dcr_improvements + the host draw instead (after
 * putting the uniform back into the stream). */
int dcr_sdrf_iteration_device_draw(dcr_graph *g, int32_t x, int32_t y, int curv_type, double tau, double uniform, int do_remove,
                                   double removal_bound, int incremental, int *out_status, int64_t *out_n_cand,
                                   int32_t out_added[2], int32_t out_removed[2], int32_t *out_u, int32_t *out_v, double *out_val);

/* Timing hooks for bench.py: accumulated device time (HIP events on the
 * handle's stream) of the curvature-pass kernels since the last reset. */
int dcr_profile_reset(dcr_graph *g);
int dcr_profile_read(dcr_graph *g, double *pass_ms_total, int64_t *pass_count);
/* SURVEY §8(d) algorithmic bytes of one BFC pass on the current graph (both difference sets of every edge charged). */
int dcr_bfc_algorithmic_bytes(dcr_graph *g, double *out_bytes);
/* The same with only the cheaper difference set's rows charged per edge: 4(d_u+d_v) + 4*sum of the row lengths of that
 * side + 8(2 + its size) + 8.  sq1, sq2 and gamma (bfc_naive.py:26-29,36-37) are degree statistics of ONE bipartite
 * graph, so a pass has to read one side only; this is the byte count bench.py's roofline fraction is quoted on. */
int dcr_bfc_algorithmic_bytes_one_sided(dcr_graph *g, double *out_bytes);

/* ---- dense float32 Balanced Forman curvature: the numerics of the reference's numba path (device pointers, caller's
 * stream).  curvature/bfc_cuda.py computes a different number from curvature/bfc_naive.py (float32 dense formula, other
 * 4-cycle term, no degree-1 rule) and it is what rewire('bfc') runs in the reference (rewiring/rewire.py:8-10), so results
 * obtained with the reference can only be reproduced with these.  A, A2 = A·A, C: row-major N x N float32; d_in / d_out:
 * column / row sums of A; pairs: the (i, j) of the non-zero entries of A as int64 [nnz][2] (C must be zero elsewhere:
 * the caller clears it).  Float64 closing expression rounded to float32 twice, as numba types the kernel.
 *   dcr_bfc_dense_f32_dev              replaces _balanced_forman_curvature   (curvature/bfc_cuda.py:11-48, launch :64)
 *   dcr_bfc_dense_post_delta_f32_dev   replaces _balanced_forman_post_delta  (curvature/bfc_cuda.py:68-141, launch :157):
 *       D[I][J] = curvature of (x, y) once (i_neighbors[I], j_neighbors[J]) is added; -1000 where the two are equal or
 *       already adjacent (:77-79); d_in_x = A[:, x].sum(), d_out_y = A[y].sum() (:147-148). */
int dcr_bfc_dense_f32_dev(const float *A_dev, const float *A2_dev, const float *d_in_dev, const float *d_out_dev, int64_t N,
                          const int64_t *pairs_dev, int64_t nnz, float *C_dev, void *hip_stream);
int dcr_bfc_dense_post_delta_f32_dev(const float *A_dev, const float *A2_dev, float d_in_x, float d_out_y, int64_t N,
                                     float *D_dev, int32_t x, int32_t y, const int32_t *i_neighbors_dev,
                                     const int32_t *j_neighbors_dev, int64_t dim_i, int64_t dim_j, void *hip_stream);

/* ---- host helper for np.random.choice(n, p=softmax(a, tau)) — sdrf_no_cuda.py:49-50, utils/softmax.py:9-10
 * Given e = exp(a * tau) and its sum S (both computed by the caller's numpy: their rounding is part of the bit-exact
 * contract), fill cdf[i] = the SEQUENTIAL float64 running sum of p_i = e_i / S, i.e. numpy.cumsum(e / S), in one fused
 * pass with numpy's operations in numpy's order, and return cdf[n-1].  The caller validates the total, draws ONE uniform
 * from the legacy stream and bisects cdf / total exactly as RandomState.choice does.  Plain host code, no GPU. */
int dcr_host_cdf_from_exp(const double *e, int64_t n, double S, double *out_cdf, double *out_total);
/* The same through the plain sequential loop only.  dcr_host_cdf_from_exp evaluates the recurrence eight elements at a
 * time in integer-valued float64 arithmetic where that is exact (csrc/dcr_host_draw.cpp) and falls back to this loop
 * elsewhere; the two must agree bit for bit (tests/test_host_cpu.py). */
int dcr_host_cdf_from_exp_plain(const double *e, int64_t n, double S, double *out_cdf, double *out_total);

/* ---- GCN aggregation (device pointers, caller's stream) --------------------
 * Replaces the propagate/scatter step of torch_geometric GCNConv (third-party,
 * call site models/gcn.py:36): C[i,:] = (bias ? bias : 0) + sum_e val[e] * B[col[e],:]
 * over CSR row i, optionally followed by ReLU.  fp32, row-major; ldb/ldc in
 * elements.  Rows are accumulated with fused multiply-adds in CSR order (rows
 * above 96 non-zeros: in a fixed strided order), so the result is deterministic
 * for a given CSR.  Asynchronous on hip_stream. */
int dcr_spmm_csr_f32_dev(const int64_t *rowptr_dev, const int32_t *col_dev, const float *val_dev,
                         const float *B_dev, float *C_dev, int64_t n_rows, int64_t n_feat, int64_t ldb, int64_t ldc,
                         const float *bias_dev, int relu, void *hip_stream);

/* Two operands in one sweep of the indices: B = [B0 | B1] and C = [C0 | C1] hold two blocks of n_feat columns side by
 * side (ldb, ldc >= 2 * n_feat); C0 = Â·B0 + bias, C1 = Â·B1 + bias, every block accumulated with the operations and in
 * the order of a dcr_spmm_csr_f32_dev call of its own (bit-identical to two calls).  The training forward of one epoch and
 * the validation forward of the previous one see the same weights, so their aggregations (models/gcn.py:36, called from
 * experiment/training_loop.py:50 and :67) share one pass over the graph: the gathers are bound by the number of row
 * requests, and a 128-byte request costs what a 64-byte one does. */
int dcr_spmm_csr_f32_pair_dev(const int64_t *rowptr_dev, const int32_t *col_dev, const float *val_dev, const float *B_dev,
                              float *C_dev, int64_t n_rows, int64_t n_feat, int64_t ldb, int64_t ldc, const float *bias_dev,
                              int relu, void *hip_stream);
/* the same with the two output blocks in matrices of their own (C1 - C0 a multiple of 4 floats; ldc the row stride of both) */
int dcr_spmm_csr_f32_pair_split_dev(const int64_t *rowptr_dev, const int32_t *col_dev, const float *val_dev, const float *B_dev,
                                    float *C0_dev, float *C1_dev, int64_t n_rows, int64_t n_feat, int64_t ldb, int64_t ldc,
                                    const float *bias_dev, int relu, void *hip_stream);
/* Only the rows rows_dev[0..n_sel) of the product: C[k,:] = row rows_dev[k] of Â·B (+ bias), each accumulated exactly as
 * dcr_spmm_csr_f32_dev accumulates it.  The reference indexes the model's output with the split masks and reads nothing else
 * (experiment/training_loop.py:50-51 log_probs[train_mask]; :64-71 log_probs[mask] of the evaluated split): the last layer's
 * aggregation (models/gcn.py:36) is evaluated at those rows. */
int dcr_spmm_csr_rows_f32_dev(const int64_t *rowptr_dev, const int32_t *col_dev, const float *val_dev, const int64_t *rows_dev,
                              int64_t n_sel, const float *B_dev, float *C_dev, int64_t n_feat, int64_t ldb, int64_t ldc,
                              const float *bias_dev, int relu, void *hip_stream);
/* Two row lists over two column blocks of one operand in one launch: output rows [0, n_first) = rows rows_dev[k] of Â·B0,
 * rows [n_first, n_sel) = rows rows_dev[k] of Â·B1, B1 = B0 + b_off2 floats (the training rows of Â·Z_train and the validation
 * rows of Â·Z_eval of one epoch, Z = [Z_train | Z_eval]: experiment/training_loop.py:50-51 and :64-71 on the same weights). */
int dcr_spmm_csr_rows2_f32_dev(const int64_t *rowptr_dev, const int32_t *col_dev, const float *val_dev, const int64_t *rows_dev,
                               int64_t n_first, int64_t n_sel, const float *B_dev, int64_t b_off2, float *C_dev, int64_t n_feat,
                               int64_t ldb, int64_t ldc, const float *bias_dev, int relu, void *hip_stream);

/* ---- GCN weight gradient on the matrix cores (device pointers, caller's stream)
 * C[M x N] = A^T * B with A [K x M] and B [K x N] row-major fp32 (lda/ldb/ldc in
 * elements), K = number of nodes: the backward of GCNConv's bias-free Linear,
 * dW = dZ^T * X (third-party torch_geometric; call site models/gcn.py:36 through
 * autograd).  Exact f32 MFMA (v_mfma_f32_32x32x2_f32), K split across workgroups,
 * partial tiles summed in a fixed order: deterministic.  The caller provides a
 * scratch buffer of dcr_atb_f32_workspace() floats.  Asynchronous on hip_stream. */
int dcr_atb_f32_workspace(int64_t K, int64_t M, int64_t N, int64_t *out_floats);
int dcr_atb_f32_dev(const float *A_dev, const float *B_dev, float *C_dev, int64_t K, int64_t M, int64_t N, int64_t lda,
                    int64_t ldb, int64_t ldc, float *workspace_dev, int64_t workspace_floats, void *hip_stream);

/* ---- ReLU + dropout between the GCN layers (device pointers, caller's stream) ------------------------------------
 * models/gcn.py:38-42 (x = act_fn(x); x = dropout(x)) as one pass per direction: y = x > 0 and kept ? x / (1 - p) : 0,
 * the keep decisions packed one bit per element into `bits` (dcr_relu_dropout_bits_words(n) 64-bit words); backward
 * scales the incoming gradient by the same bits.  Philox-4x32-10 keyed by (seed, offset): reproducible for a seed.
 * Sixteen random bits per element: an element is kept with probability 1 - floor(p * 65536) / 65536 (exact at p = 0.5). */
int dcr_relu_dropout_bits_words(int64_t n, int64_t *out_words);
int dcr_relu_dropout_fwd_f32_dev(const float *x_dev, float *y_dev, uint64_t *bits_dev, int64_t n, double p, uint64_t seed,
                                 uint64_t offset, void *hip_stream);
/* the same with the stream offset = offset + *offset_dev (a call counter the caller keeps in device memory and bumps
 * after each call): the launch parameters are then constants, so the call can be captured in a hipGraph and replayed */
int dcr_relu_dropout_fwd_f32_ctr_dev(const float *x_dev, float *y_dev, uint64_t *bits_dev, int64_t n, double p,
                                     uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *hip_stream);
int dcr_relu_dropout_bwd_f32_dev(const float *grad_out_dev, float *grad_in_dev, const uint64_t *bits_dev, int64_t n,
                                 double p, void *hip_stream);

/* ---- the same activation fused into the next layer's dense contraction ------------------------------------------------
 * models/gcn.py:36-42 between two layers (x = relu(x); x = dropout(x); next GCNConv's lin: x·W^T, W = [classes, hidden]) in
 * ONE pass over the hidden activation: z_train = dropout(relu(x))·W^T (with h_train = dropout(relu(x)) stored for the weight
 * gradient and the keep bits as above, same Philox stream as dcr_relu_dropout_fwd_f32_ctr_dev) and / or z_eval = relu(x)·W^T.
 * A null z_train (z_eval) skips that operand; a null h_train skips the stored training activation.  hidden 64 or 128, classes <= 16 (other shapes: the separate entry points).
 * Backward of the training operand: dx = keep ? (dz·W) / (1 - p) : 0. */
int dcr_act_linear_fwd_f32_dev(const float *x_dev, const float *w_dev, float *h_train_dev, float *z_train_dev, float *z_eval_dev,
                               int64_t ldz, uint64_t *bits_dev, int64_t n_rows, int hidden, int classes, double p, uint64_t seed,
                               uint64_t offset, const uint64_t *offset_dev, void *hip_stream);  /* ldz: row stride of both z */
int dcr_act_linear_bwd_f32_dev(const float *dz_dev, const float *w_dev, const uint64_t *bits_dev, float *dx_dev, int64_t n_rows,
                               int hidden, int classes, double p, void *hip_stream);
/* The same pass, also returning colsum_dev[hidden] = the column sums of dx: the bias gradient of the layer that produced x
 * (models/gcn.py:36, `bias` of the previous GCNConv) without reading dx again.  ws_dev: dcr_act_linear_bwd_workspace floats;
 * the parts are added in a fixed order (deterministic). */
int dcr_act_linear_bwd_workspace(int64_t n_rows, int hidden, int64_t *floats);
int dcr_act_linear_bwd_colsum_f32_dev(const float *dz_dev, const float *w_dev, const uint64_t *bits_dev, float *dx_dev,
                                      float *colsum_dev, float *ws_dev, int64_t ws_floats, int64_t n_rows, int hidden, int classes,
                                      double p, void *hip_stream);
/* The first GCNConv (models/gcn.py:36, on a precomputed Â·X), the activation and dropout after it (models/gcn.py:38-42) and the
 * second GCNConv's lin in ONE kernel on the matrix cores (csrc/dcr_gcn_first.hip):
 *     pre = ax · W1ᵀ + b1   [n_rows x hidden]   (written when pre_dev != NULL: the backward pass reads it)
 *     z_train = dropout_p(relu(pre)) · W2ᵀ,  z_eval = relu(pre) · W2ᵀ   [n_rows x classes, row stride ldz]; either may be NULL
 * Keep bits, Philox stream and the order of operations of the second contraction are those of dcr_act_linear_fwd_f32_dev on
 * the same pre, so dcr_act_linear_bwd_fused_f32_dev(dz, W2, bits, pre, ...) is its backward.  b1_dev may be NULL.  Shapes:
 * dcr_first_layer_fits(in_features, hidden, classes) != 0: hidden 64 or 128, classes <= 16, any input width (round 5).  Two
 * kernels behind one entry point:
 *   - W1 resident in one CU's LDS (in_features a multiple of 16, hidden x in_features floats + W2 + b1 within 160 KB: the
 *     synthetic bench shape 256 -> 128): dcr_first_layer_fwd_workspace gives 0, ws_dev may be NULL;
 *   - K-chunked (everything else, the reference's own datasets among them — Cora 1,433 -> 128, Citeseer 3,703 -> 64,
 *     utils/hyperparams.py:2-21): W1 streamed through LDS in chunks of 16384 / hidden columns, a workgroup per (64 rows, chunk),
 *     partial tiles in ws_dev, the last workgroup of a row group to arrive adds them in chunk order and runs the epilogue.
 *     ws_dev: dcr_first_layer_fwd_workspace floats, 16-byte aligned; its trailing (n_rows / 64 rounded up) words are tickets:
 *     ZERO before the first use, left zero by every launch (one launch at a time per workspace).
 * ax_dev: n_rows x ldx, ldx >= in_features rounded up to a multiple of 16, the pad columns zero (W1's pad is zero inside the
 * kernel; 0 x NaN would still be NaN).  w1_dev: hidden x in_features, row stride in_features (any alignment for the K-chunked
 * kernel).  dcr_first_layer_fwd_f32_dev is the round-4 entry point: the same call without a workspace (resident shapes only). */
int dcr_first_layer_fits(int in_features, int hidden, int classes);
int dcr_first_layer_fwd_workspace(int64_t n_rows, int in_features, int hidden, int64_t *floats);
int dcr_first_layer_fwd_ws_f32_dev(const float *ax_dev, int64_t ldx, const float *w1_dev, const float *b1_dev, const float *w2_dev,
                                   float *pre_dev, float *z_train_dev, float *z_eval_dev, int64_t ldz, uint64_t *bits_dev,
                                   uint64_t *dwords_dev, int64_t n_rows, int in_features, int hidden, int classes, double p,
                                   uint64_t seed, uint64_t offset, const uint64_t *offset_dev, float *ws_dev, int64_t ws_floats,
                                   void *hip_stream);
/* The dropout decisions of one training call of the kernels above, drawn ahead of it (round 5): dwords_dev
 * [dcr_dropout_words_count words, 8-byte aligned] gets one bit per element of the n_rows x hidden activation, packed as
 * bits_dev is — 1 where the Philox stream of (seed, offset + *offset_dev) keeps the element (F.dropout's mask,
 * models/gcn.py:40, before the activation's sign is known) — followed by a stamp of the call it belongs to (offset + *offset_dev,
 * seed, p, n_rows).  Handed to dcr_first_layer_fwd_ws_f32_dev as dwords_dev, a call with that very stamp reads its decisions
 * there and gives bit for bit what it gives with dwords_dev == NULL; a call with any other stamp draws in line as if it had
 * been given none.  Every training call given a dwords_dev also leaves, in word [dcr_dropout_words_count - 4], its own
 * offset + 1: pointing the next dcr_dropout_words_dev's offset_dev there (offset 0) draws for the call expected next without
 * reading a counter that other work may be moving.  models/gcn.py draws the next epoch's words on a side stream while the
 * current epoch's aggregations run. */
int dcr_dropout_words_count(int64_t n_rows, int hidden, int64_t *words);
int dcr_dropout_words_dev(uint64_t *dwords_dev, int64_t n_rows, int hidden, double p, uint64_t seed, uint64_t offset,
                          const uint64_t *offset_dev, void *hip_stream);
int dcr_first_layer_fwd_f32_dev(const float *ax_dev, int64_t ldx, const float *w1_dev, const float *b1_dev, const float *w2_dev,
                                float *pre_dev, float *z_train_dev, float *z_eval_dev, int64_t ldz, uint64_t *bits_dev, int64_t n_rows,
                                int in_features, int hidden, int classes, double p, uint64_t seed, uint64_t offset,
                                const uint64_t *offset_dev, void *hip_stream);

/* Backward of dcr_first_layer_fwd_f32_dev's training operand in ONE kernel (the backward of models/gcn.py:36-42 for the first
 * two layers' dense parts): from dz = d loss / d z_train [n_rows x classes, contiguous] to
 *     dw1 [hidden x in_features] = dpreᵀ · ax,  db1 [hidden] = column sums of dpre,  dw2 [classes x hidden] = dzᵀ · dropout(relu(pre)),
 * with dpre = keep ? (dz · W2) / (1 - p) : 0 never written to memory (the first layer's input needs no gradient).  bits, pre:
 * as dcr_first_layer_fwd_f32_dev left them.  ws_dev: dcr_first_layer_bwd_workspace floats.  Any in_features (round 5: the true
 * width of W1 / dW1, row stride in_features; ax_dev as in the forward call, ldx >= in_features rounded up to 16), hidden 64 or
 * 128, classes <= 16.  The input width is tiled by 256 columns (grid y), the rows by about one workgroup per CU in total.
 * Partial results are added in a fixed order: deterministic. */
int dcr_first_layer_bwd_workspace(int64_t n_rows, int in_features, int hidden, int64_t *floats);
int dcr_first_layer_bwd_f32_dev(const float *dz_dev, const float *w2_dev, const uint64_t *bits_dev, const float *pre_dev,
                                const float *ax_dev, int64_t ldx, float *dw1_dev, float *db1_dev, float *dw2_dev, float *ws_dev,
                                int64_t ws_floats, int64_t n_rows, int in_features, int hidden, int classes, double p, void *hip_stream);

/* The whole backward of dcr_act_linear_fwd_f32_dev's training operand in one pass over x, on the matrix cores: dx as above,
 * colsum_dev[hidden] = its column sums, dw_dev[classes x hidden] = dz^T · h with h = keep ? x / (1 - p) : 0 rebuilt from x and
 * the keep bits (the forward call may then pass h_train_dev = NULL and store no activation).  Replaces the weight-gradient
 * reduction of the next layer's Linear (models/gcn.py:36, dW = dz^T · h) as well.  Deterministic (fixed-order sums). */
int dcr_act_linear_bwd_fused_workspace(int64_t n_rows, int hidden, int64_t *floats);
int dcr_act_linear_bwd_fused_f32_dev(const float *dz_dev, const float *w_dev, const uint64_t *bits_dev, const float *x_dev,
                                     float *dx_dev, float *dw_dev, float *colsum_dev, float *ws_dev, int64_t ws_floats,
                                     int64_t n_rows, int hidden, int classes, double p, void *hip_stream);

/* ---- loss and accuracy on the rows an epoch reads (device pointers, caller's stream) ----
 * experiment/training_loop.py:51 F.nll_loss(log_probs[mask], y[mask]) as out_loss[0] = -mean_i lp[i, y_i] over m rows of
 * log-probabilities (row stride ld), and its backward grad[i, c] = (c == y_i) ? -g[0] / m : 0 over the contiguous [m, classes]
 * gradient (what the stock kernels produce); :64-71 log_probs[mask].max(1)[1].eq(y[mask]).sum() as out_count[0] (first maximum
 * per row). */
int dcr_nll_picked_mean_fwd_f32_dev(const float *lp_dev, int64_t ld, const int64_t *y_dev, int64_t m, int classes, float *out_loss_dev,
                                    void *ws_dev, void *hip_stream);   /* ws_dev: as dcr_head_fwd_f32_dev's (may be the same buffer) */
/* Round 5: the head of an epoch in ONE kernel per direction, from the RAW outputs o = (Â·Z + b)[rows] of the last aggregation
 * (models/gcn.py:44 log_softmax + experiment/training_loop.py:51 F.nll_loss on the training rows; :64-71 arg-max accuracy on the
 * evaluated rows — the arg-max of log-probabilities is the arg-max of the logits):
 *     out_loss[0]    = mean_i (lse_i - o_train[i, y_i]),  lse_i = log sum_c exp o_train[i, c]
 *     out_correct[0] = #{i : first arg-max of o_eval[i, :] == y_eval[i]}
 * and backward  grad[i, c] = (exp(o_train[i, c] - lse_i) - [c == y_i]) * g[0] / m_train  (contiguous [m_train, classes]) with
 * grad_bias[c] = its column sums (the gradient of the last GCNConv's bias; may be NULL).  Either half of the forward call may be
 * absent (m = 0).  classes <= 32.  Labels must be class ids in [0, classes) (what F.nll_loss accepts without ignore_index).
 * ws_dev: dcr_head_workspace bytes, 8-byte aligned, ZERO before the first use; every launch leaves its tickets zero (one launch
 * at a time per workspace).  Sums in a fixed order: deterministic. */
/* Round 5: torch.optim.Adam's update (experiment/save_models.py:78-82; amsgrad off, L2 weight decay added to the gradient) for up
 * to 8 float32 tensors in ONE launch: params / grads / exp_avg / exp_avg_sq are host arrays of device pointers, numel and
 * weight_decay host arrays.  step_dev: the step count as one float in device memory (read, then advanced by one: a captured
 * epoch replays the increment); ticket_dev: one zero word, left zero. */
int dcr_adam_step_f32_dev(int n_tensors, void *const *params_dev, const void *const *grads_dev, void *const *exp_avg_dev,
                          void *const *exp_avg_sq_dev, const int64_t *numel, const float *weight_decay, double lr, double beta1,
                          double beta2, double eps, float *step_dev, uint32_t *ticket_dev, void *hip_stream);
int dcr_head_workspace(int64_t *bytes);
int dcr_head_fwd_f32_dev(const float *o_train_dev, int64_t ld_train, const int64_t *y_train_dev, int64_t m_train, const float *o_eval_dev,
                         int64_t ld_eval, const int64_t *y_eval_dev, int64_t m_eval, int classes, float *out_loss_dev,
                         int64_t *out_correct_dev, void *ws_dev, int64_t ws_bytes, void *hip_stream);
int dcr_head_bwd_f32_dev(const float *o_train_dev, int64_t ld_train, const int64_t *y_train_dev, int64_t m_train, int classes,
                         const float *g_dev, float *grad_dev, float *grad_bias_dev, void *ws_dev, int64_t ws_bytes, void *hip_stream);
int dcr_nll_picked_mean_bwd_f32_dev(const int64_t *y_dev, int64_t m, int classes, const float *g_dev, float *grad_dev, void *hip_stream);
int dcr_count_argmax_equal_f32_dev(const float *lp_dev, int64_t ld, const int64_t *y_dev, int64_t m, int classes, int64_t *out_count_dev,
                                   void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DCR_H */
