// Internal declarations shared by the translation units of libdcr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>

#include "dcr.h"

namespace dcr {

void set_error(const std::string &msg);

#define DCR_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::dcr::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + (strrchr(__FILE__, '/') ? strrchr(__FILE__, '/') + 1 : __FILE__) + ":" + std::to_string(__LINE__) + ")"); \
            return DCR_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define DCR_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != DCR_OK) return _rc; \
    } while (0)

#define DCR_FAIL(code, msg)       \
    do {                          \
        ::dcr::set_error(msg);    \
        return (code);            \
    } while (0)

// ---- curvature-pass work bins ----------------------------------------------
// An undirected edge (u,v) needs an LDS hash set over N(u) ∪ N(v) ∪ {u,v}; the bin is chosen by
// that size so the table keeps a load factor <= 1/4 (1/2 in the last bin).  Threads per edge ("team") grow with the bin.
constexpr int NBINS = 5;
constexpr int GIANT_LIST_CAP = 65536;  // edges beyond every LDS table that one pass can list
constexpr int BIN_SLOTS[NBINS] = {128, 512, 2048, 4096, 32768};
constexpr int BIN_TEAM[NBINS] = {64, 64, 256, 512, 1024};
constexpr int MAX_TABLE_KEYS = BIN_SLOTS[NBINS - 1] / 2;  // du + dv + 2 must not exceed this
constexpr int32_t MAX_NODES = (1 << 30) - 2;              // two tag bits live above the key

// small result block mirrored in pinned host memory
struct DevResult {
    double ext_val;       // argext value
    int32_t ext_slot;     // argext slot (-1: none)
    int32_t ext_u, ext_v;
    int32_t ext_du, ext_dv;  // their degrees (saves dcr_improvements a round trip)
    int32_t add_status;   // 0 ok, 1 row overflow (nothing changed), 2 already present, 3 no edge drawn on the device (nothing changed)
    int32_t removed_u, removed_v;
    int32_t overflow_row;
    int32_t max_keys;     // largest du+dv+2 seen by classify
    int32_t work_count[NBINS];
    int32_t work_next[NBINS];  // dequeue cursors of the persistent bin kernels
    int32_t nc_count[5];       // node-centric pass: units per degree class
    int32_t nc_next[5];        // dequeue cursors of its kernels
    int32_t nc_bucket[16];     // units per degree bucket (plan phase 0)
    int32_t nc_fill[16];       // placement cursors per bucket (plan phase 1)
    int32_t flag_too_big; // an edge exceeded MAX_TABLE_KEYS and could not be put on the giant list either
    int32_t giant_count;  // edges beyond every LDS table, listed for the device-memory path (dcr_bfc_giant.hip)
    int32_t hub_count;    // nodes with more neighbours than the largest table (k_find_hubs)
    int64_t n_cand;
    int64_t imp_argmax;
    int32_t cand_i, cand_j;
    // device-side draw (dcr_sdrf_iteration_device_draw): the index np.random.choice would return, and whether the margins
    // that make it certain held (0 ok; 1 undecided / not finite: the host draws; 2 no candidates)
    int64_t draw_idx;
    double draw_total, draw_gap;
    int32_t draw_status;
    int32_t draw_pad;
    int32_t misc[8];  // scratch: has_edge/remove status, and the kernels' invariant guard record
    // two-hop pass (dcr_bfc_h2.hip): units per (class, weight bucket), placement cursors, units per class
    int32_t h2_bucket[24];
    int32_t h2_fill[24];
    int32_t h2_count[5];
    int32_t h2_failed[6];  // diagnostic: nodes whose tables filled up, per class (5: on the retry list itself)
    // triangle step, one pool per block class (0: the split class and the retry launch, 1: class M), so that the candidates of
    // a class can be taken as soon as THAT class is done: listed edges, candidates, partners; candidates the first
    // k_h2_triangles launch of the pool has taken
    int32_t h2_ntask[2], h2_ncand[2], h2_npart[2];
    int32_t h2_ncand_done[2];
    // Edit journal (round 4): every edge the device adds or removes since the two-hop pass last brought its edge set up to
    // date: {+1 / -1, u, v} per edit, in order; edit_n keeps counting past the capacity (the host then knows the bound on
    // its own and rebuilds the set).  Written by dev_add_edge / dev_remove_edge, consumed by k_h2_eset_apply.
    int32_t edit_n;
    int32_t edit_log[3 * 8];
    int32_t touched_n;  // nodes on the graph's touched list (every node whose dirty byte went from zero to non-zero since the flags
                        // were last cleared: dev_mark_dirty appends, whoever clears the flags resets this)
    int32_t h2_retry;   // units on the retry list (nodes whose tables filled up in their class, redone by the largest class)
    int32_t h2_status;  // 0 ok; 1: a table filled up or a list overflowed (the pass is then redone by the node-centric kernels);
                        // 2: the triangle step's pools were too small (run again with larger ones); 3: the pass was launched
                        // without its retry stage and some node needed it (run again with the stage)
};

// Incremental pass: which edges an edit can have changed.  BFC(a,b) is a function of deg a, deg b, N(a) ∩ N(b) and the
// adjacency between DX = N(a) \ N(b) and DY = N(b) \ N(a).  Adding or removing the edge {x,l} changes the first three
// only for edges incident to x or l, and that adjacency only for the pair {x,l} itself, i.e. only for edges {a,b} with
// a in N(x) and b in N(l) (or the other way round).  Per node, one byte: DIRTY_ENDPOINT on x and l, bit A_e on the
// members of N(x) and B_e on those of N(l) for the e-th edit since the last pass (three edits fit; dcr_sdrf_tail makes
// two); further edits fall back to DIRTY_COARSE on {x,l} ∪ N(x) ∪ N(l) (every edge with a flagged endpoint).
constexpr int EDIT_LOG_CAP = 8;
constexpr int EXT_PART_BLOCKS = 16384;  // capacity of the per-block extrema arrays
__device__ inline void journal_edit(DevResult *res, int op, int32_t u, int32_t v) {  // one thread of a single-workgroup edit kernel
    const int i = res->edit_n;
    if (i >= 0 && i < EDIT_LOG_CAP) {
        res->edit_log[3 * i] = op;
        res->edit_log[3 * i + 1] = u;
        res->edit_log[3 * i + 2] = v;
    }
    res->edit_n = i + 1;
}
constexpr unsigned DIRTY_COARSE = 0x80u, DIRTY_ENDPOINT = 0x40u;
constexpr int DIRTY_EDITS = 3;
__device__ __host__ inline bool edge_dirty(unsigned du, unsigned dv) {
    if ((du | dv) & (DIRTY_COARSE | DIRTY_ENDPOINT)) return true;
    return ((((du >> 1) & dv) | ((dv >> 1) & du)) & 0x15u) != 0u;  // some edit has its A bit on one side, its B bit on the other
}

// ---- (value, slot) extrema: first in G.edges order (= smallest slot) on ties (sdrf_no_cuda.py:27,59,61) ----
struct Ext {  // how a partial result is stored; the running pair lives in two plain registers (ext_take): passed around as a
              // struct it ended up in scratch memory (88-117 scratch accesses per kernel, round 4)
    double val;
    int32_t slot;
    int32_t pad;
};

// (bv, bs) <- the better of (bv, bs) and (v, sl); a slot < 0 means "none"
__device__ inline void ext_take(double &bv, int &bs, double v, int sl, int want_max) {
    const bool better = bs < 0 || (want_max ? v > bv : v < bv) || (v == bv && sl < bs);
    if (sl >= 0 && better) {
        bv = v;
        bs = sl;
    }
}

__device__ inline void ext_wave_reduce(double &bv, int &bs, int want_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int os = __shfl_xor(bs, off);
        ext_take(bv, bs, ov, os, want_max);
    }
}

// shv / shs: one entry per wave of the workgroup
__device__ inline void ext_block_reduce(double &bv, int &bs, int want_max, double *shv, int *shs) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    ext_wave_reduce(bv, bs, want_max);
    if (lane == 0) {
        shv[wid] = bv;
        shs[wid] = bs;
    }
    __syncthreads();
    if (wid == 0) {
        double tv = lane < nw ? shv[lane] : 0.0;
        int ts = lane < nw ? shs[lane] : -1;
        ext_wave_reduce(tv, ts, want_max);
        if (lane == 0) {
            shv[0] = tv;
            shs[0] = ts;
        }
    }
    __syncthreads();
    bv = shv[0];
    bs = shs[0];
}
__device__ inline Ext ext_make(double v, int sl) {
    Ext e;
    e.val = v;
    e.slot = sl;
    e.pad = 0;
    return e;
}

struct ImpStats {  // per (x,y) statistics for the improvement kernels; lives in device memory
    int32_t x, y, dx, dy;
    int32_t T, s1, s2;
    int32_t max1, cnt1, max2, cnt2, sec1, sec2;  // maxima of c1/c2, their multiplicity, runner-up
    int32_t table_mask;
    int32_t deg_min_is_one;
    int32_t pos_x_in_y;  // position of x inside row y
    int32_t done_rows;   // workgroups of k_imp_rows_count that are through (the last one closes the stage: statistics, scan)
    double before;
    int32_t done_draw;   // the same for k_draw_partial (the last one picks)
    int32_t pad_;
};

}  // namespace dcr

struct dcr_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0;
    int64_t n_edges = 0;    // undirected, tracked on the host
    int64_t cap_total = 0;  // adjacency slots allocated

    // HBM-resident graph: rows in insertion order, slack-padded
    int2 *rowinfo = nullptr;     // [n] {start slot, degree}
    int32_t *rowcap = nullptr;   // [n] capacity of each row
    int32_t *col = nullptr;      // [cap_total] neighbour ids
    int32_t *slot_row = nullptr; // [cap_total] owning row of each slot
    double *curv = nullptr;      // [cap_total] curvature of the undirected edge stored at slot (col > row)

    // degrees of the last arg-min edge as seen by the device; dropped by any edit
    int32_t am_x = -1, am_y = -1, am_dx = 0, am_dy = 0;
    bool am_valid = false;
    bool amax_valid = false;  // the result block holds the stale arg-max of the current graph (computed ahead of the tail)

    int curv_type_last = -1;
    bool curv_valid = false;
    uint8_t *dirty = nullptr;    // [n] node flags: an incident edge was added/removed at this node or a neighbour
    bool dirty_tracked = false;  // flags cover every edit since the last pass
    int pending_edits = 0;       // edits flagged since the last pass (the first DIRTY_EDITS get exact flags)

    // node-centric pass (dcr_bfc_nc.hip): unit lists per degree class
    int2 *nc_units[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // {node, first sub-unit}
    int64_t nc_cap[5] = {0, 0, 0, 0, 0};
    long long *nc_trace = nullptr;  // DCR_NC_TRACE diagnostic: [5 classes][16384 waves][2]
    int32_t *nc_queues = nullptr;  // dequeue cursors of the two wave-class kernels, one cache line each
    uint8_t *nc_touch = nullptr;  // [n] incremental pass: node has a flagged neighbour
    int64_t nc_touch_cap = 0;
    int2 *nc_fine_list = nullptr;  // incremental pass behind a few exactly flagged edits: the edges to recompute {owner, position}
    int64_t nc_fine_cap = 0;
    int32_t *touched = nullptr;    // [n + 64] the flagged nodes, each once (DevResult::touched_n of them)
    double sum_deg2 = 0.0;        // sum of squared degrees when the graph was created (engine choice: size of the 2-hop neighbourhoods)
    int32_t max_deg_bound = 0;    // host-side upper bound on the largest degree (exact after create / relayout)
    int pass_impl = 0;            // 0: automatic (default: two-hop kernels for full Balanced Forman passes of graphs large enough to
                                  // pay for their fixed cost, node-centric otherwise); 1: edge-centric kernels only (DCR_PASS=edge);
                                  // 2: node-centric (DCR_PASS=nc); 3: two-hop
                                  // kernels for full Balanced Forman passes, node-centric otherwise (DCR_PASS=h2)
    int last_engine = -1;         // which implementation ran the last pass: 0 two-hop, 1 edge-centric, 2 node-centric

    // two-hop pass (dcr_bfc_h2.hip)
    int32_t *h2_weight = nullptr;     // [n] sum of the neighbours' degrees
    int4 *h2_units[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // per class {node, partitions << 16 | partition, row start, degree}
    int64_t h2_units_cap[5] = {0, 0, 0, 0, 0};
    uint4 *h2_task = nullptr;         // triangle step pools (dcr_bfc_h2.hip)
    int4 *h2_cand = nullptr;
    int32_t *h2_part = nullptr;
    unsigned *h2_lists = nullptr;     // per-wave lists of the block classes' third step (dcr_bfc_h2.hip: H2List)
    int64_t h2_lists_cap = 0;
    int64_t h2_task_cap = 0, h2_cand_cap = 0, h2_part_cap = 0;
    int64_t h2_want[3] = {0, 0, 0};   // pool sizes a pass asked for (tasks, candidates, partners)
    unsigned *h2_bloom = nullptr;     // one bit per edge (prefilter of the edge set)
    int h2_bloom_bits = 0;
    int4 *h2_retry = nullptr;         // units of nodes whose tables filled up in their class (redone by the largest class)
    int64_t h2_retry_cap = 0;
    uint4 *h2_rec = nullptr;          // [cap_total] per directed slot {|sq| on the far side, max count, triangles, reverse slot}
    int64_t h2_rec_cap = 0;
    int64_t h2_weight_cap = 0;
    unsigned long long *h2_eset = nullptr;  // every undirected edge as one 64-bit key (open addressing), rebuilt per pass
    int h2_eset_bits = 0;
    bool h2_expect_retry = false;      // some pass of this graph had nodes on the retry list: the retry stage is launched with every pass
    bool h2_cleared_dirty = false;     // the last two-hop launch zeroed the incremental pass's node flags itself
    bool h2_eset_valid = false;        // the set holds the graph's edges as of the last two-hop pass; later edits are in the journal
    int h2_eset_pending = 0;           // upper bound of the edits journaled since (host-side count of the edit launches)
    int64_t h2_eset_tombs = 0;         // upper bound of the tombstones in the set (removals applied since the last rebuild)
    int32_t h2_last_count[5] = {-1, -1, -1, -1, -1};  // units per class of the previous pass (sizes the next grids)

    // edges beyond every LDS table (dcr_bfc_giant.hip): records {slot, u, v, deg u, deg v}; position map over all ids
    int32_t *giant_list = nullptr;
    int32_t *giant_pos = nullptr;
    uint32_t *giant_cnt = nullptr;
    int64_t giant_cnt_cap = 0;
    int32_t *giant_acc = nullptr;
    int32_t *hub_list = nullptr;   // {node, degree} pairs of the nodes above every table size
    int64_t hub_list_cap = 0;
    uint32_t *hub_cnt = nullptr;   // per-wave slot counters of k_hub_edges (kept all-zero between edges)
    int64_t hub_cnt_cap = 0;

    // curvature-pass work lists (edge-centric kernels)
    int32_t *work[dcr::NBINS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t work_cap = 0;

    // reductions / scans
    void *red_scratch = nullptr;  // argext partials
    // per-block (value, slot) minima and maxima left by the two-hop pass's closing kernel (k_h2_final): the first minimum
    // after the pass and the stale first maximum of the removal step then need no sweep of their own.  Valid until the next
    // edit (a removal moves values between slots) or pass.
    void *ext_part = nullptr;     // Ext[2][EXT_PART_BLOCKS]: minima, then maxima
    int ext_part_n = 0;
    bool ext_part_valid = false;
    int32_t *scan_a = nullptr, *scan_b = nullptr;
    int64_t scan_cap = 0;

    // improvement pipeline scratch (grown on demand)
    int32_t *imp_table = nullptr;  // hash keys
    int32_t *imp_posx = nullptr, *imp_posy = nullptr;
    int64_t imp_table_cap = 0;
    bool imp_table_dirty = true;   // the table is not all-empty (fresh allocation, or a pipeline that did not reach its last kernel)
    int32_t *imp_c1 = nullptr, *imp_c2 = nullptr;  // per position in row x / row y
    double *imp_b = nullptr, *imp_c = nullptr;     // class B / C improvements per position
    int32_t *imp_rowcount = nullptr, *imp_rowoff = nullptr;
    uint32_t *imp_adjbits = nullptr;               // (dx+1) x words(dy+1)
    int64_t imp_rows_cap = 0, imp_bits_cap = 0;
    double *imp_out = nullptr;                     // compacted improvements (device)
    int32_t *imp_ci = nullptr, *imp_cj = nullptr;
    int64_t imp_out_cap = 0;
    double *imp_out_h = nullptr;                   // pinned host mirrors
    int32_t *imp_ci_h = nullptr, *imp_cj_h = nullptr;
    int64_t imp_out_h_cap = 0, imp_cand_h_cap = 0;
    int64_t imp_n = 0;
    dcr::ImpStats *imp_stats = nullptr;
    double *draw_bsum = nullptr;  // device-side draw: partial sums of exp(tau * improvement)

    dcr::DevResult *dres = nullptr;  // device
    dcr::DevResult *hres = nullptr;  // pinned host

    // side streams: the work bins of a curvature pass run concurrently
    hipStream_t side[dcr::NBINS - 1] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    hipStream_t aux = nullptr;   // two-hop pass: the edge set is rebuilt here while the plan runs
    hipEvent_t ev_aux = nullptr;
    hipStream_t low[2] = {nullptr, nullptr};  // low priority (with side[2]): the fine-grained kernels of the two-hop pass
    hipEvent_t ev_aux2 = nullptr;
    hipEvent_t ev_join[dcr::NBINS - 1] = {nullptr, nullptr, nullptr, nullptr};
    int num_cu = 0;

    // profiling
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double pass_ms_total = 0.0;
    int64_t pass_count = 0;
    bool profile = false;
};

namespace dcr {

// dcr_graph.hip
int device_exclusive_scan(dcr_graph *g, const int32_t *in, int32_t *out, int64_t n, int64_t *total_out);
int ensure_scan(dcr_graph *g, int64_t n);
int relayout(dcr_graph *g);
int sync_result(dcr_graph *g);  // D2H of DevResult + stream sync
void launch_add_edge(dcr_graph *g, int32_t u, int32_t v);          // u < 0: no-op that clears add_status
void launch_remove_if_above(dcr_graph *g, double bound, int edit);  // acts on the last argext result
// add + dirty flags + conditional removal of the arg-max already in the result block, one launch (dcr_graph.hip: k_sdrf_tail)
void launch_sdrf_tail(dcr_graph *g, int32_t u, int32_t v, int edit_add, int do_remove, double bound, int edit_rem);
void launch_mark_dirty(dcr_graph *g, int32_t u, int32_t v, int edit);  // flag the edges edit number `edit` can change (>= 3: coarse)

// dcr_sdrf.hip
int launch_argext(dcr_graph *g, int want_max, int excl_u, int excl_v, hipStream_t st = nullptr);  // st: default the library stream
// the same from the per-block extrema of the last two-hop pass (g->ext_part_valid), without sweeping the edges again
int launch_argext_from_parts(dcr_graph *g, int want_max, hipStream_t st = nullptr);
int launch_argext_both(dcr_graph *g, hipStream_t st = nullptr, bool clear_dirty = false);
double nc_class_full_ms(const dcr_graph *g);   // estimates of a full pass (csrc/dcr_bfc_nc.hip): class kernels / a workgroup per edge
double nc_edges_full_ms(const dcr_graph *g);
// reductions of per-workgroup partial results shared by the GCN kernels (csrc/dcr_gemm.hip, csrc/dcr_gcn.hip)
void launch_slab_reduce(const float *part, float *C, int64_t mn, int N, int64_t ldc, int splits, hipStream_t st);
void launch_slab_reduce_cols(const float *part, float *C, int64_t mn, int N, int ncols, int64_t ldc, int splits, hipStream_t st);
void launch_parts_finish(const float *part, int64_t n_parts, int stride, int split, int columns, float *out0, float *out1, hipStream_t st);
int process_giant_edges(dcr_graph *g, int curv_type);  // dcr_bfc_giant.hip; syncs once
int process_hub_edges(dcr_graph *g, int curv_type, bool incremental);  // dcr_bfc_giant.hip; syncs once
int giant_edge(dcr_graph *g, int u, int v, int du, int dv, int64_t slot, int curv_type, bool need_cycles, int64_t *d_out6);  // result in DevResult after the next sync

// dcr_bfc.hip
int launch_curvature_pass(dcr_graph *g, int curv_type, bool incremental);
// dcr_bfc_nc.hip
int launch_curvature_pass_nc(dcr_graph *g, int curv_type, bool incremental);
// dcr_bfc_h2.hip
bool h2_can_take(const dcr_graph *g, int curv_type, bool incremental);
int launch_curvature_pass_h2(dcr_graph *g);
bool h2_grow_pools(dcr_graph *g);

template <typename T>
int dev_alloc(T **p, int64_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)(count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return DCR_ENOMEM;
    }
    *p = (T *)q;
    return DCR_OK;
}

template <typename T>
int dev_regrow(T **p, int64_t *cap, int64_t need) {
    if (need <= *cap) return DCR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    int64_t nc = need + need / 4 + 64;
    DCR_TRY(dev_alloc(p, nc));
    *cap = nc;
    return DCR_OK;
}

}  // namespace dcr
