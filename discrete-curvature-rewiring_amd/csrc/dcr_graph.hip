// Graph container of libdcr_hip.so: insertion-ordered, slack-padded adjacency rows in HBM.
//
// Replaces the networkx.Graph the reference builds and mutates at
// rewiring/sdrf_no_cuda.py:20,35,43,46,51,63,68:
//   * rows keep insertion order; add appends at the end of both rows; remove deletes in place
//     (shift-left), so "position in row" reproduces dict order;
//   * G.edges order == increasing slot index over slots whose neighbour id exceeds the row id.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_set>

#include "dcr_internal.h"

namespace dcr {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// Events that order the streams of a pass against each other.  A device-scope release when recorded (hipEventReleaseToDevice,
// -DDCR_EVENT_DEVICE_SCOPE) instead of the default system-scope fence was measured: the gap between the split class and its
// triangle step shrank from 50 to 6 us on the timeline, and the pass got 4 % SLOWER (1.053 against 1.004 ms on S100k, 10.75
// against 10.44 on S1M, four interleaved rounds: profiles/r04_event_scope_ab.txt).  The default stays.
#ifdef DCR_EVENT_DEVICE_SCOPE
#define DCR_EVENT_FLAGS (hipEventDisableTiming | hipEventReleaseToDevice)
#else
#define DCR_EVENT_FLAGS (hipEventDisableTiming)
#endif

static inline int32_t slack_for(int32_t deg) {
    int32_t s = deg / 4;
    return s < 8 ? 8 : s;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ inline int wave_find(const int32_t *row, int deg, int32_t key, int lane) {
    // position of key in row[0..deg) or -1; all 64 lanes of one wave cooperate
    int found = -1;
    for (int base = 0; base < deg; base += 64) {
        int i = base + lane;
        bool hit = (i < deg) && (row[i] == key);
        unsigned long long m = __ballot(hit);
        if (m) {
            found = base + __ffsll((long long)m) - 1;
            break;
        }
    }
    return found;
}

__device__ inline void wave_erase(int32_t *row, double *crow, int deg, int pos, int lane) {
    // delete row[pos], shifting the tail left by one (keeps insertion order).  The slot-keyed curvature values move
    // with their edges: the buffer of the last pass stays readable per edge after a removal, like the reference's
    // edge-keyed curv_dict (sdrf_no_cuda.py:57-61).
    for (int base = pos; base < deg - 1; base += 64) {
        int i = base + lane;
        int32_t t = 0;
        double c = 0.0;
        if (i < deg - 1) {
            t = row[i + 1];
            c = crow[i + 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (i < deg - 1) {
            row[i] = t;
            crow[i] = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ void dev_remove_edge(int2 *rowinfo, int32_t *col, double *curv, int32_t u, int32_t v, int lane,
                                int *status, DevResult *res) {
    int2 ru = rowinfo[u], rv = rowinfo[v];
    int pu = wave_find(col + ru.x, ru.y, v, lane);
    int pv = wave_find(col + rv.x, rv.y, u, lane);
    if (pu < 0 || pv < 0) {
        if (lane == 0) *status = 1;
        return;
    }
    wave_erase(col + ru.x, curv + ru.x, ru.y, pu, lane);
    wave_erase(col + rv.x, curv + rv.x, rv.y, pv, lane);
    if (lane == 0) {
        rowinfo[u] = make_int2(ru.x, ru.y - 1);
        rowinfo[v] = make_int2(rv.x, rv.y - 1);
        *status = 0;
        journal_edit(res, -1, u, v);
    }
}

__device__ inline void dev_add_edge(int2 *rowinfo, const int32_t *rowcap, int32_t *col, int32_t u, int32_t v, DevResult *res,
                                    int lane) {  // one wave; lane = its lane
    if (u == -2) {  // the pair picked on the device (dcr_sdrf_tail_at)
        if (res->draw_status != 0) {  // the device-side draw did not decide: nothing is edited, the host takes over
            if (lane == 0) res->add_status = 3;
            return;
        }
        u = res->cand_i;
        v = res->cand_j;
    }
    if (u < 0) {  // nothing to add (no candidates this iteration)
        if (lane == 0) res->add_status = 0;
        return;
    }
    int2 ru = rowinfo[u], rv = rowinfo[v];
    int p = (ru.y <= rv.y) ? wave_find(col + ru.x, ru.y, v, lane) : wave_find(col + rv.x, rv.y, u, lane);
    if (lane != 0) return;
    if (p >= 0) {
        res->add_status = 2;
        return;
    }
    if (ru.y >= rowcap[u] || rv.y >= rowcap[v]) {
        res->add_status = 1;
        res->overflow_row = (ru.y >= rowcap[u]) ? u : v;
        return;
    }
    col[ru.x + ru.y] = v;
    col[rv.x + rv.y] = u;
    rowinfo[u] = make_int2(ru.x, ru.y + 1);
    rowinfo[v] = make_int2(rv.x, rv.y + 1);
    res->add_status = 0;
    journal_edit(res, +1, u, v);
}

__global__ void __launch_bounds__(64) k_add_edge(int2 *rowinfo, const int32_t *rowcap, int32_t *col, int32_t u,
                                                 int32_t v, DevResult *res) {
    dev_add_edge(rowinfo, rowcap, col, u, v, res, threadIdx.x);
}

__global__ void __launch_bounds__(64) k_remove_edge(int2 *rowinfo, int32_t *col, double *curv, int32_t u, int32_t v,
                                                    DevResult *res) {
    int st = 0;
    dev_remove_edge(rowinfo, col, curv, u, v, threadIdx.x, &st, res);
    if (threadIdx.x == 0) res->misc[0] = st;
}

// tail of an SDRF iteration: remove the arg-max edge iff its (stale) curvature exceeds the bound
__global__ void __launch_bounds__(64) k_remove_if_above(int2 *rowinfo, int32_t *col, double *curv, DevResult *res,
                                                        double bound) {
    if (res->add_status == 1 || res->add_status == 3) return;  // add overflowed (host re-lays out and replays the tail) / no draw
    int lane = threadIdx.x;
    int32_t u = res->ext_u, v = res->ext_v;
    bool doit = (res->ext_slot >= 0) && (res->ext_val > bound);
    if (!doit) {
        if (lane == 0) {
            res->removed_u = -1;
            res->removed_v = -1;
        }
        return;
    }
    int st = 0;
    dev_remove_edge(rowinfo, col, curv, u, v, lane, &st, res);
    if (lane == 0) {
        res->removed_u = u;
        res->removed_v = v;
    }
}

// Incremental curvature: the per-node flags behind edge_dirty() (dcr_bfc_common.h).  Edit number `edit` (0..2) of the
// edge {u,v}: endpoint flag on u and v, bit A on the members of N(u), bit B on those of N(v); later edits: the coarse
// flag on all of them.  A node can be in both rows, so the bytes are OR-ed atomically (through their 32-bit words).
// (round 5) ... and a node whose byte goes from zero to non-zero is put on the touched list: the incremental pass lists its edges
// from the rows of those nodes instead of sweeping every adjacency slot for flags (0.14 ms of a 0.53 ms iteration at 1M nodes)
__device__ inline void dirty_or_touch(uint8_t *dirty, int k, unsigned bits, int32_t *touched, int32_t cap, DevResult *res) {
    const unsigned sh = (unsigned)(k & 3) * 8u;
    const unsigned old = atomicOr(reinterpret_cast<unsigned *>(dirty) + (k >> 2), bits << sh);
    if (((old >> sh) & 0xFFu) == 0u) {
        const int idx = atomicAdd(&res->touched_n, 1);
        if (idx >= 0 && idx < cap) touched[idx] = k;   // (each node at most once between two clears: never more than n)
    }
}
__device__ inline void dev_mark_dirty(const int2 *rowinfo, const int32_t *col, uint8_t *dirty, int32_t u, int32_t v,
                                      int edit, int tid, int nthreads, int32_t *touched, int32_t tcap, DevResult *tres) {
    if (u < 0 || v < 0) return;
    const bool exact = edit >= 0 && edit < dcr::DIRTY_EDITS;
    const unsigned bit_u = exact ? 1u << (2 * edit) : dcr::DIRTY_COARSE, bit_v = exact ? 2u << (2 * edit) : dcr::DIRTY_COARSE;
    const unsigned bit_end = exact ? dcr::DIRTY_ENDPOINT : dcr::DIRTY_COARSE;
    const int2 ru = rowinfo[u], rv = rowinfo[v];
    for (int i = tid; i < ru.y; i += nthreads) dirty_or_touch(dirty, col[ru.x + i], bit_u, touched, tcap, tres);
    for (int i = tid; i < rv.y; i += nthreads) dirty_or_touch(dirty, col[rv.x + i], bit_v, touched, tcap, tres);
    if (tid == 0) {
        dirty_or_touch(dirty, u, bit_end, touched, tcap, tres);
        dirty_or_touch(dirty, v, bit_end, touched, tcap, tres);
    }
}

__global__ void __launch_bounds__(256) k_mark_dirty(const int2 *rowinfo, const int32_t *col, uint8_t *dirty, int32_t u,
                                                     int32_t v, int edit, DevResult *res, int32_t *touched, int32_t tcap) {
    if (u == -2) {
        if (res->draw_status != 0) return;
        u = res->cand_i;
        v = res->cand_j;
    }
    dev_mark_dirty(rowinfo, col, dirty, u, v, edit, threadIdx.x, blockDim.x, touched, tcap, res);
}

// the edge the tail is about to remove (it is only known on the device)
__global__ void __launch_bounds__(256) k_mark_dirty_ext(const int2 *rowinfo, const int32_t *col, uint8_t *dirty,
                                                         DevResult *res, double bound, int edit, int32_t *touched, int32_t tcap) {
    if (res->add_status == 1 || res->add_status == 3) return;
    if (res->ext_slot >= 0 && res->ext_val > bound)
        dev_mark_dirty(rowinfo, col, dirty, res->ext_u, res->ext_v, edit, threadIdx.x, blockDim.x, touched, tcap, res);
}

// The whole tail of an SDRF iteration as ONE launch when the stale arg-max is already in the result block (sdrf_no_cuda.py:51,
// 56-66): add (u, v), flag what the add can change, flag what the removal can change while the edge is still there, remove the
// arg-max edge iff above the bound.  Four launches of one or four waves each cost their launch gaps, not their work.
__global__ void __launch_bounds__(256) k_sdrf_tail(int2 *rowinfo, const int32_t *rowcap, int32_t *col, double *curv, uint8_t *dirty,
                                                    DevResult *res, int32_t u, int32_t v, int edit_add, int do_remove, double bound,
                                                    int edit_rem, int32_t *touched, int32_t tcap) {
    const int tid = threadIdx.x;
    if (tid < 64) dev_add_edge(rowinfo, rowcap, col, u, v, res, tid);
    __threadfence_block();   // (one workgroup: its waves share the CU's cache, nothing has to reach the L2 before the barrier;
    __syncthreads();         //  the agent-scope fences that stood here wrote the L2 back twice per iteration)
    int32_t au = u, av = v;
    bool mark = true;
    if (u == -2) {
        mark = res->draw_status == 0;
        au = res->cand_i;
        av = res->cand_j;
    }
    if (mark) dev_mark_dirty(rowinfo, col, dirty, au, av, edit_add, tid, 256, touched, tcap, res);
    if (!do_remove) return;
    const int st_add = res->add_status;
    if (st_add == 1 || st_add == 3) return;  // (uniform) add overflowed: replayed after a re-layout / no draw
    const bool doit = res->ext_slot >= 0 && res->ext_val > bound;
    if (doit) dev_mark_dirty(rowinfo, col, dirty, res->ext_u, res->ext_v, edit_rem, tid, 256, touched, tcap, res);
    __threadfence_block();
    __syncthreads();
    if (tid >= 64) return;
    if (!doit) {
        if (tid == 0) {
            res->removed_u = -1;
            res->removed_v = -1;
        }
        return;
    }
    int st = 0;
    dev_remove_edge(rowinfo, col, curv, res->ext_u, res->ext_v, tid, &st, res);
    if (tid == 0) {
        res->removed_u = res->ext_u;
        res->removed_v = res->ext_v;
    }
}

__global__ void __launch_bounds__(64) k_has_edge(const int2 *rowinfo, const int32_t *col, int32_t u, int32_t v,
                                                 DevResult *res) {
    int2 ru = rowinfo[u], rv = rowinfo[v];
    int p = (ru.y <= rv.y) ? wave_find(col + ru.x, ru.y, v, threadIdx.x) : wave_find(col + rv.x, rv.y, u, threadIdx.x);
    if (threadIdx.x == 0) {
        res->misc[0] = (p >= 0);
        res->misc[1] = ru.y;
        res->misc[2] = rv.y;
    }
}

__global__ void k_relayout(const int2 *old_info, const int32_t *old_col, const double *old_curv, const int32_t *new_start,
                           int2 *new_info, int32_t *new_col, double *new_curv, int32_t *new_slot_row,
                           const int32_t *new_cap, int64_t n) {
    // one wave per row
    int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    int lane = threadIdx.x & 63;
    if (row >= n) return;
    int2 o = old_info[row];
    int ns = new_start[row];
    int nc = new_cap[row];
    for (int i = lane; i < nc; i += 64) {
        new_slot_row[ns + i] = (int32_t)row;
        if (i < o.y) {
            new_col[ns + i] = old_col[o.x + i];
            new_curv[ns + i] = old_curv[o.x + i];
        }
    }
    if (lane == 0) new_info[row] = make_int2(ns, o.y);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void launch_add_edge(dcr_graph *g, int32_t u, int32_t v) {
    g->h2_eset_pending += 1;
    g->ext_part_valid = false;
    hipLaunchKernelGGL(k_add_edge, dim3(1), dim3(64), 0, g->stream, g->rowinfo, g->rowcap, g->col, u, v, g->dres);
}

void launch_sdrf_tail(dcr_graph *g, int32_t u, int32_t v, int edit_add, int do_remove, double bound, int edit_rem) {
    g->h2_eset_pending += 2;
    g->ext_part_valid = false;
    hipLaunchKernelGGL(k_sdrf_tail, dim3(1), dim3(256), 0, g->stream, g->rowinfo, g->rowcap, g->col, g->curv, g->dirty, g->dres, u, v,
                       edit_add, do_remove, bound, edit_rem, g->touched, (int32_t)g->n);
}

void launch_remove_if_above(dcr_graph *g, double bound, int edit) {
    g->h2_eset_pending += 1;
    g->ext_part_valid = false;
    // flag the neighbourhood while the edge is still there, then remove it
    hipLaunchKernelGGL(k_mark_dirty_ext, dim3(1), dim3(256), 0, g->stream, g->rowinfo, g->col, g->dirty, g->dres, bound,
                       edit, g->touched, (int32_t)g->n);
    hipLaunchKernelGGL(k_remove_if_above, dim3(1), dim3(64), 0, g->stream, g->rowinfo, g->col, g->curv, g->dres, bound);
}

void launch_mark_dirty(dcr_graph *g, int32_t u, int32_t v, int edit) {
    if (u != -2 && (u < 0 || v < 0)) return;
    hipLaunchKernelGGL(k_mark_dirty, dim3(1), dim3(256), 0, g->stream, g->rowinfo, g->col, g->dirty, u, v, edit, g->dres, g->touched,
                       (int32_t)g->n);
}

int sync_result(dcr_graph *g) {
    DCR_HIP(hipMemcpyAsync(g->hres, g->dres, sizeof(DevResult), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    return DCR_OK;
}

static int alloc_layout(dcr_graph *g, int64_t cap_total) {
    DCR_TRY(dev_alloc(&g->col, cap_total + 64));  // padded: rows are read in aligned 16-byte pieces
    DCR_TRY(dev_alloc(&g->slot_row, cap_total));
    DCR_TRY(dev_alloc(&g->curv, cap_total));
    g->cap_total = cap_total;
    return DCR_OK;
}

// Re-pack every row with fresh slack (called when an append finds its row full).
// Row order, in-row order and the (stale) curvature values keep their relative slot order.
int relayout(dcr_graph *g) {
    g->ext_part_valid = false;  // slots move
    std::vector<int2> info((size_t)g->n);
    DCR_HIP(hipMemcpyAsync(info.data(), g->rowinfo, sizeof(int2) * (size_t)g->n, hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    std::vector<int32_t> start((size_t)g->n), cap((size_t)g->n);
    int64_t tot = 0;
    int32_t maxd = 0;
    for (int64_t u = 0; u < g->n; ++u) {
        int32_t d = info[(size_t)u].y;
        if (d > maxd) maxd = d;
        cap[(size_t)u] = d + slack_for(d) * 2;
        start[(size_t)u] = (int32_t)tot;
        tot += cap[(size_t)u];
        if (tot > INT32_MAX) DCR_FAIL(DCR_ECAPACITY, "adjacency exceeds 2^31 slots");
    }
    int32_t *d_start = nullptr;
    DCR_TRY(dev_alloc(&d_start, g->n));
    DCR_HIP(hipMemcpyAsync(d_start, start.data(), sizeof(int32_t) * (size_t)g->n, hipMemcpyHostToDevice, g->stream));
    DCR_HIP(hipMemcpyAsync(g->rowcap, cap.data(), sizeof(int32_t) * (size_t)g->n, hipMemcpyHostToDevice, g->stream));
    int2 *old_info = g->rowinfo;
    int32_t *old_col = g->col, *old_slot_row = g->slot_row;
    double *old_curv = g->curv;
    g->rowinfo = nullptr;
    DCR_TRY(dev_alloc(&g->rowinfo, g->n));
    DCR_TRY(alloc_layout(g, tot));
    DCR_HIP(hipMemsetAsync(g->curv, 0, sizeof(double) * (size_t)tot, g->stream));
    DCR_HIP(hipMemsetAsync(g->col, 0xff, sizeof(int32_t) * (size_t)tot, g->stream));
    int waves_per_block = 4;
    dim3 grid((unsigned)((g->n + waves_per_block - 1) / waves_per_block));
    hipLaunchKernelGGL(k_relayout, grid, dim3(64 * waves_per_block), 0, g->stream, old_info, old_col, old_curv, d_start,
                       g->rowinfo, g->col, g->curv, g->slot_row, g->rowcap, g->n);
    DCR_HIP(hipGetLastError());
    DCR_HIP(hipStreamSynchronize(g->stream));
    (void)hipFree(old_info);
    (void)hipFree(old_col);
    (void)hipFree(old_slot_row);
    (void)hipFree(old_curv);
    (void)hipFree(d_start);
    // work lists are sized by slots
    for (int b = 0; b < NBINS; ++b) {
        if (g->work[b]) (void)hipFree(g->work[b]);
        g->work[b] = nullptr;
    }
    g->work_cap = 0;
    g->max_deg_bound = maxd;
    return DCR_OK;
}

}  // namespace dcr

using namespace dcr;

extern "C" {

const char *dcr_last_error(void) { return g_err.c_str(); }

int dcr_device_count(int *out) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) c = 0;
    *out = c;
    return DCR_OK;
}

int dcr_graph_create(int device, int64_t n, int64_t m, const int64_t *src, const int64_t *dst, dcr_graph **out) {
    if (!out) DCR_FAIL(DCR_EINVAL, "out is null");
    *out = nullptr;
    if (n < 0 || n > MAX_NODES) DCR_FAIL(DCR_EINVAL, "num_nodes out of range (max 2^30-2)");
    if (m < 0 || (m > 0 && (!src || !dst))) DCR_FAIL(DCR_EINVAL, "bad edge arrays");

    // ---- to_networkx(to_undirected=True) on the host: keep (u,v) with v <= u, in order ----
    bool strictly_sorted = true;  // coalesced input => no duplicate can occur
    for (int64_t e = 0; e < m; ++e) {
        int64_t u = src[e], v = dst[e];
        if (u < 0 || v < 0 || u >= n || v >= n) DCR_FAIL(DCR_EINVAL, "edge endpoint out of range");
        if (u == v) DCR_FAIL(DCR_EINVAL, "self-loops are outside the SDRF boundary contract (strip them first)");
        if (e > 0 && !(src[e - 1] < u || (src[e - 1] == u && dst[e - 1] < v))) strictly_sorted = false;
    }
    std::vector<int64_t> keep;
    keep.reserve((size_t)(m / 2 + 1));
    if (strictly_sorted) {
        for (int64_t e = 0; e < m; ++e)
            if (dst[e] < src[e]) keep.push_back(e);
    } else {
        std::unordered_set<uint64_t> seen;
        seen.reserve((size_t)m);
        for (int64_t e = 0; e < m; ++e) {
            if (dst[e] > src[e]) continue;
            uint64_t key = ((uint64_t)dst[e] << 32) | (uint64_t)src[e];  // (min,max)
            if (seen.insert(key).second) keep.push_back(e);
        }
    }
    std::vector<int32_t> deg((size_t)n, 0);
    for (int64_t e : keep) {
        deg[(size_t)src[e]]++;
        deg[(size_t)dst[e]]++;
    }
    std::vector<int32_t> start((size_t)n), cap((size_t)n);
    int64_t tot = 0;
    for (int64_t u = 0; u < n; ++u) {
        cap[(size_t)u] = deg[(size_t)u] + slack_for(deg[(size_t)u]);
        start[(size_t)u] = (int32_t)tot;
        tot += cap[(size_t)u];
        if (tot > INT32_MAX) DCR_FAIL(DCR_ECAPACITY, "adjacency exceeds 2^31 slots");
    }
    std::vector<int32_t> col((size_t)tot, -1), slot_row((size_t)tot);
    std::vector<int32_t> fill((size_t)n, 0);
    for (int64_t e : keep) {
        int32_t u = (int32_t)src[e], v = (int32_t)dst[e];
        col[(size_t)start[(size_t)u] + fill[(size_t)u]++] = v;
        col[(size_t)start[(size_t)v] + fill[(size_t)v]++] = u;
    }
    std::vector<int2> info((size_t)n);
    for (int64_t u = 0; u < n; ++u) {
        info[(size_t)u] = make_int2(start[(size_t)u], deg[(size_t)u]);
        for (int32_t i = 0; i < cap[(size_t)u]; ++i) slot_row[(size_t)start[(size_t)u] + i] = (int32_t)u;
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) DCR_FAIL(DCR_EHIP, "no HIP device available");
    if (device < 0 || device >= ndev) DCR_FAIL(DCR_EINVAL, "device index out of range");
    DCR_HIP(hipSetDevice(device));

    dcr_graph *g = new dcr_graph();
    g->device = device;
    g->n = n;
    g->n_edges = (int64_t)keep.size();
    for (int64_t u = 0; u < n; ++u)
        if (deg[(size_t)u] > g->max_deg_bound) g->max_deg_bound = deg[(size_t)u];
    for (int64_t u = 0; u < n; ++u) g->sum_deg2 += (double)deg[(size_t)u] * (double)deg[(size_t)u];
    if (const char *impl = getenv("DCR_PASS"))
        g->pass_impl = (std::string(impl) == "edge") ? 1 : (std::string(impl) == "nc" || std::string(impl) == "node") ? 2 : (std::string(impl) == "h2") ? 3 : 0;
    *out = g;  // caller destroys on failure
    DCR_HIP(hipStreamCreate(&g->stream));
    DCR_HIP(hipEventCreate(&g->ev0));
    DCR_HIP(hipEventCreate(&g->ev1));
    DCR_HIP(hipEventCreateWithFlags(&g->ev_fork, DCR_EVENT_FLAGS));
    DCR_HIP(hipEventCreateWithFlags(&g->ev_aux, DCR_EVENT_FLAGS));
    DCR_HIP(hipStreamCreateWithFlags(&g->aux, hipStreamNonBlocking));
    // side[2] carries the finest-grained kernel of a pass (the smallest degree class): lowest priority, so that the
    // kernels with long units get their workgroups resident first and the fine-grained one fills in and finishes last
    int prio_low = 0, prio_high = 0;
    DCR_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    for (int b = 0; b < NBINS - 1; ++b) {
        DCR_HIP(hipStreamCreateWithPriority(&g->side[b], hipStreamNonBlocking, b == 2 ? prio_low : prio_high));
        DCR_HIP(hipEventCreateWithFlags(&g->ev_join[b], DCR_EVENT_FLAGS));
    }
    for (int b = 0; b < 2; ++b) DCR_HIP(hipStreamCreateWithPriority(&g->low[b], hipStreamNonBlocking, prio_low));
    DCR_HIP(hipEventCreateWithFlags(&g->ev_aux2, DCR_EVENT_FLAGS));
    DCR_TRY(dev_alloc(&g->rowinfo, n));
    DCR_TRY(dev_alloc(&g->rowcap, n));
    DCR_TRY(alloc_layout(g, tot));
    DCR_TRY(dev_alloc(&g->dres, 1));
    DCR_TRY(dev_alloc(&g->imp_stats, 1));
    DCR_HIP(hipMemsetAsync(g->imp_stats, 0xFF, sizeof(*g->imp_stats), g->stream));  // (pos_x_in_y = -1)
    DCR_TRY(dev_alloc(&g->draw_bsum, 256));
    DCR_TRY(dev_alloc(&g->dirty, n + 4));  // OR-ed through 32-bit words
    DCR_HIP(hipMemsetAsync(g->dirty, 0, (size_t)(n > 0 ? n : 1), g->stream));
    DCR_TRY(dev_alloc(&g->touched, n + 64));
    DCR_HIP(hipHostMalloc((void **)&g->hres, sizeof(DevResult), hipHostMallocDefault));
    std::memset(g->hres, 0, sizeof(DevResult));
    DCR_HIP(hipMemsetAsync(g->dres, 0, sizeof(DevResult), g->stream));
    DCR_HIP(hipMemsetAsync(g->curv, 0, sizeof(double) * (size_t)(tot > 0 ? tot : 1), g->stream));
    if (n > 0) {
        DCR_HIP(hipMemcpyAsync(g->rowinfo, info.data(), sizeof(int2) * (size_t)n, hipMemcpyHostToDevice, g->stream));
        DCR_HIP(hipMemcpyAsync(g->rowcap, cap.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, g->stream));
    }
    if (tot > 0) {
        DCR_HIP(hipMemcpyAsync(g->col, col.data(), sizeof(int32_t) * (size_t)tot, hipMemcpyHostToDevice, g->stream));
        DCR_HIP(hipMemcpyAsync(g->slot_row, slot_row.data(), sizeof(int32_t) * (size_t)tot, hipMemcpyHostToDevice,
                               g->stream));
    }
    DCR_HIP(hipStreamSynchronize(g->stream));
    return DCR_OK;
}

int dcr_graph_destroy(dcr_graph *g) {
    if (!g) return DCR_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    void *dev_ptrs[] = {g->rowinfo, g->rowcap, g->col, g->slot_row, g->curv, g->red_scratch, g->scan_a, g->scan_b,
                        g->imp_table, g->imp_posx, g->imp_posy, g->imp_c1, g->imp_c2, g->imp_b, g->imp_c,
                        g->imp_rowcount, g->imp_rowoff, g->imp_adjbits, g->imp_out, g->imp_ci, g->imp_cj,
                        g->imp_stats, g->draw_bsum, g->dres, g->dirty, g->nc_units[0], g->nc_units[1], g->nc_units[2],
                        g->nc_units[3], g->nc_units[4], g->nc_touch, g->nc_fine_list, g->touched, g->nc_trace, g->nc_queues, g->giant_list,
                        g->giant_pos, g->giant_cnt, g->giant_acc, g->hub_list, g->hub_cnt, g->h2_weight,
                        g->h2_units[0], g->h2_units[1], g->h2_units[2], g->h2_units[3], g->h2_units[4], g->h2_retry, g->h2_task, g->h2_cand, g->h2_part, g->h2_bloom,
                        g->h2_rec, g->h2_eset, g->ext_part, g->h2_lists};
    for (void *p : dev_ptrs)
        if (p) (void)hipFree(p);
    for (int b = 0; b < NBINS; ++b)
        if (g->work[b]) (void)hipFree(g->work[b]);
    if (g->hres) (void)hipHostFree(g->hres);
    if (g->imp_out_h) (void)hipHostFree(g->imp_out_h);
    if (g->imp_ci_h) (void)hipHostFree(g->imp_ci_h);
    if (g->imp_cj_h) (void)hipHostFree(g->imp_cj_h);
    for (int b = 0; b < NBINS - 1; ++b) {
        if (g->side[b]) (void)hipStreamDestroy(g->side[b]);
        if (g->ev_join[b]) (void)hipEventDestroy(g->ev_join[b]);
    }
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_aux) (void)hipEventDestroy(g->ev_aux);
    if (g->aux) (void)hipStreamDestroy(g->aux);
    for (int b = 0; b < 2; ++b)
        if (g->low[b]) (void)hipStreamDestroy(g->low[b]);
    if (g->ev_aux2) (void)hipEventDestroy(g->ev_aux2);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return DCR_OK;
}

int dcr_graph_num_nodes(const dcr_graph *g, int64_t *out) {
    if (!g || !out) DCR_FAIL(DCR_EINVAL, "null argument");
    *out = g->n;
    return DCR_OK;
}

int dcr_graph_num_edges(const dcr_graph *g, int64_t *out) {
    if (!g || !out) DCR_FAIL(DCR_EINVAL, "null argument");
    *out = g->n_edges;
    return DCR_OK;
}

static int check_pair(const dcr_graph *g, int32_t u, int32_t v) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (u < 0 || v < 0 || u >= g->n || v >= g->n) DCR_FAIL(DCR_EINVAL, "node id out of range");
    if (u == v) DCR_FAIL(DCR_EINVAL, "self-loops are not supported");
    return DCR_OK;
}

int dcr_graph_add_edge(dcr_graph *g, int32_t u, int32_t v) {
    DCR_TRY(check_pair(g, u, v));
    DCR_HIP(hipSetDevice(g->device));
    g->am_valid = false;
    g->amax_valid = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        launch_add_edge(g, u, v);
        DCR_HIP(hipGetLastError());
        DCR_TRY(sync_result(g));
        if (g->hres->add_status == 0) {
            g->n_edges++;
            g->max_deg_bound++;
            launch_mark_dirty(g, u, v, g->pending_edits++);
            return DCR_OK;
        }
        if (g->hres->add_status == 2) return DCR_OK;  // networkx: adding an existing edge changes nothing
        DCR_TRY(relayout(g));
    }
    DCR_FAIL(DCR_ECAPACITY, "row still full after relayout");
}

int dcr_graph_remove_edge(dcr_graph *g, int32_t u, int32_t v) {
    DCR_TRY(check_pair(g, u, v));
    DCR_HIP(hipSetDevice(g->device));
    g->am_valid = false;
    g->amax_valid = false;
    launch_mark_dirty(g, u, v, g->pending_edits++);
    g->h2_eset_pending += 1;
    g->ext_part_valid = false;
    hipLaunchKernelGGL(k_remove_edge, dim3(1), dim3(64), 0, g->stream, g->rowinfo, g->col, g->curv, u, v, g->dres);
    DCR_HIP(hipGetLastError());
    DCR_TRY(sync_result(g));
    if (g->hres->misc[0] != 0) DCR_FAIL(DCR_ENOTFOUND, "edge not in graph");
    g->n_edges--;
    return DCR_OK;
}

int dcr_graph_has_edge(dcr_graph *g, int32_t u, int32_t v, int *out) {
    if (!out) DCR_FAIL(DCR_EINVAL, "null out");
    if (g && u == v && u >= 0 && u < g->n) {
        *out = 0;
        return DCR_OK;
    }
    DCR_TRY(check_pair(g, u, v));
    DCR_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_has_edge, dim3(1), dim3(64), 0, g->stream, g->rowinfo, g->col, u, v, g->dres);
    DCR_HIP(hipGetLastError());
    DCR_TRY(sync_result(g));
    *out = g->hres->misc[0];
    return DCR_OK;
}

int dcr_graph_degree(dcr_graph *g, int32_t u, int32_t *out) {
    if (!g || !out || u < 0 || u >= g->n) DCR_FAIL(DCR_EINVAL, "bad argument");
    DCR_HIP(hipSetDevice(g->device));
    int2 ri;
    DCR_HIP(hipMemcpyAsync(&ri, g->rowinfo + u, sizeof(int2), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    *out = ri.y;
    return DCR_OK;
}

int dcr_graph_neighbors(dcr_graph *g, int32_t u, int64_t cap, int32_t *out, int64_t *n_out) {
    if (!g || !n_out || u < 0 || u >= g->n) DCR_FAIL(DCR_EINVAL, "bad argument");
    DCR_HIP(hipSetDevice(g->device));
    int2 ri;
    DCR_HIP(hipMemcpyAsync(&ri, g->rowinfo + u, sizeof(int2), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    *n_out = ri.y;
    if (ri.y > cap) DCR_FAIL(DCR_ECAPACITY, "neighbour buffer too small");
    if (ri.y > 0) {
        DCR_HIP(hipMemcpyAsync(out, g->col + ri.x, sizeof(int32_t) * (size_t)ri.y, hipMemcpyDeviceToHost, g->stream));
        DCR_HIP(hipStreamSynchronize(g->stream));
    }
    return DCR_OK;
}

static int fetch_rows(dcr_graph *g, std::vector<int2> &info, std::vector<int32_t> &col) {
    DCR_HIP(hipSetDevice(g->device));
    info.resize((size_t)g->n);
    col.resize((size_t)g->cap_total);
    if (g->n > 0)
        DCR_HIP(hipMemcpyAsync(info.data(), g->rowinfo, sizeof(int2) * (size_t)g->n, hipMemcpyDeviceToHost, g->stream));
    if (g->cap_total > 0)
        DCR_HIP(hipMemcpyAsync(col.data(), g->col, sizeof(int32_t) * (size_t)g->cap_total, hipMemcpyDeviceToHost,
                               g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    return DCR_OK;
}

int dcr_graph_edges(dcr_graph *g, int32_t *out_u, int32_t *out_v) {
    if (!g || !out_u || !out_v) DCR_FAIL(DCR_EINVAL, "null argument");
    std::vector<int2> info;
    std::vector<int32_t> col;
    DCR_TRY(fetch_rows(g, info, col));
    int64_t p = 0;
    for (int64_t u = 0; u < g->n; ++u) {
        int2 ri = info[(size_t)u];
        for (int32_t i = 0; i < ri.y; ++i) {
            int32_t v = col[(size_t)ri.x + i];
            if (v > u) {
                if (p >= g->n_edges) DCR_FAIL(DCR_ESTATE, "edge count drifted");
                out_u[p] = (int32_t)u;
                out_v[p] = v;
                ++p;
            }
        }
    }
    if (p != g->n_edges) DCR_FAIL(DCR_ESTATE, "edge count drifted");
    return DCR_OK;
}

int dcr_graph_export_edge_index(dcr_graph *g, int64_t *out) {
    // from_networkx (sdrf_no_cuda.py:68) = convert_node_labels_to_integers + to_directed().edges: the relabel
    // step re-adds edges in G.edges order, so row u lists its smaller neighbours ascending (one per earlier
    // outer node), then its larger neighbours in insertion order.
    if (!g || !out) DCR_FAIL(DCR_EINVAL, "null argument");
    std::vector<int2> info;
    std::vector<int32_t> col;
    DCR_TRY(fetch_rows(g, info, col));
    int64_t M = 2 * g->n_edges, p = 0;
    std::vector<int32_t> small;
    for (int64_t u = 0; u < g->n; ++u) {
        int2 ri = info[(size_t)u];
        small.clear();
        for (int32_t i = 0; i < ri.y; ++i)
            if (col[(size_t)ri.x + i] < u) small.push_back(col[(size_t)ri.x + i]);
        std::sort(small.begin(), small.end());
        if (p + ri.y > M) DCR_FAIL(DCR_ESTATE, "edge count drifted");
        for (int32_t v : small) {
            out[p] = u;
            out[M + p] = v;
            ++p;
        }
        for (int32_t i = 0; i < ri.y; ++i) {
            int32_t v = col[(size_t)ri.x + i];
            if (v > u) {
                out[p] = u;
                out[M + p] = v;
                ++p;
            }
        }
    }
    if (p != M) DCR_FAIL(DCR_ESTATE, "edge count drifted");
    return DCR_OK;
}

int dcr_curvature_read(dcr_graph *g, double *out_curv, int32_t *out_u, int32_t *out_v) {
    if (!g || !out_curv) DCR_FAIL(DCR_EINVAL, "null argument");
    if (!g->curv_valid) DCR_FAIL(DCR_ESTATE, "no curvature pass has run");
    std::vector<int2> info;
    std::vector<int32_t> col;
    DCR_TRY(fetch_rows(g, info, col));
    std::vector<double> cv((size_t)g->cap_total);
    if (g->cap_total > 0) {
        DCR_HIP(hipMemcpyAsync(cv.data(), g->curv, sizeof(double) * (size_t)g->cap_total, hipMemcpyDeviceToHost,
                               g->stream));
        DCR_HIP(hipStreamSynchronize(g->stream));
    }
    int64_t p = 0;
    for (int64_t u = 0; u < g->n; ++u) {
        int2 ri = info[(size_t)u];
        for (int32_t i = 0; i < ri.y; ++i) {
            int32_t v = col[(size_t)ri.x + i];
            if (v > u) {
                out_curv[p] = cv[(size_t)ri.x + i];
                if (out_u) out_u[p] = (int32_t)u;
                if (out_v) out_v[p] = v;
                ++p;
            }
        }
    }
    return DCR_OK;
}

int dcr_profile_reset(dcr_graph *g) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    g->pass_ms_total = 0.0;
    g->pass_count = 0;
    g->profile = true;
    return DCR_OK;
}

int dcr_profile_read(dcr_graph *g, double *pass_ms_total, int64_t *pass_count) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (pass_ms_total) *pass_ms_total = g->pass_ms_total;
    if (pass_count) *pass_count = g->pass_count;
    return DCR_OK;
}

}  // extern "C"
