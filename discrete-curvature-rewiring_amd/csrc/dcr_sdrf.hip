// SDRF-side kernels of libdcr_hip.so: arg-min/arg-max over edges, the candidate/improvement tensor and the
// add/remove tail of one iteration (rewiring/sdrf_no_cuda.py:27-66).
//
// Improvements.  The reference evaluates bfc_edge(x,y) twice per candidate (sdrf_no_cuda.py:41-46).  Here the
// integer ingredients of bfc_edge(x,y) on G + (i,j) are derived exactly from per-neighbour counters of the base
// graph, and the float64 closing expression (bfc_naive.py:31-40) is evaluated on those integers, so
// after - before is bit-identical to the reference while the work per candidate is O(1):
//   c1[a] = |N(i_a) ∩ DY| for i_a in DX = N(x) \ N(y) \ {y};  c2[b] = |N(j_b) ∩ DX| for j_b in DY.
//   * i in N(x), j in N(y) (neither is x or y): only (i in DX, j in DY) changes anything: c1[a]+1, c2[b]+1.
//   * i == x, j in DY: deg(x)+1, T+1, j leaves DY: c1[k] drops by [k ~ j]; needs one scan of row j.
//   * j == y, i in DX: mirrored.
//   * every other admissible pair leaves all ingredients unchanged: improvement = +0.0.
#include <cstdlib>
#include "dcr_internal.h"
#include <chrono>
#include <cstdio>

namespace dcr {

struct RowView {
    const int2 *rowinfo;
    const int32_t *col;
    const int32_t *slot_row;
};

__device__ __host__ inline double bfc_formula(int d1, int d2, int T, int s1, int s2, int gamma) {
    int dmax = d1 > d2 ? d1 : d2, dmin = d1 < d2 ? d1 : d2;
    double r = 2.0 / (double)d1;
    r = r + 2.0 / (double)d2;
    r = r - 2.0;
    r = r + (double)(2 * (int64_t)T) / (double)dmax;
    r = r + (double)T / (double)dmin;
    if (s1 == 0 || s2 == 0) return r;
    double q = 1.0 / (double)gamma;
    q = q / (double)dmax;
    q = q * (double)(s1 + s2);
    return r + q;
}

__device__ inline double bfc_value(int d1, int d2, int T, int s1, int s2, int gamma) {
    if ((d1 < d2 ? d1 : d2) == 1) return 0.0;  // bfc_naive.py:18-19
    return bfc_formula(d1, d2, T, s1, s2, gamma);
}

// ---------------------------------------------------------------------------------------------
// arg-extremum over undirected edges, first in G.edges order (= smallest slot) on ties
// ---------------------------------------------------------------------------------------------
// (Ext, ext_better and the reductions live in dcr_internal.h: the two-hop pass's closing kernel leaves per-block extrema too)
__global__ void __launch_bounds__(256) k_argext_edges(RowView g, int64_t cap_total, const double *curv, int want_max,
                                                       int excl_u, int excl_v, const DevResult *res, Ext *partial) {
    __shared__ double shv[4];
    __shared__ int shs[4];
    if (excl_u == -2) {  // the edge picked on the device (dcr_sdrf_tail_at)
        excl_u = res->cand_i;
        excl_v = res->cand_j;
    }
    double best_v = 0.0;
    int best_s = -1;
    // four slots per thread and round, their chains of dependent loads (owner row -> row extent -> neighbour ->
    // value) in flight together; slots are visited in increasing order per thread, so "first" extremum is preserved by
    // ext_better's (value, slot) order
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < cap_total; s0 += 4 * stride) {
        int u[4], v[4];
        int2 ru[4];
        bool ok[4];
        double cv[4];
        // everything addressed by the slot itself in one round (owner row, neighbour, value: all in bounds for every slot,
        // slack included), the row extent in a second one
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t s = s0 + q * stride;
            ok[q] = s < cap_total;
            u[q] = ok[q] ? g.slot_row[s] : 0;
            v[q] = ok[q] ? g.col[s] : -1;
            cv[q] = ok[q] ? curv[s] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ru[q] = ok[q] ? g.rowinfo[u[q]] : make_int2(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t s = s0 + q * stride;
            if (!ok[q] || (int)(s - ru[q].x) >= ru[q].y || v[q] <= u[q] || (u[q] == excl_u && v[q] == excl_v)) continue;
            ext_take(best_v, best_s, cv[q], (int)s, want_max);
        }
    }
    ext_block_reduce(best_v, best_s, want_max, shv, shs);
    if (threadIdx.x == 0) partial[blockIdx.x] = ext_make(best_v, best_s);
}

// both extrema in ONE sweep, left as per-workgroup partials where the two-hop pass's closing kernel leaves its per-wave ones
// (g->ext_part): after a node-centric or incremental pass the arg-min of the loop (sdrf_no_cuda.py:27) and the stale arg-max of
// its removal step (:57-61) were two sweeps of 13 us each
// (round 5: clear_words — the incremental pass's node flags, which nothing reads between the pass and this sweep: their fill
//  launch rode in front of it before)
__global__ void __launch_bounds__(256) k_argext_edges_both(RowView g, int64_t cap_total, const double *curv, Ext *part_min, Ext *part_max,
                                                            unsigned *clear_words, int64_t n_clear_words, DevResult *res) {
    __shared__ double shv[4];
    __shared__ int shs[4];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_clear_words; i += (int64_t)gridDim.x * blockDim.x) clear_words[i] = 0u;
    if (n_clear_words > 0 && blockIdx.x == 0 && threadIdx.x == 0) res->touched_n = 0;   // (the list of flagged nodes goes with the flags)
    double lo_v = 0.0, hi_v = 0.0;
    int lo_s = -1, hi_s = -1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < cap_total; s0 += 4 * stride) {
        int u[4], v[4];
        int2 ru[4];
        bool ok[4];
        double cv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t s = s0 + q * stride;
            ok[q] = s < cap_total;
            u[q] = ok[q] ? g.slot_row[s] : 0;
            v[q] = ok[q] ? g.col[s] : -1;
            cv[q] = ok[q] ? curv[s] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ru[q] = ok[q] ? g.rowinfo[u[q]] : make_int2(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t s = s0 + q * stride;
            if (!ok[q] || (int)(s - ru[q].x) >= ru[q].y || v[q] <= u[q]) continue;
            ext_take(lo_v, lo_s, cv[q], (int)s, 0);
            ext_take(hi_v, hi_s, cv[q], (int)s, 1);
        }
    }
    ext_block_reduce(lo_v, lo_s, 0, shv, shs);
    if (threadIdx.x == 0) part_min[blockIdx.x] = ext_make(lo_v, lo_s);
    __syncthreads();
    ext_block_reduce(hi_v, hi_s, 1, shv, shs);
    if (threadIdx.x == 0) part_max[blockIdx.x] = ext_make(hi_v, hi_s);
}

// the one-workgroup reduction of the partial extrema into the result block (any workgroup size)
__device__ inline void argext_final_body(const RowView &g, const Ext *partial, int nparts, int want_max, DevResult *res) {
    __shared__ double shv[16];
    __shared__ int shs[16];
    double best_v = 0.0;
    int best_s = -1;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        const Ext e = partial[i];
        ext_take(best_v, best_s, e.val, e.slot, want_max);
    }
    ext_block_reduce(best_v, best_s, want_max, shv, shs);
    if (threadIdx.x == 0) {
        res->ext_val = best_v;
        res->ext_slot = best_s;
        const int eu = best_s >= 0 ? g.slot_row[best_s] : -1, ev = best_s >= 0 ? g.col[best_s] : -1;
        res->ext_u = eu;
        res->ext_v = ev;
        res->ext_du = best_s >= 0 ? g.rowinfo[eu].y : 0;
        res->ext_dv = best_s >= 0 ? g.rowinfo[ev].y : 0;
    }
}

__global__ void __launch_bounds__(1024) k_argext_final(RowView g, const Ext *partial, int nparts, int want_max,
                                                        DevResult *res) {
    argext_final_body(g, partial, nparts, want_max, res);
}

// first maximum of a plain array (np.argmax of the improvements, utils/softmax.py:7)
__global__ void __launch_bounds__(256) k_argmax_array(const double *a, int64_t n, Ext *partial, int64_t *idx_hi) {
    __shared__ double shv[4];
    __shared__ int shs[4];
    double best_v = 0.0;
    int best_s = -1;
    // indices may exceed int32 only beyond 2^31 candidates, which the row capacity already excludes
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        ext_take(best_v, best_s, a[i], (int)i, 1);
    }
    ext_block_reduce(best_v, best_s, 1, shv, shs);
    if (threadIdx.x == 0) partial[blockIdx.x] = ext_make(best_v, best_s);
    (void)idx_hi;
}

__global__ void __launch_bounds__(256) k_argmax_final(const Ext *partial, int nparts, DevResult *res) {
    __shared__ double shv[4];
    __shared__ int shs[4];
    double best_v = 0.0;
    int best_s = -1;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        const Ext e = partial[i];
        ext_take(best_v, best_s, e.val, e.slot, 1);
    }
    ext_block_reduce(best_v, best_s, 1, shv, shs);
    if (threadIdx.x == 0) res->imp_argmax = best_s;
}

static inline int nparts_threads(int nparts) { return nparts > 1024 ? 1024 : nparts > 256 ? 512 : 256; }  // threads of the one-workgroup reduction
constexpr int ARGEXT_BLOCKS = 1024;  // (8192 blocks: 92 us instead of 33 on S100k, the per-block reduction dominates)

int launch_argext(dcr_graph *g, int want_max, int excl_u, int excl_v, hipStream_t st) {
    if (!st) st = g->stream;
    g->amax_valid = false;  // the ext fields of the result block are about to be overwritten
    if (!g->red_scratch) {
        Ext *p = nullptr;
        DCR_TRY(dev_alloc(&p, ARGEXT_BLOCKS));
        g->red_scratch = p;
    }
    RowView vw{g->rowinfo, g->col, g->slot_row};
    int64_t blocks = (g->cap_total + 255) / 256;
    if (blocks > ARGEXT_BLOCKS) blocks = ARGEXT_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_argext_edges, dim3((unsigned)blocks), dim3(256), 0, st, vw, g->cap_total, g->curv,
                       want_max, excl_u, excl_v, g->dres, (Ext *)g->red_scratch);
    hipLaunchKernelGGL(k_argext_final, dim3(1), dim3(nparts_threads((int)blocks)), 0, st, vw, (const Ext *)g->red_scratch, (int)blocks,
                       want_max, g->dres);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// one sweep for both extrema; the partials stay valid until the next edit (as the closing kernel's of the two-hop pass)
int launch_argext_both(dcr_graph *g, hipStream_t st, bool clear_dirty) {
    if (!st) st = g->stream;
    if (!g->ext_part) {
        Ext *p = nullptr;
        DCR_TRY(dev_alloc(&p, 2 * EXT_PART_BLOCKS));
        g->ext_part = p;
    }
    RowView vw{g->rowinfo, g->col, g->slot_row};
    int64_t blocks = (g->cap_total + 255) / 256;
    if (blocks > ARGEXT_BLOCKS) blocks = ARGEXT_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_argext_edges_both, dim3((unsigned)blocks), dim3(256), 0, st, vw, g->cap_total, g->curv, (Ext *)g->ext_part,
                       (Ext *)g->ext_part + EXT_PART_BLOCKS, reinterpret_cast<unsigned *>(g->dirty),
                       clear_dirty ? (int64_t)(g->n + 3) / 4 : (int64_t)0, g->dres);   // (the flags hold n + 4 bytes)
    DCR_HIP(hipGetLastError());
    g->ext_part_n = (int)blocks;
    g->ext_part_valid = true;
    return DCR_OK;
}

int launch_argext_from_parts(dcr_graph *g, int want_max, hipStream_t st) {
    if (!st) st = g->stream;
    g->amax_valid = false;  // the ext fields of the result block are about to be overwritten
    RowView vw{g->rowinfo, g->col, g->slot_row};
    const Ext *parts = (const Ext *)g->ext_part + (want_max ? EXT_PART_BLOCKS : 0);
    hipLaunchKernelGGL(k_argext_final, dim3(1), dim3(nparts_threads(g->ext_part_n)), 0, st, vw, parts, g->ext_part_n, want_max, g->dres);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// ---------------------------------------------------------------------------------------------
// improvement pipeline
// ---------------------------------------------------------------------------------------------
constexpr unsigned T_EMPTY = 0xFFFFFFFFu;

__device__ inline unsigned g_hash(unsigned key, unsigned mask) { return (key * 0x9E3779B1u >> 7) & mask; }

// slot index of key in the global table, or -1
__device__ inline int g_find(const int32_t *keys, unsigned mask, int key) {
    unsigned h = g_hash((unsigned)key, mask);
    while (true) {
        const int e = keys[h];
        if (e == key) return (int)h;
        if ((unsigned)e == T_EMPTY) return -1;
        h = (h + 1) & mask;
    }
}

// four look-ups with their first probes in flight together (the rows swept by the improvement kernels are walked by one
// wave each: the longest row sets the kernel's duration, and its steps are chains of dependent loads)
__device__ inline void g_find4(const int32_t *keys, unsigned mask, const int (&key)[4], const bool (&valid)[4], int (&out)[4]) {
    unsigned h[4];
    int e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = g_hash((unsigned)key[q], mask);
        e[q] = valid[q] ? keys[h[q]] : (int)T_EMPTY;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[q] = -1;
        if (!valid[q]) continue;
        while (true) {
            if (e[q] == key[q]) { out[q] = (int)h[q]; break; }
            if ((unsigned)e[q] == T_EMPTY) break;
            h[q] = (h[q] + 1) & mask;
            e[q] = keys[h[q]];
        }
    }
}

struct ImpBuf {
    int32_t *keys, *posx, *posy;  // hash table over N(x) ∪ N(y) minus {x,y}
    int32_t *c1, *c2;             // [dx], [dy]
    int32_t *clsx, *clsy;         // 0: only in own row (DX / DY), 1: in both rows (triangle), 2: the other endpoint
    double *impb, *impc;          // class B (i == x) per b, class C (j == y) per a
    int32_t *rowcount, *rowoff;   // [dx+1]
    uint32_t *adjbits;            // [dx+1][words]
    ImpStats *st;
};

// K1 (round 4: one launch for both rows; the pipeline was eleven launch-bound kernels, 45 us of launch gaps in 115): row x and
// row y go into the table concurrently — a key met in both rows (a triangle node) is inserted by whichever thread comes first
// and found by the other; which of the two it was does not matter, posx / posy are separate arrays.  The classes (0: in its
// own row only, 1: in both, 2: the other endpoint) are resolved by the next kernel, when the table is complete.
// Round 5: the removal step's stale arg-max (the reduction of the partial maxima the pass's closing kernel left; it depends on
// nothing in the pipeline and writes other fields of the result block) rides on this launch as ONE MORE workgroup instead of
// being a launch of its own between the pipeline and the draw (5.4 us of the iteration: profiles/r05_step_timeline.txt).
__global__ void __launch_bounds__(256) k_imp_insert(RowView g, ImpBuf B, int x, int y, unsigned mask, const Ext *amax_parts, int amax_n,
                                                     DevResult *res) {
    const int nblocks = (int)gridDim.x - (amax_parts ? 1 : 0);
    if ((int)blockIdx.x == nblocks) {  // (uniform: the extra workgroup)
        argext_final_body(g, amax_parts, amax_n, 1, res);
        return;
    }
    const int2 rx = g.rowinfo[x], ry = g.rowinfo[y];
    const int total = rx.y + ry.y;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += nblocks * blockDim.x) {
        const bool isx = t < rx.y;
        const int p = isx ? t : t - rx.y;
        const int k = g.col[(isx ? rx.x : ry.x) + p];
        if (isx) {
            B.c1[p] = 0;
            B.clsx[p] = k == y ? 2 : 0;
        } else {
            B.c2[p] = 0;
            B.clsy[p] = k == x ? 2 : 0;
            if (k == x) B.st->pos_x_in_y = p;
        }
        if (k == (isx ? y : x)) continue;  // the endpoints themselves are never in the table
        unsigned h = g_hash((unsigned)k, mask);
        while (true) {
            const unsigned old = atomicCAS((unsigned *)&B.keys[h], T_EMPTY, (unsigned)k);
            if (old == T_EMPTY || old == (unsigned)k) break;
            h = (h + 1) & mask;
        }
        if (isx) B.posx[h] = p;
        else B.posy[h] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        B.st->x = x;
        B.st->y = y;
        B.st->dx = rx.y;
        B.st->dy = ry.y;
        B.st->table_mask = (int)mask;
        B.st->T = 0;
        B.st->deg_min_is_one = (rx.y < ry.y ? rx.y : ry.y) == 1;
        B.st->done_rows = 0;
        B.st->done_draw = 0;
        // (pos_x_in_y: written above by the thread that meets x in row y; reset to -1 by the pipeline's last kernel)
    }
}

// empty table (only after a fresh allocation or a pipeline that did not run to its end: k_imp_emit leaves the table empty)
__global__ void __launch_bounds__(256) k_imp_clear(ImpBuf B, int64_t ts) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < ts; i += nth) {
        B.keys[i] = -1;
        B.posx[i] = -1;
        B.posy[i] = -1;
    }
    if (tid == 0) B.st->pos_x_in_y = -1;
}

struct Mx {
    int m, c, s;  // maximum, how many attain it, largest value below it
};

__device__ inline Mx mx_join(Mx a, Mx b) {
    Mx r;
    if (a.m == b.m) {
        r.m = a.m;
        r.c = a.c + b.c;
        r.s = a.s > b.s ? a.s : b.s;
    } else if (a.m > b.m) {
        r.m = a.m;
        r.c = a.c;
        r.s = a.s > b.m ? a.s : b.m;
    } else {
        r.m = b.m;
        r.c = b.c;
        r.s = b.s > a.m ? b.s : a.m;
    }
    return r;
}

__device__ inline Mx mx_add(Mx a, int v) {
    Mx b;
    b.m = v;
    b.c = 1;
    b.s = -1;
    return mx_join(a, b);
}

__device__ inline void block_stats(const int32_t *c, int n, int *cnt_pos, Mx *mx, int *shi) {
    int pos = 0;
    Mx m;
    m.m = -1;
    m.c = 0;
    m.s = -1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = c[i];
        pos += v > 0;
        m = mx_add(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_xor(pos, off);
        Mx o;
        o.m = __shfl_xor(m.m, off);
        o.c = __shfl_xor(m.c, off);
        o.s = __shfl_xor(m.s, off);
        m = mx_join(m, o);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) {
        shi[wid * 4 + 0] = pos;
        shi[wid * 4 + 1] = m.m;
        shi[wid * 4 + 2] = m.c;
        shi[wid * 4 + 3] = m.s;
    }
    __syncthreads();
    pos = 0;
    m.m = -1;
    m.c = 0;
    m.s = -1;
    for (int w = 0; w < nw; ++w) {
        pos += shi[w * 4];
        Mx o;
        o.m = shi[w * 4 + 1];
        o.c = shi[w * 4 + 2];
        o.s = shi[w * 4 + 3];
        m = mx_join(m, o);
    }
    *cnt_pos = pos;
    *mx = m;
}

// |sq1|, |sq2|, maxima (with multiplicity and runner-up), triangles and the base curvature; called by one whole workgroup
__device__ inline void imp_close_stats(ImpBuf B, int curv_type, int *shi) {
    ImpStats *st = B.st;
    const int dx = st->dx, dy = st->dy;
    int s1, s2;
    Mx m1, m2;
    block_stats(B.c1, dx, &s1, &m1, shi);
    block_stats(B.c2, dy, &s2, &m2, shi);
    // triangles: the members of row x that are in row y too (class 1)
    int T = 0;
    for (int a = threadIdx.x; a < dx; a += blockDim.x) T += B.clsx[a] == 1;
    for (int off = 32; off > 0; off >>= 1) T += __shfl_xor(T, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = T;
    __syncthreads();
    if (threadIdx.x == 0) {
        T = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) T += shi[w];
        st->T = T;
        st->s1 = s1;
        st->s2 = s2;
        st->max1 = m1.m < 0 ? 0 : m1.m;
        st->cnt1 = m1.c;
        st->sec1 = m1.s < 0 ? 0 : m1.s;
        st->max2 = m2.m < 0 ? 0 : m2.m;
        st->cnt2 = m2.c;
        st->sec2 = m2.s < 0 ? 0 : m2.s;
        const int gam = st->max1 > st->max2 ? st->max1 : st->max2;
        double before;
        switch (curv_type) {
            case DCR_CURV_1D: before = (double)(4 - dx - dy); break;
            case DCR_CURV_AUGMENTED: before = (double)(4 - dx - dy + 3 * T); break;
            case DCR_CURV_HAANTJES: before = (double)T; break;
            default: before = bfc_value(dx, dy, T, s1, s2, gam);
        }
        st->before = before;
    }
    __syncthreads();
}

// K4: class B (i == x, j = y_nb[b] in DY) for workgroups [0,dy), class C (j == y, i = x_nb[a] in DX) for [dy,dy+dx).
// (Round 4: a workgroup per element, its four waves sharing the row — a wave per element left the kernel waiting 22 us for
//  the one wave that streams a hub's 1,400-entry row.)
__global__ void __launch_bounds__(256) k_imp_bc(RowView g, ImpBuf B, int x, int y, unsigned mask, int curv_type) {
    __shared__ int dec_sh, maxadj_sh;
    const ImpStats st = *B.st;
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x;
    if (wv >= st.dx + st.dy) return;
    const bool isB = wv < st.dy;
    const int p = isB ? wv : wv - st.dy;
    const int cls = isB ? B.clsy[p] : B.clsx[p];
    double *out = isB ? B.impb : B.impc;
    if (cls != 0) {
        if (threadIdx.x == 0) out[p] = 0.0;
        return;
    }
    if (curv_type != DCR_CURV_BFC) {
        // 4 - d1 - d2 (+3T): one degree grows, one triangle appears (classical_curvatures.py:14-28)
        const double d = curv_type == DCR_CURV_1D ? -1.0 : curv_type == DCR_CURV_AUGMENTED ? 2.0 : 1.0;
        if (threadIdx.x == 0) out[p] = d;
        return;
    }
    if (threadIdx.x == 0) {
        dec_sh = 0;
        maxadj_sh = 0;
    }
    __syncthreads();
    const int2 rown = g.rowinfo[isB ? y : x];
    const int node = g.col[rown.x + p];  // j for class B, i for class C
    const int2 rn = g.rowinfo[node];
    const int32_t *cnt_other = isB ? B.c1 : B.c2;  // counters of the side that loses edges
    const int max_other = isB ? st.max1 : st.max2;
    int n_dec0 = 0, n_maxadj = 0;
    for (int base = 0; base < rn.y; base += 4 * 256) {
        int w[4], h[4];
        bool in[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = base + 256 * q + (int)threadIdx.x;
            in[q] = t < rn.y;
            w[q] = in[q] ? g.col[rn.x + t] : -1;
        }
        g_find4(B.keys, mask, w, in, h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool dec0 = false, mxa = false;
            if (h[q] >= 0) {
                const int px = B.posx[h[q]], py = B.posy[h[q]];
                const bool other_only = isB ? (px >= 0 && py < 0) : (py >= 0 && px < 0);
                if (other_only) {
                    const int c = cnt_other[isB ? px : py];
                    dec0 = (c == 1);
                    mxa = (c == max_other);
                }
            }
            n_dec0 += __popcll(__ballot(dec0));
            n_maxadj += __popcll(__ballot(mxa));
        }
    }
    if (lane == 0) {  // (the per-wave totals are uniform over the wave: one lane adds them)
        if (n_dec0) atomicAdd(&dec_sh, n_dec0);
        if (n_maxadj) atomicAdd(&maxadj_sh, n_maxadj);
    }
    __syncthreads();
    n_dec0 = dec_sh;
    n_maxadj = maxadj_sh;
    if (threadIdx.x != 0) return;
    int s1, s2, m1, m2, d1, d2;
    if (isB) {
        const int cb = B.c2[p];
        s1 = st.s1 - n_dec0;
        m1 = (st.max1 > 0 && n_maxadj == st.cnt1) ? st.max1 - 1 : st.max1;
        s2 = st.s2 - (cb > 0);
        m2 = (cb == st.max2 && st.cnt2 == 1) ? st.sec2 : st.max2;
        d1 = st.dx + 1;
        d2 = st.dy;
    } else {
        const int ca = B.c1[p];
        s2 = st.s2 - n_dec0;
        m2 = (st.max2 > 0 && n_maxadj == st.cnt2) ? st.max2 - 1 : st.max2;
        s1 = st.s1 - (ca > 0);
        m1 = (ca == st.max1 && st.cnt1 == 1) ? st.sec1 : st.max1;
        d1 = st.dx;
        d2 = st.dy + 1;
    }
    const double after = bfc_value(d1, d2, st.T + 1, s1, s2, m1 > m2 ? m1 : m2);
    out[p] = after - st.before;
}

// K2 (round 4: one launch for what were k_imp_count, k_imp_stats, k_imp_rows and k_imp_scan).  One workgroup per candidate row
// a (i = x_nb[a]; a == dx: i = x) streams row i ONCE: every entry is looked up in the table; an entry that is in row y only is
// a 4-cycle x - i - w - y (c1[a] grows, and c2 of the node it lands on: bfc_naive.py:26-29,36-37), an entry that is in
// row y at all rules the pair (i, that neighbour) out (sdrf_no_cuda.py:35: has_edge) — a bit in the row's bitmap, as do i == j,
// w == y and w == x.  The workgroup also settles the classes of its row and of a share of row y.  The LAST workgroup to
// finish closes the stage: statistics of c1 / c2, triangles, base curvature, exclusive scan of the rows' candidate counts.
__global__ void __launch_bounds__(256) k_imp_rows_count(RowView g, ImpBuf B, int x, int y, unsigned mask, int words, int curv_type,
                                                         DevResult *res) {
    extern __shared__ uint32_t bits[];
    __shared__ int cnt_sh, c1_sh, last_sh;
    __shared__ int shi[16 * 4];
    ImpStats *stp = B.st;
    const int dx = stp->dx, dy = stp->dy;
    const int pos_x_in_y = stp->pos_x_in_y;  // x is an endpoint, so it is not in the table
    const int a = blockIdx.x;
    const int2 rx = g.rowinfo[x], ry = g.rowinfo[y];
    const int i = a < dx ? g.col[rx.x + a] : x;
    for (int w = threadIdx.x; w < words; w += blockDim.x) bits[w] = 0u;
    // class of row a: 2 = the other endpoint, 1 = also in row y (a triangle node), 0 = in row x only.  Every thread looks i up
    // itself (one address: a broadcast) — behind one thread and a barrier the whole workgroup waited for two dependent reads.
    int cls = 3, b_i = -1;  // b_i: position of i in row y + [y] (the pair (i, i) is ruled out: sdrf_no_cuda.py:35, i != j)
    if (i == y) {
        cls = 2;
        b_i = dy;
    } else if (i == x) {
        b_i = pos_x_in_y;
    } else {
        const int h = g_find(B.keys, mask, i);
        if (h >= 0) b_i = B.posy[h];
        if (a < dx) cls = b_i >= 0 ? 1 : 0;
    }
    if (threadIdx.x == 0) {
        cnt_sh = 0;
        c1_sh = 0;
        // (what the closing workgroup reads of this one — clsx, rowcount, c1 — is stored through the L2 with agent-scope relaxed
        //  atomics: see the ticket below)
        if (a < dx) __hip_atomic_store(&B.clsx[a], cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool counting = curv_type == DCR_CURV_BFC && cls == 0;
    const int2 ri = g.rowinfo[i];
    int c = 0;
    for (int base = 0; base < ri.y; base += 4 * 256) {  // four look-ups in flight per thread (a hub's row is 1,400 entries)
        int w[4], h[4];
        bool in[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = base + 256 * q + (int)threadIdx.x;
            in[q] = t < ri.y;
            w[q] = in[q] ? g.col[ri.x + t] : -1;
            in[q] = in[q] && w[q] != x && w[q] != y;  // (the endpoints are not in the table)
        }
        g_find4(B.keys, mask, w, in, h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int b = -1;
            if (w[q] == y && base + 256 * q + (int)threadIdx.x < ri.y) b = dy;
            else if (w[q] == x && base + 256 * q + (int)threadIdx.x < ri.y) b = pos_x_in_y;
            else if (in[q] && h[q] >= 0) {
                b = B.posy[h[q]];
                if (counting && b >= 0 && B.posx[h[q]] < 0) {  // in N(y) only
                    ++c;
                    atomicAdd(&B.c2[b], 1);
                }
            }
            if (b >= 0) atomicOr(&bits[b >> 5], 1u << (b & 31));
        }
    }
    if (threadIdx.x == 0 && b_i >= 0) atomicOr(&bits[b_i >> 5], 1u << (b_i & 31));  // i == j
    // classes of a share of row y (every position is settled by exactly one workgroup; nothing in this kernel reads them)
    for (int b = a + (int)threadIdx.x * (int)gridDim.x; b < dy; b += (int)blockDim.x * (int)gridDim.x) {
        const int k = g.col[ry.x + b];
        if (k != x) {
            const int h = g_find(B.keys, mask, k);
            B.clsy[b] = (h >= 0 && B.posx[h] >= 0) ? 1 : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&c1_sh, c);
    __syncthreads();
    int adm = 0;
    const int nb = dy + 1;
    for (int w = threadIdx.x; w < words; w += blockDim.x) {
        uint32_t v = bits[w];
        const int lo = w * 32;
        const uint32_t live = (nb - lo >= 32) ? 0xFFFFFFFFu : ((1u << (nb - lo)) - 1u);
        adm += __popc(~v & live);
        B.adjbits[(size_t)a * words + w] = v;
    }
    for (int off = 32; off > 0; off >>= 1) adm += __shfl_xor(adm, off);
    if ((threadIdx.x & 63) == 0 && adm) atomicAdd(&cnt_sh, adm);
    __syncthreads();
    if (threadIdx.x == 0) {
        // Round 5: an agent-scope release fence writes back the whole L2 of its XCD (csrc/dcr_gcn_first.hip found 45 us of them in one
        // kernel; here 1,400 workgroups' fences were 11 of this kernel's 44 us: the round-4 timing-only build).  The three words the
        // closing workgroup reads of this one go through the L2 instead (sc1 stores), c2 is updated by device-scope atomics anyway,
        // and all that orders them against the ticket is the wait for the stores (a workgroup-scope release).
        __hip_atomic_store(&B.rowcount[a], cnt_sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a < dx) __hip_atomic_store(&B.c1[a], counting ? c1_sh : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        last_sh = atomicAdd(&stp->done_rows, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_sh) return;  // uniform
    // every other workgroup's results are visible from here on: they went through the L2 (sc1 stores, device-scope atomics) before
    // the tickets; this workgroup only has to drop what its own caches may hold (an agent-scope ACQUIRE: buffer_inv, no write-back
    // of the L2 as __threadfence() has it)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    imp_close_stats(B, curv_type, shi);
    // exclusive scan of rowcount[0 .. rows): the candidates are emitted row by row (sdrf_no_cuda.py:32-37: outer loop over x_nb)
    const int rows = (int)gridDim.x;
    __shared__ int wsum[4];
    __shared__ int carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < rows; base += 256) {
        const int r = base + (int)threadIdx.x;
        const int v = r < rows ? __hip_atomic_load(&B.rowcount[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const int carry = carry_sh;
        if (r < rows) B.rowoff[r] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_sh = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) res->n_cand = carry_sh;
}

// K7: write the admitted candidates of row a, in j order, with their improvements.
// (Round 4: the pipeline's last user of the hash table leaves it EMPTY for the next iteration — nothing here reads it, and the
//  separate clearing kernel at the head of the pipeline is gone.)
__global__ void __launch_bounds__(256) k_imp_emit(RowView g, ImpBuf B, int x, int y, int words, int curv_type,
                                                   double *out, int32_t *ci, int32_t *cj, int64_t ts) {
    __shared__ int wsum[4];
    __shared__ int carry_sh;
    const ImpStats st = *B.st;
    const int a = blockIdx.x;
    for (int64_t i = (int64_t)a * 256 + threadIdx.x; i < ts; i += (int64_t)gridDim.x * 256) {
        B.keys[i] = -1;
        B.posx[i] = -1;
        B.posy[i] = -1;
    }
    if (a == 0 && threadIdx.x == 0) B.st->pos_x_in_y = -1;
    if (B.rowcount[a] == 0) return;
    const int2 rx = g.rowinfo[x], ry = g.rowinfo[y];
    const int i = a < st.dx ? g.col[rx.x + a] : x;
    const int cls_a = a < st.dx ? B.clsx[a] : 3;
    const int c1a = (a < st.dx && cls_a == 0) ? B.c1[a] : 0;
    const int gam = st.max1 > st.max2 ? st.max1 : st.max2;
    const int nb = st.dy + 1;
    const int64_t row_base = B.rowoff[a];
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    // one thread per candidate position b (256 per round): its place in the output is the number of admitted positions
    // before it (ballots inside the wave, wave totals through LDS).  (One thread per 32-bit bitmap word, as this kernel
    // first did it, leaves 5 of 256 threads busy for a typical 150-neighbour endpoint, 32 closing expressions each.)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int bbase = 0; bbase < nb; bbase += 256) {
        const int b = bbase + threadIdx.x;
        bool ok = false;
        if (b < nb) ok = ((B.adjbits[(size_t)a * words + (b >> 5)] >> (b & 31)) & 1u) == 0u;
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wsum[wid] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wid; ++q) woff += wsum[q];
        const int carry = carry_sh;
        if (ok) {
            const int64_t o = row_base + carry + woff + __popcll(m & lt);
            const int j = b < st.dy ? g.col[ry.x + b] : y;
            double val;
            if (a == st.dx) {
                val = B.impb[b];  // i == x: admitted only for j in DY
            } else if (b == st.dy) {
                val = B.impc[a];  // j == y: admitted only for i in DX
            } else if (curv_type == DCR_CURV_BFC && cls_a == 0 && B.clsy[b] == 0) {
                const int c2b = B.c2[b];
                const int mm = (c1a + 1 > c2b + 1 ? c1a + 1 : c2b + 1);
                const double after = bfc_value(st.dx, st.dy, st.T, st.s1 + (c1a == 0), st.s2 + (c2b == 0),
                                               mm > gam ? mm : gam);
                val = after - st.before;
            } else {
                val = 0.0;
            }
            out[o] = val;
            ci[o] = i < j ? i : j;  // sorted((i, j)), sdrf_no_cuda.py:37
            cj[o] = i < j ? j : i;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_sh = carry + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

template <typename T>
static int pinned_regrow(T **p, int64_t *cap, int64_t need) {
    if (need <= *cap) return DCR_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    int64_t nc = need + need / 4 + 1024;
    void *q = nullptr;
    hipError_t e = hipHostMalloc(&q, (size_t)nc * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return DCR_ENOMEM;
    }
    *p = (T *)q;
    *cap = nc;
    return DCR_OK;
}

// the candidate drawn on the host, by index: (k, l) never leaves the device (dcr_sdrf_tail_at)
__global__ void k_pick_candidate(const int32_t *ci, const int32_t *cj, int64_t index, DevResult *res) {
    const int32_t a = ci[index], b = cj[index];
    res->cand_i = a < b ? a : b;
    res->cand_j = a < b ? b : a;
    res->draw_status = 0;
}

// ---- the draw of sdrf_no_cuda.py:49-50 on the device -------------------------------------------------------------------
// np.random.choice(n, p = softmax(improvements, tau)) is: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') with
// ONE uniform u from the global stream (the host draws it: it does not depend on anything computed here).  The index is the
// first i with cdf_i / t > u, i.e. with P_i > u · P_n for the exact prefix sums P of e_i = exp(tau · improvement_i): numpy's
// floating-point chain (its exp, the pairwise sum S, e / S, the sequential cumsum, the division by t) and the sums formed
// here each stay within (n + 64) · 2^-53 · P_n of those exact values, so when u · P_n is further than a margin of
// (n + 1024) · 2^-51 · P_n from both P_{i-1} and P_i the comparison numpy makes at i - 1 and at i has the outcome found here
// and i IS the index numpy returns.  When it is closer (probability ~ 1e-10 per draw), or anything is not finite, or there is
// no candidate, nothing is decided here: draw_status says so, the tail kernels leave the graph alone, and the host runs the
// exact path (numpy's own exp and sums) for this iteration.
constexpr int DRAW_BLOCKS = 256;

// tau = +inf: softmax is one-hot at the first arg-max of the improvements (utils/softmax.py:5-8), so the index np.random.choice
// returns is that arg-max whatever the uniform (which it still consumes: the caller has taken it).  Candidate count from the
// result block: nothing here needs a host value.
__global__ void __launch_bounds__(256) k_argmax_array_dev(const double *a, const DevResult *res, Ext *partial) {
    __shared__ double shv[4];
    __shared__ int shs[4];
    const int64_t n = res->n_cand;
    double best_v = 0.0;
    int best_s = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        ext_take(best_v, best_s, a[i], (int)i, 1);
    }
    ext_block_reduce(best_v, best_s, 1, shv, shs);
    if (threadIdx.x == 0) partial[blockIdx.x] = ext_make(best_v, best_s);
}

__global__ void k_draw_from_argmax(const int32_t *ci, const int32_t *cj, DevResult *res) {
    const int64_t n = res->n_cand, idx = res->imp_argmax;
    if (n <= 0) {
        res->draw_status = 2;
        res->draw_idx = -1;
        return;
    }
    if (idx < 0 || idx >= n) {  // (not a number anywhere: the host path decides what numpy does with it)
        res->draw_status = 1;
        res->draw_idx = -1;
        return;
    }
    const int32_t a = ci[idx], b = cj[idx];
    res->cand_i = a < b ? a : b;
    res->cand_j = a < b ? b : a;
    res->draw_idx = idx;
    res->draw_status = 0;
}

__device__ void draw_pick_block(const double *__restrict__ imp, const int32_t *__restrict__ ci, const int32_t *__restrict__ cj,
                                DevResult *res, double tau, double u, const double *bsum, double margin_scale);
__device__ void draw_block_sum(const double *__restrict__ imp, const DevResult *res, double tau, double *__restrict__ bsum);

// Block sums of exp(tau * improvement); the LAST block to finish picks the index (round 4: k_draw_pick was a launch of its own).
__global__ void __launch_bounds__(256) k_draw_partial(const double *__restrict__ imp, const int32_t *__restrict__ ci,
                                                       const int32_t *__restrict__ cj, DevResult *res, double tau, double u,
                                                       double *bsum, double margin_scale, ImpStats *st) {
    __shared__ int last_sh;
    draw_block_sum(imp, res, tau, bsum);
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the block sum's store (sc1) has completed before the ticket is taken
        last_sh = atomicAdd(&st->done_draw, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_sh) return;  // uniform
    // (no agent fence: the block sums are read through the L2 — sc1 loads in draw_pick_block — and everything else this
    //  workgroup reads was written by earlier kernels)
    draw_pick_block(imp, ci, cj, res, tau, u, bsum, margin_scale);
}

__device__ void draw_block_sum(const double *__restrict__ imp, const DevResult *res, double tau, double *__restrict__ bsum) {
    __shared__ double red[256];
    const int64_t n = res->n_cand;
    const int64_t L = (n + DRAW_BLOCKS - 1) / DRAW_BLOCKS, l = (L + 255) / 256;
    const int64_t b0 = (int64_t)blockIdx.x * L, t0 = b0 + (int64_t)threadIdx.x * l;
    const int64_t bend = b0 + L < n ? b0 + L : n;
    const int64_t t1 = t0 + l < bend ? t0 + l : bend;
    double s = 0.0;
    for (int64_t k = t0; k < t1; ++k) s += exp(imp[k] * tau);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) __hip_atomic_store(&bsum[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (through the L2: no agent fence below)
}

__device__ void draw_pick_block(const double *__restrict__ imp, const int32_t *__restrict__ ci, const int32_t *__restrict__ cj,
                                DevResult *res, double tau, double u, const double *bsum, double margin_scale) {
    __shared__ double pre[256];
    __shared__ int found;
    const int t = threadIdx.x;
    const int64_t n = res->n_cand;
    if (n <= 0) {
        if (t == 0) {
            res->draw_status = 2;
            res->draw_idx = -1;
        }
        return;
    }
    // inclusive prefix of the block sums (256 values: one thread adds them in order)
    pre[t] = __hip_atomic_load(&bsum[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (other workgroups' sums)
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int k = 0; k < DRAW_BLOCKS; ++k) {
            a += pre[k];
            pre[k] = a;
        }
        found = DRAW_BLOCKS;
    }
    __syncthreads();
    const double total = pre[DRAW_BLOCKS - 1];
    const double T = u * total;
    // (false for NaN too; far from the subnormal range, where the relative error bounds behind the margin do not hold)
    const bool fine = total > 1e-250 && total < 1e300;
    if (fine && pre[t] > T) atomicMin(&found, t);
    __syncthreads();
    const int bs = found;
    if (!fine || bs >= DRAW_BLOCKS) {
        if (t == 0) {
            res->draw_status = 1;
            res->draw_idx = -1;
            res->draw_total = total;
        }
        return;
    }
    const double base = bs == 0 ? 0.0 : pre[bs - 1];
    __syncthreads();
    // inside block bs: the threads' segment sums, then the segment that crosses T is walked element by element
    const int64_t L = (n + DRAW_BLOCKS - 1) / DRAW_BLOCKS, l = (L + 255) / 256;
    const int64_t b0 = (int64_t)bs * L, t0 = b0 + (int64_t)t * l;
    const int64_t bend = b0 + L < n ? b0 + L : n;
    const int64_t t1 = t0 + l < bend ? t0 + l : bend;
    double s = 0.0;
    for (int64_t k = t0; k < t1; ++k) s += exp(imp[k] * tau);
    pre[t] = s;
    __syncthreads();
    if (t == 0) {
        const double margin = (double)(n + 1024) * 0x1p-51 * total * margin_scale;
        double run = base;
        int64_t idx = -1;
        double below = 0.0, above = 0.0;
        for (int k = 0; k < 256 && idx < 0; ++k) {
            if (run + pre[k] > T) {
                const int64_t s0 = b0 + (int64_t)k * l, s1 = s0 + l < bend ? s0 + l : bend;
                for (int64_t q = s0; q < s1; ++q) {
                    const double prev = run;
                    run += exp(imp[q] * tau);
                    if (run > T) {
                        idx = q;
                        below = T - prev;
                        above = run - T;
                        break;
                    }
                }
                break;  // (not found inside a segment whose sum crosses: rounding; left undecided)
            }
            run += pre[k];
        }
        const bool ok = idx >= 0 && above > margin && (idx == 0 || below > margin);
        res->draw_total = total;
        res->draw_gap = idx >= 0 ? (above < below || idx == 0 ? above : below) / total : 0.0;
        res->draw_idx = ok ? idx : -1;
        res->draw_status = ok ? 0 : 1;
        if (ok) {
            const int32_t a = ci[idx], b = cj[idx];
            res->cand_i = a < b ? a : b;
            res->cand_j = a < b ? b : a;
        }
    }
}

}  // namespace dcr

using namespace dcr;

// the improvement pipeline of one edge (x, y), enqueued on the library stream: candidates and their improvements end up in
// g->imp_out / imp_ci / imp_cj, their number in the result block (n_cand); *upper_out = the bound (dx + 1)(dy + 1) on it
// amax_from_parts: the stale arg-max of the removal step reduced from the pass's partial maxima by one more workgroup of the
// first launch (the caller has checked g->ext_part_valid and sets g->amax_valid)
static int imp_enqueue(dcr_graph *g, int32_t x, int32_t y, int curv_type, int64_t *upper_out, bool amax_from_parts = false) {
    int dx, dy;
    if (g->am_valid && g->am_x == x && g->am_y == y) {  // the arg-min step already brought the degrees over
        dx = g->am_dx;
        dy = g->am_dy;
    } else {
        int2 rxy[2];
        DCR_HIP(hipMemcpyAsync(&rxy[0], g->rowinfo + x, sizeof(int2), hipMemcpyDeviceToHost, g->stream));
        DCR_HIP(hipMemcpyAsync(&rxy[1], g->rowinfo + y, sizeof(int2), hipMemcpyDeviceToHost, g->stream));
        DCR_HIP(hipStreamSynchronize(g->stream));
        dx = rxy[0].y;
        dy = rxy[1].y;
    }
    const int64_t keys = (int64_t)dx + dy;
    int64_t ts = 64;
    while (ts < 4 * keys) ts <<= 1;
    const unsigned mask = (unsigned)(ts - 1);
    const int rows = dx + 1;
    const int words = (dy + 1 + 31) / 32;
    if ((size_t)words * 4 > 150 * 1024) DCR_FAIL(DCR_ECAPACITY, "deg(y) too large for the candidate bitmap");
    const int64_t upper = (int64_t)(dx + 1) * (dy + 1);
    if (upper > INT32_MAX) DCR_FAIL(DCR_ECAPACITY, "more than 2^31 candidate pairs");

    // scratch
    if (ts > g->imp_table_cap) {
        for (int32_t **p : {&g->imp_table, &g->imp_posx, &g->imp_posy}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
            DCR_TRY(dev_alloc(p, ts));
        }
        g->imp_table_cap = ts;
        g->imp_table_dirty = true;
    }
    const int64_t rows_need = (int64_t)(dx > dy ? dx : dy) + 2;
    if (rows_need > g->imp_rows_cap) {
        const int64_t nc = 2 * rows_need + 64;
        for (int32_t **p : {&g->imp_c1, &g->imp_c2, &g->imp_rowcount, &g->imp_rowoff, &g->scan_a, &g->scan_b}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
            DCR_TRY(dev_alloc(p, nc));
        }
        for (double **p : {&g->imp_b, &g->imp_c}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
            DCR_TRY(dev_alloc(p, nc));
        }
        g->imp_rows_cap = nc;
    }
    if ((int64_t)rows * words > g->imp_bits_cap) DCR_TRY(dev_regrow(&g->imp_adjbits, &g->imp_bits_cap, 2 * (int64_t)rows * words));
    if (upper > g->imp_out_cap) {
        if (g->imp_out) (void)hipFree(g->imp_out);
        if (g->imp_ci) (void)hipFree(g->imp_ci);
        if (g->imp_cj) (void)hipFree(g->imp_cj);
        g->imp_out = nullptr;
        g->imp_ci = g->imp_cj = nullptr;
        const int64_t nc = 2 * upper + 1024;   // (doubling: a re-allocation is a device synchronisation + two driver calls, ~0.3 ms —
                                               //  a few of them inside a 20-iteration measurement are 1-2 % of it)
        DCR_TRY(dev_alloc(&g->imp_out, nc));
        DCR_TRY(dev_alloc(&g->imp_ci, nc));
        DCR_TRY(dev_alloc(&g->imp_cj, nc));
        g->imp_out_cap = nc;
    }

    ImpBuf B;
    B.keys = g->imp_table;
    B.posx = g->imp_posx;
    B.posy = g->imp_posy;
    B.c1 = g->imp_c1;
    B.c2 = g->imp_c2;
    B.clsx = g->scan_a;
    B.clsy = g->scan_b;
    B.impb = g->imp_b;
    B.impc = g->imp_c;
    B.rowcount = g->imp_rowcount;
    B.rowoff = g->imp_rowoff;
    B.adjbits = g->imp_adjbits;
    B.st = g->imp_stats;
    RowView vw{g->rowinfo, g->col, g->slot_row};

    // Five launches (round 4; eleven before): insert both rows | per candidate row: classes, 4-cycle counters, admissibility
    // bitmap, and the stage's closing statistics + scan by its last workgroup | classes B and C | emit (and leave the table
    // empty) | the draw (its caller).  The table is only cleared here when the previous pipeline did not leave it empty.
    if (g->imp_table_dirty) {
        const unsigned blocks = (unsigned)((g->imp_table_cap + 255) / 256 > 1024 ? 1024 : (g->imp_table_cap + 255) / 256);
        hipLaunchKernelGGL(k_imp_clear, dim3(blocks ? blocks : 1), dim3(256), 0, g->stream, B, g->imp_table_cap);
    }
    g->imp_table_dirty = true;  // (until k_imp_emit below has been enqueued)
    const int gi = (dx + dy + 255) / 256 > 0 ? (dx + dy + 255) / 256 : 1;
    const Ext *amax_parts = amax_from_parts ? (const Ext *)g->ext_part + EXT_PART_BLOCKS : nullptr;
    hipLaunchKernelGGL(k_imp_insert, dim3(gi + (amax_parts ? 1 : 0)), dim3(256), 0, g->stream, vw, B, x, y, mask, amax_parts,
                       amax_parts ? g->ext_part_n : 0, g->dres);
    hipLaunchKernelGGL(k_imp_rows_count, dim3(rows), dim3(256), (size_t)words * 4, g->stream, vw, B, x, y, mask, words, curv_type,
                       g->dres);
    if (dx + dy > 0)
        hipLaunchKernelGGL(k_imp_bc, dim3(dx + dy), dim3(256), 0, g->stream, vw, B, x, y, mask, curv_type);
    hipLaunchKernelGGL(k_imp_emit, dim3(rows), dim3(256), 0, g->stream, vw, B, x, y, words, curv_type, g->imp_out,
                       g->imp_ci, g->imp_cj, ts);
    g->imp_table_dirty = false;
    DCR_HIP(hipGetLastError());
    *upper_out = upper;
    return DCR_OK;
}

extern "C" {

int dcr_argext(dcr_graph *g, int want_max, int32_t excl_u, int32_t excl_v, int32_t *out_u, int32_t *out_v,
               double *out_val) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (!g->curv_valid) DCR_FAIL(DCR_ESTATE, "dcr_argext needs a curvature pass first");
    DCR_HIP(hipSetDevice(g->device));
    if (excl_u > excl_v) {
        int32_t t = excl_u;
        excl_u = excl_v;
        excl_v = t;
    }
    DCR_TRY(launch_argext(g, want_max, excl_u, excl_v));
    DCR_TRY(sync_result(g));
    if (g->hres->ext_slot < 0) DCR_FAIL(DCR_ENOTFOUND, "graph has no (eligible) edges");
    if (out_u) *out_u = g->hres->ext_u;
    if (out_v) *out_v = g->hres->ext_v;
    if (out_val) *out_val = g->hres->ext_val;
    return DCR_OK;
}

int dcr_improvements(dcr_graph *g, int32_t x, int32_t y, int curv_type, int want_candidates, int64_t *n_out,
                     const double **out_improvement, const int32_t **out_ci, const int32_t **out_cj) {
    if (!g || !n_out) DCR_FAIL(DCR_EINVAL, "null argument");
    if (x < 0 || y < 0 || x >= g->n || y >= g->n || x == y) DCR_FAIL(DCR_EINVAL, "bad node ids");
    if (curv_type < DCR_CURV_BFC || curv_type > DCR_CURV_HAANTJES) DCR_FAIL(DCR_EINVAL, "unknown curvature type");
    DCR_HIP(hipSetDevice(g->device));
#ifdef DCR_IMP_TIMING
    static double t_enq = 0, t_sync = 0; static long t_n = 0;
    const auto T0 = std::chrono::steady_clock::now();
#endif
    int64_t upper = 0;
    DCR_TRY(imp_enqueue(g, x, y, curv_type, &upper));
    // one host sync: the result block and the values go out together; the candidate count is not known yet, so the
    // copy is sized by its upper bound (dx+1)(dy+1), which the real count nearly reaches on a sparse graph
    if (upper > 0 && out_improvement) {
        DCR_TRY(pinned_regrow(&g->imp_out_h, &g->imp_out_h_cap, upper));
        DCR_HIP(hipMemcpyAsync(g->imp_out_h, g->imp_out, sizeof(double) * (size_t)upper, hipMemcpyDeviceToHost, g->stream));
    }
    if (upper > 0 && want_candidates) {
        if (upper > g->imp_cand_h_cap) {
            int64_t c1 = g->imp_cand_h_cap, c2 = g->imp_cand_h_cap;
            DCR_TRY(pinned_regrow(&g->imp_ci_h, &c1, upper));
            DCR_TRY(pinned_regrow(&g->imp_cj_h, &c2, upper));
            g->imp_cand_h_cap = c1 < c2 ? c1 : c2;
        }
        DCR_HIP(hipMemcpyAsync(g->imp_ci_h, g->imp_ci, sizeof(int32_t) * (size_t)upper, hipMemcpyDeviceToHost, g->stream));
        DCR_HIP(hipMemcpyAsync(g->imp_cj_h, g->imp_cj, sizeof(int32_t) * (size_t)upper, hipMemcpyDeviceToHost, g->stream));
    }
#ifdef DCR_IMP_TIMING
    const auto T1 = std::chrono::steady_clock::now();
#endif
    DCR_TRY(sync_result(g));
#ifdef DCR_IMP_TIMING
    const auto T2 = std::chrono::steady_clock::now();
    t_enq += std::chrono::duration<double, std::micro>(T1 - T0).count();
    t_sync += std::chrono::duration<double, std::micro>(T2 - T1).count();
    if (++t_n % 50 == 0) fprintf(stderr, "[imp timing] enqueue %.1f us, sync wait %.1f us (avg of %ld)\n", t_enq / t_n, t_sync / t_n, t_n);
#endif
    const int64_t n = g->hres->n_cand;
    if (n < 0 || n > upper) DCR_FAIL(DCR_ESTATE, "candidate count outside its bound");
    g->imp_n = n;
    *n_out = n;
    // The removal step looks for the highest STALE curvature (sdrf_no_cuda.py:57-61), which does not depend on the edge
    // about to be drawn (that one is excluded there only because it has no stale value): compute it now, while the host
    // draws, and let dcr_sdrf_tail* pick it up.  Any pass or edit in between drops it.
    if (g->curv_valid && n > 0) {
        if (g->ext_part_valid) DCR_TRY(launch_argext_from_parts(g, 1));  // left by the two-hop pass's closing kernel
        else DCR_TRY(launch_argext(g, 1, -1, -1));
        g->amax_valid = true;
    }
    if (out_improvement) *out_improvement = n > 0 ? g->imp_out_h : nullptr;
    if (out_ci) *out_ci = (n > 0 && want_candidates) ? g->imp_ci_h : nullptr;
    if (out_cj) *out_cj = (n > 0 && want_candidates) ? g->imp_cj_h : nullptr;
    return DCR_OK;
}

int dcr_improvements_argmax(dcr_graph *g, int64_t *out_index) {
    if (!g || !out_index) DCR_FAIL(DCR_EINVAL, "null argument");
    if (g->imp_n <= 0) DCR_FAIL(DCR_ESTATE, "no candidates from the last dcr_improvements call");
    DCR_HIP(hipSetDevice(g->device));
    if (!g->red_scratch) {
        Ext *p = nullptr;
        DCR_TRY(dev_alloc(&p, ARGEXT_BLOCKS));
        g->red_scratch = p;
    }
    int64_t blocks = (g->imp_n + 255) / 256;
    if (blocks > ARGEXT_BLOCKS) blocks = ARGEXT_BLOCKS;
    hipLaunchKernelGGL(k_argmax_array, dim3((unsigned)blocks), dim3(256), 0, g->stream, g->imp_out, g->imp_n,
                       (Ext *)g->red_scratch, (int64_t *)nullptr);
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, g->stream, (const Ext *)g->red_scratch, (int)blocks,
                       g->dres);
    DCR_HIP(hipGetLastError());
    DCR_TRY(sync_result(g));
    *out_index = g->hres->imp_argmax;
    return DCR_OK;
}

int dcr_candidate_at(dcr_graph *g, int64_t index, int32_t *out_i, int32_t *out_j) {
    if (!g || !out_i || !out_j) DCR_FAIL(DCR_EINVAL, "null argument");
    if (index < 0 || index >= g->imp_n) DCR_FAIL(DCR_EINVAL, "candidate index out of range");
    DCR_HIP(hipSetDevice(g->device));
    DCR_HIP(hipMemcpyAsync(out_i, g->imp_ci + index, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipMemcpyAsync(out_j, g->imp_cj + index, sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    return DCR_OK;
}

// The tail of an iteration in two halves, so that the NEXT iteration's curvature pass can be enqueued behind it without a
// host round trip in between (dcr_sdrf_tail_at_pass_argmin): enqueue (add, dirty flags, stale arg-max, conditional remove)
// and finish (read the result block after some later synchronisation).
struct TailCall {
    int32_t add_k, add_l;
    int do_remove;
    double bound;
    bool adding, have_amax;
    int edit_add, edit_rem;
};

static int tail_prepare(dcr_graph *g, int32_t add_k, int32_t add_l, int do_remove, double removal_bound, TailCall *tc) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (do_remove && !g->curv_valid) DCR_FAIL(DCR_ESTATE, "removal needs a curvature pass first");
    tc->adding = add_k >= 0 || add_k == -2;  // -2: the pair k_pick_candidate left in the result block
    if (add_k >= 0) {
        if (add_l < 0 || add_k >= g->n || add_l >= g->n || add_k == add_l) DCR_FAIL(DCR_EINVAL, "bad edge to add");
        if (add_k > add_l) {
            int32_t t = add_k;
            add_k = add_l;
            add_l = t;
        }
    }
    tc->add_k = add_k;
    tc->add_l = add_l;
    tc->do_remove = do_remove;
    tc->bound = removal_bound;
    DCR_HIP(hipSetDevice(g->device));
    g->am_valid = false;
    tc->have_amax = g->amax_valid;  // computed on this graph before the add: nothing to exclude
    g->amax_valid = false;
    // edit numbers for the incremental pass's flags (edge_dirty): the add, then the removal
    tc->edit_add = g->pending_edits;
    tc->edit_rem = g->pending_edits + (tc->adding ? 1 : 0);
    g->pending_edits += (tc->adding ? 1 : 0) + (do_remove ? 1 : 0);
    return DCR_OK;
}

static int tail_enqueue(dcr_graph *g, const TailCall &tc, bool first_attempt) {
    if (tc.adding && (!tc.do_remove || (tc.have_amax && first_attempt))) {  // nothing to compute in between: one launch
        launch_sdrf_tail(g, tc.add_k, tc.add_l, tc.edit_add, tc.do_remove, tc.bound, tc.edit_rem);
        DCR_HIP(hipGetLastError());
        return DCR_OK;
    }
    launch_add_edge(g, tc.add_k, tc.add_l);
    launch_mark_dirty(g, tc.add_k, tc.add_l, tc.edit_add);  // after the append: the new neighbours are flagged too
    if (tc.do_remove) {
        if (!(tc.have_amax && first_attempt))
            DCR_TRY(launch_argext(g, 1, tc.adding ? tc.add_k : -1, tc.adding ? tc.add_l : -1));
        launch_remove_if_above(g, tc.bound, tc.edit_rem);
    }
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// after a synchronisation that brought the result block over: host-side bookkeeping and outputs
static void tail_finish(dcr_graph *g, const TailCall &tc, int32_t out_removed[2], double *out_max_val, bool max_val_known) {
    if (tc.adding && g->hres->add_status == 0) {
        g->n_edges++;
        g->max_deg_bound++;
    }
    int32_t ru = -1, rv = -1;
    if (tc.do_remove) {
        ru = g->hres->removed_u;
        rv = g->hres->removed_v;
        if (ru >= 0) g->n_edges--;
        if (out_max_val) *out_max_val = (max_val_known && g->hres->ext_slot >= 0) ? g->hres->ext_val : 0.0;
    }
    if (out_removed) {
        out_removed[0] = ru;
        out_removed[1] = rv;
    }
}

static int sdrf_tail_impl(dcr_graph *g, int32_t add_k, int32_t add_l, int do_remove, double removal_bound,
                          int32_t out_removed[2], double *out_max_val) {
    TailCall tc;
    DCR_TRY(tail_prepare(g, add_k, add_l, do_remove, removal_bound, &tc));
    for (int attempt = 0; attempt < 2; ++attempt) {
        DCR_TRY(tail_enqueue(g, tc, attempt == 0));
        DCR_TRY(sync_result(g));
        if (g->hres->add_status != 1) break;
        if (attempt == 1) DCR_FAIL(DCR_ECAPACITY, "row still full after relayout");
        DCR_TRY(relayout(g));
    }
    tail_finish(g, tc, out_removed, out_max_val, true);
    return DCR_OK;
}

int dcr_sdrf_tail_at(dcr_graph *g, int64_t cand_index, int do_remove, double removal_bound, int32_t out_added[2],
                     int32_t out_removed[2], double *out_max_val) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (cand_index < 0 || cand_index >= g->imp_n) DCR_FAIL(DCR_EINVAL, "candidate index out of range");
    DCR_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_pick_candidate, dim3(1), dim3(1), 0, g->stream, g->imp_ci, g->imp_cj, cand_index, g->dres);
    DCR_TRY(sdrf_tail_impl(g, -2, -2, do_remove, removal_bound, out_removed, out_max_val));
    if (out_added) {
        out_added[0] = g->hres->cand_i;
        out_added[1] = g->hres->cand_j;
    }
    return DCR_OK;
}

int dcr_sdrf_tail(dcr_graph *g, int32_t add_k, int32_t add_l, int do_remove, double removal_bound,
                  int32_t out_removed[2], double *out_max_val) {
    return sdrf_tail_impl(g, add_k, add_l, do_remove, removal_bound, out_removed, out_max_val);
}

// Tail of iteration i and the head of iteration i + 1 in one call with ONE host synchronisation: add the drawn
// candidate, conditional removal (sdrf_no_cuda.py:51,56-66), then the curvature pass of the next iteration and its first
// minimum (:24,:27).  The pass is enqueued right behind the edit; the result block carries the tail's outcome and the
// arg-min over together.  (A row overflow of the add — rare: rows carry slack — is seen only then: the rows are laid out
// again, the tail replayed and the pass redone.)
int dcr_sdrf_tail_at_pass_argmin(dcr_graph *g, int64_t cand_index, int do_remove, double removal_bound, int curv_type,
                                 int incremental, int32_t out_added[2], int32_t out_removed[2], int32_t *out_u, int32_t *out_v,
                                 double *out_val) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (cand_index < 0 || cand_index >= g->imp_n) DCR_FAIL(DCR_EINVAL, "candidate index out of range");
    DCR_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_pick_candidate, dim3(1), dim3(1), 0, g->stream, g->imp_ci, g->imp_cj, cand_index, g->dres);
    TailCall tc;
    DCR_TRY(tail_prepare(g, -2, -2, do_remove, removal_bound, &tc));
    for (int attempt = 0; attempt < 2; ++attempt) {
        DCR_TRY(tail_enqueue(g, tc, attempt == 0));
        // the pass below sizes its launches by this upper bound (which class kernels run at all): count the pending add in,
        // on the replay too — relayout() has just reset the bound to the exact maximum BEFORE the add
        g->max_deg_bound++;
        const int rc = dcr_curvature_pass_argmin(g, curv_type, incremental, out_u, out_v, out_val);  // synchronises
        g->max_deg_bound--;  // (tail_finish counts it once the add is known to have happened)
        if (g->hres->add_status != 1) {
            if (rc != DCR_OK) {
                // the edit did happen on the device (the result block is from a completed synchronisation unless the
                // failure was the HIP call itself): keep the host's edge count and degree bound in step before reporting
                if (rc != DCR_EHIP) tail_finish(g, tc, out_removed, nullptr, false);
                return rc;
            }
            break;
        }
        if (attempt == 1) DCR_FAIL(DCR_ECAPACITY, "row still full after relayout");
        DCR_TRY(relayout(g));  // nothing was edited (the removal is skipped when the add overflows): lay out, replay
    }
    tail_finish(g, tc, out_removed, nullptr, false);
    if (out_added) {
        out_added[0] = g->hres->cand_i;
        out_added[1] = g->hres->cand_j;
    }
    return DCR_OK;
}

// One whole iteration of the loop body for finite tau without moving the improvements to the host: the improvement pipeline
// of the edge (x, y) found by the previous call, the draw on the device (k_draw_*: the host supplies the uniform it has taken
// from numpy's stream), the tail (add, conditional removal) and the NEXT iteration's curvature pass and first minimum
// (sdrf_no_cuda.py:29-66, then :24,:27).  Two host round trips, each carrying the 1 KB result block only.  *out_status: 0
// done; 1 the draw was left undecided, 2 there was no candidate — in both cases NOTHING was edited (and no pass run) and the
// caller runs the iteration through dcr_improvements / dcr_sdrf_tail* instead.
int dcr_sdrf_iteration_device_draw(dcr_graph *g, int32_t x, int32_t y, int curv_type, double tau, double uniform, int do_remove,
                                   double removal_bound, int incremental, int *out_status, int64_t *out_n_cand,
                                   int32_t out_added[2], int32_t out_removed[2], int32_t *out_u, int32_t *out_v, double *out_val) {
    if (!g || !out_status || !out_n_cand) DCR_FAIL(DCR_EINVAL, "null argument");
    if (x < 0 || y < 0 || x >= g->n || y >= g->n || x == y) DCR_FAIL(DCR_EINVAL, "bad node ids");
    if (curv_type < DCR_CURV_BFC || curv_type > DCR_CURV_HAANTJES) DCR_FAIL(DCR_EINVAL, "unknown curvature type");
    const bool tau_inf = tau > 1.7e308;  // +inf: the first arg-max
    if (!(tau == tau) || tau < -1.7e308 || !(uniform >= 0.0 && uniform < 1.0))
        DCR_FAIL(DCR_EINVAL, "device draw: tau finite or +inf and a uniform in [0, 1) expected");
    if (do_remove && !g->curv_valid) DCR_FAIL(DCR_ESTATE, "removal needs a curvature pass first");
    DCR_HIP(hipSetDevice(g->device));
    // the stale arg-max of the removal step does not depend on the edge about to be drawn (see dcr_improvements), nor on the
    // improvement pipeline: beside it, on a stream of its own (it writes other fields of the result block)
    const bool amax = g->curv_valid && do_remove;
    const bool amax_parts = amax && g->ext_part_valid;  // one small reduction over what the pass's closing kernel left: in line
    if (amax && !amax_parts) {
        DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
        DCR_HIP(hipStreamWaitEvent(g->side[0], g->ev_fork, 0));
        DCR_TRY(launch_argext(g, 1, -1, -1, g->side[0]));
        DCR_HIP(hipEventRecord(g->ev_join[0], g->side[0]));
    }
    int64_t upper = 0;
    if (amax_parts) g->amax_valid = false;  // the ext fields of the result block are about to be overwritten
    DCR_TRY(imp_enqueue(g, x, y, curv_type, &upper, amax_parts));
    if (amax_parts) {
        g->amax_valid = true;   // (reduced by one more workgroup of the pipeline's first launch)
    } else if (amax) {
        DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[0], 0));
        g->amax_valid = true;
    }
    // (DCR_DRAW_MARGIN_SCALE widens the margin: the tests use it to send draws down the undecided path)
    const double margin_scale = getenv("DCR_DRAW_MARGIN_SCALE") ? atof(getenv("DCR_DRAW_MARGIN_SCALE")) : 1.0;
    if (tau_inf && margin_scale <= 1.0) {
        if (!g->red_scratch) {
            Ext *p = nullptr;
            DCR_TRY(dev_alloc(&p, ARGEXT_BLOCKS));
            g->red_scratch = p;
        }
        // (red_scratch is free again: the stale arg-max beside the pipeline has been joined above)
        hipLaunchKernelGGL(k_argmax_array_dev, dim3(256), dim3(256), 0, g->stream, g->imp_out, g->dres, (Ext *)g->red_scratch);
        hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, g->stream, (const Ext *)g->red_scratch, 256, g->dres);
        hipLaunchKernelGGL(k_draw_from_argmax, dim3(1), dim3(1), 0, g->stream, g->imp_ci, g->imp_cj, g->dres);
    } else {
        hipLaunchKernelGGL(k_draw_partial, dim3(DRAW_BLOCKS), dim3(256), 0, g->stream, g->imp_out, g->imp_ci, g->imp_cj, g->dres, tau,
                           uniform, g->draw_bsum, margin_scale >= 1.0 ? margin_scale : 1.0, g->imp_stats);
    }
    // A host round trip here, without any transfer or host arithmetic: enqueuing the tail and the pass behind a stream that
    // is still working through the small kernels above cost 0.2 ms per iteration more than enqueuing them on an idle one
    // (measured, interleaved in one run: 1.89 against 1.59 ms), and the draw's verdict comes over with it, so an undecided
    // draw costs no wasted pass.
    // (Round 5: not in front of an INCREMENTAL pass — three launches on this one stream, 0.08 ms: there the round trip costs a
    //  tenth of the iteration and saves nothing; an undecided draw makes the tail a no-op (it reads the verdict in the result
    //  block), the pass then finds nothing flagged, and the verdict comes over with the pass's result.)
    static const int sync_env = getenv("DCR_DRAW_SYNC") ? atoi(getenv("DCR_DRAW_SYNC")) : -1;   // 1: always, 0: never (A/B aid)
    if (sync_env < 0 ? !incremental : sync_env != 0) {
        DCR_TRY(sync_result(g));
        g->imp_n = g->hres->n_cand;
        *out_n_cand = g->hres->n_cand;
        *out_status = g->hres->draw_status;
        if (g->hres->draw_status != 0) {
            if (out_removed) out_removed[0] = out_removed[1] = -1;
            if (out_added) out_added[0] = out_added[1] = -1;
            return DCR_OK;
        }
    }
    TailCall tc;
    DCR_TRY(tail_prepare(g, -2, -2, do_remove, removal_bound, &tc));
    for (int attempt = 0; attempt < 2; ++attempt) {
        DCR_TRY(tail_enqueue(g, tc, attempt == 0));
        g->max_deg_bound++;  // (see dcr_sdrf_tail_at_pass_argmin)
        const int rc = dcr_curvature_pass_argmin(g, curv_type, incremental, out_u, out_v, out_val);  // synchronises
        g->max_deg_bound--;
        if (g->hres->add_status != 1) {
            if (rc != DCR_OK) {
                if (rc != DCR_EHIP && g->hres->add_status != 3) tail_finish(g, tc, out_removed, nullptr, false);
                return rc;
            }
            break;
        }
        if (attempt == 1) DCR_FAIL(DCR_ECAPACITY, "row still full after relayout");
        DCR_TRY(relayout(g));
    }
    g->imp_n = g->hres->n_cand;
    *out_n_cand = g->hres->n_cand;
    *out_status = g->hres->draw_status;
    if (g->hres->draw_status != 0 || g->hres->add_status == 3) {  // nothing was edited; the pass ran on the same graph
        if (g->hres->draw_status == 0) DCR_FAIL(DCR_ESTATE, "device draw: edit skipped without a draw status");
        if (out_removed) out_removed[0] = out_removed[1] = -1;
        if (out_added) out_added[0] = out_added[1] = -1;
        return DCR_OK;
    }
    tail_finish(g, tc, out_removed, nullptr, false);
    if (out_added) {
        out_added[0] = g->hres->cand_i;
        out_added[1] = g->hres->cand_j;
    }
    return DCR_OK;
}

}  // extern "C"
