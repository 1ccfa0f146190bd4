// Edges beyond every LDS table: both endpoints with more than 8,190 neighbours, or one with more than 16,382
// (deg(u) + deg(v) + 2 > 16,384 keys).  They are rare (hub-to-hub edges) but a graph may have them, and the reference
// (curvature/bfc_naive.py:7-40) has no such limit, so they get a path of their own with the scratch in device memory:
// a position map over ALL node ids instead of a hash set, one edge at a time, every phase a full-grid kernel.
//
//   mark    pos[k] = i + 1 for the i-th member k of N(a), a = the endpoint with more neighbours; counters cleared
//   sweep   members of N(b): in N(a) -> triangle (T, bfc_naive.py:25) and flagged; b itself flagged
//   stream  the rows of DY = N(b) \ N(a) \ {a}, a workgroup per row: an unflagged hit z closes a 4-cycle a-z-w-b;
//           hits per row = |N(w) ∩ DX|, hits per z (its counter) = |N(z) ∩ DY|  (bfc_naive.py:26-29, 36-37)
//   finish  closing expression with the shared bfc_formula (left-to-right float64), written to the edge's slot
//   unmark  pos back to zero
#include "dcr_bfc_common.h"
#include "dcr_internal.h"

namespace dcr {

struct GiantAcc {
    int T, s_table, s_rows, gam;
};

constexpr unsigned GFLAG = 0x80000000u;

__global__ void __launch_bounds__(256) k_giant_mark(const int2 *rowinfo, const int32_t *col, int n, int a, int b,
                                                     int32_t *pos, unsigned *cnt, GiantAcc *acc) {
    const int2 ra = rowinfo[a];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < ra.y; i += nth) {
        const int k = col[ra.x + i];
        if (k >= 0 && k < n) pos[k] = (int32_t)i + 1;
        cnt[i] = 0u;
    }
    if (tid == 0) *acc = GiantAcc{0, 0, 0, 0};
}

__global__ void __launch_bounds__(256) k_giant_sweep(const int2 *rowinfo, const int32_t *col, int n, int a, int b,
                                                      const int32_t *pos, unsigned *cnt, GiantAcc *acc) {
    const int2 rb = rowinfo[b];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    int t = 0;
    for (int64_t j = tid; j < rb.y; j += nth) {
        const int k = col[rb.x + j];
        if (k < 0 || k >= n || k == a) continue;
        const int p = pos[k];
        if (p) {
            ++t;
            cnt[p - 1] = GFLAG;  // common neighbour: never counted as a 4-cycle corner
        }
    }
    if (t) atomicAdd(&acc->T, t);
    if (tid == 0) {
        const int p = pos[b];
        if (p) cnt[p - 1] = GFLAG;  // b is a neighbour of a and a member of every streamed row
    }
}

__global__ void __launch_bounds__(256) k_giant_stream(const int2 *rowinfo, const int32_t *col, int n, int a, int b,
                                                       const int32_t *pos, unsigned *cnt, GiantAcc *acc) {
    __shared__ int row_hits;
    const int2 rb = rowinfo[b];
    for (int j = blockIdx.x; j < rb.y; j += gridDim.x) {  // uniform per workgroup
        const int k = col[rb.x + j];
        if (k < 0 || k >= n || k == a || pos[k] != 0) continue;  // not a member of DY
        const int2 rk = rowinfo[k];
        if (threadIdx.x == 0) row_hits = 0;
        __syncthreads();
        int hits = 0;
        for (int i = threadIdx.x; i < rk.y; i += blockDim.x) {
            const int x = col[rk.x + i];
            if (x < 0 || x >= n) continue;
            const int p = pos[x];
            if (!p) continue;
            const unsigned old = atomicAdd(&cnt[p - 1], 1u);
            if (old & GFLAG) continue;
            ++hits;
            if (old == 0u) atomicAdd(&acc->s_table, 1);
            else atomicMax(&acc->gam, (int)old + 1);
        }
        if (hits) atomicAdd(&row_hits, hits);
        __syncthreads();
        if (threadIdx.x == 0 && row_hits > 0) {
            atomicAdd(&acc->s_rows, 1);
            atomicMax(&acc->gam, row_hits);
        }
        __syncthreads();
    }
}

// out6 (optional): {deg u, deg v, T, |sq| on u's side, |sq| on v's side, gamma} as dcr_bfc_ingredients reports them
__global__ void k_giant_finish(const int2 *rowinfo, int u, int v, int a, int64_t slot, int curv_type, double *curv,
                               const GiantAcc *acc, int64_t *out6) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int du = rowinfo[u].y, dv = rowinfo[v].y;
    const int su = a == u ? acc->s_table : acc->s_rows, sv = a == u ? acc->s_rows : acc->s_table;
    if (out6) {
        out6[0] = du; out6[1] = dv; out6[2] = acc->T; out6[3] = su; out6[4] = sv; out6[5] = acc->gam;
    }
    if (curv && slot >= 0) {
        double r;
        if (curv_type == DCR_CURV_BFC) r = bfc_formula(du, dv, acc->T, su, sv, acc->gam);
        else if (curv_type == DCR_CURV_AUGMENTED) r = (double)(4 - du - dv + 3 * acc->T);
        else r = (double)acc->T;
        curv[slot] = r;
    }
}

__global__ void __launch_bounds__(256) k_giant_unmark(const int2 *rowinfo, const int32_t *col, int n, int a, int32_t *pos) {
    const int2 ra = rowinfo[a];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < ra.y; i += nth) {
        const int k = col[ra.x + i];
        if (k >= 0 && k < n) pos[k] = 0;
    }
}

static int ensure_giant(dcr_graph *g, int d_table) {
    if (!g->giant_pos) {
        DCR_TRY(dev_alloc(&g->giant_pos, g->n > 0 ? g->n : 1));
        DCR_HIP(hipMemsetAsync(g->giant_pos, 0, sizeof(int32_t) * (size_t)(g->n > 0 ? g->n : 1), g->stream));
    }
    DCR_TRY(dev_regrow(&g->giant_cnt, &g->giant_cnt_cap, (int64_t)d_table + 1));
    if (!g->giant_acc) DCR_TRY(dev_alloc(&g->giant_acc, 4));
    return DCR_OK;
}

// One edge {u,v} (degrees as the host knows them: only used to pick sides and size grids; the kernels read rowinfo).
// Asynchronous on the graph's stream.  slot < 0: no curvature write (ingredients only).
int giant_edge(dcr_graph *g, int u, int v, int du, int dv, int64_t slot, int curv_type, bool need_cycles, int64_t *d_out6) {
    const int a = du >= dv ? u : v, b = du >= dv ? v : u;
    const int da = du >= dv ? du : dv, db = du >= dv ? dv : du;
    DCR_TRY(ensure_giant(g, da));
    GiantAcc *acc = reinterpret_cast<GiantAcc *>(g->giant_acc);
    const int n = (int)g->n;
    const unsigned ga = (unsigned)((da + 255) / 256 > 1024 ? 1024 : (da + 255) / 256 < 1 ? 1 : (da + 255) / 256);
    const unsigned gb = (unsigned)((db + 255) / 256 > 1024 ? 1024 : (db + 255) / 256 < 1 ? 1 : (db + 255) / 256);
    hipLaunchKernelGGL(k_giant_mark, dim3(ga), dim3(256), 0, g->stream, g->rowinfo, g->col, n, a, b, g->giant_pos,
                       g->giant_cnt, acc);
    hipLaunchKernelGGL(k_giant_sweep, dim3(gb), dim3(256), 0, g->stream, g->rowinfo, g->col, n, a, b, g->giant_pos,
                       g->giant_cnt, acc);
    if (need_cycles) {
        const unsigned gs = (unsigned)(db > 4096 ? 4096 : db < 1 ? 1 : db);
        hipLaunchKernelGGL(k_giant_stream, dim3(gs), dim3(256), 0, g->stream, g->rowinfo, g->col, n, a, b, g->giant_pos,
                           g->giant_cnt, acc);
    }
    hipLaunchKernelGGL(k_giant_finish, dim3(1), dim3(64), 0, g->stream, g->rowinfo, u, v, a, slot, curv_type,
                       slot >= 0 ? g->curv : nullptr, acc, d_out6);
    hipLaunchKernelGGL(k_giant_unmark, dim3(ga), dim3(256), 0, g->stream, g->rowinfo, g->col, n, a, g->giant_pos);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// After the classify kernel: the edges it put on the giant list, one after the other.  One host sync to learn how many.
int process_giant_edges(dcr_graph *g, int curv_type) {
    DCR_TRY(sync_result(g));
    const int count = g->hres->giant_count;
    if (count <= 0) return DCR_OK;
    if (count > GIANT_LIST_CAP) return DCR_OK;  // flag_too_big is set: the caller reports it
    std::vector<int32_t> rec((size_t)count * 5);
    DCR_HIP(hipMemcpyAsync(rec.data(), g->giant_list, sizeof(int32_t) * rec.size(), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    for (int i = 0; i < count; ++i) {
        const int32_t *r = &rec[(size_t)i * 5];
        if (r[1] < 0 || r[2] < 0 || r[1] >= g->n || r[2] >= g->n || r[0] < 0 || r[0] >= g->cap_total)
            DCR_FAIL(DCR_ESTATE, "corrupt giant-edge record");
        DCR_TRY(giant_edge(g, r[1], r[2], r[3], r[4], r[0], curv_type, curv_type == DCR_CURV_BFC, nullptr));
    }
    return DCR_OK;
}

}  // namespace dcr
