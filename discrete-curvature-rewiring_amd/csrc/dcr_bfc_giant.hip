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

// ---------------------------------------------------------------------------------------------------------------
// Hub-to-small-node edges: hub h has more neighbours than any LDS table holds (> NC_MAXD), the other endpoint v has
// at most HUB_OTHER_MAX.  Owned by v they would cost a sweep over all of N(h) and the rows of almost all its members
// PER EDGE.  Instead h's position map is marked once and every such edge {h,v} takes one wave: sweep N(v) against the
// map (triangles, DY = N(v) \ N(h)), stream the few rows of DY against the map, per-slot counters in a per-wave array
// in device memory that is kept all-zero between edges (the slots an edge touched are remembered in LDS and cleared).
// ---------------------------------------------------------------------------------------------------------------
constexpr int HUB_TOUCH_CAP = 1024;

__global__ void __launch_bounds__(256) k_find_hubs(const int2 *rowinfo, int n, int32_t *hub_list, int64_t cap,
                                                    DevResult *res) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t u = tid; u < n; u += nth) {
        const int d = rowinfo[u].y;
        if (d > NC_MAXD) {
            const int idx = atomicAdd(&res->hub_count, 1);
            if (idx < cap) {
                hub_list[2 * (int64_t)idx] = (int32_t)u;
                hub_list[2 * (int64_t)idx + 1] = d;
            }
        }
    }
}

__device__ inline int wave_sum(int x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__device__ inline int wave_max(int x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int y = __shfl_xor(x, off);
        x = y > x ? y : x;
    }
    return x;
}
__device__ inline void hub_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ void __launch_bounds__(256) k_hub_edges(View g, int h, const int32_t *pos, unsigned *cnt_all, int64_t cnt_stride,
                                                    int curv_type, double *curv) {
    __shared__ int2 desc_all[4][HUB_OTHER_MAX + 2];
    __shared__ int touched_all[4][HUB_TOUCH_CAP];
    __shared__ int ntouch_all[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;
    int2 *desc = desc_all[wid];
    int *touched = touched_all[wid];
    int *ntouch = &ntouch_all[wid];
    unsigned *cnt = cnt_all + (int64_t)wave * cnt_stride;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int2 rh = g.rowinfo[h];
    if (!row_ok(g, rh, 30, h, 0)) return;
    for (int i = wave; i < rh.y; i += nwaves) {  // one edge {h, v} per wave and round; everything below is uniform
        const int v = g.col[rh.x + i];
        if (v < 0 || v >= g.n || v == h) continue;
        const int2 rv = g.rowinfo[v];
        if (!row_ok(g, rv, 31, v, h) || rv.y < 1 || rv.y > HUB_OTHER_MAX) continue;  // other paths take the rest
        if (g.dirty && !edge_dirty(g.dirty[h], g.dirty[v])) continue;
        // sweep N(v): where h sits in row v, triangles (flagged in the counters), descriptors of the rows of DY
        int T = 0, posh = -1, nrows = 0, nt = 0;
        for (int base = 0; base < rv.y; base += 64) {
            const int j = base + lane;
            const int k = j < rv.y ? g.col[rv.x + j] : -1;
            const bool ish = k == h;
            const unsigned long long mh = __ballot(ish);
            if (mh) posh = base + __ffsll((long long)mh) - 1;
            const bool valid = k >= 0 && k < g.n && !ish;
            const int p = valid ? pos[k] : 0;
            const unsigned long long mt = __ballot(p != 0);
            T += __popcll(mt);
            if (MODE == MODE_BFC) {
                if (p) {
                    cnt[p - 1] = GFLAG;  // common neighbour: never counted as a 4-cycle corner
                    touched[nt + __popcll(mt & lt)] = p - 1;
                }
                nt += __popcll(mt);
                const bool member = valid && p == 0;
                const unsigned long long mm = __ballot(member);
                if (member) {
                    int2 rk = g.rowinfo[k];
                    if (!row_ok(g, rk, 32, k, v)) rk = make_int2(0, 0);
                    desc[nrows + __popcll(mm & lt)] = rk;
                }
                nrows += __popcll(mm);
            }
        }
        int s_table = 0, s_rows = 0, gam = 0;
        const bool trivial = curv_type == DCR_CURV_BFC && rv.y == 1;  // bfc_naive.py:18-19
        if (MODE == MODE_BFC && !trivial) {
            if (lane == 0) {
                cnt[i] = GFLAG;  // v itself: a neighbour of h and a member of every streamed row
                touched[nt] = i;
                *ntouch = nt + 1;
            }
            hub_wave_sync();  // flags and descriptors before the atomics / reads of other lanes
            for (int r = 0; r < nrows; ++r) {
                const int2 rk = desc[r];
                int hits = 0;
                for (int t = lane; t < rk.y; t += 64) {
                    const int x = g.col[rk.x + t];
                    if (x < 0 || x >= g.n) continue;
                    const int px = pos[x];
                    if (!px) continue;
                    const unsigned old = atomicAdd(&cnt[px - 1], 1u);
                    if (old & GFLAG) continue;
                    ++hits;
                    if (old == 0u) {
                        ++s_table;  // a new member of sq on the hub's side; remember the slot for the clean-up
                        const int at = atomicAdd(ntouch, 1);
                        if (at < HUB_TOUCH_CAP) touched[at] = px - 1;
                    } else {
                        gam = (int)old + 1 > gam ? (int)old + 1 : gam;
                    }
                }
                const int row_hits = wave_sum(hits);
                if (row_hits > 0) {
                    ++s_rows;
                    gam = row_hits > gam ? row_hits : gam;
                }
            }
            s_table = wave_sum(s_table);
            gam = wave_max(gam);
            hub_wave_sync();
            // back to all-zero counters
            const int ntl = *ntouch;
            if (ntl <= HUB_TOUCH_CAP) {
                for (int t = lane; t < ntl; t += 64) cnt[touched[t]] = 0u;
            } else {  // more distinct slots than the list holds: walk everything this edge can have touched again
                for (int j = lane; j < rv.y; j += 64) {
                    const int k = g.col[rv.x + j];
                    const int p = (k >= 0 && k < g.n && k != h) ? pos[k] : 0;
                    if (p) cnt[p - 1] = 0u;
                }
                if (lane == 0) cnt[i] = 0u;
                for (int r = 0; r < nrows; ++r) {
                    const int2 rk = desc[r];
                    for (int t = lane; t < rk.y; t += 64) {
                        const int x = g.col[rk.x + t];
                        const int px = (x >= 0 && x < g.n) ? pos[x] : 0;
                        if (px) cnt[px - 1] = 0u;
                    }
                }
            }
            hub_wave_sync();
        }
        if (lane == 0) {
            int64_t slot = -1;
            if (h < v) slot = (int64_t)rh.x + i;
            else if (posh >= 0) slot = (int64_t)rv.x + posh;
            if (slot < 0 || slot >= g.cap_total) {
                row_ok(g, make_int2(-1, posh), 33, h, v);  // adjacency not symmetric: report, never write
            } else if (MODE == MODE_BFC) {
                curv[slot] = trivial ? 0.0 : bfc_formula(rh.y, rv.y, T, s_table, s_rows, gam);
            } else {
                curv[slot] = curv_type == DCR_CURV_AUGMENTED ? (double)(4 - rh.y - rv.y + 3 * T) : (double)T;
            }
        }
    }
}

int process_hub_edges(dcr_graph *g, int curv_type, bool incremental) {
    const int64_t cap = g->cap_total / (NC_MAXD + 1) + 2;  // there cannot be more nodes above NC_MAXD neighbours
    if (cap > g->hub_list_cap) {
        if (g->hub_list) (void)hipFree(g->hub_list);
        g->hub_list = nullptr;
        DCR_TRY(dev_alloc(&g->hub_list, 2 * cap));
        g->hub_list_cap = cap;
    }
    DCR_HIP(hipMemsetAsync(&g->dres->hub_count, 0, sizeof(int32_t), g->stream));
    const int n = (int)g->n;
    hipLaunchKernelGGL(k_find_hubs, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256 < 1 ? 1 : (n + 255) / 256)),
                       dim3(256), 0, g->stream, g->rowinfo, n, g->hub_list, g->hub_list_cap, g->dres);
    DCR_HIP(hipGetLastError());
    DCR_TRY(sync_result(g));
    const int count = g->hres->hub_count;
    if (count <= 0) return DCR_OK;
    if (count > g->hub_list_cap) DCR_FAIL(DCR_ESTATE, "more hubs than adjacency slots allow");
    std::vector<int32_t> hubs((size_t)count * 2);
    DCR_HIP(hipMemcpyAsync(hubs.data(), g->hub_list, sizeof(int32_t) * hubs.size(), hipMemcpyDeviceToHost, g->stream));
    DCR_HIP(hipStreamSynchronize(g->stream));
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, incremental ? g->dirty : nullptr,
            (int32_t)g->n, 1, nullptr};
    for (int q = 0; q < count; ++q) {
        const int h = hubs[(size_t)q * 2], dh = hubs[(size_t)q * 2 + 1];
        if (h < 0 || h >= g->n || dh <= NC_MAXD) DCR_FAIL(DCR_ESTATE, "corrupt hub record");
        DCR_TRY(ensure_giant(g, dh));
        // waves in flight: one edge each, a counter array of dh words each, at most 512 MB of them
        int64_t waves = ((int64_t)512 << 20) / ((int64_t)dh * 4);
        if (waves > 2048) waves = 2048;
        if (waves > dh) waves = dh;
        waves = (waves / 4) * 4;
        if (waves < 4) waves = 4;
        const int64_t need = waves * (int64_t)dh;
        if (need > g->hub_cnt_cap) {
            if (g->hub_cnt) (void)hipFree(g->hub_cnt);
            g->hub_cnt = nullptr;
            DCR_TRY(dev_alloc(&g->hub_cnt, need));
            g->hub_cnt_cap = need;
            DCR_HIP(hipMemsetAsync(g->hub_cnt, 0, sizeof(uint32_t) * (size_t)need, g->stream));
        }
        const unsigned ga = (unsigned)((dh + 255) / 256 > 1024 ? 1024 : (dh + 255) / 256);
        hipLaunchKernelGGL(k_giant_mark, dim3(ga), dim3(256), 0, g->stream, g->rowinfo, g->col, n, h, h, g->giant_pos,
                           g->giant_cnt, reinterpret_cast<GiantAcc *>(g->giant_acc));
        if (curv_type == DCR_CURV_BFC)
            hipLaunchKernelGGL((k_hub_edges<MODE_BFC>), dim3((unsigned)(waves / 4)), dim3(256), 0, g->stream, vw, h,
                               g->giant_pos, g->hub_cnt, (int64_t)dh, curv_type, g->curv);
        else
            hipLaunchKernelGGL((k_hub_edges<MODE_TRI>), dim3((unsigned)(waves / 4)), dim3(256), 0, g->stream, vw, h,
                               g->giant_pos, g->hub_cnt, (int64_t)dh, curv_type, g->curv);
        hipLaunchKernelGGL(k_giant_unmark, dim3(ga), dim3(256), 0, g->stream, g->rowinfo, g->col, n, h, g->giant_pos);
        DCR_HIP(hipGetLastError());
    }
    return DCR_OK;
}

}  // namespace dcr
