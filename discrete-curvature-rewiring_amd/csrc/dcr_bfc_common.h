// Device helpers shared by the two curvature-pass implementations (dcr_bfc.hip: edge-centric; dcr_bfc_nc.hip:
// node-centric).
#pragma once
#include "dcr_internal.h"

namespace dcr {

constexpr unsigned EMPTY = 0xFFFFFFFFu;

enum { MODE_BFC = 0, MODE_TRI = 1, MODE_BYTES = 2 };

struct View {
    const int2 *rowinfo;
    const int32_t *col;
    const int32_t *slot_row;
    int64_t cap_total;
    int32_t *guard;  // [8] first violated invariant: code, block, item data (debug / safety net)
    const uint8_t *dirty;  // incremental pass: only edges with a flagged endpoint are recomputed (nullptr: all)
    int32_t n;             // number of nodes
    int32_t nc_handles;    // 1: the node-centric kernels take every edge within their degree limits
    long long *trace;      // diagnostic build aid (DCR_NC_TRACE): per wave {first, last} s_memrealtime stamps, else nullptr
};

__device__ inline bool row_ok(const View &g, const int2 rk, int code, int a, int b) {
    const bool ok = rk.x >= 0 && rk.y >= 0 && (int64_t)rk.x + rk.y <= g.cap_total;
    if (!ok && atomicCAS(&g.guard[0], 0, code) == 0) {
        g.guard[1] = rk.x; g.guard[2] = rk.y; g.guard[3] = a; g.guard[4] = b; g.guard[5] = blockIdx.x;
        g.guard[6] = threadIdx.x;
    }
    return ok;
}

// bfc_naive.py:31-32 / 39-40, left to right in float64; compiled with -ffp-contract=off
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

template <int SLOTS>
__device__ inline unsigned hash_slot(unsigned key) {
    constexpr int BITS = __builtin_ctz(SLOTS);
    return (key * 0x9E3779B1u) >> (32 - BITS);
}

// Rows are read in aligned 16-byte pieces: lane q of a row's lane set fetches col[a0 + 4q .. a0 + 4q + 3], a0 = row
// start rounded down to a multiple of 4 (the col allocation is padded, so the last piece never leaves it).
__device__ inline int4 load_piece(const int32_t *col, int a) {
    return *reinterpret_cast<const int4 *>(col + a);
}

// popcount of the four per-element hit ballots, restricted to the lanes selected by `sel`
__device__ inline int count_hits(unsigned m, unsigned long long sel, int shift) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += __popcll((__ballot((m >> j) & 1u) >> shift) & sel);
    return c;
}

constexpr int LONG_ROW = 60;  // longer rows are streamed by a whole wave; shorter ones fit four 64-byte group steps

// ---- node-centric pass: which edges it takes (dcr_bfc_nc.hip) -------------------------------------------------
constexpr int NC_CLASSES = 5;
constexpr int NC_MAXD = 8190;       // largest degree whose neighbour table fits the biggest class
constexpr int NC_MAXOTHER = 16382;  // largest degree of the other endpoint (15-bit per-slot counters)

// Edges between a hub above every table size and a node of moderate degree are swept from the hub's side with a
// position map in device memory (dcr_bfc_giant.hip, k_hub_edges): the rows streamed are then those of the small
// endpoint's neighbours.  (Owned by the small endpoint they would stream every neighbour row of the hub, per edge.)
constexpr int HUB_OTHER_MAX = 1022;
__device__ __host__ inline bool hub_takes(int da, int db) {
    const int dmax = da > db ? da : db, dmin = da < db ? da : db;
    return dmax > NC_MAXD && dmin <= HUB_OTHER_MAX;
}

__device__ __host__ inline bool nc_takes(int da, int db) {
    if (hub_takes(da, db)) return false;
    const int dmin = da < db ? da : db, dmax = da < db ? db : da;
    return dmin <= NC_MAXD && dmax <= NC_MAXOTHER;
}

}  // namespace dcr
