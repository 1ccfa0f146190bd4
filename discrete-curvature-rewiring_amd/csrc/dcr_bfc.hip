// Curvature pass of libdcr_hip.so — the hot path.
//
// Replaces compute_curvature_graph(G, curv_type) at rewiring/sdrf_no_cuda.py:24, i.e. E calls of
// bfc_naive.bfc_edge (curvature/bfc_naive.py:7-40) or compute_curvature_edge
// (curvature/classical_curvatures.py:14-28), with one classify kernel plus one kernel per work bin.
//
// Per undirected edge (u,v), a team of threads (one wave for small neighbourhoods, a whole workgroup
// for hubs):
//   1. stages N(u) ∪ N(v) ∪ {u,v} as a tagged hash set in LDS (tag bits: "in N(u)", "in N(v)");
//      T = |N(u) ∩ N(v)| falls out of the second insertion sweep              (bfc_naive.py:22-25)
//   2. for every k in N(u) \ N(v), k != v: streams row k from HBM (coalesced, one wave per row),
//      probes each neighbour against the LDS set and counts the lanes that hit "N(v) only" with
//      ballot + popcount.  count > 0 puts k in sq1, max count is gamma         (bfc_naive.py:26-27,36)
//   3. same for v's side                                                        (bfc_naive.py:28-29,37)
//   4. lane 0 evaluates the float64 closing expression in the reference's order (bfc_naive.py:31-40)
// u and v themselves are stored with both tags, which removes them from every set difference the
// reference writes out explicitly (k != v2, "- (S1 ∪ {v1})", the "- 1" in gamma).
#include <cstdio>
#include <cstdlib>

#include "dcr_bfc_common.h"

namespace dcr {

constexpr unsigned TAG_U = 1u << 30;
constexpr unsigned TAG_V = 1u << 31;
constexpr unsigned KEY_MASK = (1u << 30) - 1u;

struct WorkLists {
    int32_t *w[NBINS];
    int32_t *giant;  // records of the edges no bin can hold (nullptr: report them as too big)
};

// keys (= du + dv + 2) admitted per bin: load factor <= 1/4 except the last bin (<= 1/2)
__host__ __device__ constexpr int bin_max_keys(int b) {
    return b == 0 ? 32 : b == 1 ? 128 : b == 2 ? 512 : b == 3 ? 2048 : 16384;
}

template <int SLOTS>
__device__ inline void table_insert(unsigned *tab, unsigned entry) {
    unsigned h = hash_slot<SLOTS>(entry & KEY_MASK);
    while (true) {
        unsigned old = atomicCAS(&tab[h], EMPTY, entry);
        if (old == EMPTY) return;
        h = (h + 1) & (SLOTS - 1);
    }
}

// returns the entry holding key, or EMPTY
template <int SLOTS>
__device__ inline unsigned table_lookup(const unsigned *tab, unsigned key) {
    unsigned h = hash_slot<SLOTS>(key);
    while (true) {
        unsigned e = tab[h];
        if (e == EMPTY || (e & KEY_MASK) == key) return e;
        h = (h + 1) & (SLOTS - 1);
    }
}

// Which of the (up to four) entries of a piece lie inside [lo, hi) and hit an entry whose tag pattern is `other`;
// returned as a 4-bit mask.  The four first probes are issued together.  With `cnt`, every hit also bumps the
// counter of the table slot it landed on (the 4-cycle seen from the other endpoint's side).
template <int SLOTS>
__device__ inline unsigned probe_piece(const unsigned *tab, unsigned *cnt, const int4 w, int a, int lo, int hi,
                                       unsigned other) {
    const unsigned k0 = (unsigned)w.x, k1 = (unsigned)w.y, k2 = (unsigned)w.z, k3 = (unsigned)w.w;
    const bool v0 = a >= lo && a < hi, v1 = a + 1 >= lo && a + 1 < hi, v2 = a + 2 >= lo && a + 2 < hi,
               v3 = a + 3 >= lo && a + 3 < hi;
    unsigned h0 = hash_slot<SLOTS>(k0), h1 = hash_slot<SLOTS>(k1), h2 = hash_slot<SLOTS>(k2), h3 = hash_slot<SLOTS>(k3);
    unsigned e0 = v0 ? tab[h0] : EMPTY, e1 = v1 ? tab[h1] : EMPTY, e2 = v2 ? tab[h2] : EMPTY, e3 = v3 ? tab[h3] : EMPTY;
    while (e0 != EMPTY && (e0 & KEY_MASK) != k0) { h0 = (h0 + 1) & (SLOTS - 1); e0 = tab[h0]; }
    while (e1 != EMPTY && (e1 & KEY_MASK) != k1) { h1 = (h1 + 1) & (SLOTS - 1); e1 = tab[h1]; }
    while (e2 != EMPTY && (e2 & KEY_MASK) != k2) { h2 = (h2 + 1) & (SLOTS - 1); e2 = tab[h2]; }
    while (e3 != EMPTY && (e3 & KEY_MASK) != k3) { h3 = (h3 + 1) & (SLOTS - 1); e3 = tab[h3]; }
    const bool t0 = e0 != EMPTY && (e0 >> 30) == other, t1 = e1 != EMPTY && (e1 >> 30) == other,
               t2 = e2 != EMPTY && (e2 >> 30) == other, t3 = e3 != EMPTY && (e3 >> 30) == other;
    if (cnt) {
        if (t0) atomicAdd(&cnt[h0], 1u);
        if (t1) atomicAdd(&cnt[h1], 1u);
        if (t2) atomicAdd(&cnt[h2], 1u);
        if (t3) atomicAdd(&cnt[h3], 1u);
    }
    return (t0 ? 1u : 0u) | (t1 ? 2u : 0u) | (t2 ? 4u : 0u) | (t3 ? 8u : 0u);
}

template <int TEAM>
__device__ inline void team_sync() {
    __syncthreads();  // single-wave teams: the compiler lowers this to a wait, no s_barrier
}

struct Ingredients {
    int du, dv, T, s1, s2, gamma;
    double bytes;   // both difference sets charged
    double bytes1;  // the same with only the cheaper side's rows charged (what a one-sided count has to read)
};

// Whole team cooperates; the result is valid in thread 0.
//
// 4-cycle counting is one-sided.  With DX = N(u) \ N(v) \ {v} and DY = N(v) \ N(u) \ {u}, the reference counts for
// k in DX the neighbours of k inside DY (bfc_naive.py:26-27,36) and for k' in DY the neighbours inside DX (:28-29,37).
// Both are degrees in the same bipartite graph between DX and DY, so only the rows of the cheaper side (smaller sum
// of degrees) are streamed: a row's hit count is its own degree, and every hit bumps a counter on the table slot of the
// node it landed on, which after the sweep is that node's degree seen from the other side.
constexpr int MAXR = 4;       // rows of one endpoint held per lane while both sides are sized (deg <= MAXR * TEAM)

template <int SLOTS, int TEAM, int MODE, int DESC_CAP>
__device__ inline Ingredients edge_ingredients(const View &g, int u, int v, unsigned *tab, unsigned *cnt, int *red,
                                               int2 *desc, int *cnts) {
    constexpr int NW = TEAM / 64;
    constexpr bool ONE_SIDED = DESC_CAP > 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int2 ru = g.rowinfo[u], rv = g.rowinfo[v];
    if (!row_ok(g, ru, 3, u, v)) ru = make_int2(0, 0);
    if (!row_ok(g, rv, 4, u, v)) rv = make_int2(0, 0);
    if (ru.y + rv.y + 2 > (ONE_SIDED ? DESC_CAP : SLOTS / 2) || (ONE_SIDED && (ru.y > MAXR * TEAM || rv.y > MAXR * TEAM))) {
        row_ok(g, make_int2(-1, ru.y + rv.y), 5, u, v);
        ru.y = 0;
        rv.y = 0;
    }
    const int32_t *rowu = g.col + ru.x, *rowv = g.col + rv.x;

    for (int i = tid; i < SLOTS; i += TEAM) {
        tab[i] = EMPTY;
        if (ONE_SIDED) cnt[i] = 0u;
    }
    team_sync<TEAM>();
    if (tid == 0) {
        table_insert<SLOTS>(tab, (unsigned)u | TAG_U | TAG_V);
        table_insert<SLOTS>(tab, (unsigned)v | TAG_U | TAG_V);
    }
    for (int i = tid; i < ru.y; i += TEAM) {
        int k = rowu[i];
        if (k != v) table_insert<SLOTS>(tab, (unsigned)k | TAG_U);
    }
    team_sync<TEAM>();
    int tcount = 0;
    for (int i = tid; i < rv.y; i += TEAM) {
        unsigned k = (unsigned)rowv[i];
        if ((int)k == u) continue;
        unsigned h = hash_slot<SLOTS>(k);
        while (true) {
            unsigned e = tab[h];
            if (e == EMPTY) {
                e = atomicCAS(&tab[h], EMPTY, k | TAG_V);
                if (e == EMPTY) break;
            }
            if ((e & KEY_MASK) == k) {  // already there from N(u): a triangle
                atomicOr(&tab[h], TAG_V);
                ++tcount;
                break;
            }
            h = (h + 1) & (SLOTS - 1);
        }
    }
    team_sync<TEAM>();

    Ingredients out;
    out.du = ru.y;
    out.dv = rv.y;
    out.s1 = out.s2 = out.gamma = 0;
    out.bytes = 0.0;
    int s_rows = 0, s_slots = 0, gam = 0;  // per-lane partials
    long long len_u = 0, len_v = 0;        // sum of row lengths over DX / DY (per-lane partials, then totals)
    int n_u = 0, n_v = 0;                  // |DX|, |DY|
    int T = 0;
    bool rows_are_u = true;                // which endpoint's side was streamed row by row

    if (ONE_SIDED) {
        // ---- size both sides: fetch {start, length} of every member of DX and DY, lane-parallel ----------------
        int2 ku[MAXR], kv[MAXR];
        if (MODE != MODE_TRI) {
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                ku[r] = make_int2(0, -1);
                kv[r] = make_int2(0, -1);
                const int p = r * TEAM + tid;
                if (p < ru.y) {
                    const int k = rowu[p];
                    if ((table_lookup<SLOTS>(tab, (unsigned)k) >> 30) == 1u) {  // in N(u) only
                        ku[r] = g.rowinfo[k];
                        if (!row_ok(g, ku[r], 1, k, p)) ku[r] = make_int2(0, 0);
                        len_u += ku[r].y;
                        ++n_u;
                    }
                }
                if (p < rv.y) {
                    const int k = rowv[p];
                    if ((table_lookup<SLOTS>(tab, (unsigned)k) >> 30) == 2u) {  // in N(v) only
                        kv[r] = g.rowinfo[k];
                        if (!row_ok(g, kv[r], 2, k, p)) kv[r] = make_int2(0, 0);
                        len_v += kv[r].y;
                        ++n_v;
                    }
                }
            }
        }
        // team totals: triangles, side sizes
        int t = tcount;
        for (int off = 32; off > 0; off >>= 1) {
            t += __shfl_xor(t, off);
            len_u += __shfl_xor(len_u, off);
            len_v += __shfl_xor(len_v, off);
            n_u += __shfl_xor(n_u, off);
            n_v += __shfl_xor(n_v, off);
        }
        if (NW > 1) {
            if (lane == 0) {
                red[wid] = t;
                red[NW + wid] = n_u;
                red[2 * NW + wid] = n_v;
                ((long long *)(red + 4 * NW))[wid] = len_u;
                ((long long *)(red + 4 * NW))[NW + wid] = len_v;
            }
            if (tid == 0) {
                cnts[0] = 0;
                cnts[1] = 0;
            }
            team_sync<TEAM>();
            t = n_u = n_v = 0;
            len_u = len_v = 0;
            for (int w = 0; w < NW; ++w) {
                t += red[w];
                n_u += red[NW + w];
                n_v += red[2 * NW + w];
                len_u += ((long long *)(red + 4 * NW))[w];
                len_v += ((long long *)(red + 4 * NW))[NW + w];
            }
            team_sync<TEAM>();
        } else {
            if (tid == 0) {
                cnts[0] = 0;
                cnts[1] = 0;
            }
            team_sync<TEAM>();
        }
        T = t;
        out.T = T;
        if (MODE == MODE_TRI) return out;
        if (MODE == MODE_BYTES) {
            // rows of u and v, the row of every non-triangle neighbour, row-pointer pairs, output
            out.bytes = 4.0 * (double)(ru.y + rv.y) + 4.0 * (double)(len_u + len_v) + 8.0 * (double)(2 + n_u + n_v) + 8.0;
            const bool cheap_u = len_u <= len_v;
            out.bytes1 = 4.0 * (double)(ru.y + rv.y) + 4.0 * (double)(cheap_u ? len_u : len_v) +
                         8.0 * (double)(2 + (cheap_u ? n_u : n_v)) + 8.0;
            return out;
        }
        // ---- stream the cheaper side ---------------------------------------------------------------------------
        const bool scan_u = len_u <= len_v;  // identical in every lane (team totals)
        rows_are_u = scan_u;
        const unsigned other = scan_u ? 2u : 1u;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int2 rk = scan_u ? ku[r] : kv[r];
            const bool want = rk.y >= 0;
            const bool is_long = want && rk.y > LONG_ROW;
            const bool is_short = want && !is_long;
            const unsigned long long ms = __ballot(is_short), ml = __ballot(is_long);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (ms) {
                int base = 0;
                const int leader = __ffsll((long long)ms) - 1;
                if (lane == leader) base = atomicAdd(&cnts[0], __popcll(ms));
                base = __shfl(base, leader);
                if (is_short) desc[base + __popcll(ms & below)] = rk;
            }
            if (ml) {
                int base = 0;
                const int leader = __ffsll((long long)ml) - 1;
                if (lane == leader) base = atomicAdd(&cnts[1], __popcll(ml));
                base = __shfl(base, leader);
                if (is_long) desc[DESC_CAP - 1 - (base + __popcll(ml & below))] = rk;
            }
        }
        team_sync<TEAM>();
        int nshort = __builtin_amdgcn_readfirstlane(cnts[0]);
        int nlong = __builtin_amdgcn_readfirstlane(cnts[1]);
        if (nshort < 0 || nlong < 0 || nshort + nlong > DESC_CAP) {
            row_ok(g, make_int2(-1, nshort), 8, nlong, u);
            nshort = nlong = 0;
        }
        // (b) short rows (<= LONG_ROW entries, i.e. at most four 64-byte steps): a 4-lane group per row, so one
        //     wave-instruction works on 16 rows and each group reads 64 contiguous bytes per step.  All steps of a
        //     row are loaded before the first probe (up to four 16-byte loads in flight per lane).
        {
            constexpr int NG = TEAM / 4;
            const int gid = tid >> 2, gl = tid & 3, gsh = (lane >> 2) << 2;
#ifdef DCR_ABLATE_SHORT
            for (int b0 = 0; b0 < 0; b0 += NG) {
#else
            for (int b0 = 0; b0 < nshort; b0 += NG) {
#endif
                const int i = b0 + gid;
                const int2 rk = i < nshort ? desc[i] : make_int2(0, 0);
                const int hi = rk.x + rk.y;
                const int a = (rk.x & ~3) + 4 * gl;
                int4 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[q] = make_int4(0, 0, 0, 0);
                    if (a + 16 * q < hi) w[q] = load_piece(g.col, a + 16 * q);
                }
                int c = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool act = a + 16 * q < hi;
                    if (!__any(act)) break;
                    unsigned m = 0;
                    if (act) m = probe_piece<SLOTS>(tab, cnt, w[q], a + 16 * q, rk.x, hi, other);
                    c += count_hits(m, 0xFull, gsh);
                }
                if (gl == 0 && c > 0) {
                    ++s_rows;
                    gam = c > gam ? c : gam;
                }
            }
        }
        // (c) long rows: a whole wave per row, 1 KiB per wave-instruction, next piece in flight while probing
#ifdef DCR_ABLATE_LONG
        for (int i = wid; i < 0; i += NW) {
#else
        for (int i = wid; i < nlong; i += NW) {
#endif
            const int2 rl = desc[DESC_CAP - 1 - i];
            const int hi = rl.x + rl.y;
            int al = (rl.x & ~3) + 4 * lane;
            int4 wl = make_int4(0, 0, 0, 0);
            if (al < hi) wl = load_piece(g.col, al);
            int c = 0;
            for (int a0 = rl.x & ~3; a0 < hi; a0 += 256) {
                const int an = al + 256;
                int4 wn = make_int4(0, 0, 0, 0);
                if (an < hi) wn = load_piece(g.col, an);
                unsigned m = 0;
                if (al < hi) m = probe_piece<SLOTS>(tab, cnt, wl, al, rl.x, hi, other);
                c += count_hits(m, ~0ull, 0);
                al = an;
                wl = wn;
            }
            if (lane == 0 && c > 0) {
                ++s_rows;
                gam = c > gam ? c : gam;
            }
        }
        team_sync<TEAM>();
        // (d) the other side, from the slot counters
        for (int i = tid; i < SLOTS; i += TEAM) {
            const unsigned e = tab[i];
            if (e != EMPTY && (e >> 30) == other) {
                const int c = (int)cnt[i];
                if (c > 0) {
                    ++s_slots;
                    gam = c > gam ? c : gam;
                }
            }
        }
    } else {
        // ---- largest bin: the 128 KiB table leaves no LDS for descriptors or counters; both sides are streamed,
        //      one wave per row ---------------------------------------------------------------------------------
        int t = tcount;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (NW > 1) {
            if (lane == 0) red[wid] = t;
            team_sync<TEAM>();
            t = 0;
            for (int w = 0; w < NW; ++w) t += red[w];
            team_sync<TEAM>();
        }
        T = t;
        out.T = T;
        if (MODE == MODE_TRI) return out;
        for (int side = 0; side < 2; ++side) {
            const int32_t *row = side == 0 ? rowu : rowv;
            const int deg = side == 0 ? ru.y : rv.y;
            const unsigned mine = side == 0 ? 1u : 2u, other = side == 0 ? 2u : 1u;
            int scount = 0;
            for (int p = wid; p < deg; p += NW) {
                const int k = row[p];
                if ((table_lookup<SLOTS>(tab, (unsigned)k) >> 30) != mine) continue;
                int2 rk = g.rowinfo[k];
                if (!row_ok(g, rk, 9, k, p)) rk = make_int2(0, 0);
                if (MODE == MODE_BYTES) {
                    if (lane == 0) {
                        if (side == 0) {
                            len_u += rk.y;
                            n_u += 1;
                        } else {
                            len_v += rk.y;
                            n_v += 1;
                        }
                    }
                    continue;
                }
                const int hi = rk.x + rk.y;
                int c = 0;
                for (int al = (rk.x & ~3) + 4 * lane; __any(al < hi); al += 256) {
                    unsigned m = 0;
                    if (al < hi) m = probe_piece<SLOTS>(tab, nullptr, load_piece(g.col, al), al, rk.x, hi, other);
                    c += count_hits(m, ~0ull, 0);
                }
                if (lane == 0 && c > 0) {
                    ++scount;
                    gam = c > gam ? c : gam;
                }
            }
            if (side == 0) s_rows = scount; else s_slots = scount;
        }
    }
    // per-lane partials -> wave -> team
    for (int off = 32; off > 0; off >>= 1) {
        s_rows += __shfl_xor(s_rows, off);
        s_slots += __shfl_xor(s_slots, off);
        const int og = __shfl_xor(gam, off);
        gam = og > gam ? og : gam;
        if (!ONE_SIDED) {
            n_u += __shfl_xor(n_u, off);
            len_u += __shfl_xor(len_u, off);
            n_v += __shfl_xor(n_v, off);
            len_v += __shfl_xor(len_v, off);
        }
    }
    if (NW > 1) {
        if (lane == 0) {
            red[wid] = s_rows;
            red[NW + wid] = s_slots;
            red[2 * NW + wid] = gam;
            red[3 * NW + wid] = n_u;
            ((long long *)(red + 4 * NW))[wid] = len_u;
        }
        team_sync<TEAM>();
        s_rows = s_slots = gam = 0;
        int nn = 0;
        long long ll = 0;
        for (int w = 0; w < NW; ++w) {
            s_rows += red[w];
            s_slots += red[NW + w];
            gam = red[2 * NW + w] > gam ? red[2 * NW + w] : gam;
            nn += red[3 * NW + w];
            ll += ((long long *)(red + 4 * NW))[w];
        }
        if (!ONE_SIDED) {
            n_u = nn;
            len_u = ll;
        }
        team_sync<TEAM>();
        if (!ONE_SIDED && MODE == MODE_BYTES) {  // the other side's totals, through the same scratch
            if (lane == 0) {
                red[3 * NW + wid] = n_v;
                ((long long *)(red + 4 * NW))[wid] = len_v;
            }
            team_sync<TEAM>();
            nn = 0;
            ll = 0;
            for (int w = 0; w < NW; ++w) {
                nn += red[3 * NW + w];
                ll += ((long long *)(red + 4 * NW))[w];
            }
            n_v = nn;
            len_v = ll;
            team_sync<TEAM>();
        }
    }
    out.s1 = rows_are_u ? s_rows : s_slots;  // |sq1| (u's side), |sq2| (v's side)
    out.s2 = rows_are_u ? s_slots : s_rows;
    out.gamma = gam;
    if (!ONE_SIDED && MODE == MODE_BYTES) {
        out.bytes = 4.0 * (double)(ru.y + rv.y) + 4.0 * (double)(len_u + len_v) + 8.0 * (double)(2 + n_u + n_v) + 8.0;
        const bool cheap_u = len_u <= len_v;
        out.bytes1 = 4.0 * (double)(ru.y + rv.y) + 4.0 * (double)(cheap_u ? len_u : len_v) +
                     8.0 * (double)(2 + (cheap_u ? n_u : n_v)) + 8.0;
    }
    return out;
}

template <int SLOTS, int TEAM, int MODE, int DESC_CAP>
__device__ inline void pass_item(const View &g, int item, int count, const int32_t *work, int curv_type, double *curv,
                                 double *bytes_total, unsigned *tab, unsigned *cnt, int *red, int2 *desc,
                                 int *cnts) {
    const int s = work[item];
    if (s < 0 || s >= g.cap_total) {  // cannot happen (classify writes valid slots); never chase a bad index
        row_ok(g, make_int2(-1, s), 6, item, count);
        return;
    }
    const int u = g.slot_row[s];
    const int v = g.col[s];
    Ingredients q = edge_ingredients<SLOTS, TEAM, MODE, DESC_CAP>(g, u, v, tab, cnt, red, desc, cnts);
    if (threadIdx.x == 0) {
        if (MODE == MODE_BFC) {
            curv[s] = bfc_formula(q.du, q.dv, q.T, q.s1, q.s2, q.gamma);
        } else if (MODE == MODE_TRI) {
            curv[s] = curv_type == DCR_CURV_AUGMENTED ? (double)(4 - q.du - q.dv + 3 * q.T) : (double)q.T;
        } else {
            atomicAdd(bytes_total, q.bytes);
            atomicAdd(bytes_total + 1, q.bytes1);
        }
    }
}

// Persistent grid over one bin's work list.
//   * single-wave teams (TEAM == 64) dequeue CHUNK items at a time with one atomic whose result is broadcast
//     by readfirstlane: register-only, no LDS hand-off, no barrier;
//   * multi-wave teams take items round-robin (item = block, block + grid, ...): every value that steers control
//     flow around the team barriers is provably uniform.  (An LDS-broadcast dequeue for multi-wave teams hung /
//     faulted on gfx950 in round 1 and was removed; see DESIGN.md.)
template <int SLOTS, int TEAM, int MODE, int DESC_CAP, int CHUNK>
__global__ void __launch_bounds__(TEAM) k_edge_pass(View g, const int32_t *work, const int32_t *work_count,
                                                     int32_t *work_next, int curv_type, double *curv,
                                                     double *bytes_total) {
    __shared__ unsigned tab[SLOTS];
    __shared__ unsigned cnt[DESC_CAP > 0 ? SLOTS : 1];
    __shared__ int2 desc[DESC_CAP > 0 ? DESC_CAP : 1];
    __shared__ int red[8 * (TEAM / 64) + 2];
    __shared__ int cnts[2];
    const int count = *work_count;
    if (count < 0 || count > g.cap_total) {  // cannot happen; never walk a list with a corrupt length
        row_ok(g, make_int2(-1, count), 7, 0, 0);
        return;
    }
    if (TEAM == 64) {
        const int max_rounds = count / CHUNK + 2;
        for (int round = 0; round < max_rounds; ++round) {
            int first = 0;
            if (threadIdx.x == 0) first = atomicAdd(work_next, CHUNK);
            first = __builtin_amdgcn_readfirstlane(first);
            if (first >= count || first < 0) break;
            const int last = first + CHUNK < count ? first + CHUNK : count;
            for (int item = first; item < last; ++item)
                pass_item<SLOTS, TEAM, MODE, DESC_CAP>(g, item, count, work, curv_type, curv, bytes_total, tab, cnt, red,
                                                       desc, cnts);
        }
    } else {
        for (int item = blockIdx.x; item < count; item += gridDim.x)
            pass_item<SLOTS, TEAM, MODE, DESC_CAP>(g, item, count, work, curv_type, curv, bytes_total, tab, cnt, red,
                                                   desc, cnts);
    }
}

// one edge given directly (dcr_curvature_edge / dcr_bfc_ingredients)
template <int SLOTS, int TEAM, int DESC_CAP>
__global__ void __launch_bounds__(TEAM) k_edge_single(View g, int u, int v, int64_t *out6) {
    __shared__ unsigned tab[SLOTS];
    __shared__ unsigned cnt[DESC_CAP > 0 ? SLOTS : 1];
    __shared__ int2 desc[DESC_CAP > 0 ? DESC_CAP : 1];
    __shared__ int red[8 * (TEAM / 64) + 2];
    __shared__ int cnts[2];
    Ingredients q = edge_ingredients<SLOTS, TEAM, MODE_BFC, DESC_CAP>(g, u, v, tab, cnt, red, desc, cnts);
    if (threadIdx.x == 0) {
        out6[0] = q.du; out6[1] = q.dv; out6[2] = q.T; out6[3] = q.s1; out6[4] = q.s2; out6[5] = q.gamma;
    }
}

// Classify every adjacency slot: undirected edges are the slots whose neighbour id exceeds the row id.
// Trivial cases are finished here ('1d'; BFC with a degree-1 endpoint, bfc_naive.py:18-19); the rest go to
// the work list of the bin that fits their neighbourhood.  A block owns CLASSIFY_CHUNK consecutive slots:
// it counts its items per bin, reserves list space with one global atomic per bin, then writes.
constexpr int CLASSIFY_CHUNK = 4096;

__device__ inline int classify_slot(const View &g, int64_t s, int64_t cap_total, int curv_type, int mode, double *curv,
                                    DevResult *res, double *bytes_total, bool finish_trivial, int32_t *giant) {
    if (s >= cap_total) return -1;
    const int u = g.slot_row[s];
    const int2 ru = g.rowinfo[u];
    if ((int)(s - ru.x) >= ru.y) return -1;
    const int v = g.col[s];
    if (v <= u) return -1;
    if (g.dirty && !edge_dirty(g.dirty[u], g.dirty[v])) return -1;  // no edit can have changed it: the stored value is still exact
    const int dv = g.rowinfo[v].y, du = ru.y;
    if (g.nc_handles && (nc_takes(du, dv) || hub_takes(du, dv))) return -1;  // the node-centric / hub kernels own this edge
    if (mode != MODE_BYTES && curv_type == DCR_CURV_1D) {
        if (finish_trivial) curv[s] = (double)(4 - du - dv);
        return -1;
    }
    if (curv_type == DCR_CURV_BFC && (du < dv ? du : dv) == 1) {
        if (finish_trivial) {
            if (mode == MODE_BYTES) {
                atomicAdd(bytes_total, 24.0);
                atomicAdd(bytes_total + 1, 24.0);
            }
            else curv[s] = 0.0;
        }
        return -1;
    }
    const int keys = du + dv + 2;
    int bin = NBINS;
    for (int b = NBINS - 1; b >= 0; --b)
        if (keys <= bin_max_keys(b)) bin = b;
    if (bin == NBINS) {  // beyond the largest LDS table: listed for the device-memory path (dcr_bfc_giant.hip)
        if (mode == MODE_BYTES || !giant) {
            res->flag_too_big = 1;
        } else if (finish_trivial) {
            const int idx = atomicAdd(&res->giant_count, 1);
            if (idx < GIANT_LIST_CAP) {
                int32_t *r = giant + (size_t)idx * 5;
                r[0] = (int32_t)s; r[1] = u; r[2] = v; r[3] = du; r[4] = dv;
            } else {
                res->flag_too_big = 1;
            }
        }
        return -1;
    }
    return bin;
}

__global__ void __launch_bounds__(256) k_classify(View g, int64_t cap_total, int curv_type, int mode, double *curv,
                                                   WorkLists wl, DevResult *res, double *bytes_total) {
    __shared__ int cnt[NBINS];
    __shared__ int base[NBINS];
    const int lane = threadIdx.x & 63;
    const int64_t chunk0 = (int64_t)blockIdx.x * CLASSIFY_CHUNK;
    if (threadIdx.x < NBINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    // phase 1: count per bin (and finish the trivial edges)
    int mycnt[NBINS] = {0, 0, 0, 0, 0};
    for (int o = threadIdx.x; o < CLASSIFY_CHUNK; o += 256) {
        const int bin = classify_slot(g, chunk0 + o, cap_total, curv_type, mode, curv, res, bytes_total, true, wl.giant);
#pragma unroll
        for (int b = 0; b < NBINS; ++b) mycnt[b] += (bin == b);
    }
#pragma unroll
    for (int b = 0; b < NBINS; ++b) {
        int t = mycnt[b];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0 && t) atomicAdd(&cnt[b], t);
    }
    __syncthreads();
    if (threadIdx.x < NBINS) {
        const int c = cnt[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(&res->work_count[threadIdx.x], c) : 0;
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    // phase 2: write the items
    for (int o = threadIdx.x; o < CLASSIFY_CHUNK; o += 256) {
        const int64_t s = chunk0 + o;
        const int bin = classify_slot(g, s, cap_total, curv_type, mode, curv, res, bytes_total, false, wl.giant);
#pragma unroll
        for (int b = 0; b < NBINS; ++b) {
            const unsigned long long m = __ballot(bin == b);
            if (m == 0) continue;
            int off = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) off = atomicAdd(&cnt[b], __popcll(m));
            off = __shfl(off, leader);
            if (bin == b) wl.w[b][base[b] + off + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)s;
        }
    }
}

__global__ void k_clear_counts(DevResult *res, int keep_guard) {
    if (threadIdx.x < 8 && !keep_guard) res->misc[threadIdx.x] = 0;
    if (threadIdx.x < NBINS) {
        res->work_count[threadIdx.x] = 0;
        res->work_next[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        res->flag_too_big = 0;
        res->giant_count = 0;
    }
}

static int ensure_work(dcr_graph *g) {
    if (g->work_cap >= g->cap_total / 2 + 64 && g->work[0]) return DCR_OK;
    int64_t cap = g->cap_total / 2 + 64;  // at most one undirected edge per two directed slots
    for (int b = 0; b < NBINS; ++b) {
        if (g->work[b]) (void)hipFree(g->work[b]);
        g->work[b] = nullptr;
        DCR_TRY(dev_alloc(&g->work[b], cap));
    }
    g->work_cap = cap;
    return DCR_OK;
}

// descriptors per team: one per possible member of a difference set (<= keys of the bin); the largest bin has
// no LDS left next to its 128 KiB table and streams rows one wave at a time instead
__host__ __device__ constexpr int bin_desc_cap(int b) { return b == 4 ? 0 : bin_max_keys(b); }
__host__ __device__ constexpr int bin_chunk(int b) { return b == 0 ? 16 : b == 1 ? 8 : b == 2 ? 2 : 1; }

template <int B, int MODE>
static void launch_bin(dcr_graph *g, const View &vw, int curv_type, double *bytes_total, int num_cu,
                       hipStream_t st) {
    constexpr int SLOTS = BIN_SLOTS[B];
    constexpr int TEAM = BIN_TEAM[B];
    constexpr int LDS = SLOTS * 4 * (bin_desc_cap(B) > 0 ? 2 : 1) + bin_desc_cap(B) * 8 + 512;
    // persistent grid: as many teams as fit a CU (wave slots, LDS), items dequeued dynamically
    int per_cu = (160 * 1024) / LDS;
    const int by_waves = 32 / (TEAM / 64);
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    const int grid = num_cu * per_cu;
    hipLaunchKernelGGL((k_edge_pass<SLOTS, TEAM, MODE, bin_desc_cap(B), bin_chunk(B)>), dim3(grid), dim3(TEAM), 0, st,
                       vw, g->work[B], &g->dres->work_count[B], &g->dres->work_next[B], curv_type, g->curv,
                       bytes_total);
}

template <int MODE>
static int run_pass(dcr_graph *g, int curv_type, double *bytes_total, bool incremental = false, bool nc_rest = false) {
    DCR_TRY(ensure_work(g));
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, incremental ? g->dirty : nullptr,
            (int32_t)g->n, nc_rest ? 1 : 0, nullptr};
    WorkLists wl;
    for (int b = 0; b < NBINS; ++b) wl.w[b] = g->work[b];
    // can an edge exceed the largest table at all?  (max_deg_bound is an upper bound of every degree)
    const bool giant_possible = MODE != MODE_BYTES && 2 * (int64_t)g->max_deg_bound + 2 > bin_max_keys(NBINS - 1);
    if (giant_possible && !g->giant_list) DCR_TRY(dev_alloc(&g->giant_list, (int64_t)GIANT_LIST_CAP * 5));
    wl.giant = giant_possible ? g->giant_list : nullptr;
    if (g->num_cu <= 0) {
        g->num_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0)
            g->num_cu = prop.multiProcessorCount;
    }
    const int num_cu = g->num_cu;
    hipLaunchKernelGGL(k_clear_counts, dim3(1), dim3(64), 0, g->stream, g->dres, nc_rest ? 1 : 0);
    int64_t blocks = (g->cap_total + CLASSIFY_CHUNK - 1) / CLASSIFY_CHUNK;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_classify, dim3((unsigned)blocks), dim3(256), 0, g->stream, vw, g->cap_total, curv_type, MODE,
                       g->curv, wl, g->dres, bytes_total);
    if (curv_type != DCR_CURV_1D || MODE == MODE_BYTES) {
        // the bins are independent: fork them onto side streams so their tails overlap; heaviest first
        static const bool serial = getenv("DCR_SERIAL_BINS") != nullptr;  // debugging aid: one stream
        hipStream_t s0 = g->stream, s1 = g->stream, s2 = g->stream, s3 = g->stream;
        if (!serial) {
            DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
            for (int b = 0; b < NBINS - 1; ++b) DCR_HIP(hipStreamWaitEvent(g->side[b], g->ev_fork, 0));
            s0 = g->side[0]; s1 = g->side[1]; s2 = g->side[2]; s3 = g->side[3];
        }
        launch_bin<3, MODE>(g, vw, curv_type, bytes_total, num_cu, g->stream);
        launch_bin<4, MODE>(g, vw, curv_type, bytes_total, num_cu, s0);
        launch_bin<2, MODE>(g, vw, curv_type, bytes_total, num_cu, s1);
        launch_bin<1, MODE>(g, vw, curv_type, bytes_total, num_cu, s2);
        launch_bin<0, MODE>(g, vw, curv_type, bytes_total, num_cu, s3);
        if (!serial) {
            for (int b = 0; b < NBINS - 1; ++b) {
                DCR_HIP(hipEventRecord(g->ev_join[b], g->side[b]));
                DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[b], 0));
            }
        }
    }
    DCR_HIP(hipGetLastError());
    if (giant_possible) DCR_TRY(process_giant_edges(g, curv_type));
    return DCR_OK;
}

int launch_curvature_pass(dcr_graph *g, int curv_type, bool incremental) {
    g->last_engine = 2;
    if (h2_can_take(g, curv_type, incremental)) {  // DCR_PASS=h2: full Balanced Forman passes by the two-hop kernels
        g->last_engine = 0;
        return launch_curvature_pass_h2(g);
    }
    if (curv_type == DCR_CURV_1D || g->pass_impl == 1) {
        g->last_engine = 1;
        if (curv_type == DCR_CURV_BFC || curv_type == DCR_CURV_1D)
            return run_pass<MODE_BFC>(g, curv_type, nullptr, incremental);
        return run_pass<MODE_TRI>(g, curv_type, nullptr, incremental);
    }
    // node-centric kernels; the edge-centric ones only for edges beyond their degree limits (two hubs with more
    // than NC_MAXD neighbours each), which cannot exist while the largest degree is within the limit
    DCR_TRY(launch_curvature_pass_nc(g, curv_type, incremental));
    if (g->max_deg_bound > NC_MAXD) {
        if (curv_type == DCR_CURV_BFC) DCR_TRY(run_pass<MODE_BFC>(g, curv_type, nullptr, incremental, true));
        else DCR_TRY(run_pass<MODE_TRI>(g, curv_type, nullptr, incremental, true));
        DCR_TRY(process_hub_edges(g, curv_type, incremental));  // hub-to-small-node edges (dcr_bfc_giant.hip)
    }
    return DCR_OK;
}

template <int B>
static void launch_single(dcr_graph *g, const View &vw, int u, int v, int64_t *out6) {
    hipLaunchKernelGGL((k_edge_single<BIN_SLOTS[B], BIN_TEAM[B], bin_desc_cap(B)>), dim3(1), dim3(BIN_TEAM[B]), 0,
                       g->stream, vw, u, v, out6);
}

}  // namespace dcr

using namespace dcr;

extern "C" {

static int curvature_pass_impl(dcr_graph *g, int curv_type, bool want_incremental, bool with_argmin = false);

// the per-node flags of the incremental pass and, with them, the list of flagged nodes (DevResult::touched_n: whoever zeroes the
// flags resets the list — a node is appended when its byte leaves zero)
static void clear_dirty_flags(dcr_graph *g) {
    (void)hipMemsetAsync(g->dirty, 0, (size_t)(g->n > 0 ? g->n : 1), g->stream);
    (void)hipMemsetAsync(&g->dres->touched_n, 0, sizeof(int32_t), g->stream);
}

int dcr_curvature_pass(dcr_graph *g, int curv_type) { return curvature_pass_impl(g, curv_type, false); }

int dcr_curvature_pass_incremental(dcr_graph *g, int curv_type) { return curvature_pass_impl(g, curv_type, true); }

int dcr_curvature_pass_argmin(dcr_graph *g, int curv_type, int incremental, int32_t *out_u, int32_t *out_v,
                              double *out_val) {
    DCR_TRY(curvature_pass_impl(g, curv_type, incremental != 0, true));
    if (g->hres->ext_slot < 0) DCR_FAIL(DCR_ENOTFOUND, "graph has no edges");
    g->am_x = g->hres->ext_u;
    g->am_y = g->hres->ext_v;
    g->am_dx = g->hres->ext_du;
    g->am_dy = g->hres->ext_dv;
    g->am_valid = true;
    if (out_u) *out_u = g->hres->ext_u;
    if (out_v) *out_v = g->hres->ext_v;
    if (out_val) *out_val = g->hres->ext_val;
    return DCR_OK;
}

static int curvature_pass_impl(dcr_graph *g, int curv_type, bool want_incremental, bool with_argmin) {
    if (!g) DCR_FAIL(DCR_EINVAL, "null graph");
    if (curv_type < DCR_CURV_BFC || curv_type > DCR_CURV_HAANTJES) DCR_FAIL(DCR_EINVAL, "unknown curvature type");
    DCR_HIP(hipSetDevice(g->device));
    g->amax_valid = false;
    g->ext_part_valid = false;  // (set again by the two-hop pass's closing kernel)
    // first minimum in G.edges order, same host sync as the pass: from the closing kernel's per-block minima when it left some
    bool dirty_clear_pending = false;   // the node flags still to be zeroed: by the sweep for the extrema when one follows
    auto argmin_after_pass = [&]() -> int {
        // (a pass of the node-centric kernels — incremental ones above all — leaves no partial extrema: ONE sweep takes both,
        //  the stale arg-max of the removal step then is a 5 us reduction as behind the two-hop pass; DCR_ARGEXT_BOTH=0: two sweeps)
        static const bool both = !(getenv("DCR_ARGEXT_BOTH") && atoi(getenv("DCR_ARGEXT_BOTH")) == 0);
        if (!g->ext_part_valid && both) DCR_TRY(launch_argext_both(g, nullptr, dirty_clear_pending));
        else if (dirty_clear_pending) clear_dirty_flags(g);
        dirty_clear_pending = false;
        return g->ext_part_valid ? launch_argext_from_parts(g, 0) : launch_argext(g, 0, -1, -1);
    };
    // incremental is only sound on top of a complete buffer of the same curvature kind whose later edits were all
    // recorded in the dirty flags (dcr_graph_add_edge / _remove_edge / dcr_sdrf_tail do that)
    const bool incremental = want_incremental && g->curv_valid && g->curv_type_last == curv_type && g->dirty_tracked;
    if (g->profile) DCR_HIP(hipEventRecord(g->ev0, g->stream));
    g->h2_cleared_dirty = false;
    DCR_TRY(launch_curvature_pass(g, curv_type, incremental));
    // (the two-hop launch zeroes the flags in its first kernel: one fill launch less at the tail of every pass)
    if (!(g->last_engine == 0 && g->h2_cleared_dirty)) {
        if (with_argmin) dirty_clear_pending = true;
        else clear_dirty_flags(g);
    }
    g->dirty_tracked = true;
    g->pending_edits = 0;
    if (g->profile) DCR_HIP(hipEventRecord(g->ev1, g->stream));
    if (with_argmin) DCR_TRY(argmin_after_pass());
    DCR_TRY(sync_result(g));
    if (g->last_engine == 0) {
        for (int c = 0; c < 5; ++c) g->h2_last_count[c] = g->hres->h2_count[c];
        for (int again = 0; again < 6 && g->hres->misc[0] == 0; ++again) {
            if (g->hres->h2_status == 3) {
                // launched without its retry stage, and some node's tables filled up: nothing was written; from now on every
                // pass of this graph carries the stage
                g->h2_expect_retry = true;
            } else if (g->hres->h2_status == 2 && h2_grow_pools(g)) {
                // the pools of its triangle step were too small (dense neighbourhoods): run it again with what it asked for
            } else {
                break;
            }
            g->ext_part_valid = false;
            DCR_TRY(launch_curvature_pass(g, curv_type, false));
            if (g->profile) DCR_HIP(hipEventRecord(g->ev1, g->stream));
            if (with_argmin) DCR_TRY(argmin_after_pass());
            DCR_TRY(sync_result(g));
            for (int c = 0; c < 5; ++c) g->h2_last_count[c] = g->hres->h2_count[c];
        }
        if (g->hres->h2_status != 0) g->ext_part_valid = false;  // (the closing kernel wrote nothing)
        // (advisor, round 4) a two-hop pass that ended with a status may have left its edge set partly patched (k_h2_eset_apply
        // clears the journal before it applies it): the next two-hop pass rebuilds the set instead of trusting it
        if (g->hres->h2_status != 0) g->h2_eset_valid = false;
        if (g->hres->h2_status != 0 && g->hres->misc[0] == 0) {
            // a table of the two-hop pass filled up (keys of a split node hashed unevenly) or a unit list overflowed:
            // nothing it wrote is kept, the node-centric kernels redo the whole pass
            const int keep = g->pass_impl;
            g->ext_part_valid = false;
            g->pass_impl = 2;
            const int rc = launch_curvature_pass(g, curv_type, false);
            g->pass_impl = keep;
            DCR_TRY(rc);
            if (g->profile) DCR_HIP(hipEventRecord(g->ev1, g->stream));
            if (with_argmin) DCR_TRY(argmin_after_pass());
            DCR_TRY(sync_result(g));
        }
    }
    if (g->profile) {
        float ms = 0.f;
        DCR_HIP(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->pass_ms_total += ms;
        g->pass_count += 1;
    }
    if (g->hres->misc[0] != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "curvature kernel invariant %d violated: row {%d,%d} a=%d b=%d block=%d thread=%d",
                 g->hres->misc[0], g->hres->misc[1], g->hres->misc[2], g->hres->misc[3], g->hres->misc[4],
                 g->hres->misc[5], g->hres->misc[6]);
        DCR_FAIL(DCR_EHIP, buf);
    }
    if (g->hres->flag_too_big)
        DCR_FAIL(DCR_ECAPACITY, "more than 65536 edges with deg(u)+deg(v)+2 > 16384 (beyond every LDS table) in one pass");
    g->curv_type_last = curv_type;
    g->curv_valid = true;
    return DCR_OK;
}

int dcr_pass_engine(dcr_graph *g, int *out) {
    if (!g || !out) DCR_FAIL(DCR_EINVAL, "null argument");
    *out = g->last_engine;
    return DCR_OK;
}

int dcr_bfc_ingredients(dcr_graph *g, int32_t u, int32_t v, int64_t out6[6]) {
    if (!g || !out6) DCR_FAIL(DCR_EINVAL, "null argument");
    if (u < 0 || v < 0 || u >= g->n || v >= g->n || u == v) DCR_FAIL(DCR_EINVAL, "bad node ids");
    DCR_HIP(hipSetDevice(g->device));
    int32_t du, dv;
    DCR_TRY(dcr_graph_degree(g, u, &du));
    DCR_TRY(dcr_graph_degree(g, v, &dv));
    const int keys = du + dv + 2;
    int64_t *d_out = nullptr;
    DCR_TRY(dev_alloc(&d_out, 6));
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, nullptr, (int32_t)g->n, 0, nullptr};
    if (keys <= bin_max_keys(0)) launch_single<0>(g, vw, u, v, d_out);
    else if (keys <= bin_max_keys(1)) launch_single<1>(g, vw, u, v, d_out);
    else if (keys <= bin_max_keys(2)) launch_single<2>(g, vw, u, v, d_out);
    else if (keys <= bin_max_keys(3)) launch_single<3>(g, vw, u, v, d_out);
    else if (keys <= bin_max_keys(4)) launch_single<4>(g, vw, u, v, d_out);
    else {  // beyond every LDS table: device-memory path (the pair must be an edge for its flags to mean anything)
        int rc = giant_edge(g, u, v, du, dv, -1, DCR_CURV_BFC, true, d_out);
        if (rc != DCR_OK) {
            (void)hipFree(d_out);
            return rc;
        }
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out6, d_out, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) DCR_FAIL(DCR_EHIP, hipGetErrorString(e));
    return DCR_OK;
}

int dcr_curvature_edge(dcr_graph *g, int32_t u, int32_t v, int curv_type, double *out) {
    if (!out) DCR_FAIL(DCR_EINVAL, "null out");
    int64_t q[6];
    DCR_TRY(dcr_bfc_ingredients(g, u, v, q));
    const int du = (int)q[0], dv = (int)q[1], T = (int)q[2];
    switch (curv_type) {
        case DCR_CURV_1D: *out = (double)(4 - du - dv); break;
        case DCR_CURV_AUGMENTED: *out = (double)(4 - du - dv + 3 * T); break;
        case DCR_CURV_HAANTJES: *out = (double)T; break;
        case DCR_CURV_BFC:
            // the closing expression is evaluated by the same inline function the kernels use
            *out = (du < dv ? du : dv) == 1 ? 0.0 : bfc_formula(du, dv, T, (int)q[3], (int)q[4], (int)q[5]);
            break;
        default: DCR_FAIL(DCR_EINVAL, "unknown curvature type");
    }
    return DCR_OK;
}

static int algorithmic_bytes(dcr_graph *g, double out2[2]) {
    DCR_HIP(hipSetDevice(g->device));
    double *d_total = nullptr;
    DCR_TRY(dev_alloc(&d_total, 2));
    DCR_HIP(hipMemsetAsync(d_total, 0, 2 * sizeof(double), g->stream));
    int rc = run_pass<MODE_BYTES>(g, DCR_CURV_BFC, d_total);
    if (rc == DCR_OK) {
        hipError_t e = hipMemcpyAsync(out2, d_total, 2 * sizeof(double), hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = DCR_EHIP;
        }
    }
    (void)hipFree(d_total);
    return rc;
}

int dcr_bfc_algorithmic_bytes(dcr_graph *g, double *out_bytes) {
    if (!g || !out_bytes) DCR_FAIL(DCR_EINVAL, "null argument");
    double b[2];
    DCR_TRY(algorithmic_bytes(g, b));
    *out_bytes = b[0];
    return DCR_OK;
}

int dcr_bfc_algorithmic_bytes_one_sided(dcr_graph *g, double *out_bytes) {
    if (!g || !out_bytes) DCR_FAIL(DCR_EINVAL, "null argument");
    double b[2];
    DCR_TRY(algorithmic_bytes(g, b));
    *out_bytes = b[1];
    return DCR_OK;
}

}  // extern "C"
