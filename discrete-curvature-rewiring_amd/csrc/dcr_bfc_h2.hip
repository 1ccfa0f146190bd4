// Two-hop curvature pass of libdcr_hip.so: the default implementation of a FULL Balanced Forman pass.
//
// Replaces compute_curvature_graph(G, 'bfc') at rewiring/sdrf_no_cuda.py:24, i.e. E calls of bfc_naive.bfc_edge
// (curvature/bfc_naive.py:7-40), with the same integers and the same float64 closing expression as the other two
// implementations (dcr_bfc_nc.hip, dcr_bfc.hip) — only the way the integers are counted differs.
//
// For an edge {u,v}: DX = N(u) \ N(v) \ {v}, DY = N(v) \ N(u) \ {u}.  bfc_naive.py:26-29,36-37 needs, for every w in DY,
// c(w) = |N(w) ∩ DX| (|sq2| = how many are positive, gamma = the largest; the same from v's side gives |sq1|).  Now
//
//     c(w) = |N(w) ∩ N(u)|  -  1  -  |N(w) ∩ N(u) ∩ N(v)|
//
// (v itself is a common neighbour of w and u; the last term are the triangle partners of {u,v} adjacent to w).  The first
// term M_u(w) does not depend on v: it is row u of A·A.  So instead of streaming, per edge, the rows of all of DY against
// N(u) (1.08 G adjacency entries per pass on the 100k-node bench graph), every node u
//   A. builds M_u once as a hash map in LDS by streaming the rows of its neighbours (Σ_u Σ_{k∈N(u)} d_k = Σ d² entries:
//      0.117 G on that graph), members of N(u) flagged;
//   B. streams the same rows once more: for neighbour v and w in row v, a flagged w is a triangle (T, bfc_naive.py:25),
//      w = u gives the slot of the reverse entry, anything else contributes M_u(w) - 1.  96 % of the 2-hop entries have
//      M_u(w) = 1 and contribute nothing;
//   C. only for edges with T > 0 AND some positive count: the triangle partners' rows are streamed against a small
//      per-edge table of the positive candidates to subtract the third term (5 % of the edges).
// Each node writes, per adjacency slot u->v, {|sq| on v's side, max count, T, reverse slot}; a final kernel joins the two
// records of an edge and evaluates the float64 closing expression in the reference's order (bfc_naive.py:31-40).
//
// Nodes are grouped by K = deg + 1 + Σ neighbour degrees (an upper bound of the keys of M_u): up to 1,280 keys a wave
// owns a node and a private 2,048-slot table; up to 5,120 (10,240) a workgroup of 8 (16) waves shares a table of 8,192
// (16,384) slots; beyond that the KEYS of a node are split by a second hash into P partitions, each a unit of its own
// (every unit streams all rows but keeps only its share; results meet in the record through integer atomics), which
// also spreads a hub over P workgroups.  Units are laid out heaviest first.  Bounds: HBM / L2 row streaming
// (2 x 4 B x Σ d² per pass) and LDS atomics; no MFMA (integer set counting).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "dcr_bfc_common.h"

namespace dcr {

constexpr unsigned H2_EMPTY = 0xFFFFFFFFu;
constexpr int H2_MAXDEG = 8190;  // flagged neighbours live in every partition's table; 15-bit counters
constexpr int H2_CLASSES = 3;
constexpr int H2_WB = 8;         // weight buckets per class (units are laid out heaviest bucket first)
#ifndef H2_Q
#define H2_Q 2                   // 16-byte pieces per lane in flight in the streaming loops
#endif

__host__ __device__ constexpr int h2_cap(int c) { return c == 0 ? 2048 : c == 1 ? 8192 : 16384; }
__host__ __device__ constexpr int h2_maxkeys(int c) { return c == 0 ? 1280 : c == 1 ? 5120 : 10240; }
__host__ __device__ constexpr int h2_waves(int c) { return c == 0 ? 1 : c == 1 ? 8 : 16; }
__host__ __device__ constexpr int h2_ecap(int c) { return c == 0 ? 128 : 256; }
constexpr int H2_WPB0 = 2;  // waves (= nodes in flight) per workgroup of the wave class

__device__ inline void h2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int NW>
__device__ inline void h2_sync() {
    if (NW == 1) h2_wave_sync();
    else __syncthreads();
}

// ---- the table: open addressing over 4-slot buckets (one 16-byte LDS read settles almost every access) -------------
template <int CAP>
__device__ inline unsigned h2_bucket(unsigned key) {
    constexpr int BITS = __builtin_ctz(CAP / 4);
    unsigned prod;  // Fibonacci hashing of the low 24 bits with the full-rate 24-bit multiply (see dcr_bfc_nc.hip)
    asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(prod) : "v"(key));
    return (prod >> (24 - BITS)) & ((1u << BITS) - 1);
}
// which partition of a split node a key belongs to: a hash independent of the bucket hash
__device__ inline int h2_part(unsigned key, int nparts) {
    return (int)((((key * 0x85EBCA6Bu) >> 16) * (unsigned)nparts) >> 16);
}
__device__ inline int h2_match(const uint4 e, unsigned w) {
    return e.x == w ? 0 : e.y == w ? 1 : e.z == w ? 2 : e.w == w ? 3 : -1;
}

// slot of w, continuing from its (already read) home bucket; -1: absent
template <int CAP>
__device__ inline int h2_find_from(const unsigned *key, unsigned b, uint4 e, unsigned w) {
    const uint4 *tb = reinterpret_cast<const uint4 *>(key);
    for (int walk = 0; walk < CAP / 4; ++walk) {
        const int pos = h2_match(e, w);
        if (pos >= 0) return (int)(b * 4) + pos;
        if (e.w == H2_EMPTY) return -1;  // slots of a bucket fill in order: a free last slot means the key never spilled
        b = (b + 1) & (CAP / 4 - 1);
        e = tb[b];
    }
    return -1;
}
template <int CAP>
__device__ inline int h2_find(const unsigned *key, unsigned w) {
    const unsigned b = h2_bucket<CAP>(w);
    return h2_find_from<CAP>(key, b, reinterpret_cast<const uint4 *>(key)[b], w);
}

// slot of w, inserting it if absent; -1: the table is full (reported by the caller, never loops forever).
// A lane claims the first free slot it saw; slots never empty again, so filled slots always form a prefix of a bucket and a
// key is never stored twice (a racing lane with the same key meets it on its way up).
template <int CAP>
__device__ inline int h2_insert_from(unsigned *key, unsigned b, uint4 e, unsigned w) {
    const uint4 *tb = reinterpret_cast<const uint4 *>(key);
    for (int walk = 0; walk < CAP / 4; ++walk) {
        const int pos = h2_match(e, w);
        if (pos >= 0) return (int)(b * 4) + pos;
        int ep = e.x == H2_EMPTY ? 0 : e.y == H2_EMPTY ? 1 : e.z == H2_EMPTY ? 2 : e.w == H2_EMPTY ? 3 : 4;
        for (; ep < 4; ++ep) {
            const unsigned old = atomicCAS(&key[b * 4 + ep], H2_EMPTY, w);
            if (old == H2_EMPTY || old == w) return (int)(b * 4) + ep;
        }
        b = (b + 1) & (CAP / 4 - 1);
        e = tb[b];
    }
    return -1;
}
template <int CAP>
__device__ inline int h2_insert(unsigned *key, unsigned w) {
    const unsigned b = h2_bucket<CAP>(w);
    return h2_insert_from<CAP>(key, b, reinterpret_cast<const uint4 *>(key)[b], w);
}

// per-slot state, two 16-bit halves per word: bit 15 = member of N(u), bits 0-14 = occurrences in the neighbours' rows
__device__ inline void h2_cnt_flag(unsigned *cnt, int s) { atomicOr(&cnt[s >> 1], 0x8000u << ((s & 1) * 16)); }
__device__ inline void h2_cnt_add(unsigned *cnt, int s) { atomicAdd(&cnt[s >> 1], 1u << ((s & 1) * 16)); }
__device__ inline unsigned h2_cnt_get(const unsigned *cnt, int s) { return (cnt[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu; }

template <int ECAP>
struct H2Scratch {
    int2 desc[64];   // {start, length} of the rows of the current batch
    int poff[66];    // exclusive prefix of their piece counts; poff[64] = total
    int rowT[64], rowPos[64], rowMx[64], rowRev[64];  // phase B accumulators per row of the batch
    unsigned ekey[ECAP];  // step C: the positive candidates of one edge ...
    unsigned eval[ECAP];  // ... count << 16 | triangle partners adjacent to it
};

__device__ inline unsigned h2_piece_mask(int a, int lo, int hi) {
    const int s = lo - a, t = hi - a;
    const unsigned head = s > 0 ? (0xFu << s) & 0xFu : 0xFu;
    const unsigned tail = t < 4 ? (1u << t) - 1u : 0xFu;
    return head & tail;
}

// row of piece j of the flat piece list of a batch (same scheme as dcr_bfc_nc.hip: two ballots bracket the rows of a
// 64-piece step, a few shuffles or a short bisection settle the lane's own)
__device__ inline int h2_piece_row(const int *poff, int poff_lane, int j, int j_first, int j_last) {
    const int rf = __popcll(__ballot(poff_lane <= j_first)) - 1;
    const int rl = __popcll(__ballot(poff_lane <= j_last)) - 1;
    int r = rf;
    const int span = rl - rf;
    if (span <= 4) {
        for (int b = rf + 1; b <= rl; ++b) r += (__shfl(poff_lane, b) <= j);
    } else {
        for (int step = span >= 32 ? 32 : span >= 16 ? 16 : span >= 8 ? 8 : 4; step > 0; step >>= 1)
            if (r + step <= rl && poff[r + step] <= j) r += step;
    }
    return r;
}

template <int ECAP>
__device__ inline unsigned h2_ehash(unsigned key) {
    constexpr int BITS = __builtin_ctz(ECAP);
    return (key * 0x9E3779B1u) >> (32 - BITS);
}

// ---- step C: one edge {u,v} with triangles and positive counts, by one wave ------------------------------------------
// Returns {|{w : c(w) > 0}|, max c(w)} over the w of row v that live in this unit's table, c(w) = M_u(w) - 1 - |N(w) ∩ Tset|.
template <int CAP, int ECAP>
__device__ inline int2 h2_edge_with_triangles(const View &g, int u, int2 rv, const unsigned *key, const unsigned *cnt,
                                              H2Scratch<ECAP> *sc, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const int32_t *rowv = g.col + rv.x;
    // how many positive candidates does the row hold?  They are handled ECAP / 4 at a time (table load <= 1/4).
    int nheavy = 0;
    for (int base = 0; base < rv.y; base += 64) {
        const int i = base + lane;
        const int w = i < rv.y ? rowv[i] : -1;
        bool hv = false;
        if (w >= 0 && w != u) {
            const int s = h2_find<CAP>(key, (unsigned)w);
            if (s >= 0) {
                const unsigned c16 = h2_cnt_get(cnt, s);
                hv = !(c16 & 0x8000u) && (c16 & 0x7FFFu) >= 2u;
            }
        }
        nheavy += __popcll(__ballot(hv));
    }
    const int rounds = (nheavy + ECAP / 4 - 1) / (ECAP / 4);
    int pos = 0, mx = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        for (int i = lane; i < ECAP; i += 64) {
            sc->ekey[i] = H2_EMPTY;
            sc->eval[i] = 0u;
        }
        h2_wave_sync();
        for (int base = 0; base < rv.y; base += 64) {
            const int i = base + lane;
            const int w = i < rv.y ? rowv[i] : -1;
            if (w >= 0 && w != u) {
                const int s = h2_find<CAP>(key, (unsigned)w);
                if (s >= 0) {
                    const unsigned c16 = h2_cnt_get(cnt, s);
                    const bool hv = !(c16 & 0x8000u) && (c16 & 0x7FFFu) >= 2u;
                    if (hv && (rounds == 1 || (int)(((unsigned)w * 0xC2B2AE35u) >> 8) % rounds == rd)) {
                        unsigned e = h2_ehash<ECAP>((unsigned)w);
                        bool placed = false;
                        for (int walk = 0; walk < ECAP; ++walk) {
                            const unsigned old = atomicCAS(&sc->ekey[e], H2_EMPTY, (unsigned)w);
                            if (old == H2_EMPTY || old == (unsigned)w) {
                                sc->eval[e] = (c16 & 0x7FFFu) << 16;  // (a row holds each id once: one writer per slot)
                                placed = true;
                                break;
                            }
                            e = (e + 1) & (ECAP - 1);
                        }
                        if (!placed) *status = 1;
                    }
                }
            }
        }
        h2_wave_sync();
        // the rows of the triangle partners t (flagged members of row v), one after the other, against the small table
        for (int base = 0; base < rv.y; base += 64) {
            const int i = base + lane;
            const int w = i < rv.y ? rowv[i] : -1;
            bool isT = false;
            if (w >= 0 && w != u) {
                const int s = h2_find<CAP>(key, (unsigned)w);
                isT = s >= 0 && (h2_cnt_get(cnt, s) & 0x8000u);
            }
            unsigned long long mT = __ballot(isT);
            while (mT) {
                const int b = __ffsll((long long)mT) - 1;
                mT &= mT - 1;
                const int t = __shfl(w, b);
                int2 rt = make_int2(0, 0);
                if (t >= 0 && t < g.n) rt = g.rowinfo[t];
                if (!row_ok(g, rt, 31, t, u)) rt = make_int2(0, 0);
                const int hi = rt.x + rt.y;
                for (int a = (rt.x & ~3) + 4 * lane; a < hi; a += 256) {
                    const int4 p = load_piece(g.col, a);
                    const unsigned m = h2_piece_mask(a, rt.x, hi);
                    const unsigned x[4] = {(unsigned)p.x, (unsigned)p.y, (unsigned)p.z, (unsigned)p.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (!((m >> jj) & 1u)) continue;
                        unsigned e = h2_ehash<ECAP>(x[jj]);
                        for (int walk = 0; walk < ECAP; ++walk) {
                            const unsigned kx = sc->ekey[e];
                            if (kx == x[jj]) {
                                atomicAdd(&sc->eval[e], 1u);
                                break;
                            }
                            if (kx == H2_EMPTY) break;
                            e = (e + 1) & (ECAP - 1);
                        }
                    }
                }
            }
        }
        h2_wave_sync();
        for (int e = lane; e < ECAP; e += 64) {
            if (sc->ekey[e] != H2_EMPTY) {
                const unsigned val = sc->eval[e];
                const int c = (int)(val >> 16) - 1 - (int)(val & 0xFFFFu);
                if (c > 0) {
                    ++pos;
                    mx = c > mx ? c : mx;
                }
            }
        }
        h2_wave_sync();
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_xor(pos, off);
        const int o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    return make_int2(pos, mx);
}

// ---- phases A and B: the rows of the neighbours of u, in batches of 64 rows per wave --------------------------------
// Wave `wid` of NW takes the rows i = wid, wid + NW, ... of row u (strided: a hub's heaviest rows, adjacent at the
// front of its row, spread over the waves); lane l of the batch starting at `base` stands for row base + l * NW + wid.
template <int CAP, int NW, int ECAP, bool PARTS, int PHASE>
__device__ inline void h2_stream(const View &g, int u, int2 ru, int part, int nparts, unsigned *key, unsigned *cnt,
                                 H2Scratch<ECAP> *sc, uint4 *rec, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const int wid = NW == 1 ? 0 : (int)(threadIdx.x >> 6);
    const uint4 *tb = reinterpret_cast<const uint4 *>(key);
    for (int base = 0; base < ru.y; base += 64 * NW) {
        const int i = base + lane * NW + wid;
        int k = -1;
        int2 rk = make_int2(0, 0);
        if (i < ru.y) {
            k = g.col[ru.x + i];
            if (k >= 0 && k < g.n && k != u) {
                rk = g.rowinfo[k];
                if (!row_ok(g, rk, 32, k, u)) rk = make_int2(0, 0);
            } else {
                k = -1;
            }
        }
        const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
        int incl = np;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int P = __shfl(incl, 63);
        if (P == 0) continue;  // uniform: no row in this wave's share of the batch
        const int poff_lane = incl - np;
        sc->desc[lane] = rk;
        sc->poff[lane] = poff_lane;
        if (lane == 0) sc->poff[64] = P;
        if (PHASE == 1) {
            sc->rowT[lane] = 0;
            sc->rowPos[lane] = 0;
            sc->rowMx[lane] = 0;
            sc->rowRev[lane] = -1;
        }
        h2_wave_sync();
        for (int j0 = 0; j0 < P; j0 += 64 * H2_Q) {
            int4 w[H2_Q];
            int rr[H2_Q], aa[H2_Q];
#pragma unroll
            for (int q = 0; q < H2_Q; ++q) {
                const int j = j0 + 64 * q + lane;
                rr[q] = -1;
                aa[q] = 0;
                w[q] = make_int4(0, 0, 0, 0);
                const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
                if (jf >= P) continue;  // uniform
                const int r = h2_piece_row(sc->poff, poff_lane, j < P ? j : jl, jf, jl);
                if (j < P) {
                    const int2 d = sc->desc[r];
                    const int a = (d.x & ~3) + 4 * (j - sc->poff[r]);
                    w[q] = load_piece(g.col, a);
                    rr[q] = r;
                    aa[q] = a;
                }
            }
#pragma unroll
            for (int q = 0; q < H2_Q; ++q) {
                if (rr[q] < 0) continue;
                const int2 d = sc->desc[rr[q]];
                const unsigned vm = h2_piece_mask(aa[q], d.x, d.x + d.y);
                const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
                bool take[4];
                unsigned b[4];
                uint4 e[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    take[jj] = ((vm >> jj) & 1u) && kk[jj] != (unsigned)u && kk[jj] < (unsigned)g.n;
                    if (PHASE == 0 && PARTS) take[jj] = take[jj] && h2_part(kk[jj], nparts) == part;
                    if (PHASE == 1 && ((vm >> jj) & 1u) && kk[jj] == (unsigned)u) sc->rowRev[rr[q]] = aa[q] + jj;
                    b[jj] = h2_bucket<CAP>(kk[jj]);
                    e[jj] = take[jj] ? tb[b[jj]] : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (!take[jj]) continue;
                    if (PHASE == 0) {
                        const int s = h2_insert_from<CAP>(key, b[jj], e[jj], kk[jj]);
                        if (s < 0) *status = 1;
                        else h2_cnt_add(cnt, s);
                    } else {
                        const int s = h2_find_from<CAP>(key, b[jj], e[jj], kk[jj]);
                        if (s >= 0) {
                            const unsigned c16 = h2_cnt_get(cnt, s);
                            if (c16 & 0x8000u) {
                                atomicAdd(&sc->rowT[rr[q]], 1);
                            } else {
                                const int c = (int)(c16 & 0x7FFFu) - 1;
                                if (c > 0) {
                                    atomicAdd(&sc->rowPos[rr[q]], 1);
                                    atomicMax(&sc->rowMx[rr[q]], c);
                                }
                            }
                        }
                    }
                }
            }
        }
        h2_wave_sync();
        if (PHASE == 1) {
            int T = sc->rowT[lane], pos = sc->rowPos[lane], mx = sc->rowMx[lane];
            const int rev = sc->rowRev[lane];
            // triangles AND positive counts: the counts of this edge still include the triangle partners (step C)
            unsigned long long todo = __ballot(k >= 0 && T > 0 && pos > 0);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int2 rv = make_int2(__shfl(rk.x, l), __shfl(rk.y, l));
                const int2 pm = h2_edge_with_triangles<CAP, ECAP>(g, u, rv, key, cnt, sc, status);
                if (lane == l) {
                    pos = pm.x;
                    mx = pm.y;
                }
            }
            if (k >= 0) {
                const int64_t slot = (int64_t)ru.x + i;
                if (rev < 0 || rev >= g.cap_total) {
                    row_ok(g, make_int2(-1, rev), 33, u, k);  // adjacency not symmetric: report, never publish
                } else if (!PARTS) {
                    rec[slot] = make_uint4((unsigned)pos, (unsigned)mx, (unsigned)T, (unsigned)rev);
                } else {
                    unsigned *r4 = reinterpret_cast<unsigned *>(rec + slot);
                    if (pos) atomicAdd(&r4[0], (unsigned)pos);
                    if (mx) atomicMax(&r4[1], (unsigned)mx);
                    if (part == 0) {  // the flagged neighbours live in every partition's table: T and the slot are whole
                        r4[2] = (unsigned)T;
                        r4[3] = (unsigned)rev;
                    }
                }
            }
            h2_wave_sync();
        }
    }
}

// one unit: node u, key partition `part` of `nparts`, by NW waves sharing `key` / `cnt`
template <int CAP, int NW, int ECAP, bool PARTS>
__device__ inline void h2_node(const View &g, int u, int2 ru, int part, int nparts, unsigned *key, unsigned *cnt,
                               H2Scratch<ECAP> *sc, uint4 *rec, int32_t *status) {
    const int tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    constexpr int NT = 64 * NW;
    uint4 *k4 = reinterpret_cast<uint4 *>(key);
    uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
    for (int i = tid; i < CAP / 4; i += NT) k4[i] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
    for (int i = tid; i < CAP / 8; i += NT) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    h2_sync<NW>();
    for (int i = tid; i < ru.y; i += NT) {  // the members of N(u), flagged (in every partition's table)
        const int k = g.col[ru.x + i];
        if (k >= 0 && k < g.n && k != u) {
            const int s = h2_insert<CAP>(key, (unsigned)k);
            if (s < 0) *status = 1;
            else h2_cnt_flag(cnt, s);
        }
    }
    h2_sync<NW>();
    h2_stream<CAP, NW, ECAP, PARTS, 0>(g, u, ru, part, nparts, key, cnt, sc, rec, status);
    h2_sync<NW>();
    h2_stream<CAP, NW, ECAP, PARTS, 1>(g, u, ru, part, nparts, key, cnt, sc, rec, status);
    h2_sync<NW>();  // the table is rewritten by the next unit
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
// wave class: a wave owns a unit and its private table; units are taken grid-stride (heaviest first in the list)
template <int CAP, int ECAP>
__global__ void __launch_bounds__(64 * H2_WPB0) k_h2_wave(View g, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                          uint4 *rec, int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned key_all[H2_WPB0][CAP];
    __shared__ __attribute__((aligned(16))) unsigned cnt_all[H2_WPB0][CAP / 2];
    __shared__ H2Scratch<ECAP> sc_all[H2_WPB0];
    const int wid = threadIdx.x >> 6;
    const int total = *count;
    if (total < 0 || total > unit_cap) {
        row_ok(g, make_int2(-1, total), 34, 0, 0);
        return;
    }
    for (int64_t it = (int64_t)blockIdx.x * H2_WPB0 + wid; it < total; it += (int64_t)gridDim.x * H2_WPB0) {
        const int2 un = units[it];
        const int u = un.x;
        if (u < 0 || u >= g.n) {
            row_ok(g, make_int2(-1, u), 35, (int)it, total);
            continue;
        }
        const int2 ru = g.rowinfo[u];
        if (!row_ok(g, ru, 36, u, (int)it) || ru.y <= 0 || ru.y > H2_MAXDEG) continue;
        h2_node<CAP, 1, ECAP, false>(g, u, ru, 0, 1, key_all[wid], cnt_all[wid], &sc_all[wid], rec, status);
    }
}

// block classes: a workgroup of W waves shares one table per unit
template <int CAP, int W, int ECAP, bool PARTS>
__global__ void __launch_bounds__(64 * W) k_h2_block(View g, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                      uint4 *rec, int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned key[CAP];
    __shared__ __attribute__((aligned(16))) unsigned cnt[CAP / 2];
    __shared__ H2Scratch<ECAP> sc_all[W];
    const int wid = threadIdx.x >> 6;
    const int total = *count;
    if (total < 0 || total > unit_cap) {  // uniform
        row_ok(g, make_int2(-1, total), 37, 0, 0);
        return;
    }
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {  // every value steering the barriers is uniform
        const int2 un = units[it];
        const int u = un.x;
        const int nparts = PARTS ? (int)((unsigned)un.y >> 16) : 1, part = PARTS ? (un.y & 0xFFFF) : 0;
        bool ok = u >= 0 && u < g.n && nparts >= 1 && part < nparts;
        int2 ru = make_int2(0, 0);
        if (ok) {
            ru = g.rowinfo[u];
            ok = row_ok(g, ru, 38, u, (int)it) && ru.y > 0 && ru.y <= H2_MAXDEG;
        }
        if (!ok) continue;
        h2_node<CAP, W, ECAP, PARTS>(g, u, ru, part, nparts, key, cnt, &sc_all[wid], rec, status);
    }
}

// Σ_{k in N(u)} deg(k) for every u: a thread per adjacency slot, run-length sums inside a wave (rows are contiguous),
// one atomic per run
__global__ void __launch_bounds__(256) k_h2_weight(View g, int32_t *weight) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int u = -1, val = 0;
    if (s < g.cap_total) {
        u = g.slot_row[s];
        if (u >= 0 && u < g.n) {
            const int2 ru = g.rowinfo[u];
            if ((int)(s - ru.x) < ru.y && s >= ru.x) {
                const int v = g.col[s];
                if (v >= 0 && v < g.n) val = g.rowinfo[v].y;
            }
        } else {
            u = -1;
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(val, off), uu = __shfl_up(u, off);
        if (lane >= off && uu == u) val += t;
    }
    const int un = __shfl_down(u, 1);
    if (u >= 0 && val > 0 && (lane == 63 || un != u)) atomicAdd(&weight[u], val);
}

// ---- plan: class, partitions and weight bucket per node; units laid out heaviest bucket first inside each class -------
struct H2Lists {
    int2 *units[H2_CLASSES];
    int64_t cap[H2_CLASSES];
};
constexpr int H2_NB = H2_CLASSES * H2_WB;
constexpr int H2_PLAN_THREADS = 1024;

__device__ inline void h2_classify(int d, int S, int &cls, int &wb, int &nparts) {
    const int64_t K = (int64_t)d + 1 + S;
    nparts = 1;
    if (K <= h2_maxkeys(0)) {
        cls = 0;
        wb = (int)(K * H2_WB / (h2_maxkeys(0) + 1));
    } else if (K <= h2_maxkeys(1)) {
        cls = 1;
        wb = (int)((K - h2_maxkeys(0)) * H2_WB / (h2_maxkeys(1) - h2_maxkeys(0) + 1));
    } else {
        cls = 2;
        if (K > h2_maxkeys(2)) {  // split the keys: deg + 1 flagged neighbours in every part, a quarter of slack for the hash
            const int64_t room = h2_maxkeys(2) - d - 1;
            nparts = (int)(((int64_t)S * 5 / 4 + room - 1) / room);
            if (nparts > 65535) nparts = 65535;  // (cannot fit then: the table reports it and the pass falls back)
        }
        wb = S < 16384 ? 0 : S < 32768 ? 1 : S < 65536 ? 2 : S < 131072 ? 3 : S < 262144 ? 4 : S < 524288 ? 5 : S < 1048576 ? 6 : 7;
    }
    if (wb > H2_WB - 1) wb = H2_WB - 1;
    if (wb < 0) wb = 0;
}

template <int PHASE>
__global__ void __launch_bounds__(H2_PLAN_THREADS) k_h2_plan(View g, const int32_t *weight, H2Lists L, DevResult *res) {
    __shared__ int blk_count[H2_NB];
    __shared__ int blk_base[H2_NB];
    const int u = blockIdx.x * H2_PLAN_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < H2_NB) blk_count[threadIdx.x] = 0;
    __syncthreads();
    int bkt = -1, cls = -1, wb = 0, nparts = 0;
    if (u < g.n) {
        const int d = g.rowinfo[u].y;
        if (d > 0 && d <= H2_MAXDEG) {
            h2_classify(d, weight[u], cls, wb, nparts);
            bkt = cls * H2_WB + wb;
        }
    }
    const int nunits = bkt < 0 ? 0 : nparts;
    int my_off = 0;
    for (int b = 0; b < H2_NB; ++b) {
        const unsigned long long m = __ballot(bkt == b);
        if (m == 0) continue;
        int incl = bkt == b ? nunits : 0;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int tot = __shfl(incl, 63);
        int wave_off = 0;
        if (lane == 0) wave_off = atomicAdd(&blk_count[b], tot);
        wave_off = __shfl(wave_off, 0);
        if (bkt == b) my_off = wave_off + incl - nunits;
    }
    __syncthreads();
    if (threadIdx.x < H2_NB) {
        const int b = threadIdx.x, c = blk_count[b];
        if (PHASE == 0) {
            if (c) atomicAdd(&res->h2_bucket[b], c);
        } else {
            int before = 0;
            for (int h = b + 1; h < H2_NB; ++h)  // heavier buckets of the same class come first
                if (h / H2_WB == b / H2_WB) before += res->h2_bucket[h];
            blk_base[b] = before + (c ? atomicAdd(&res->h2_fill[b], c) : 0);
        }
    }
    if (PHASE == 0) return;
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + H2_CLASSES) {
        const int c = (int)threadIdx.x - 64;
        int64_t tot = 0;
        for (int b = c * H2_WB; b < (c + 1) * H2_WB; ++b) tot += res->h2_bucket[b];
        if (tot > L.cap[c]) {  // the list cannot hold them: nothing of this class runs, the pass is redone elsewhere
            res->h2_status = 1;
            tot = 0;
        }
        res->h2_count[c] = (int)tot;
    }
    __syncthreads();
    if (bkt >= 0) {
        const int64_t first = (int64_t)blk_base[bkt] + my_off;
        if (first >= 0 && first + nunits <= L.cap[cls]) {
            for (int j = 0; j < nunits; ++j) L.units[cls][first + j] = make_int2(u, (nparts << 16) | j);
        }  // (else: reported through h2_count / h2_status above)
    }
}

__global__ void k_h2_clear(DevResult *res) {
    if (threadIdx.x < 8) res->misc[threadIdx.x] = 0;
    if (threadIdx.x < H2_NB) {
        res->h2_bucket[threadIdx.x] = 0;
        res->h2_fill[threadIdx.x] = 0;
    }
    if (threadIdx.x < H2_CLASSES) res->h2_count[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        res->h2_status = 0;
        res->flag_too_big = 0;
    }
}

// ---- join the two records of every edge and evaluate the closing expression ------------------------------------------
__global__ void __launch_bounds__(256) k_h2_final(View g, const uint4 *rec, double *curv) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= g.cap_total) return;
    const int u = g.slot_row[s];
    if (u < 0 || u >= g.n) return;
    const int2 ru = g.rowinfo[u];
    if (s < ru.x || (int)(s - ru.x) >= ru.y) return;
    const int v = g.col[s];
    if (v <= u || v >= g.n) return;  // the value lives at the slot whose neighbour id exceeds the row id
    const int dv = g.rowinfo[v].y;
    if ((ru.y < dv ? ru.y : dv) == 1) {  // bfc_naive.py:18-19
        curv[s] = 0.0;
        return;
    }
    const uint4 a = rec[s];  // from u's side: statistics over N(v) \ N(u) \ {u}
    const int64_t r = (int64_t)a.w;
    if (r < 0 || r >= g.cap_total || g.col[r] != u || g.slot_row[r] != v) {
        row_ok(g, make_int2(-1, (int)a.w), 39, u, v);
        return;
    }
    const uint4 b = rec[r];  // from v's side: statistics over N(u) \ N(v) \ {v}
    if (a.z != b.z) {        // both sides count the same triangles
        row_ok(g, make_int2(-1, (int)a.z), 40, u, v);
        return;
    }
    const int gam = (int)(a.y > b.y ? a.y : b.y);
    curv[s] = bfc_formula(ru.y, dv, (int)a.z, (int)b.x, (int)a.x, gam);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool h2_can_take(const dcr_graph *g, int curv_type, bool incremental) {
    return curv_type == DCR_CURV_BFC && !incremental && g->max_deg_bound <= H2_MAXDEG && g->cap_total < (int64_t)1 << 31;
}

static int ensure_h2(dcr_graph *g) {
    DCR_TRY(dev_regrow(&g->h2_weight, &g->h2_weight_cap, g->n + 64));
    DCR_TRY(dev_regrow(&g->h2_rec, &g->h2_rec_cap, g->cap_total + 64));
    const int64_t need[H2_CLASSES] = {g->n + 64, g->n + 64, g->n + g->cap_total / 4 + 64};
    for (int c = 0; c < H2_CLASSES; ++c) DCR_TRY(dev_regrow(&g->h2_units[c], &g->h2_units_cap[c], need[c]));
    return DCR_OK;
}

static unsigned h2_grid(const dcr_graph *g, int c, int64_t per_block) {
    // one unit per workgroup slot when the count of the previous pass is known (the graph changes by an edge or two per
    // SDRF iteration), else a grid-stride launch over a few rounds of workgroups; any grid is correct
    const int64_t units = g->h2_last_count[c] >= 0 ? (int64_t)g->h2_last_count[c] + g->h2_last_count[c] / 32 + 8
                                                   : (int64_t)g->num_cu * 32 * per_block;
    int64_t grid = (units + per_block - 1) / per_block;
    static const int64_t cap = getenv("DCR_H2_GRID") ? atoll(getenv("DCR_H2_GRID")) : 0;  // tuning aid: workgroups per CU
    if (cap > 0 && grid > cap * g->num_cu) grid = cap * g->num_cu;
    if (grid < 1) grid = 1;
    return (unsigned)grid;
}

int launch_curvature_pass_h2(dcr_graph *g) {
    DCR_TRY(ensure_h2(g));
    if (g->num_cu <= 0) {
        g->num_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0)
            g->num_cu = prop.multiProcessorCount;
    }
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, nullptr, (int32_t)g->n, 0, nullptr};
    H2Lists L;
    for (int c = 0; c < H2_CLASSES; ++c) {
        L.units[c] = g->h2_units[c];
        L.cap[c] = g->h2_units_cap[c];
    }
    int32_t *status = &g->dres->h2_status;
    hipLaunchKernelGGL(k_h2_clear, dim3(1), dim3(64), 0, g->stream, g->dres);
    DCR_HIP(hipMemsetAsync(g->h2_weight, 0, sizeof(int32_t) * (size_t)(g->n > 0 ? g->n : 1), g->stream));
    const int64_t sblocks = (g->cap_total + 255) / 256;
    if (sblocks > 0) hipLaunchKernelGGL(k_h2_weight, dim3((unsigned)sblocks), dim3(256), 0, g->stream, vw, g->h2_weight);
    const int64_t pblocks = (g->n + H2_PLAN_THREADS - 1) / H2_PLAN_THREADS;
    if (pblocks > 0) {
        hipLaunchKernelGGL(k_h2_plan<0>, dim3((unsigned)pblocks), dim3(H2_PLAN_THREADS), 0, g->stream, vw, g->h2_weight, L,
                           g->dres);
        hipLaunchKernelGGL(k_h2_plan<1>, dim3((unsigned)pblocks), dim3(H2_PLAN_THREADS), 0, g->stream, vw, g->h2_weight, L,
                           g->dres);
    }
    // records of split nodes are accumulated with atomics: start from zero
    DCR_HIP(hipMemsetAsync(g->h2_rec, 0, sizeof(uint4) * (size_t)(g->cap_total > 0 ? g->cap_total : 1), g->stream));
    static const bool serial = getenv("DCR_SERIAL_BINS") != nullptr;
    hipStream_t s1 = g->stream, s2 = g->stream;
    if (!serial) {
        DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
        for (int b = 0; b < 2; ++b) DCR_HIP(hipStreamWaitEvent(g->side[b], g->ev_fork, 0));
        s1 = g->side[0];
        s2 = g->side[1];
    }
    // heaviest class first on the main stream; the classes are independent
    hipLaunchKernelGGL((k_h2_block<h2_cap(2), h2_waves(2), h2_ecap(2), true>), dim3(h2_grid(g, 2, 1)),
                       dim3(64 * h2_waves(2)), 0, g->stream, vw, g->h2_units[2], &g->dres->h2_count[2], g->h2_units_cap[2],
                       g->h2_rec, status);
    hipLaunchKernelGGL((k_h2_block<h2_cap(1), h2_waves(1), h2_ecap(1), false>), dim3(h2_grid(g, 1, 1)),
                       dim3(64 * h2_waves(1)), 0, s1, vw, g->h2_units[1], &g->dres->h2_count[1], g->h2_units_cap[1],
                       g->h2_rec, status);
    hipLaunchKernelGGL((k_h2_wave<h2_cap(0), h2_ecap(0)>), dim3(h2_grid(g, 0, H2_WPB0)),
                       dim3(64 * H2_WPB0), 0, s2, vw, g->h2_units[0], &g->dres->h2_count[0], g->h2_units_cap[0], g->h2_rec,
                       status);
    if (!serial) {
        for (int b = 0; b < 2; ++b) {
            DCR_HIP(hipEventRecord(g->ev_join[b], g->side[b]));
            DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[b], 0));
        }
    }
    if (sblocks > 0) hipLaunchKernelGGL(k_h2_final, dim3((unsigned)sblocks), dim3(256), 0, g->stream, vw, g->h2_rec, g->curv);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

}  // namespace dcr
