// Two-hop curvature pass of libdcr_hip.so: an ALTERNATIVE implementation of a full Balanced Forman pass (DCR_PASS=h2).
// Same bits as the node-centric kernels, 9x fewer adjacency entries streamed, but slower on MI355X as built (3.0 ms against
// 1.98 ms on the bench graph, DESIGN.md §4.1b has the counters): the map of a node grows with its 2-HOP neighbourhood, so
// LDS admits 8 waves per CU where the node-centric kernels run 24, and every visit pays an LDS atomic.
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
// The map.  96 % of the 2-hop entries occur once, and what is rare per lane (a second occurrence, a collision) happens in
// nearly every 64-lane instruction, so the common case must be branch-free and the rest must not run at 1/64 occupancy:
//   * a direct-mapped FRONT table takes the first key that hashes to a slot with ONE compare-and-swap (phase A) and
//     answers "seen exactly once, not a neighbour" with ONE read (phase B);
//   * everything else — later occurrences, keys whose front slot is taken by another key, the flagged members of N(u) —
//     lives in an OVERFLOW table (4-slot buckets, 15-bit counters, as in dcr_bfc_nc.hip); M_u(w) = its counter, plus
//     one if w also holds its front slot;
//   * lanes that need the overflow table do not walk it on the spot: they queue (key, row) in LDS and the wave works the
//     queue off 64 items at a time.
// (First cut, bucket table only, every lane walking it in line: 720 M vector instructions per pass, as many as the
// node-centric kernels execute for nine times the entries.)
//
// Nodes are grouped by K = deg + 1 + Σ neighbour degrees (an upper bound of the keys of M_u): up to 1,280 keys a wave
// owns a node and private tables (2,048 front slots); up to 5,120 (10,240) a workgroup of 4 (8) waves shares tables of
// 8,192 (16,384) front slots; beyond that the KEYS of a node are split by a second hash into P partitions, each a unit
// of its own (every unit streams all rows but keeps only its share; results meet in the record through integer
// atomics), which also spreads a hub over P workgroups.  Units are laid out heaviest first.  Bounds: HBM / L2 row
// streaming (2 x 4 B x Σ d² per pass) and LDS atomics; no MFMA (integer set counting).
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

__host__ __device__ constexpr int h2_capd(int c) { return c == 0 ? 2048 : c == 1 ? 8192 : 16384; }  // front slots
__host__ __device__ constexpr int h2_cap(int c) { return h2_capd(c) / 2; }                            // overflow slots
__host__ __device__ constexpr int h2_maxkeys(int c) { return c == 0 ? 1280 : c == 1 ? 5120 : 10240; }
#ifndef H2_W1
#define H2_W1 4
#endif
#ifndef H2_W2
#define H2_W2 8
#endif
__host__ __device__ constexpr int h2_waves(int c) { return c == 0 ? 1 : c == 1 ? H2_W1 : H2_W2; }
__host__ __device__ constexpr int h2_ecap(int c) { return c == 0 ? 128 : 256; }
constexpr int H2_WPB0 = 2;  // waves (= nodes in flight) per workgroup of the wave class
constexpr unsigned H2_FLAG = 0x80000000u;  // on a front entry: the key also has an overflow entry (node ids stay below 2^30)
constexpr int H2_QCAP = 192;               // queued overflow items per wave (worked off when fewer than 64 slots are left)

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
template <int CAPD>
__device__ inline unsigned h2_dslot(unsigned key) {  // front slot: another 24-bit multiplier than the bucket hash
    constexpr int BITS = __builtin_ctz(CAPD);
    unsigned prod;
    asm("v_mul_u32_u24 %0, 0x85ebcb, %1" : "=v"(prod) : "v"(key));
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
    unsigned ekey[ECAP];  // step C: candidate ids and counts ...
    unsigned eval[ECAP];  // ... triangle partners and corrections
    unsigned qw[H2_QCAP];       // queued overflow work: key ...
    unsigned char qr[H2_QCAP];  // ... and row of the batch
};

// the three LDS arrays of a unit
struct H2Tab {
    unsigned *front;  // [CAPD] direct-mapped: key, | H2_FLAG once the key has an overflow entry
    unsigned *key;    // [CAP]  overflow keys, 4-slot buckets
    unsigned *cnt;    // [CAP / 2] their state, 16 bits each
};

// M_u(w) and whether w is a member of N(u); {0, false} for a key this unit does not hold
template <int CAPD, int CAP>
__device__ inline int h2_query(const H2Tab t, unsigned w, bool &nbr) {
    nbr = false;
    const unsigned e = t.front[h2_dslot<CAPD>(w)];
    if (e == w) return 1;
    if (e == H2_EMPTY) return 0;  // every key that was ever touched found its front slot taken, or took it
    const int s = h2_find<CAP>(t.key, w);
    if (s < 0) return 0;
    const unsigned c16 = h2_cnt_get(t.cnt, s);
    nbr = (c16 & 0x8000u) != 0u;
    return (int)(c16 & 0x7FFFu) + ((e & ~H2_FLAG) == w ? 1 : 0);
}

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

// ---- the edge set: every undirected edge as one 64-bit key in an open-addressing table in device memory -------------------
// Rebuilt by every pass (one CAS per edge).  Step C asks it "is w adjacent to t?" a few times per edge with triangles,
// where the first cut streamed the whole (hub) row of every triangle partner: 2.3 ms of a 4.7 ms pass on the bench graph.
constexpr unsigned long long H2_ESET_EMPTY = ~0ull;
__device__ inline unsigned long long h2_edge_key(int a, int b) {
    return a < b ? ((unsigned long long)(unsigned)a << 32) | (unsigned)b : ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
}
__device__ inline unsigned long long h2_eset_slot(unsigned long long key, int bits) {
    return (key * 0x9E3779B97F4A7C15ull) >> (64 - bits);
}
struct H2EdgeSet {
    unsigned long long *tab;
    int bits;
};
__device__ inline bool h2_eset_has(const H2EdgeSet es, int a, int b) {
    const unsigned long long key = h2_edge_key(a, b), mask = (1ull << es.bits) - 1ull;
    unsigned long long h = h2_eset_slot(key, es.bits);
    for (unsigned long long walk = 0; walk <= mask; ++walk) {
        const unsigned long long k = es.tab[h];
        if (k == key) return true;
        if (k == H2_ESET_EMPTY) return false;
        h = (h + 1) & mask;
    }
    return false;
}
__global__ void __launch_bounds__(256) k_h2_eset_build(View g, H2EdgeSet es, int32_t *status) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= g.cap_total) return;
    const int u = g.slot_row[s];
    if (u < 0 || u >= g.n) return;
    const int2 ru = g.rowinfo[u];
    if (s < ru.x || (int)(s - ru.x) >= ru.y) return;
    const int v = g.col[s];
    if (v <= u || v >= g.n) return;
    const unsigned long long key = h2_edge_key(u, v), mask = (1ull << es.bits) - 1ull;
    unsigned long long h = h2_eset_slot(key, es.bits);
    for (unsigned long long walk = 0; walk <= mask; ++walk) {
        const unsigned long long old = atomicCAS(&es.tab[h], H2_ESET_EMPTY, key);
        if (old == H2_ESET_EMPTY || old == key) return;
        h = (h + 1) & mask;
    }
    *status = 1;
}

// ---- step C: one edge {u,v} with triangles and positive counts, by one wave ------------------------------------------
// Returns {|{w : c(w) > 0}|, max c(w)} over the w of row v that live in this unit's table, c(w) = M_u(w) - 1 - |N(w) ∩ Tset|.
// One sweep of row v lists the candidates (w outside N(u) with M_u(w) >= 2) and the triangle partners (flagged w); every
// (candidate, partner) pair is then one probe of the edge set.  Lists longer than LCAP are taken LCAP at a time.
template <int CAPD, int CAP, int ECAP>
__device__ inline int2 h2_edge_with_triangles(const View &g, const H2EdgeSet es, int u, int2 rv, const H2Tab t,
                                              H2Scratch<ECAP> *sc, int32_t *status) {
    constexpr int LCAP = ECAP / 2;  // candidates: ekey[0..LCAP) ids, ekey[LCAP..) counts; partners: eval[0..LCAP); corrections: eval[LCAP..)
    const int lane = threadIdx.x & 63;
    const int32_t *rowv = g.col + rv.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    int pos = 0, mx = 0;
    int ncand_total = 0, npart_total = 0;
    for (int cbase = 0, first = 1; first || cbase < ncand_total; cbase += LCAP, first = 0) {
        for (int pbase = 0, pfirst = 1; pfirst || pbase < npart_total; pbase += LCAP, pfirst = 0) {
            // sweep row v: candidates number cbase.. and partners number pbase.. go to the lists
            int nc = 0, np = 0;  // running totals (uniform)
            for (int base = 0; base < rv.y; base += 64) {
                const int i = base + lane;
                const int w = i < rv.y ? rowv[i] : -1;
                bool isC = false, isT = false;
                int M = 0;
                if (w >= 0 && w != u) {
                    M = h2_query<CAPD, CAP>(t, (unsigned)w, isT);
                    isC = !isT && M >= 2;
                }
                const unsigned long long mC = __ballot(isC), mT = __ballot(isT);
                const int ic = nc + __popcll(mC & below) - cbase, it = np + __popcll(mT & below) - pbase;
                if (isC && ic >= 0 && ic < LCAP) {
                    sc->ekey[ic] = (unsigned)w;
                    sc->ekey[LCAP + ic] = (unsigned)(M - 1);
                    if (pbase == 0) sc->eval[LCAP + ic] = 0u;  // corrections accumulate over the partner rounds
                }
                if (isT && it >= 0 && it < LCAP) sc->eval[it] = (unsigned)w;
                nc += __popcll(mC);
                np += __popcll(mT);
            }
            ncand_total = nc;
            npart_total = np;
            h2_wave_sync();
            const int ncl = nc - cbase < LCAP ? nc - cbase : LCAP, npl = np - pbase < LCAP ? np - pbase : LCAP;
            const int pairs = (ncl > 0 && npl > 0) ? ncl * npl : 0;
            for (int p = lane; p < pairs; p += 64) {
                const int ci = p / npl, tj = p - ci * npl;
                if (h2_eset_has(es, (int)sc->ekey[ci], (int)sc->eval[tj])) atomicAdd(&sc->eval[LCAP + ci], 1u);
            }
            h2_wave_sync();
        }
        const int ncl = ncand_total - cbase < LCAP ? ncand_total - cbase : LCAP;
        for (int ci = lane; ci < ncl; ci += 64) {
            const int c = (int)sc->ekey[LCAP + ci] - (int)sc->eval[LCAP + ci];
            if (c > 0) {
                ++pos;
                mx = c > mx ? c : mx;
            }
        }
        h2_wave_sync();
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_xor(pos, off);
        const int o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    (void)status;
    return make_int2(pos, mx);
}

// ---- phases A and B: the rows of the neighbours of u, in batches of 64 rows per wave --------------------------------
// the queued overflow work of one wave (see the file comment): n is uniform and lives in a register
template <int CAPD, int CAP, int ECAP, int PHASE>
__device__ inline void h2_drain(const H2Tab t, H2Scratch<ECAP> *sc, int &n, int32_t *status) {
    const int lane = threadIdx.x & 63;
    h2_wave_sync();
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i >= n) continue;
        const unsigned w = sc->qw[i];
        const unsigned ds = h2_dslot<CAPD>(w);
        const unsigned e = t.front[ds];
        const bool in_front = (e & ~H2_FLAG) == w;
        if (PHASE == 0) {
            if (in_front && !(e & H2_FLAG)) atomicOr(&t.front[ds], H2_FLAG);  // a second occurrence of the slot's own key
            const int s = h2_insert<CAP>(t.key, w);
            if (s < 0) *status = 1;
            else h2_cnt_add(t.cnt, s);
        } else {
            const int s = h2_find<CAP>(t.key, w);
            if (s >= 0) {
                const int r = sc->qr[i];
                const unsigned c16 = h2_cnt_get(t.cnt, s);
                if (c16 & 0x8000u) {
                    atomicAdd(&sc->rowT[r], 1);
                } else {
                    const int c = (int)(c16 & 0x7FFFu) + (in_front ? 1 : 0) - 1;
                    if (c > 0) {
                        atomicAdd(&sc->rowPos[r], 1);
                        atomicMax(&sc->rowMx[r], c);
                    }
                }
            }
        }
    }
    h2_wave_sync();
    n = 0;
}

// Wave `wid` of NW takes the rows i = wid, wid + NW, ... of row u (strided: a hub's heaviest rows, adjacent at the
// front of its row, spread over the waves); lane l of the batch starting at `base` stands for row base + l * NW + wid.
template <int CAPD, int CAP, int NW, int ECAP, bool PARTS, int PHASE>
__device__ inline void h2_stream(const View &g, const H2EdgeSet es, int u, int2 ru, int part, int nparts, const H2Tab t,
                                 H2Scratch<ECAP> *sc, uint4 *rec, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const int wid = NW == 1 ? 0 : (int)(threadIdx.x >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    int qn = 0;  // queued overflow items (uniform)
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
            const int tt = __shfl_up(incl, off);
            if (lane >= off) incl += tt;
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
                if (j0 + 64 * q >= P) continue;  // uniform
                unsigned vm = 0u;
                if (rr[q] >= 0) {
                    const int2 d = sc->desc[rr[q]];
                    vm = h2_piece_mask(aa[q], d.x, d.x + d.y);
                }
                const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
                bool slow[4];
                // the common case, branch-free: one compare-and-swap (A) or one read (B) on the front table per entry
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const bool in = (vm >> jj) & 1u;
                    bool take = in && kk[jj] != (unsigned)u && kk[jj] < (unsigned)g.n;
                    if (PHASE == 0 && PARTS) take = take && h2_part(kk[jj], nparts) == part;
                    if (PHASE == 1 && in && kk[jj] == (unsigned)u) sc->rowRev[rr[q]] = aa[q] + jj;
                    slow[jj] = false;
                    if (take) {
                        unsigned *fs = &t.front[h2_dslot<CAPD>(kk[jj])];
                        if (PHASE == 0) {
                            slow[jj] = atomicCAS(fs, H2_EMPTY, kk[jj]) != H2_EMPTY;
                        } else {
                            const unsigned e = *fs;
                            slow[jj] = e != kk[jj] && e != H2_EMPTY;
                        }
                    }
                }
                // whatever needs the overflow table is queued
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned long long m = __ballot(slow[jj]);
                    if (m == 0) continue;  // uniform
                    if (qn > H2_QCAP - 64) h2_drain<CAPD, CAP, ECAP, PHASE>(t, sc, qn, status);
                    if (slow[jj]) {
                        const int idx = qn + __popcll(m & below);
                        sc->qw[idx] = kk[jj];
                        if (PHASE == 1) sc->qr[idx] = (unsigned char)rr[q];
                    }
                    qn += __popcll(m);
                }
            }
        }
        if (PHASE == 1) h2_drain<CAPD, CAP, ECAP, PHASE>(t, sc, qn, status);  // the row totals are read next
        h2_wave_sync();
        if (PHASE == 1) {
            int T = sc->rowT[lane], pos = sc->rowPos[lane], mx = sc->rowMx[lane];
            const int rev = sc->rowRev[lane];
            // triangles AND positive counts: the counts of this edge still include the triangle partners (step C)
            unsigned long long todo = __ballot(k >= 0 && T > 0 && pos > 0);
#ifdef H2_NO_STEPC  // timing-only build (results wrong)
            todo = 0;
#endif
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int2 rv = make_int2(__shfl(rk.x, l), __shfl(rk.y, l));
                const int2 pm = h2_edge_with_triangles<CAPD, CAP, ECAP>(g, es, u, rv, t, sc, status);
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
    if (qn > 0) h2_drain<CAPD, CAP, ECAP, PHASE>(t, sc, qn, status);  // (after the loop: a wave's last batches may be empty)
}

// one unit: node u, key partition `part` of `nparts`, by NW waves sharing the tables `t`
template <int CAPD, int CAP, int NW, int ECAP, bool PARTS>
__device__ inline void h2_node(const View &g, const H2EdgeSet es, int u, int2 ru, int part, int nparts, const H2Tab t,
                               H2Scratch<ECAP> *sc, uint4 *rec, int32_t *status) {
    const int tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    constexpr int NT = 64 * NW;
    uint4 *f4 = reinterpret_cast<uint4 *>(t.front);
    uint4 *k4 = reinterpret_cast<uint4 *>(t.key);
    uint4 *c4 = reinterpret_cast<uint4 *>(t.cnt);
    for (int i = tid; i < CAPD / 4; i += NT) f4[i] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
    for (int i = tid; i < CAP / 4; i += NT) k4[i] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
    for (int i = tid; i < CAP / 8; i += NT) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    h2_sync<NW>();
    for (int i = tid; i < ru.y; i += NT) {  // the members of N(u): flagged overflow entries (in every partition's tables)
        const int k = g.col[ru.x + i];
        if (k >= 0 && k < g.n && k != u) {
            const int s = h2_insert<CAP>(t.key, (unsigned)k);
            if (s < 0) *status = 1;
            else h2_cnt_flag(t.cnt, s);
            atomicCAS(&t.front[h2_dslot<CAPD>((unsigned)k)], H2_EMPTY, (unsigned)k | H2_FLAG);  // (taken: k stays overflow-only)
        }
    }
    h2_sync<NW>();
#ifndef H2_NO_PHASEA
    h2_stream<CAPD, CAP, NW, ECAP, PARTS, 0>(g, es, u, ru, part, nparts, t, sc, rec, status);
#endif
    h2_sync<NW>();
#ifndef H2_NO_PHASEB
    h2_stream<CAPD, CAP, NW, ECAP, PARTS, 1>(g, es, u, ru, part, nparts, t, sc, rec, status);
#endif
    h2_sync<NW>();  // the tables are rewritten by the next unit
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
// wave class: a wave owns a unit and its private table; units are taken grid-stride (heaviest first in the list)
template <int CAPD, int CAP, int ECAP>
__global__ void __launch_bounds__(64 * H2_WPB0) k_h2_wave(View g, H2EdgeSet es, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                          uint4 *rec, int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned front_all[H2_WPB0][CAPD];
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
        h2_node<CAPD, CAP, 1, ECAP, false>(g, es, u, ru, 0, 1, H2Tab{front_all[wid], key_all[wid], cnt_all[wid]}, &sc_all[wid],
                                           rec, status);
    }
}

// block classes: a workgroup of W waves shares one table per unit
template <int CAPD, int CAP, int W, int ECAP, bool PARTS>
__global__ void __launch_bounds__(64 * W) k_h2_block(View g, H2EdgeSet es, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                      uint4 *rec, int32_t *status) {
    __shared__ __attribute__((aligned(16))) unsigned front[CAPD];
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
        h2_node<CAPD, CAP, W, ECAP, PARTS>(g, es, u, ru, part, nparts, H2Tab{front, key, cnt}, &sc_all[wid], rec, status);
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
#ifdef H2_NO_PHASEB  // timing-only build: there are no records to join
    return;
#endif
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
    return g->pass_impl == 3 && curv_type == DCR_CURV_BFC && !incremental && g->max_deg_bound <= H2_MAXDEG && g->cap_total < (int64_t)1 << 31;
}

static int ensure_h2(dcr_graph *g) {
    DCR_TRY(dev_regrow(&g->h2_weight, &g->h2_weight_cap, g->n + 64));
    DCR_TRY(dev_regrow(&g->h2_rec, &g->h2_rec_cap, g->cap_total + 64));
    // edge set: at most cap_total / 2 undirected edges, load <= 1/4
    int bits = 10;
    while ((1ll << bits) < 2 * g->cap_total) ++bits;
    if (g->h2_eset_bits != bits || !g->h2_eset) {
        if (g->h2_eset) (void)hipFree(g->h2_eset);
        g->h2_eset = nullptr;
        DCR_TRY(dev_alloc(&g->h2_eset, (int64_t)1 << bits));
        g->h2_eset_bits = bits;
    }
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
    const H2EdgeSet es{g->h2_eset, g->h2_eset_bits};
    DCR_HIP(hipMemsetAsync(g->h2_eset, 0xFF, sizeof(unsigned long long) << g->h2_eset_bits, g->stream));
    if (sblocks > 0) hipLaunchKernelGGL(k_h2_eset_build, dim3((unsigned)sblocks), dim3(256), 0, g->stream, vw, es, status);
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
    hipLaunchKernelGGL((k_h2_block<h2_capd(2), h2_cap(2), h2_waves(2), h2_ecap(2), true>), dim3(h2_grid(g, 2, 1)),
                       dim3(64 * h2_waves(2)), 0, g->stream, vw, es, g->h2_units[2], &g->dres->h2_count[2], g->h2_units_cap[2],
                       g->h2_rec, status);
    hipLaunchKernelGGL((k_h2_block<h2_capd(1), h2_cap(1), h2_waves(1), h2_ecap(1), false>), dim3(h2_grid(g, 1, 1)),
                       dim3(64 * h2_waves(1)), 0, s1, vw, es, g->h2_units[1], &g->dres->h2_count[1], g->h2_units_cap[1],
                       g->h2_rec, status);
    hipLaunchKernelGGL((k_h2_wave<h2_capd(0), h2_cap(0), h2_ecap(0)>), dim3(h2_grid(g, 0, H2_WPB0)),
                       dim3(64 * H2_WPB0), 0, s2, vw, es, g->h2_units[0], &g->dres->h2_count[0], g->h2_units_cap[0], g->h2_rec,
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
