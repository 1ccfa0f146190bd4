// Node-centric curvature pass of libdcr_hip.so (the default implementation of dcr_curvature_pass).
//
// Replaces compute_curvature_graph(G, curv_type) at rewiring/sdrf_no_cuda.py:24, i.e. E calls of
// bfc_naive.bfc_edge (curvature/bfc_naive.py:7-40) or compute_curvature_edge (classical_curvatures.py:14-28).
//
// Every undirected edge {u,v} is owned by its higher-degree endpoint u (ties: smaller id).  The neighbour set N(u) is
// staged ONCE as a hash set of 4-slot buckets in LDS and reused for all edges u owns; per edge one wave (or a quarter /
// half of one: up to four edges of the same owner are processed side by side) then only has to
//   1. look the members of N(v) up in that set: hits are the triangles T (bfc_naive.py:25) and get a per-edge flag,
//      misses are DY = N(v) \ N(u) \ {u};
//   2. stream the rows of the members of DY from HBM.  Their aligned 16-byte pieces form one flat list that the 64
//      lanes share evenly whatever the row lengths (four loads in flight per lane); every entry is probed against the
//      set: an unflagged hit z is a 4-cycle u-z-w-v.  The hits of one row are |N(w) ∩ DX| (an LDS counter per row), and
//      every hit bumps a 15-bit counter on z's table slot, which after the sweep is |N(z) ∩ DY|.  sq1, sq2 and gamma
//      (bfc_naive.py:26-29,36-37) are degree statistics of that one bipartite graph between DX = N(u) \ N(v) \ {v}
//      and DY, so nothing else has to be read;
//   3. park the integers (T, |sq1|, |sq2|, gamma) per position of the unit; after its (up to 16) edges the lanes that
//      stand for them evaluate the float64 closing expression together (bfc_naive.py:31-40, reference operation order).
// Compared with the edge-centric kernels in dcr_bfc.hip there is no per-edge table build, no sizing of both sides,
// no descriptor list of the unstreamed side, no final table scan and, for hubs, no workgroup barrier per edge.
//
// Work units are 16 (4 for hubs) positions of a row, strided over the row; nodes are grouped by degree: up to 254
// neighbours a wave owns a unit and its private table ("wave classes"), above that a workgroup builds one table and each
// of its waves takes a unit ("block classes").  A two-phase plan lays the units out heaviest first; persistent waves
// pull them from eight cursors.  DESIGN.md §4.1 has the measurements behind each of these choices.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dcr_bfc_common.h"

namespace dcr {

__host__ __device__ constexpr int nc_slots(int c) { return c == 0 ? 256 : c == 1 ? 512 : c == 2 ? 2048 : c == 3 ? 8192 : 16384; }
__host__ __device__ constexpr int nc_maxdeg(int c) {
    return c == 0 ? 62 : c == 1 ? 254 : c == 2 ? 1022 : c == 3 ? 4094 : NC_MAXD;
}
// waves per workgroup; the last class (64 KiB table + 32 KiB of slot state per wave) fits two
__host__ __device__ constexpr int nc_waves(int c) { return c == 2 ? 8 : c == 4 ? 2 : 4; }

__device__ inline int nc_class_of(int d) {
    return d <= nc_maxdeg(0) ? 0 : d <= nc_maxdeg(1) ? 1 : d <= nc_maxdeg(2) ? 2 : d <= nc_maxdeg(3) ? 3
           : d <= nc_maxdeg(4) ? 4 : -1;
}

// compiler-level ordering of one wave's LDS traffic (the hardware executes a wave's DS instructions in order)
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the higher-degree endpoint owns an edge (its table is the big one, the rows streamed are those of the
// lower-degree endpoint's neighbours); degree-1 edges are finished by the smaller id without any lookup
__device__ inline bool nc_owns(int a, int da, int b, int db, bool trivial_rule) {
    if (!nc_takes(da, db)) return false;  // left to the edge-centric kernels
    const bool a_can = da <= NC_MAXD, b_can = db <= NC_MAXD;  // only nodes within the table sizes get units
    if (!a_can) return false;
    if (!b_can) return true;
    if (trivial_rule && (da < db ? da : db) == 1) return a < b;
    return da > db || (da == db && a < b);
}

// The neighbour table is an open-addressing hash set of 4-slot buckets (16 bytes): a key lives in the first free
// slot of its home bucket, or of the next bucket when that one is full.  One 16-byte LDS read settles almost every
// lookup (hit, or a free slot in the bucket = miss); with single-slot linear probing some lane of every
// wave-instruction needed a second and a third probe, and that loop alone was 15 % of the pass.
// A free slot.  Not 0xFFFFFFFF: that is what the slack behind a row holds, and the fast test below compares raw row
// pieces (slack included) with the bucket contents.
#ifndef NC_Q
#define NC_Q 3  // pieces per lane and round of the streaming loops (S100k / S1M pass ms: 1: 2.05 / 23.4, 2: 1.98 / 22.0, 3: 1.97 / 21.8, 4: 2.04 / 22.4)
#endif
constexpr unsigned NC_EMPTY = 0xFFFFFFFDu;

template <int SLOTS>
__device__ inline unsigned hash_bucket(unsigned key) {
    constexpr int BITS = __builtin_ctz(SLOTS / 4);
#ifdef DCR_HASH_MUL32
    return (key * 0x9E3779B1u) >> (32 - BITS);
#else
    // Fibonacci hashing in a 24-bit word: v_mul_u32_u24 is a full-rate instruction, the 32-bit v_mul_lo_u32 is
    // quarter rate and this runs once per streamed neighbour id.  Only the low 24 bits of the id are hashed (the
    // bucket stores the whole id, so ids that differ above bit 23 merely share a home bucket).
    // (inline assembly: the compiler drops the 24-bit mask as not demanded and then picks the 32-bit multiply)
    unsigned prod;
    asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(prod) : "v"(key));
    return (prod >> (24 - BITS)) & ((1u << BITS) - 1);
#endif
}

// position of key inside the bucket (0..3) or -1; go_on: not found and the bucket is full (its slots fill in order,
// so the last one taken means all taken): the key may have spilled into the next bucket
__device__ inline int bucket_match(const uint4 e, unsigned key, bool &go_on) {
    const int pos = e.x == key ? 0 : e.y == key ? 1 : e.z == key ? 2 : e.w == key ? 3 : -1;
    go_on = pos < 0 && e.w != NC_EMPTY;
    return pos;
}

template <int SLOTS>
__device__ inline void nc_insert(unsigned *tab, unsigned key, int *spilled) {
    unsigned b = hash_bucket<SLOTS>(key);
    while (true) {
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const unsigned old = atomicCAS(&tab[b * 4 + pos], NC_EMPTY, key);
            if (old == NC_EMPTY || old == key) return;
        }
        *spilled = 1;  // some key of this table lives outside its home bucket
        b = (b + 1) & (SLOTS / 4 - 1);
    }
}

// slot of key, or -1 (the table is at most half full, so the walk always meets a bucket with a free slot)
template <int SLOTS>
__device__ inline int nc_find(const unsigned *tab, unsigned key) {
    unsigned b = hash_bucket<SLOTS>(key);
    while (true) {
        bool go_on;
        const int pos = bucket_match(*reinterpret_cast<const uint4 *>(tab + b * 4), key, go_on);
        if (pos >= 0) return (int)(b * 4) + pos;
        if (!go_on) return -1;
        b = (b + 1) & (SLOTS / 4 - 1);
    }
}

// per-edge slot state, two 16-bit halves per word: bit 15 = "member of N(v) or v itself", bits 0-14 = hits
__device__ inline void cnt_flag(unsigned *cnt, int h) { atomicOr(&cnt[h >> 1], 0x8000u << ((h & 1) * 16)); }
__device__ inline unsigned cnt_add(unsigned *cnt, int h) {
    const unsigned sh = (unsigned)(h & 1) * 16u;
    return (atomicAdd(&cnt[h >> 1], 1u << sh) >> sh) & 0xFFFFu;
}

// One aligned piece of a streamed row: which of its four entries (those selected by `vmask`) may be in the table.
//
// Almost every streamed id is NOT in the table, so the test is kept to the bare minimum: hash, one 16-byte LDS read, four
// equality compares and a "bucket full" compare per id (nc_probe_flags).  Pieces are compared raw; ids outside the row,
// slack included, are masked off afterwards.  What passes the test is queued for the full look-up (slot position, walk
// to the next bucket, slot counter update), see NC_QCAP below.
#ifdef NC_DIRTY_STATS  // diagnostic build (tools/build_variant.sh dstats -DNC_DIRTY_STATS): what an incremental pass's units hold
__device__ unsigned long long nc_dstats[4 * 8];  // per class of table size: units, units with an edge to compute, such edges, their rows' entries
#endif
#ifdef NC_STATS  // diagnostic build (tools/build_variant.sh stats -DNC_STATS): how often the fast test fails
__device__ unsigned long long nc_stats[8];
#define NC_STATS_COUNT(look_any)                                                             \
    {                                                                                        \
        const unsigned long long act = __ballot(true), lk = __ballot(look_any);              \
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {                        \
            atomicAdd(&nc_stats[0], 1ull);                                                   \
            atomicAdd(&nc_stats[1], lk ? 1ull : 0ull);                                       \
            atomicAdd(&nc_stats[2], (unsigned long long)__popcll(act));                      \
            atomicAdd(&nc_stats[3], (unsigned long long)__popcll(lk));                       \
        }                                                                                    \
    }
#else
#define NC_STATS_COUNT(look_any)
#endif

// ovf (uniform): the table has keys outside their home buckets, so a full home bucket without a match is not yet a miss
// vskip: the other endpoint v of the edge.  It is in the table (a neighbour of u) and in EVERY streamed row (their nodes
// are neighbours of v), and it is never counted (flagged): without this test each row would take the slow path once.
__device__ inline bool bucket_look(const uint4 e, unsigned key, bool ovf, unsigned vskip) {
    bool look = (e.x == key) | (e.y == key) | (e.z == key) | (e.w == key);
    if (ovf) look |= e.w != NC_EMPTY;
    return look & (key != vskip);
}

// which of the four entries of a piece (those selected by vmask) need the full look-up: bit j set = entry j
template <int SLOTS>
__device__ inline unsigned nc_probe_flags(const unsigned *tab, const int4 w, unsigned vmask, bool ovf, unsigned vskip) {
    const unsigned k[4] = {(unsigned)w.x, (unsigned)w.y, (unsigned)w.z, (unsigned)w.w};
    const uint4 *tb = reinterpret_cast<const uint4 *>(tab);
    uint4 e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = tb[hash_bucket<SLOTS>(k[j])];
    unsigned f = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) f |= (bucket_look(e[j], k[j], ovf, vskip) ? 1u : 0u) << j;
    NC_STATS_COUNT((f & vmask) != 0u)
    return f & vmask;
}

// which of the four entries of the piece at slot a lie inside the row [lo, hi)
__device__ inline unsigned piece_mask(int a, int lo, int hi) {
    const int s = lo - a, t = hi - a;  // t >= 1 for every piece of the row
    const unsigned head = s > 0 ? (0xFu << s) & 0xFu : 0xFu;
    const unsigned tail = t < 4 ? (1u << t) - 1u : 0xFu;
    return head & tail;
}

// Row of piece j in the flat piece list.  Lane r still holds poff[r] of "its" row in a register, so the rows of the
// first and the last piece of a 64-piece step come from two ballots.  A few row starts in between: a lane counts how
// many are at or below its own j; many: bisection over that range only (poff mirrors the prefix sums in LDS).
__device__ inline int piece_row(const int *poff, int poff_lane, int j, int j_first, int j_last) {
    const int rf = __popcll(__ballot(poff_lane <= j_first)) - 1;
    const int rl = __popcll(__ballot(poff_lane <= j_last)) - 1;
    int r = rf;
    const int span = rl - rf;  // uniform
    if (span <= 4) {
        for (int b = rf + 1; b <= rl; ++b) r += (__shfl(poff_lane, b) <= j);
    } else {
        for (int step = span >= 32 ? 32 : span >= 16 ? 16 : span >= 8 ? 8 : 4; step > 0; step >>= 1)
            if (r + step <= rl && poff[r + step] <= j) r += step;
    }
    return r;
}

struct NcEdge {
    int T, s1, s2, gam, posu;
};

// Full look-ups are not done in line.  A streamed id that may be in the table is rare per lane (1.2 % on the bench graph)
// but 42 % of the wave-level probes have one in SOME lane, and the look-up (walk, validity, counter update) then ran with
// ~7 of 64 lanes active: a quarter of the pass's vector instructions.  The lanes queue (id, row) in LDS instead and the
// wave works the queue off 64 items at a time.
#ifndef NC_QCAP
#define NC_QCAP 128
#endif

// per-wave scratch in LDS
struct NcScratch {
    int2 desc[64];   // {start, length} of the DY rows of the current batch
    int poff[66];    // exclusive prefix of their piece counts; poff[64] = total
    int rowcnt[64];  // hits per row
    int acc[2];      // wave totals: |sq| on the table side, gamma
    int acc4[4][2];  // the same per edge of a batch
    int res[4][5];   // batch results per edge: T, |sq| table side, |sq| row side, gamma, position of u in row v
    int fin[32][5];  // the same per position of the unit, read back when the closing expressions are evaluated
    int spilled;     // set while the table is built: some key lives outside its home bucket
    unsigned qk[NC_QCAP];       // queued full look-ups: streamed id ...
    unsigned char qrow[NC_QCAP];  // ... and its row of the batch
};

// push the flagged entries of one piece (ALL lanes call this; lanes without a piece pass f = 0).  qn is uniform; `drain`
// works the queue off and resets qn (called when fewer than 64 slots are left: one entry position adds at most 64 items)
template <typename Drain>
__device__ inline void nc_queue_push(NcScratch *sc, int &qn, unsigned f, const int4 w, int row, Drain drain) {
    const unsigned k[4] = {(unsigned)w.x, (unsigned)w.y, (unsigned)w.z, (unsigned)w.w};
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool p = (f >> j) & 1u;
        const unsigned long long m = __ballot(p);
        if (m == 0) continue;  // uniform
        if (qn > NC_QCAP - 64) drain();
        if (p) {
            const int idx = qn + __popcll(m & below);
            sc->qk[idx] = k[j];
            sc->qrow[idx] = (unsigned char)row;
        }
        qn += __popcll(m);
    }
}

// One edge {u,v} owned by u, by one wave.  `tab` holds N(u); cnt and sc are this wave's scratch.
template <int SLOTS, int MODE>
__device__ inline NcEdge nc_edge(const View &g, int u, int v, int2 rv, const unsigned *tab, unsigned *cnt,
                                 NcScratch *sc, bool ovf) {
    const int lane = threadIdx.x & 63;
    const int32_t *rowv = g.col + rv.x;
    NcEdge out;
    out.T = out.s1 = out.s2 = out.gam = 0;
    out.posu = -1;
    if (MODE == MODE_BFC) {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = lane; i < SLOTS / 8; i += 64) c4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) {
            sc->acc[0] = 0;
            sc->acc[1] = 0;
        }
        wave_sync();
        const int hv = nc_find<SLOTS>(tab, (unsigned)v);  // v is a neighbour of u: never counted as a hit
        if (lane == 0 && hv >= 0) cnt_flag(cnt, hv);
    }
    // sweep 1 over N(v): triangles (flagged), where u sits in row v
    int T = 0, posu = -1;
    int k_first = -1;         // first 64 members of N(v) and whether they are in N(u): reused by sweep 2
    bool in_first = false;
    for (int base = 0; base < rv.y; base += 64) {
        const int i = base + lane;
        const int k = i < rv.y ? rowv[i] : -1;
        const bool isu = k == u;
        const unsigned long long mu = __ballot(isu);
        if (mu) posu = base + __ffsll((long long)mu) - 1;
        const int h = (k >= 0 && !isu) ? nc_find<SLOTS>(tab, (unsigned)k) : -1;
        if (MODE == MODE_BFC && h >= 0) cnt_flag(cnt, h);
        T += __popcll(__ballot(h >= 0));
        if (base == 0) {
            k_first = k;
            in_first = h >= 0;
        }
    }
    out.T = T;
    out.posu = posu;
    if (MODE != MODE_BFC) return out;
    wave_sync();
    // sweep 2 over N(v): the rows of DY, 64 members at a time.  Their aligned 16-byte pieces form one flat list that
    // the lanes share evenly, whatever the row lengths: lane t takes pieces t, t + 64, ... and keeps four loads in
    // flight; the row of a piece is found by bisection of the prefix sums of the rows' piece counts.
    int s1 = 0, gam = 0, s2 = 0;
    int qn = 0;  // queued full look-ups (uniform)
    auto drain = [&]() {
        wave_sync();
        for (int qb = 0; qb < qn; qb += 64) {
            const int qi = qb + lane;
            if (qi < qn) {
                const int h = nc_find<SLOTS>(tab, sc->qk[qi]);
                if (h >= 0) {
                    const unsigned old = cnt_add(cnt, h);
                    if (!(old & 0x8000u)) {  // not a member of N(v): a 4-cycle u-z-w-v
                        atomicAdd(&sc->rowcnt[sc->qrow[qi]], 1);
                        s1 += (old == 0u);
                        gam = (int)old + 1 > gam ? (int)old + 1 : gam;
                    }
                }
            }
        }
        wave_sync();
        qn = 0;
    };
    for (int base = 0; base < rv.y; base += 64) {
        const int i = base + lane;
        int k = k_first;
        bool in_nu = in_first;
        if (base > 0) {
            k = i < rv.y ? rowv[i] : -1;
            in_nu = k >= 0 && nc_find<SLOTS>(tab, (unsigned)k) >= 0;
        }
        const bool member = k >= 0 && k < g.n && k != u && !in_nu;
        int2 rk = make_int2(0, 0);
        if (member) {
            rk = g.rowinfo[k];
            if (!row_ok(g, rk, 11, k, v)) rk = make_int2(0, 0);
        }
        const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
        int incl = np;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int P = __shfl(incl, 63);
        sc->desc[lane] = rk;
        sc->poff[lane] = incl - np;
        const int poff_lane = incl - np;
        sc->rowcnt[lane] = 0;
        if (lane == 0) sc->poff[64] = P;
        wave_sync();
        for (int j0 = 0; j0 < P; j0 += 64 * NC_Q) {
            int4 w[NC_Q];
            int rr[NC_Q], aa[NC_Q];
#pragma unroll
            for (int q = 0; q < NC_Q; ++q) {
                const int j = j0 + 64 * q + lane;
                rr[q] = -1;
                aa[q] = 0;
                w[q] = make_int4(0, 0, 0, 0);
                const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
                if (jf >= P) continue;  // uniform
                const int r = piece_row(sc->poff, poff_lane, j < P ? j : jl, jf, jl);
                if (j < P) {
                    const int2 d = sc->desc[r];
                    const int a = (d.x & ~3) + 4 * (j - sc->poff[r]);
                    w[q] = load_piece(g.col, a);
                    rr[q] = r;
                    aa[q] = a;
                }
            }
#pragma unroll
            for (int q = 0; q < NC_Q; ++q) {
                if (j0 + 64 * q >= P) continue;  // uniform
                unsigned f = 0u;
                if (rr[q] >= 0) {
                    const int2 d = sc->desc[rr[q]];
                    f = nc_probe_flags<SLOTS>(tab, w[q], piece_mask(aa[q], d.x, d.x + d.y), ovf, (unsigned)v);
                }
                if (__ballot(f != 0u)) nc_queue_push(sc, qn, f, w[q], rr[q], drain);  // uniform; rare per lane, common per wave
                __builtin_amdgcn_sched_barrier(0);  // one probe sequence live at a time: keeps the wave at 8 per SIMD
            }
        }
        if (qn > 0) drain();
        wave_sync();
        const int c = sc->rowcnt[lane];
        s2 += __popcll(__ballot(c > 0));
        gam = c > gam ? c : gam;
        wave_sync();  // the scratch is rewritten by the next batch
    }
    // lane partials -> wave totals through two LDS words (hits are rare: few lanes have anything to add)
    if (s1 > 0) atomicAdd(&sc->acc[0], s1);
    if (gam > 0) atomicMax(&sc->acc[1], gam);
    wave_sync();
    out.s1 = sc->acc[0];
    out.gam = sc->acc[1];
    out.s2 = s2;
    wave_sync();
    return out;
}

// One sub-unit of row u: nc_lanes positions, strided by the number of sub-units of the row (p = sub + l * nsub), so
// that the expensive edges of a hub (those to other hubs sit next to each other at the front of its row) spread over
// all of its sub-units.  Lane l stands for the edge to the neighbour at its position.
// edges processed side by side by one wave (their neighbour rows share the 64 lanes): 4 for the smallest class,
// 2 for the next, 1 (no batching) for the block classes
__host__ __device__ constexpr int nc_batch_for_slots(int slots) { return slots <= 256 ? 4 : slots <= 512 ? 2 : 1; }

// B edges {u, v_b} owned by u, by one wave: lane group b (64 / B lanes) sweeps N(v_b); the rows of all the DY sets form
// one flat piece list.  `v`, `rv` are those of the lane's group (rv.y == 0: the group has no edge; every rv.y <= 64 / B).
// Results go to sc->res[b].
template <int SLOTS, int B>
__device__ inline void nc_edge_batch(const View &g, int u, int v, int2 rv, const unsigned *tab, unsigned *cnt,
                                     NcScratch *sc, bool ovf) {
    constexpr int G = 64 / B;
    const int lane = threadIdx.x & 63, grp = lane / G, gl = lane % G;
    const unsigned long long gmask = ((1ull << G) - 1ull) << (grp * G);
    unsigned *cntg = cnt + grp * (SLOTS / 2);
    uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
    for (int i = lane; i < B * SLOTS / 8; i += 64) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < B) {
        sc->acc4[lane][0] = 0;
        sc->acc4[lane][1] = 0;
    }
    wave_sync();
    const bool act = rv.y > 0;
    if (act && gl == 0) {
        const int hv = nc_find<SLOTS>(tab, (unsigned)v);  // v is a neighbour of u: never counted as a hit
        if (hv >= 0) cnt_flag(cntg, hv);
    }
    // sweep over N(v_b): triangles (flagged), where u sits in row v_b, the members of DY
    const int k = (act && gl < rv.y) ? g.col[rv.x + gl] : -1;
    const bool isu = k == u;
    const unsigned long long mu = __ballot(isu) & gmask;
    const int posu = mu ? __ffsll((long long)mu) - 1 - grp * G : -1;
    const int h = (k >= 0 && !isu) ? nc_find<SLOTS>(tab, (unsigned)k) : -1;
    if (h >= 0) cnt_flag(cntg, h);
    const int T = __popcll(__ballot(h >= 0) & gmask);
    const bool member = k >= 0 && k < g.n && !isu && h < 0;
    int2 rk = make_int2(0, 0);
    if (member) {
        rk = g.rowinfo[k];
        if (!row_ok(g, rk, 11, k, v)) rk = make_int2(0, 0);
    }
    const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
    int incl = np;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const int P = __shfl(incl, 63);
    sc->desc[lane] = rk;
    sc->poff[lane] = incl - np;
    const int poff_lane = incl - np;
    sc->rowcnt[lane] = 0;
    if (lane == 0) sc->poff[64] = P;
    wave_sync();
    int qn = 0;  // queued full look-ups (uniform); the row of an item tells its edge: group = row / G
    auto drain = [&]() {
        wave_sync();
        for (int qb = 0; qb < qn; qb += 64) {
            const int qi = qb + lane;
            if (qi < qn) {
                const int h = nc_find<SLOTS>(tab, sc->qk[qi]);
                if (h >= 0) {
                    const int row = sc->qrow[qi], qg = row / G;
                    const unsigned old = cnt_add(cnt + qg * (SLOTS / 2), h);
                    if (!(old & 0x8000u)) {
                        atomicAdd(&sc->rowcnt[row], 1);
                        if (old == 0u) atomicAdd(&sc->acc4[qg][0], 1);   // a new member of sq on the table side
                        else atomicMax(&sc->acc4[qg][1], (int)old + 1);  // (a first hit's 1 is covered by the row-side maximum)
                    }
                }
            }
        }
        wave_sync();
        qn = 0;
    };
    for (int j0 = 0; j0 < P; j0 += 64 * NC_Q) {
        int4 w[NC_Q];
        int rr[NC_Q], aa[NC_Q];
#pragma unroll
        for (int q = 0; q < NC_Q; ++q) {
            const int j = j0 + 64 * q + lane;
            rr[q] = -1;
            aa[q] = 0;
            w[q] = make_int4(0, 0, 0, 0);
            const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
            if (jf >= P) continue;  // uniform
            const int r = piece_row(sc->poff, poff_lane, j < P ? j : jl, jf, jl);
            if (j < P) {
                const int2 d = sc->desc[r];
                const int a = (d.x & ~3) + 4 * (j - sc->poff[r]);
                w[q] = load_piece(g.col, a);
                rr[q] = r;
                aa[q] = a;
            }
        }
#pragma unroll
        for (int q = 0; q < NC_Q; ++q) {
            if (j0 + 64 * q >= P) continue;  // uniform
            const int gr = (rr[q] < 0 ? 0 : rr[q]) / G;
            const unsigned vgr = (unsigned)__shfl(v, gr * G);  // the edge this row belongs to (all lanes converged here)
            unsigned f = 0u;
            if (rr[q] >= 0) {
                const int2 d = sc->desc[rr[q]];
                f = nc_probe_flags<SLOTS>(tab, w[q], piece_mask(aa[q], d.x, d.x + d.y), ovf, vgr);
            }
            if (__ballot(f != 0u)) nc_queue_push(sc, qn, f, w[q], rr[q], drain);  // uniform
            __builtin_amdgcn_sched_barrier(0);  // one probe sequence live at a time
        }
    }
    if (qn > 0) drain();
    wave_sync();
    const int c = sc->rowcnt[lane];
    const int s2 = __popcll(__ballot(c > 0) & gmask);
    if (c > 0) atomicMax(&sc->acc4[grp][1], c);
    wave_sync();
    if (gl == 0) {
        sc->res[grp][0] = T;
        sc->res[grp][1] = sc->acc4[grp][0];
        sc->res[grp][2] = s2;
        sc->res[grp][3] = sc->acc4[grp][1];
        sc->res[grp][4] = posu;
    }
    wave_sync();
}

// positions per sub-unit: 16 where edges are cheap (amortises the table build and the closing expression), 8 for the
// block classes, whose edges cost thousands of entries each (more, smaller units: shorter tails)
// (measured on S100k / S1M, pass ms: positions per unit in the wave classes 8: 2.22, 16: 2.07, 32: 2.09; in the block
//  classes 2: 2.11, 4: 2.07 / 23.5, 8: 2.03 / 22.6, 16: 2.10 / 22.1; units per dequeue in the smallest class 1: 2.13,
//  2: 2.07, 4: 2.05, 8: 2.12)
#ifndef NC_LANES_SMALL
#define NC_LANES_SMALL 16
#endif
#ifndef NC_LANES_BIG
#define NC_LANES_BIG 8
#endif
#ifndef NC_CHUNK0
#define NC_CHUNK0 4
#endif
__host__ __device__ constexpr int nc_lanes(int c) { return c < 2 ? NC_LANES_SMALL : NC_LANES_BIG; }
__host__ __device__ constexpr int nc_lanes_for_slots(int slots) { return slots <= 512 ? NC_LANES_SMALL : NC_LANES_BIG; }
// (Round 5 measured smaller units for the incremental pass behind an SDRF edit — about a hundred edges over a few hundred units,
//  the pass as long as its longest unit: 4 positions per unit take the class of 63-254 neighbours from 178 to 107 us, one
//  position to 101, profiles/r05_incremental_lanes_ab.txt — and then replaced the class kernels there by the edge-by-edge pass
//  at the end of this file.)

template <int SLOTS, int MODE>
__device__ inline void nc_chunk(const View &g, int u, int2 ru, int sub, int nsub, const unsigned *tab, unsigned *cnt,
                                NcScratch *sc, const int *spilled, int curv_type, double *curv) {
    const bool ovf = __builtin_amdgcn_readfirstlane(*spilled) != 0;  // uniform, in a scalar register
    const int lane = threadIdx.x & 63;
    const int p = sub + lane * nsub;
    int v = -1;
    int2 rv = make_int2(0, 0);
    bool own = false;
    if (lane < nc_lanes_for_slots(SLOTS) && p < ru.y) {
        v = g.col[ru.x + p];
        if (v >= 0 && v < g.n && v != u) {
            rv = g.rowinfo[v];
            if (!row_ok(g, rv, 12, v, u)) rv = make_int2(0, 0);
            own = rv.y > 0 && nc_owns(u, ru.y, v, rv.y, curv_type == DCR_CURV_BFC);
            if (own && g.dirty) own = edge_dirty(g.dirty[u], g.dirty[v]);
        }
    }
#ifdef NC_DIRTY_STATS
    if (g.dirty) {
        const unsigned long long m = __ballot(own);
        int len = own ? rv.y : 0;
        for (int off = 32; off > 0; off >>= 1) len += __shfl_xor(len, off);
        if (lane == 0) {
            const int c = SLOTS == 256 ? 0 : SLOTS == 512 ? 1 : SLOTS == 2048 ? 2 : 3;
            atomicAdd(&nc_dstats[c * 8 + 0], 1ull);
            atomicAdd(&nc_dstats[c * 8 + 1], m ? 1ull : 0ull);
            atomicAdd(&nc_dstats[c * 8 + 2], (unsigned long long)__popcll(m));
            atomicAdd(&nc_dstats[c * 8 + 3], (unsigned long long)len);
        }
    }
#endif
    const bool trivial = curv_type == DCR_CURV_BFC && (ru.y < rv.y ? ru.y : rv.y) == 1;  // bfc_naive.py:18-19
    if (own && trivial && u < v) {
        curv[ru.x + p] = 0.0;  // the slot is in the owner's own row: nothing to look up
        own = false;
    }
    // (a degree-1 edge whose smaller endpoint is too big to own units takes the general path below, which finds
    //  the slot in the other row; its value is forced to 0 at the end)
    bool done = false;
    constexpr int B = nc_batch_for_slots(SLOTS);
    if constexpr (B > 1 && MODE == MODE_BFC) {
        // edges whose other endpoint has at most 64 / B neighbours are processed B at a time: the per-edge set-up
        // (sweeps, prefix sums, LDS round trips, the chain of dependent loads) is paid once per batch
        constexpr int G = 64 / B;
        const int grp = lane / G;
        unsigned long long small = __ballot(own && !trivial && rv.y <= G);
        while (small) {
            int lsel[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                lsel[b] = small ? __ffsll((long long)small) - 1 : -1;
                small &= small - 1;  // (0 & anything stays 0)
            }
            int src = -1;
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (grp == b) src = lsel[b];
            const int vg = __shfl(v, src < 0 ? 0 : src);
            int2 rvg = make_int2(__shfl(rv.x, src < 0 ? 0 : src), __shfl(rv.y, src < 0 ? 0 : src));
            if (src < 0) rvg.y = 0;
            nc_edge_batch<SLOTS, B>(g, u, vg, rvg, tab, cnt, sc, ovf);
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (lane == lsel[b]) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) sc->fin[lane][t] = sc->res[b][t];
                    done = true;
                }
            wave_sync();
        }
    }
    unsigned long long todo = __ballot(own && !done);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int ve = __shfl(v, l);
        const int2 rve = make_int2(__shfl(rv.x, l), __shfl(rv.y, l));
        const NcEdge r = nc_edge<SLOTS, MODE>(g, u, ve, rve, tab, cnt, sc, ovf);
        if (lane == l) {
            sc->fin[lane][0] = r.T; sc->fin[lane][1] = r.s1; sc->fin[lane][2] = r.s2; sc->fin[lane][3] = r.gam;
            sc->fin[lane][4] = r.posu;
        }
    }
    wave_sync();
    if (own) {
        const int my_T = sc->fin[lane][0], my_s1 = sc->fin[lane][1], my_s2 = sc->fin[lane][2], my_gam = sc->fin[lane][3],
                  my_posu = sc->fin[lane][4];
        int64_t slot = -1;
        if (u < v) slot = (int64_t)ru.x + p;
        else if (my_posu >= 0) slot = (int64_t)rv.x + my_posu;
        if (slot < 0 || slot >= g.cap_total) {
            row_ok(g, make_int2(-1, my_posu), 13, u, v);  // adjacency not symmetric: report, never write
        } else if (MODE == MODE_BFC) {
            curv[slot] = trivial ? 0.0 : bfc_formula(ru.y, rv.y, my_T, my_s1, my_s2, my_gam);
        } else {
            curv[slot] = curv_type == DCR_CURV_AUGMENTED ? (double)(4 - ru.y - rv.y + 3 * my_T) : (double)my_T;
        }
    }
}

// waves per SIMD the wave kernels are compiled for (measured on S100k: 5 -> 2.32 ms, 6 -> 2.05, 7 -> 2.23, 8 -> 2.39:
// below 6 the load latency shows, above it the spills cost more than the extra waves hide)
#ifndef NC_WAVE_OCC
#define NC_WAVE_OCC 6
#endif
#ifndef NC_QUEUES_N
#define NC_QUEUES_N 8
#endif
constexpr int NC_QUEUES = NC_QUEUES_N;         // dequeue cursors per wave-class kernel
constexpr int NC_QUEUE_STRIDE = 32;  // ints between cursors: one 128-byte line each

// ---- wave classes: a wave owns a node; persistent waves pull CHUNK nodes at a time ----------------------------
template <int SLOTS, int MODE, int CHUNK>
__global__ void __launch_bounds__(256, NC_WAVE_OCC) k_nc_wave(View g, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                  int32_t *next, int curv_type, double *curv) {
    constexpr int WPB = 4;
    __shared__ __attribute__((aligned(16))) unsigned tab_all[WPB][SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned cnt_all[WPB][nc_batch_for_slots(SLOTS) * SLOTS / 2];
    __shared__ NcScratch sc_all[WPB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned *tab = tab_all[wid], *cnt = cnt_all[wid];
    NcScratch *sc = &sc_all[wid];
    const int total = *count;
    if (total < 0 || total > unit_cap) {  // cannot happen; never walk a list with a corrupt length
        row_ok(g, make_int2(-1, total), 14, 0, 0);
        return;
    }
    long long *tr = g.trace ? g.trace + 2 * ((SLOTS == nc_slots(0) ? 0 : 16384) + (int)(blockIdx.x * WPB + wid) % 16384) : nullptr;
    if (tr && lane == 0) tr[0] = (long long)__builtin_amdgcn_s_memrealtime();
    // The unit list is dealt round-robin to NC_QUEUES queues, each with its own cursor on its own cache line: one
    // cursor for all waves saturates at ~90 dequeues per microsecond on MI355X and was 40 % of the pass.  A wave
    // starts on the queue of its workgroup (blockIdx % 8: workgroups that share an XCD) and moves on when it is empty.
    for (int qi = 0; qi < NC_QUEUES; ++qi) {
        const int q = (int)((blockIdx.x + qi) % NC_QUEUES);
        const int count_q = total > q ? (total - q + NC_QUEUES - 1) / NC_QUEUES : 0;
        int32_t *cursor = next + q * NC_QUEUE_STRIDE;
        const int max_rounds = count_q / CHUNK + 2;
        for (int round = 0; round < max_rounds; ++round) {
            int first = 0;
            if (lane == 0) first = atomicAdd(cursor, CHUNK);
            first = __builtin_amdgcn_readfirstlane(first);
            if (first >= count_q || first < 0) break;
            const int last = first + CHUNK < count_q ? first + CHUNK : count_q;
            for (int t = first; t < last; ++t) {
                const int it = q + t * NC_QUEUES;
                const int2 un = units[it];
                const int u = un.x, sub = un.y;
                if (u < 0 || u >= g.n || sub < 0) {
                    row_ok(g, make_int2(-1, u), 15, it, total);
                    continue;
                }
                int2 ru = g.rowinfo[u];
                if (!row_ok(g, ru, 16, u, it) || ru.y > SLOTS / 2 - 2) continue;  // the plan keeps the load <= 1/2
                const int nsub = (ru.y + nc_lanes_for_slots(SLOTS) - 1) / nc_lanes_for_slots(SLOTS);
                if (sub >= nsub) continue;
                for (int i = lane; i < SLOTS; i += 64) tab[i] = NC_EMPTY;
                if (lane == 0) sc->spilled = 0;
                wave_sync();
                for (int i = lane; i < ru.y; i += 64) {
                    const int k = g.col[ru.x + i];
                    if (k >= 0) nc_insert<SLOTS>(tab, (unsigned)k, &sc->spilled);
                }
                wave_sync();
                nc_chunk<SLOTS, MODE>(g, u, ru, sub, nsub, tab, cnt, sc, &sc->spilled, curv_type, curv);
            }
            if (tr && lane == 0) tr[1] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    }
}

// ---- block classes: one table per workgroup, wave w takes sub-unit sub0 + w of the row --------------------------
// The unit loop of one class, run by a workgroup of BW waves of which the first W work on units (W < BW only in the
// combined kernel below).  LDS comes from the caller.
template <int SLOTS, int W, int BW, int MODE>
__device__ inline void nc_block_units(const View &g, const int2 *units, const int32_t *count, int64_t unit_cap,
                                      int32_t *next, int curv_type, double *curv, unsigned *tab, unsigned *cnt_base,
                                      NcScratch *sc_all, int *it_sh, int trace_slot) {
    const int wid = threadIdx.x >> 6;
    const int total = *count;
    if (total < 0 || total > unit_cap) {  // uniform
        row_ok(g, make_int2(-1, total), 17, 0, 0);
        return;
    }
    long long *tr = g.trace ? g.trace + 2 * (trace_slot * 16384 + (int)(blockIdx.x * BW + wid) % 16384) : nullptr;
    if (tr && (threadIdx.x & 63) == 0 && wid < W) tr[0] = (long long)__builtin_amdgcn_s_memrealtime();
    // dynamic dequeue: thread 0 pulls the next unit and publishes it through LDS between two barriers, so every value
    // that steers control flow around the barriers is uniform in the workgroup
    for (int round = 0; round <= total; ++round) {
        if (threadIdx.x == 0) *it_sh = atomicAdd(next, 1);
        __syncthreads();
        const int it = *it_sh;
        if (it >= total || it < 0) {
            __syncthreads();  // nobody may still be reading it_sh when the caller's next loop writes it
            break;
        }
        const int2 un = units[it];
        const int u = un.x, sub0 = un.y;
        bool ok = u >= 0 && u < g.n && sub0 >= 0;
        int2 ru = make_int2(0, 0);
        if (ok) {
            ru = g.rowinfo[u];
            ok = row_ok(g, ru, 18, u, it) && ru.y <= SLOTS / 2 - 2;
        }
        if (!ok) {  // uniform
            __syncthreads();
            continue;
        }
        for (int i = threadIdx.x; i < SLOTS; i += 64 * BW) tab[i] = NC_EMPTY;
        if (threadIdx.x == 0) sc_all[0].spilled = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < ru.y; i += 64 * BW) {
            const int k = g.col[ru.x + i];
            if (k >= 0) nc_insert<SLOTS>(tab, (unsigned)k, &sc_all[0].spilled);
        }
        __syncthreads();
        const int nsub = (ru.y + nc_lanes_for_slots(SLOTS) - 1) / nc_lanes_for_slots(SLOTS);
        const int sub = sub0 + wid;
        if (wid < W && sub < nsub)
            nc_chunk<SLOTS, MODE>(g, u, ru, sub, nsub, tab, cnt_base + wid * (SLOTS / 2), &sc_all[wid], &sc_all[0].spilled,
                                  curv_type, curv);
        if (tr && (threadIdx.x & 63) == 0 && wid < W) tr[1] = (long long)__builtin_amdgcn_s_memrealtime();
        __syncthreads();  // the table and the unit index are rewritten by the next round
    }
}

template <int SLOTS, int W, int MODE>
__global__ void __launch_bounds__(64 * W) k_nc_block(View g, const int2 *units, const int32_t *count, int64_t unit_cap,
                                                      int32_t *next, int curv_type, double *curv) {
    __shared__ __attribute__((aligned(16))) unsigned tab[SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned cnt_all[W * (SLOTS / 2)];
    __shared__ NcScratch sc_all[W];
    __shared__ int it_sh;
    nc_block_units<SLOTS, W, W, MODE>(g, units, count, unit_cap, next, curv_type, curv, tab, cnt_all, sc_all, &it_sh,
                                      SLOTS == 2048 ? 2 : 3);
}

// The two largest classes in ONE kernel.  Their workgroups need most of a CU's LDS (table 64 KB + 2 x 32 KB of slot
// state, or 32 KB + 4 x 16 KB); launched as kernels of their own, the second one found no CU with that much LDS free
// until the other persistent kernels had finished (measured on the 1M-node graph: it started at the very end of the
// pass and added its whole run time).  One launch, first on the main stream, 4 waves: the workgroups take the units of
// the 16,384-slot class first (waves 2 and 3 only help to build the table), then those of the 8,192-slot class.
template <int MODE>
__global__ void __launch_bounds__(256) k_nc_block_big(View g, const int2 *units4, const int32_t *count4, int64_t cap4,
                                                       int32_t *next4, const int2 *units3, const int32_t *count3,
                                                       int64_t cap3, int32_t *next3, int curv_type, double *curv) {
    __shared__ __attribute__((aligned(16))) unsigned tab[16384];
    __shared__ __attribute__((aligned(16))) unsigned cnt_all[2 * (16384 / 2)];  // = 4 * (8192 / 2)
    __shared__ NcScratch sc_all[4];
    __shared__ int it_sh;
    nc_block_units<16384, 2, 4, MODE>(g, units4, count4, cap4, next4, curv_type, curv, tab, cnt_all, sc_all, &it_sh, 4);
    nc_block_units<8192, 4, 4, MODE>(g, units3, count3, cap3, next3, curv_type, curv, tab, cnt_all, sc_all, &it_sh, 3);
}

// ---- plan: one thread per node appends its units to the list of its degree class -----------------------------
struct NcLists {
    int2 *units[NC_CLASSES];
    int64_t cap[NC_CLASSES];
};

// Units are laid out heaviest first inside each class (by degree bucket of the owning node), so that with dynamic
// dequeue the long-running units start at once and the cheap ones fill the tail.
constexpr int NC_BUCKETS = 16;
__device__ inline int nc_bucket_lo(int b) {
    constexpr int lo[NC_BUCKETS] = {1, 13, 17, 25, 33, 49, 63, 97, 129, 193, 255, 511, 767, 1023, 2047, 4095};
    return lo[b];
}
__device__ inline int nc_bucket_of(int d) {
    int b = 0;
#pragma unroll
    for (int i = 1; i < NC_BUCKETS; ++i) b += (d >= nc_bucket_lo(i));
    return b;
}
__device__ inline int nc_bucket_class(int b) { return b < 6 ? 0 : b < 10 ? 1 : b < 13 ? 2 : b < 15 ? 3 : 4; }

// PHASE 0 counts the units per bucket; PHASE 1 places them (class list = its buckets, heaviest first).  Reservations
// are aggregated wave -> workgroup (LDS) -> one global atomic per bucket and workgroup: the few bucket counters are
// shared by every node, and one atomic per wave on them made the plan cost as much as 3 % of the pass.
constexpr int PLAN_THREADS = 1024;

template <int PHASE>
__global__ void __launch_bounds__(PLAN_THREADS) k_nc_plan(View g, NcLists L, const uint8_t *touch, DevResult *res) {
    __shared__ int blk_count[NC_BUCKETS];   // units of this workgroup per bucket
    __shared__ int blk_base[NC_BUCKETS];    // PHASE 1: where this workgroup's units of a bucket start in the class list
    const int u = blockIdx.x * PLAN_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < NC_BUCKETS) blk_count[threadIdx.x] = 0;
    __syncthreads();
    int bkt = -1, d = 0;
    if (u < g.n) {
        d = g.rowinfo[u].y;
        if (d > 0 && d <= NC_MAXD && (!g.dirty || touch[u])) bkt = nc_bucket_of(d);
    }
    const int cls = bkt < 0 ? -1 : nc_bucket_class(bkt);
    const int L_ = cls < 0 ? 16 : nc_lanes(cls);
    const int nsub = (d + L_ - 1) / L_;
    // wave classes: one unit per sub-unit; block classes: one unit per group of W sub-units
    const int W = cls >= 2 ? nc_waves(cls) : 1;
    const int nunits = cls < 0 ? 0 : (nsub + W - 1) / W;
    int my_off = 0;  // offset of this node's units inside its workgroup's share of the bucket
    for (int b = 0; b < NC_BUCKETS; ++b) {
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
    // (round 5: the bucket totals come into LDS with ONE load per bucket, all in flight together; the sums over them below ran as
    //  chains of up to 15 dependent L2 round trips per thread — 46 us for this kernel against 7 for PHASE 0)
    __shared__ int bucket_sh[NC_BUCKETS];
    if (PHASE == 1 && threadIdx.x >= 128 && threadIdx.x < 128 + NC_BUCKETS) bucket_sh[threadIdx.x - 128] = res->nc_bucket[threadIdx.x - 128];
    __syncthreads();
    if (threadIdx.x < NC_BUCKETS) {
        const int b = threadIdx.x, c = blk_count[b];
        if (PHASE == 0) {
            if (c) atomicAdd(&res->nc_bucket[b], c);
        } else {
            int before = 0;
            for (int h = b + 1; h < NC_BUCKETS; ++h)  // heavier buckets of the same class come first
                if (nc_bucket_class(h) == nc_bucket_class(b)) before += bucket_sh[h];
            blk_base[b] = before + (c ? atomicAdd(&res->nc_fill[b], c) : 0);
        }
    }
    if (PHASE == 0) return;
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + NC_CLASSES) {
        int tot = 0;
        for (int b = 0; b < NC_BUCKETS; ++b)
            if (nc_bucket_class(b) == (int)threadIdx.x - 64) tot += bucket_sh[b];
        res->nc_count[threadIdx.x - 64] = tot;
    }
    __syncthreads();
    if (bkt >= 0) {
        const int first = blk_base[bkt] + my_off;
        if (first < 0 || (int64_t)first + nunits > L.cap[cls]) {
            row_ok(g, make_int2(-1, first), 19, u, nunits);
        } else {
            for (int j = 0; j < nunits; ++j) L.units[cls][first + j] = make_int2(u, j * W);
        }
    }
}

// incremental pass: the nodes with at least one edge that an edit can have changed (either endpoint may own it)
__global__ void __launch_bounds__(256) k_nc_touch(View g, uint8_t *touch) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= g.cap_total) return;
    const int u = g.slot_row[s];
    const int2 ru = g.rowinfo[u];
    if ((int)(s - ru.x) >= ru.y) return;
    const int v = g.col[s];
    if (v >= 0 && v < g.n && edge_dirty(g.dirty[u], g.dirty[v])) touch[u] = 1;
}

// (round 5: also the dequeue cursors and, for an incremental pass, the per-node flags k_nc_touch sets — two fill launches of
//  4 us each at the head of every pass before)
__global__ void __launch_bounds__(256) k_nc_clear(DevResult *res, int32_t *queues, int n_queue_ints, unsigned *touch_words, int64_t n_touch_words) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    for (int64_t i = tid; i < n_queue_ints; i += nth) queues[i] = 0;
    for (int64_t i = tid; i < n_touch_words; i += nth) touch_words[i] = 0u;
    if (blockIdx.x != 0) return;
    if (threadIdx.x < 8) res->misc[threadIdx.x] = 0;
    if (threadIdx.x < 16) {
        res->nc_bucket[threadIdx.x] = 0;
        res->nc_fill[threadIdx.x] = 0;
    }
    if (threadIdx.x < NC_CLASSES) res->nc_next[threadIdx.x] = 0;
    if (threadIdx.x == 0) res->flag_too_big = 0;
    if (threadIdx.x < NC_CLASSES) res->nc_count[threadIdx.x] = 0;
}

static int ensure_nc(dcr_graph *g) {
    // units per class, from the smallest degree a member node can have and its sub-units (16 or 4 positions each)
    const int64_t need[NC_CLASSES] = {g->n * 4 + 64, g->cap_total / 3 + 64, g->cap_total / 6 + 64,
                                      g->cap_total / 3 + 64, g->cap_total / 6 + 64};
    for (int c = 0; c < NC_CLASSES; ++c) {
        if (g->nc_cap[c] < need[c]) {
            if (g->nc_units[c]) (void)hipFree(g->nc_units[c]);
            g->nc_units[c] = nullptr;
            DCR_TRY(dev_alloc(&g->nc_units[c], need[c]));
            g->nc_cap[c] = need[c];
        }
    }
    if (g->nc_touch_cap < g->n + 64) {
        if (g->nc_touch) (void)hipFree(g->nc_touch);
        g->nc_touch = nullptr;
        DCR_TRY(dev_alloc(&g->nc_touch, g->n + 64));
        g->nc_touch_cap = g->n + 64;
    }
    return DCR_OK;
}

template <int C, int MODE>
static void launch_nc_wave(dcr_graph *g, const View &vw, int curv_type, hipStream_t st) {
    constexpr int SLOTS = nc_slots(C);
    constexpr int CHUNK = C == 0 ? NC_CHUNK0 : 1;
    constexpr int LDS = 4 * (SLOTS * 4 + nc_batch_for_slots(SLOTS) * SLOTS * 2 + (int)sizeof(NcScratch));
    int per_cu = (160 * 1024) / LDS;
    if (per_cu > 8) per_cu = 8;  // 32 wave slots per CU, 4 waves per workgroup
    if (per_cu < 1) per_cu = 1;
    // no more workgroups than there can be work for (small graphs): units <= nodes of the class + their slots / 16
    int64_t grid = (int64_t)g->num_cu * per_cu;
    const int64_t nodes_bound = C == 0 ? g->n : g->cap_total / (nc_maxdeg(C - 1) + 1);
    const int64_t by_work = (nodes_bound + g->cap_total / NC_LANES_SMALL) / (4 * CHUNK) + 1;
    if (grid > by_work) grid = by_work;
    // incremental pass after a few exactly flagged edits (an SDRF step): a handful of units, and a full persistent grid
    // costs more to start and drain than they do (pass 0.34 -> 0.26 ms with one workgroup per CU)
    if (vw.dirty && g->pending_edits <= DIRTY_EDITS && grid > g->num_cu) grid = g->num_cu;
    hipLaunchKernelGGL((k_nc_wave<SLOTS, MODE, CHUNK>), dim3((unsigned)grid), dim3(256), 0, st, vw, g->nc_units[C],
                       &g->dres->nc_count[C], g->nc_cap[C], g->nc_queues + C * NC_QUEUES * NC_QUEUE_STRIDE, curv_type,
                       g->curv);
}

template <int C, int MODE>
static void launch_nc_block(dcr_graph *g, const View &vw, int curv_type, hipStream_t st) {
    constexpr int SLOTS = nc_slots(C);
    constexpr int W = nc_waves(C);
    constexpr int LDS = SLOTS * 4 + W * (SLOTS * 2 + (int)sizeof(NcScratch));
    int per_cu = (160 * 1024) / LDS;
    if (per_cu > 32 / W) per_cu = 32 / W;
    if (per_cu < 1) per_cu = 1;
    hipLaunchKernelGGL((k_nc_block<SLOTS, W, MODE>), dim3(g->num_cu * per_cu), dim3(64 * W), 0, st, vw, g->nc_units[C],
                       &g->dres->nc_count[C], g->nc_cap[C], &g->dres->nc_next[C], curv_type, g->curv);
}

template <int MODE>
static void launch_nc_block_big(dcr_graph *g, const View &vw, int curv_type, hipStream_t st) {
    hipLaunchKernelGGL((k_nc_block_big<MODE>), dim3(g->num_cu), dim3(256), 0, st, vw, g->nc_units[4],
                       &g->dres->nc_count[4], g->nc_cap[4], &g->dres->nc_next[4], g->nc_units[3], &g->dres->nc_count[3],
                       g->nc_cap[3], &g->dres->nc_next[3], curv_type, g->curv);
}

template <int MODE>
static int run_nc(dcr_graph *g, int curv_type, bool incremental) {
    DCR_TRY(ensure_nc(g));
    if (g->num_cu <= 0) {
        g->num_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0)
            g->num_cu = prop.multiProcessorCount;
    }
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, incremental ? g->dirty : nullptr,
            (int32_t)g->n, 1, nullptr};
    static const bool want_trace = getenv("DCR_NC_TRACE") != nullptr;
    if (want_trace) {
        if (!g->nc_trace) DCR_TRY(dev_alloc(&g->nc_trace, NC_CLASSES * 16384 * 2));
        DCR_HIP(hipMemsetAsync(g->nc_trace, 0, sizeof(long long) * NC_CLASSES * 16384 * 2, g->stream));
        vw.trace = g->nc_trace;
    }
    NcLists L;
    for (int c = 0; c < NC_CLASSES; ++c) {
        L.units[c] = g->nc_units[c];
        L.cap[c] = g->nc_cap[c];
    }
    if (!g->nc_queues) DCR_TRY(dev_alloc(&g->nc_queues, 2 * NC_QUEUES * NC_QUEUE_STRIDE));
    {
        const int64_t touch_words = incremental ? (g->n + 3) / 4 : 0;   // (the buffer holds n + 64 bytes)
        int64_t cb = (touch_words + 1023) / 1024;
        if (cb < 1) cb = 1;
        if (cb > 256) cb = 256;
        hipLaunchKernelGGL(k_nc_clear, dim3((unsigned)cb), dim3(256), 0, g->stream, g->dres, g->nc_queues, 2 * NC_QUEUES * NC_QUEUE_STRIDE,
                           reinterpret_cast<unsigned *>(g->nc_touch), touch_words);
    }
    if (incremental) {
        const int64_t blocks = (g->cap_total + 255) / 256;
        if (blocks > 0) hipLaunchKernelGGL(k_nc_touch, dim3((unsigned)blocks), dim3(256), 0, g->stream, vw, g->nc_touch);
    }
    const int64_t pblocks = (g->n + PLAN_THREADS - 1) / PLAN_THREADS;
    if (pblocks > 0) {
        hipLaunchKernelGGL(k_nc_plan<0>, dim3((unsigned)pblocks), dim3(PLAN_THREADS), 0, g->stream, vw, L, g->nc_touch,
                           g->dres);
        hipLaunchKernelGGL(k_nc_plan<1>, dim3((unsigned)pblocks), dim3(PLAN_THREADS), 0, g->stream, vw, L, g->nc_touch,
                           g->dres);
    }
    // the classes are independent: fork them onto side streams; the rarest, longest-running units first
    static const bool serial = getenv("DCR_SERIAL_BINS") != nullptr;  // debugging aid: one stream
    hipStream_t s1 = g->stream, s2 = g->stream, s3 = g->stream;
    if (!serial) {
        DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
        s1 = g->side[0]; s2 = g->side[1]; s3 = g->side[2];
    }
    // a class whose smallest degree exceeds the (host-tracked upper bound of the) largest degree has no units: on
    // small graphs that saves the dispatch of up to four full persistent grids.  (Each side stream's wait is enqueued
    // right before its kernel: the chip is through the plan before the host is through these calls.)
    if (g->max_deg_bound > nc_maxdeg(3)) launch_nc_block_big<MODE>(g, vw, curv_type, g->stream);  // classes 4 and 3
    else if (g->max_deg_bound > nc_maxdeg(2)) launch_nc_block<3, MODE>(g, vw, curv_type, g->stream);
    if (!serial) DCR_HIP(hipStreamWaitEvent(s1, g->ev_fork, 0));
    if (g->max_deg_bound > nc_maxdeg(1)) launch_nc_block<2, MODE>(g, vw, curv_type, s1);
    if (!serial) DCR_HIP(hipStreamWaitEvent(s2, g->ev_fork, 0));
    if (g->max_deg_bound > nc_maxdeg(0)) launch_nc_wave<1, MODE>(g, vw, curv_type, s2);
    if (!serial) DCR_HIP(hipStreamWaitEvent(s3, g->ev_fork, 0));
    launch_nc_wave<0, MODE>(g, vw, curv_type, s3);
    if (!serial) {
        for (int b = 0; b < 3; ++b) {
            DCR_HIP(hipEventRecord(g->ev_join[b], g->side[b]));
            DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[b], 0));
        }
    }
    DCR_HIP(hipGetLastError());
#ifdef NC_DIRTY_STATS
    if (incremental) {
        unsigned long long h[32];
        DCR_HIP(hipStreamSynchronize(g->stream));
        DCR_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(nc_dstats), sizeof(h)));
        for (int c = 0; c < 4; ++c)
            fprintf(stderr, "[nc dirty stats] class %d: units %llu, with work %llu, edges %llu, row entries %llu\n", c, h[c * 8], h[c * 8 + 1],
                    h[c * 8 + 2], h[c * 8 + 3]);
        unsigned long long z[32] = {0};
        DCR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(nc_dstats), z, sizeof(z)));
    }
#endif
#ifdef NC_STATS
    {
        unsigned long long h[8];
        DCR_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(nc_stats), sizeof(h)));
        fprintf(stderr, "[nc stats] piece probes (wave level) %llu, slow path %llu (%.1f%%), lanes active %.1f of 64, lanes with a candidate per slow probe %.2f\n",
                h[0], h[1], 100.0 * h[1] / (h[0] ? h[0] : 1), (double)h[2] / (h[0] ? h[0] : 1), (double)h[3] / (h[1] ? h[1] : 1));
        unsigned long long z[8] = {0};
        DCR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(nc_stats), z, sizeof(z)));
    }
#endif
    if (want_trace) {  // per class: when did the waves start / make their last progress (100 MHz ticks -> microseconds)
        std::vector<long long> h(NC_CLASSES * 16384 * 2);
        DCR_HIP(hipStreamSynchronize(g->stream));
        DCR_HIP(hipMemcpy(h.data(), g->nc_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long long t0 = 0;
        for (size_t i = 0; i < h.size(); i += 2)
            if (h[i] && (!t0 || h[i] < t0)) t0 = h[i];
        for (int c = 0; c < NC_CLASSES; ++c) {
            std::vector<double> st, en;
            for (int w = 0; w < 16384; ++w) {
                const long long a = h[2 * (c * 16384 + w)], b = h[2 * (c * 16384 + w) + 1];
                if (!a) continue;
                st.push_back((a - t0) * 0.01);
                en.push_back(((b ? b : a) - t0) * 0.01);
            }
            if (st.empty()) continue;
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            auto q = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            fprintf(stderr, "[nc trace] class %d: %zu waves; start p0 %.0f p50 %.0f p100 %.0f us; last progress p10 %.0f p50 %.0f p90 %.0f p99 %.0f p100 %.0f us\n",
                    c, st.size(), st.front(), q(st, 0.5), st.back(), q(en, 0.1), q(en, 0.5), q(en, 0.9), q(en, 0.99), en.back());
        }
    }
    return DCR_OK;
}

// =====================================================================================================================
// Round 5: the incremental pass behind a few exactly flagged edits (an SDRF iteration), edge by edge
// =====================================================================================================================
// Such a pass recomputes about a hundred edges (tools/build_variant.sh dstats -DNC_DIRTY_STATS).  Through the class kernels
// above it cost what their longest unit costs: a wave builds a table and streams its unit's edges one after the other, the
// rows of a long edge batch by batch (class of 63-254 neighbours: 178 us with units of 16 positions, 102 with 4), behind a
// sweep over the slots for the touched nodes and the two plan launches.  Here: ONE sweep over the slots lists the edges to
// recompute (owner, position in its row) — the same ownership rule, the same dirty test as nc_chunk — and ONE kernel gives
// every listed edge a workgroup of four waves: the table of the owner's neighbours is built by all of them (one size, the
// largest class's, for every owner), the two sweeps over the other endpoint's row are dealt to the waves in batches of 64
// rows (the slot counters are LDS atomics, every statistic a sum or a maximum: nc_edge with its loops strided by the wave
// index).  Same integers, same closing expression.  Measured: the pass 0.240 -> 0.076 ms, the iteration 0.375 -> 0.21 ms.
// DCR_NC_FINE=0: the class kernels, as for a pass behind more (coarsely flagged) edits.
constexpr int NCF_W = 4;   // (table sizes: 1,024 / 4,096 / 16,384 slots, the smallest that holds the graph's largest degree at half load —
                           //  a hub-to-hub edit flags tens of thousands of edges, and a smaller table is more workgroups per CU)
struct NcFineAcc {
    int T, posu, s1, gam, s2, spilled;
};

__global__ void __launch_bounds__(256) k_nc_fine_list(View g, int curv_type, double *curv, int2 *list, int64_t cap, int32_t *count) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= g.cap_total) return;
    const int u = g.slot_row[s];
    if (u < 0 || u >= g.n) return;
    const int2 ru = g.rowinfo[u];
    if (s < ru.x || (int)(s - ru.x) >= ru.y || ru.y > NC_MAXD) return;
    const int v = g.col[s];
    if (v < 0 || v >= g.n || v == u) return;
    if (g.dirty && !edge_dirty(g.dirty[u], g.dirty[v])) return;  // no edit can have changed it: the stored value is still exact
    const int2 rv = g.rowinfo[v];
    if (!row_ok(g, rv, 12, v, u)) return;
    const bool bfc = curv_type == DCR_CURV_BFC;
    if (!(rv.y > 0 && nc_owns(u, ru.y, v, rv.y, bfc))) return;
    if (bfc && (ru.y < rv.y ? ru.y : rv.y) == 1 && u < v) {  // bfc_naive.py:18-19 (as nc_chunk: the slot is in the owner's own row)
        curv[s] = 0.0;
        return;
    }
    const int idx = atomicAdd(count, 1);
    if (idx >= 0 && idx < cap) list[idx] = make_int2(u, (int)(s - ru.x));
}

// The same list from the rows of the FLAGGED nodes (the touched list the edit kernels keep: every node whose dirty byte left zero
// since the flags were cleared, csrc/dcr_graph.hip) — a wave per node.  A flagged edge has both endpoints flagged (an edited
// node's neighbours carry its A or B bit), so its owner's row is among them.  Used for graphs of 4 M slots and more (S1M: the
// sweep's 141 us -> 45 us; on S100k the sweep's 11 us are less than this kernel's chain of dependent reads).
__global__ void __launch_bounds__(256) k_nc_fine_list_rows(View g, int curv_type, double *curv, const int32_t *touched, int64_t tcap,
                                                           const DevResult *res, int2 *list, int64_t cap, int32_t *count) {
    const int lane = threadIdx.x & 63;
    const int64_t nt = res->touched_n;
    if (nt < 0 || nt > tcap) {  // uniform: more appends than nodes — some clearing of the flags did not reset the list
        row_ok(g, make_int2(-1, (int)nt), 25, 0, 0);
        return;
    }
    const bool bfc = curv_type == DCR_CURV_BFC;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < nt; t += (int64_t)gridDim.x * 4) {
        const int u = touched[t];
        if (u < 0 || u >= g.n) continue;
        const int2 ru = g.rowinfo[u];
        if (!row_ok(g, ru, 26, u, (int)t) || ru.y > NC_MAXD) continue;
        const unsigned du = g.dirty[u];
        for (int base = 0; base < ru.y; base += 64) {
            const int p = base + lane;
            if (p >= ru.y) continue;
            const int v = g.col[ru.x + p];
            if (v < 0 || v >= g.n || v == u) continue;
            if (!edge_dirty(du, g.dirty[v])) continue;
            const int2 rv = g.rowinfo[v];
            if (!row_ok(g, rv, 12, v, u)) continue;
            if (!(rv.y > 0 && nc_owns(u, ru.y, v, rv.y, bfc))) continue;
            if (bfc && (ru.y < rv.y ? ru.y : rv.y) == 1 && u < v) {  // bfc_naive.py:18-19
                curv[ru.x + p] = 0.0;
                continue;
            }
            const int idx = atomicAdd(count, 1);
            if (idx >= 0 && idx < cap) list[idx] = make_int2(u, p);
        }
    }
}

// nc_edge by the NCF_W waves of a workgroup: `tab` (N(u)), `cnt` and `acc` are the workgroup's, `sc` this wave's
template <int SLOTS, int MODE>
__device__ inline NcEdge nc_edge_wg(const View &g, int u, int v, int2 rv, const unsigned *tab, unsigned *cnt, NcScratch *sc,
                                    NcFineAcc *acc, bool ovf) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int32_t *rowv = g.col + rv.x;
    NcEdge out;
    out.T = out.s1 = out.s2 = out.gam = 0;
    out.posu = -1;
    if (MODE == MODE_BFC) {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = threadIdx.x; i < SLOTS / 8; i += 64 * NCF_W) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (threadIdx.x == 0) {
        acc->T = 0;
        acc->posu = -1;
        acc->s1 = 0;
        acc->gam = 0;
        acc->s2 = 0;
    }
    __syncthreads();
    if (MODE == MODE_BFC && threadIdx.x == 0) {
        const int hv = nc_find<SLOTS>(tab, (unsigned)v);  // v is a neighbour of u: never counted as a hit
        if (hv >= 0) cnt_flag(cnt, hv);
    }
    // sweep 1 over N(v): triangles (flagged), where u sits in row v
    int T = 0;
    for (int base = 64 * wid; base < rv.y; base += 64 * NCF_W) {
        const int i = base + lane;
        const int k = i < rv.y ? rowv[i] : -1;
        const bool isu = k == u;
        const unsigned long long mu = __ballot(isu);
        if (mu && lane == 0) acc->posu = base + __ffsll((long long)mu) - 1;
        const int h = (k >= 0 && !isu) ? nc_find<SLOTS>(tab, (unsigned)k) : -1;
        if (MODE == MODE_BFC && h >= 0) cnt_flag(cnt, h);
        T += __popcll(__ballot(h >= 0));
    }
    if (lane == 0 && T) atomicAdd(&acc->T, T);
    __syncthreads();  // every flag is set before the first hit is counted
    out.T = acc->T;
    out.posu = acc->posu;
    if (MODE != MODE_BFC) return out;
    // sweep 2 over N(v): the rows of DY, 64 members at a time, the batches dealt to the waves
    int s1 = 0, gam = 0, s2 = 0;
    int qn = 0;  // queued full look-ups (uniform per wave)
    auto drain = [&]() {
        wave_sync();
        for (int qb = 0; qb < qn; qb += 64) {
            const int qi = qb + lane;
            if (qi < qn) {
                const int h = nc_find<SLOTS>(tab, sc->qk[qi]);
                if (h >= 0) {
                    const unsigned old = cnt_add(cnt, h);
                    if (!(old & 0x8000u)) {  // not a member of N(v): a 4-cycle u-z-w-v
                        atomicAdd(&sc->rowcnt[sc->qrow[qi]], 1);
                        s1 += (old == 0u);
                        gam = (int)old + 1 > gam ? (int)old + 1 : gam;
                    }
                }
            }
        }
        wave_sync();
        qn = 0;
    };
    for (int base = 64 * wid; base < rv.y; base += 64 * NCF_W) {
        const int i = base + lane;
        const int k = i < rv.y ? rowv[i] : -1;
        const bool in_nu = k >= 0 && nc_find<SLOTS>(tab, (unsigned)k) >= 0;
        const bool member = k >= 0 && k < g.n && k != u && !in_nu;
        int2 rk = make_int2(0, 0);
        if (member) {
            rk = g.rowinfo[k];
            if (!row_ok(g, rk, 11, k, v)) rk = make_int2(0, 0);
        }
        const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
        int incl = np;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int P = __shfl(incl, 63);
        sc->desc[lane] = rk;
        sc->poff[lane] = incl - np;
        const int poff_lane = incl - np;
        sc->rowcnt[lane] = 0;
        if (lane == 0) sc->poff[64] = P;
        wave_sync();
        for (int j0 = 0; j0 < P; j0 += 64 * NC_Q) {
            int4 w[NC_Q];
            int rr[NC_Q], aa[NC_Q];
#pragma unroll
            for (int q = 0; q < NC_Q; ++q) {
                const int j = j0 + 64 * q + lane;
                rr[q] = -1;
                aa[q] = 0;
                w[q] = make_int4(0, 0, 0, 0);
                const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
                if (jf >= P) continue;  // uniform
                const int r = piece_row(sc->poff, poff_lane, j < P ? j : jl, jf, jl);
                if (j < P) {
                    const int2 d = sc->desc[r];
                    const int a = (d.x & ~3) + 4 * (j - sc->poff[r]);
                    w[q] = load_piece(g.col, a);
                    rr[q] = r;
                    aa[q] = a;
                }
            }
#pragma unroll
            for (int q = 0; q < NC_Q; ++q) {
                if (j0 + 64 * q >= P) continue;  // uniform
                unsigned f = 0u;
                if (rr[q] >= 0) {
                    const int2 d = sc->desc[rr[q]];
                    f = nc_probe_flags<SLOTS>(tab, w[q], piece_mask(aa[q], d.x, d.x + d.y), ovf, (unsigned)v);
                }
                if (__ballot(f != 0u)) nc_queue_push(sc, qn, f, w[q], rr[q], drain);  // uniform; rare per lane, common per wave
            }
        }
        if (qn > 0) drain();
        wave_sync();
        const int c = sc->rowcnt[lane];
        s2 += __popcll(__ballot(c > 0));
        gam = c > gam ? c : gam;
        wave_sync();  // the scratch is rewritten by the next batch
    }
    if (s1 > 0) atomicAdd(&acc->s1, s1);
    if (gam > 0) atomicMax(&acc->gam, gam);
    if (lane == 0 && s2 > 0) atomicAdd(&acc->s2, s2);
    __syncthreads();
    out.s1 = acc->s1;
    out.gam = acc->gam;
    out.s2 = acc->s2;
    return out;
}

template <int NCF_SLOTS, int MODE>
__global__ void __launch_bounds__(64 * NCF_W) k_nc_fine_edges(View g, const int2 *list, const int32_t *count, int64_t cap, int curv_type,
                                                               double *curv) {
    __shared__ __attribute__((aligned(16))) unsigned tab[NCF_SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned cnt[NCF_SLOTS / 2];
    __shared__ NcScratch sc_all[NCF_W];
    __shared__ NcFineAcc acc;
    const int total = *count;
    if (total < 0 || total > cap) {  // uniform (more edges than slots / 2: the adjacency is not symmetric)
        row_ok(g, make_int2(-1, total), 21, 0, 0);
        return;
    }
    for (int e = blockIdx.x; e < total; e += gridDim.x) {
        const int2 it = list[e];
        const int u = it.x, p = it.y;
        bool ok = u >= 0 && u < g.n && p >= 0;
        int2 ru = make_int2(0, 0), rv = make_int2(0, 0);
        int v = -1;
        if (ok) {
            ru = g.rowinfo[u];
            ok = row_ok(g, ru, 22, u, e) && p < ru.y;
            if (ok && ru.y > NCF_SLOTS / 2 - 2) {  // the host's bound on the largest degree picked this table: cannot happen; never skip silently
                row_ok(g, make_int2(-1, ru.y), 24, u, e);
                ok = false;
            }
        }
        if (ok) {
            v = g.col[ru.x + p];
            ok = v >= 0 && v < g.n && v != u;
        }
        if (ok) {
            rv = g.rowinfo[v];
            ok = row_ok(g, rv, 23, v, u) && rv.y > 0;
        }
        if (!ok) continue;  // uniform
        for (int i = threadIdx.x; i < NCF_SLOTS; i += 64 * NCF_W) tab[i] = NC_EMPTY;
        if (threadIdx.x == 0) acc.spilled = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < ru.y; i += 64 * NCF_W) {
            const int k = g.col[ru.x + i];
            if (k >= 0) nc_insert<NCF_SLOTS>(tab, (unsigned)k, &acc.spilled);
        }
        __syncthreads();
        const bool ovf = acc.spilled != 0;
        const NcEdge r = nc_edge_wg<NCF_SLOTS, MODE>(g, u, v, rv, tab, cnt, &sc_all[threadIdx.x >> 6], &acc, ovf);
        if (threadIdx.x == 0) {
            int64_t slot = -1;
            if (u < v) slot = (int64_t)ru.x + p;
            else if (r.posu >= 0) slot = (int64_t)rv.x + r.posu;
            if (slot < 0 || slot >= g.cap_total) {
                row_ok(g, make_int2(-1, r.posu), 13, u, v);  // adjacency not symmetric: report, never write
            } else if (MODE == MODE_BFC) {
                const bool trivial = (ru.y < rv.y ? ru.y : rv.y) == 1;  // bfc_naive.py:18-19
                curv[slot] = trivial ? 0.0 : bfc_formula(ru.y, rv.y, r.T, r.s1, r.s2, r.gam);
            } else {
                curv[slot] = curv_type == DCR_CURV_AUGMENTED ? (double)(4 - ru.y - rv.y + 3 * r.T) : (double)r.T;
            }
        }
        __syncthreads();  // table, counters and totals are rewritten for the next edge
    }
}

template <int MODE>
static int run_nc_fine(dcr_graph *g, int curv_type, bool incremental) {
    if (g->num_cu <= 0) {
        g->num_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0) g->num_cu = prop.multiProcessorCount;
    }
    const int64_t need = g->cap_total / 2 + 64;  // every edge has one owner
    if (g->nc_fine_cap < need) {
        if (g->nc_fine_list) (void)hipFree(g->nc_fine_list);
        g->nc_fine_list = nullptr;
        DCR_TRY(dev_alloc(&g->nc_fine_list, need));
        g->nc_fine_cap = need;
    }
    if (!g->nc_queues) DCR_TRY(dev_alloc(&g->nc_queues, 2 * NC_QUEUES * NC_QUEUE_STRIDE));
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, incremental ? g->dirty : nullptr, (int32_t)g->n, 1, nullptr};
    hipLaunchKernelGGL(k_nc_clear, dim3(1), dim3(256), 0, g->stream, g->dres, g->nc_queues, 0, (unsigned *)nullptr, (int64_t)0);
    // (the sweep over every slot costs 11 us per 2.6 M slots, the rows of the flagged nodes a chain of five dependent reads, 14 us
    //  whatever the graph's size: S100k 0.203 / 0.208 ms per iteration sweep / rows, S1M 0.540 / 0.452 — by rows from 4 M slots;
    //  DCR_NC_FINE_SWEEP=1 / 0 forces one: A/B aid)
    const char *sweep_env = getenv("DCR_NC_FINE_SWEEP");
    const bool by_rows = sweep_env ? atoi(sweep_env) == 0 : g->cap_total >= 4000000;
    const int64_t blocks = (g->cap_total + 255) / 256;
    if (incremental && by_rows && g->touched)
        hipLaunchKernelGGL(k_nc_fine_list_rows, dim3((unsigned)(2 * g->num_cu)), dim3(256), 0, g->stream, vw, curv_type, g->curv, g->touched,
                           (int64_t)g->n, g->dres, g->nc_fine_list, g->nc_fine_cap, &g->dres->nc_count[0]);
    else if (blocks > 0)
        hipLaunchKernelGGL(k_nc_fine_list, dim3((unsigned)blocks), dim3(256), 0, g->stream, vw, curv_type, g->curv, g->nc_fine_list,
                           g->nc_fine_cap, &g->dres->nc_count[0]);
    // (max_deg_bound: the host's upper bound on every degree, raised by one per edit; owners have at most NC_MAXD neighbours)
    const int64_t dmax = g->max_deg_bound < NC_MAXD ? g->max_deg_bound : NC_MAXD;
#define DCR_NCF_LAUNCH(SLOTS_, PER_CU_)                                                                                          \
    hipLaunchKernelGGL((k_nc_fine_edges<SLOTS_, MODE>), dim3((unsigned)(g->num_cu * (PER_CU_))), dim3(64 * NCF_W), 0, g->stream, vw, \
                       g->nc_fine_list, &g->dres->nc_count[0], g->nc_fine_cap, curv_type, g->curv)
    if (dmax <= 1024 / 2 - 2) DCR_NCF_LAUNCH(1024, 8);
    else if (dmax <= 4096 / 2 - 2) DCR_NCF_LAUNCH(4096, 4);
    else DCR_NCF_LAUNCH(16384, 1);
#undef DCR_NCF_LAUNCH
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// Estimates of a full Balanced Forman pass, milliseconds on one MI355X (fitted on 31 graphs of four families, tools/probe_engine_choice.py;
// E edges, s = sum d^2 / n the mean size of a 2-hop neighbourhood, dmax the largest degree): the class kernels of this file (round 4's
// model, unchanged) and the edge-by-edge kernels — a workgroup per edge costs its chain of dependent reads plus what it streams.
double nc_class_full_ms(const dcr_graph *g) {
    const double n = (double)(g->n > 0 ? g->n : 1), E = (double)g->n_edges, s = g->sum_deg2 / n;
    const double dmax = (double)(g->max_deg_bound < 400 ? g->max_deg_bound : 400);
    return 0.127 + 0.438e-6 * E + 1.135e-9 * E * s + 0.201 * dmax / 400.0;
}
double nc_edges_full_ms(const dcr_graph *g) {
    const double n = (double)(g->n > 0 ? g->n : 1), E = (double)g->n_edges, s = g->sum_deg2 / n;
    return 0.012 + E * (5.0e-6 + 4.2e-9 * s);
}

int launch_curvature_pass_nc(dcr_graph *g, int curv_type, bool incremental) {
    const char *fine_env = getenv("DCR_NC_FINE");   // (read per call: the tests run both routes in one process)
    const bool fine_on = !(fine_env && atoi(fine_env) == 0);
    if (incremental && fine_on && g->pending_edits <= DIRTY_EDITS && !getenv("DCR_NC_TRACE")) {
        if (curv_type == DCR_CURV_BFC) return run_nc_fine<MODE_BFC>(g, curv_type, true);
        return run_nc_fine<MODE_TRI>(g, curv_type, true);
    }
    // A FULL pass of a small graph, edge by edge too: the class kernels are launch-bound there (plans, four persistent grids and
    // their joins: 0.17-0.4 ms whatever the graph holds), a workgroup per edge is not — Cora's size (5 k edges) 0.19 -> 0.04 ms,
    // 25 k edges 0.31 -> 0.17, break-even near 50 k edges (profiles/r05_engine_choice.txt).  Taken when its estimate is the
    // lowest (nc_edges_full_ms against the two models of h2_can_take; automatic engine choice only: DCR_PASS=nc keeps the class
    // kernels); DCR_NC_FINE_FULL=<slots> forces it for graphs of at most that many adjacency slots (A/B aid).
    bool full_edges = false;
    if (!incremental && fine_on && !getenv("DCR_NC_TRACE")) {
        const char *full_env = getenv("DCR_NC_FINE_FULL");
        if (full_env) full_edges = g->cap_total <= atoll(full_env);
        else full_edges = g->pass_impl == 0 && nc_edges_full_ms(g) < nc_class_full_ms(g);
    }
    if (full_edges) {
        if (curv_type == DCR_CURV_BFC) return run_nc_fine<MODE_BFC>(g, curv_type, false);
        return run_nc_fine<MODE_TRI>(g, curv_type, false);
    }
    if (curv_type == DCR_CURV_BFC) return run_nc<MODE_BFC>(g, curv_type, incremental);
    return run_nc<MODE_TRI>(g, curv_type, incremental);
}

}  // namespace dcr
