// Two-hop curvature pass of libdcr_hip.so: full Balanced Forman passes from row u of A·A instead of per-edge
// neighbourhood streaming (round 3: rebuilt around Bloom bitmaps; the round-2 hash-map version of this file was slower
// than the node-centric kernels and is gone).
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
// term M_u(w) does not depend on v: it is row u of A·A, the number of times w is met when the rows of the neighbours of
// u are read once.  The node-centric kernels find it by intersecting N(w) with N(u) for every (edge, w): 1.05 G adjacency
// entries per pass on the 100k-node bench graph; here every node reads its neighbours' rows: Σ d² = 0.117 G entries.
//
// 97.6 % of the 2-hop keys of a node occur once (c = 0: nothing to count) and only 5 % of the entries belong to keys
// that repeat, so exact state is kept for those only:
//   A. first sweep of the neighbours' rows: test-and-set one bit per entry in a Bloom bitmap B1 (LDS, ~16 bits per
//      entry); an entry that finds its bit set is a repeat (or, rarely, a collision) and sets the bit of a second,
//      smaller bitmap B2.  No table, no queue, one LDS atomic per entry;
//   B. second sweep: an entry whose B2 bit is set — every occurrence of every repeated key, first ones included, plus a
//      few collisions — goes to an exact table EX of (key → rows that hold it); the members of N(u) are seeded into
//      B1, B2 and EX beforehand, flagged, so a hit on one of them is a triangle (bfc_naive.py:25);
//   C. with EX complete, c(w) for the occurrence of w in row i is |rows(w) \ {i} \ rows adjacent to row i|.
// Nodes of at most 64 neighbours (97 % of the bench graph's nodes, 85 % of its entries) are taken by one wave each:
// rows(w) is a 64-bit mask, the rows adjacent to row i (the triangle partners of edge {u, v_i}) another, the B-sweep
// leaves a short list of candidate occurrences and step C is one pass over that list — no third sweep, no edge-set
// probes.  Larger nodes are taken by a workgroup with a counting table (key → occurrences) and a third sweep for the
// per-row statistics; the triangle term is subtracted per edge with triangles AND positive counts by probing an edge
// hash set in device memory; hubs whose repeated keys exceed the table are split by key hash into partitions (units of
// their own, results met by integer atomics).  A node whose tables fill up (dense neighbourhoods: sizes are typical,
// not worst case) is put on a retry list and redone by the largest class with worst-case partitions; only if that fails
// too does the whole pass fall back to the node-centric kernels.
// Each node writes, per adjacency slot u->v, {|sq| on v's side, max count, T, reverse slot}; a final kernel joins the two
// records of an edge and evaluates the float64 closing expression in the reference's order (bfc_naive.py:31-40).
// Bounds: HBM / L2 row streaming (2-3 x 4 B x Σ d² per pass) and LDS atomics; no MFMA (integer set counting).
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dcr_bfc_common.h"

namespace dcr {

constexpr unsigned H2_EMPTY = 0xFFFFFFFFu;
constexpr unsigned H2_NBR = 0x80000000u;  // on a key of the small classes' EX: member of N(u) (node ids stay below 2^30)
constexpr int H2_MAXDEG = 5000;           // flagged neighbours live in every partition's table (5,500 keys in the largest class:
                                          // a hub near the limit is split into many partitions, one workgroup each)
constexpr int H2_CLASSES = 5;             // 0-2: one wave per node (<= 64 neighbours), 3-4: one workgroup per unit
constexpr int H2_WB = 4;                  // weight buckets per class (units are laid out heaviest bucket first)
// third word of an edge's record: the triangle count with the degree of the row's node above it (both below 2^16: H2_MAXDEG).
// The closing kernel reads the OTHER endpoint's degree out of the partner record it fetches anyway, instead of a random read of
// that node's row header: one random read per edge is left of the four it made in round 3.
__device__ inline unsigned h2_rec_t(int T, int deg) { return (unsigned)T | ((unsigned)deg << 16); }
static_assert(H2_MAXDEG < 65536, "record packing");
constexpr int H2_SMALL_DEG = 64;
#ifndef H2_Q
#define H2_Q 2                            // 16-byte pieces per lane in flight in the streaming loops
#endif
#ifndef H2_QL
#define H2_QL 2                           // the same for the split class (16 waves per unit, a few pieces per lane and sweep)
#endif
constexpr int H2_QCAP = 128;              // queued exact-path items per wave (worked off when fewer than 64 slots are left)
__host__ __device__ constexpr int h2_wpb(int c) { return c == 2 ? 2 : 4; }  // waves (= nodes in flight) per workgroup of the wave classes

// per class: log2 of the bits of B1; largest weight W = Σ neighbour degrees (the entries of one sweep)
__host__ __device__ constexpr int h2_l1(int c) { return c == 0 ? 14 : c == 1 ? 15 : c == 2 ? 16 : c == 3 ? 16 : 17; }
__host__ __device__ constexpr int h2_maxw(int c) { return c == 0 ? 1024 : c == 1 ? 2048 : c == 2 ? 4096 : c == 3 ? 8192 : INT_MAX; }
__host__ __device__ constexpr int h2_exs(int c) { return c == 0 ? 128 : c == 1 ? 256 : c == 2 ? 512 : c == 3 ? 4096 : 8192; }
__host__ __device__ constexpr int h2_clcap(int c) { return c == 0 ? 192 : c == 1 ? 384 : 768; }
__host__ __device__ constexpr int h2_waves(int c) { return c == 3 ? 4 : 16; }
__host__ __device__ constexpr int h2_keycap(int c) { return c == 3 ? 2800 : 5500; }  // keys a block table takes (4-slot buckets)
constexpr int H2_PLCAP = 192;              // triangle partners of one batch of rows kept in LDS (more: their rows are read again)
constexpr int H2_WALK = 32;                // longest probe sequence of the exact tables

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

// ---- hashing: Fibonacci hashing of the low 24 bits with the full-rate 24-bit multiply (see dcr_bfc_nc.hip) ----------
__device__ inline unsigned h2_mul24(unsigned key) {
    unsigned prod;
    asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(prod) : "v"(key));
    return prod;
}
__device__ inline unsigned h2_mul24b(unsigned key) {  // an independent multiplier for the exact tables
    unsigned prod;
    asm("v_mul_u32_u24 %0, 0x85ebcb, %1" : "=v"(prod) : "v"(key));
    return prod;
}
template <int L1>
__device__ inline unsigned h2_bit(unsigned key) {
    return (h2_mul24(key) >> (24 - L1)) & ((1u << L1) - 1u);
}
// which partition of a split node a key belongs to: a hash independent of the others (NOT a multiplier whose leading digits
// equal those of h2_mul24b: the keys of one partition then share the leading bits of their bucket index and fill 1 / nparts
// of the table)
__device__ inline int h2_part(unsigned key, int nparts) {
    return (int)((((key * 0xC2B2AE35u) >> 16) * (unsigned)nparts) >> 16);
}

// ---- the bitmaps ------------------------------------------------------------------------------------------------------
// B1: TWO bits per 2-hop entry in ONE 32-bit word — the word and the first bit from the first hash, the second bit from an
// independent one (a blocked Bloom filter); "seen" = both were set; B2 (a quarter of the bits, indexed by the first hash):
// seen again.  (Round 4: with one bit per entry in 16, one key in sixteen that occurs ONCE found its bit set by another key
// and went down the exact path — half of all exact-path items, and the exact tables of the heaviest nodes of every wave class
// ran at 85 % load; with two bits a singleton needs both taken: 1-2 %.  Both bits go in ONE atomic: two atomics were tried
// first and were WRONG — two lanes holding the same key in one instruction are served in an order of the hardware's choosing,
// not necessarily the same for the second atomic, and then neither lane sees both bits set.)
#ifndef H2_TWO_HASH
#define H2_TWO_HASH 1
#endif
__device__ inline unsigned h2_mul24c(unsigned key) {  // a third multiplier: independent of the bitmaps' first hash and of the tables'
    unsigned prod;
    asm("v_mul_u32_u24 %0, 0xb55a4f, %1" : "=v"(prod) : "v"(key));
    return prod;
}
// the entry's bits inside word (bit index >> 5) of B1
__device__ inline unsigned h2_word_mask(unsigned key, unsigned b) {
    unsigned m = 1u << (b & 31u);
    if (H2_TWO_HASH) m |= 1u << ((h2_mul24c(key) >> 19) & 31u);
    return m;
}
template <int L1>
__device__ inline void h2_seed(unsigned *b1, unsigned *b2, unsigned key) {  // a member of N(u): every occurrence is exact
    const unsigned b = h2_bit<L1>(key);
    atomicOr(&b1[b >> 5], h2_word_mask(key, b));
    const unsigned c = b >> 2;
    atomicOr(&b2[c >> 5], 1u << (c & 31u));
}
__device__ inline bool h2_again(const unsigned *b2, unsigned b) {
    const unsigned c = b >> 2;
    return (b2[c >> 5] >> (c & 31u)) & 1u;
}
// The four entries of a piece at once, branch-free up to the rare second step: entry by entry, every returning atomic was
// followed by a wait and a branch (20 round trips of LDS latency per sweep of a small node).  `valid`: 4-bit mask.
template <int L1>
__device__ inline void h2_mark4(unsigned *b1, unsigned *b2, const unsigned kk[4], unsigned valid) {
    unsigned b[4], m[4], old[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        b[jj] = h2_bit<L1>(kk[jj]);
        m[jj] = ((valid >> jj) & 1u) ? h2_word_mask(kk[jj], b[jj]) : 0u;  // (no bit for an entry that does not count: the atomic changes nothing)
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) old[jj] = atomicOr(&b1[b[jj] >> 5], m[jj]);
    unsigned any = 0u;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        old[jj] = (m[jj] != 0u && (old[jj] & m[jj]) == m[jj]) ? 1u : 0u;
        any |= old[jj];
    }
    if (any) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (old[jj]) {
                const unsigned c = b[jj] >> 2;
                atomicOr(&b2[c >> 5], 1u << (c & 31u));
            }
    }
}
// which of the four entries have their B2 bit set (four independent reads)
template <int L1>
__device__ inline unsigned h2_again4(const unsigned *b2, const unsigned kk[4], unsigned valid) {
    unsigned c[4], wd[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        c[jj] = h2_bit<L1>(kk[jj]) >> 2;
        wd[jj] = b2[c[jj] >> 5];
    }
    unsigned f = 0u;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) f |= ((wd[jj] >> (c[jj] & 31u)) & 1u) << jj;
    return f & valid;
}
__device__ inline unsigned h2_eq4(const unsigned kk[4], unsigned u) {
    return (kk[0] == u ? 1u : 0u) | (kk[1] == u ? 2u : 0u) | (kk[2] == u ? 4u : 0u) | (kk[3] == u ? 8u : 0u);
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

// The rows of a batch (descriptors and piece prefix sums in LDS) as one flat list of aligned 16-byte pieces shared evenly
// by the 64 lanes.  body(piece, valid-entry mask, row, slot of the piece's first entry) is called by ALL lanes, converged
// (lanes without a piece pass mask 0, row -1).
template <typename Body>
__device__ inline void h2_for_pieces(const int32_t *col, const int2 *desc, const int *poff, int poff_lane, int P, Body body) {
    const int lane = threadIdx.x & 63;
    for (int j0 = 0; j0 < P; j0 += 64 * H2_Q) {
        int4 w[H2_Q];
        int rr[H2_Q], aa[H2_Q];
        unsigned vm[H2_Q];
#pragma unroll
        for (int q = 0; q < H2_Q; ++q) {
            const int j = j0 + 64 * q + lane;
            rr[q] = -1;
            aa[q] = 0;
            vm[q] = 0u;
            w[q] = make_int4(0, 0, 0, 0);
            const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
            if (jf >= P) continue;  // uniform
            const int r = h2_piece_row(poff, poff_lane, j < P ? j : jl, jf, jl);
            if (j < P) {
                const int2 d = desc[r];
                const int a = (d.x & ~3) + 4 * (j - poff[r]);
                w[q] = load_piece(col, a);
                rr[q] = r;
                aa[q] = a;
                vm[q] = h2_piece_mask(a, d.x, d.x + d.y);
            }
        }
#pragma unroll
        for (int q = 0; q < H2_Q; ++q) {
            if (j0 + 64 * q >= P) continue;  // uniform
            body(w[q], vm[q], rr[q], aa[q]);
        }
    }
}

// The same with the exact-path queue: flags(piece, mask, row, slot) -> which of the four entries go to the queue;
// push(index, key, row) stores one; drain() works the queue off and resets qn.  The drain is called from ONE place per
// iteration (inlined at every entry position it was 4 x H2_Q copies of the table walks per sweep, and the kernels
// outgrew the instruction cache: 135-190 KB each).
template <int Q, typename Flags, typename Push, typename Drain>
__device__ inline void h2_for_pieces_queued(const int32_t *col, const int2 *desc, const int *poff, int poff_lane, int P, int &qn,
                                            Flags flags, Push push, Drain drain) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int j0 = 0; j0 < P; j0 += 64 * Q) {
        int4 w[Q];
        int rr[Q];
        unsigned fl[Q];
        unsigned anyf = 0u;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int j = j0 + 64 * q + lane;
            rr[q] = -1;
            fl[q] = 0u;
            w[q] = make_int4(0, 0, 0, 0);
            const int jf = j0 + 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
            if (jf >= P) continue;  // uniform
            const int r = h2_piece_row(poff, poff_lane, j < P ? j : jl, jf, jl);
            int a = 0;
            unsigned vm = 0u;
            if (j < P) {
                const int2 d = desc[r];
                a = (d.x & ~3) + 4 * (j - poff[r]);
                w[q] = load_piece(col, a);
                rr[q] = r;
                vm = h2_piece_mask(a, d.x, d.x + d.y);
            }
            fl[q] = flags(w[q], vm, rr[q], a);
            anyf |= fl[q];
        }
        if (__ballot(anyf != 0u) == 0ull) continue;  // uniform
#pragma unroll 1
        while (true) {
            bool stop = false;  // uniform: the queue is full, the rest waits for the drain
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const bool p = (fl[q] >> jj) & 1u;
                    const unsigned long long m = __ballot(p);
                    if (m == 0ull) continue;  // uniform
                    if (stop || qn + __popcll(m) > H2_QCAP) {
                        stop = true;
                        continue;
                    }
                    if (p) {
                        push(qn + __popcll(m & below), kk[jj], rr[q]);
                        fl[q] &= ~(1u << jj);
                    }
                    qn += __popcll(m);
                }
            }
            if (!stop) break;
            drain();
        }
    }
}

// piece counts of the rows held by the lanes -> exclusive prefix in `excl`, total returned
__device__ inline int h2_prefix(int np, int &excl) {
    const int lane = threadIdx.x & 63;
    int incl = np;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    excl = incl - np;
    return __shfl(incl, 63);
}

// ---- the retry list: nodes whose tables filled up in their class -------------------------------------------------------
// partitions of a node in the largest class: `safe` sizes them for the worst case (every key repeats: W / 2 keys, a
// quarter of slack for the partition hash), else for what power-law graphs need (W / 4)
__host__ __device__ inline int h2_parts_for(int d, int64_t W, bool safe) {
    const int64_t room = h2_keycap(4) - d;  // the flagged neighbours live in every partition's table
    if (room <= 0) return 65536;
    const int64_t keys = safe ? W / 2 + W / 8 + 64 : W / 3 + 64;
    int64_t p = (keys + room - 1) / room;
    // ... and B1 wants 8 (16) bits per entry of the partition, or its collisions fill B2 and the table with keys that occur once
    const int64_t per = safe ? (1 << h2_l1(4)) / 16 : (1 << h2_l1(4)) / 8;
    if (p < (W + per - 1) / per) p = (W + per - 1) / per;
    if (p < 1) p = 1;
    if (p > 65535) p = 65536;
    return (int)p;
}
struct H2Retry {
    int4 *units;
    int64_t cap;
    int32_t *weight;  // bit 31: the node is on the list already (several partitions of a split node may fail)
    DevResult *res;
};
__device__ inline void h2_retry_push(const H2Retry rt, int u, int d, int cls) {  // one lane calls this
    atomicAdd(&rt.res->h2_failed[cls], 1);
    const unsigned old = atomicOr(reinterpret_cast<unsigned *>(&rt.weight[u]), 0x80000000u);
    if (old & 0x80000000u) return;
    const int nparts = h2_parts_for(d, (int64_t)old, true);
    const int first = atomicAdd(&rt.res->h2_retry, nparts);
    if (nparts > 65535 || first < 0 || (int64_t)first + nparts > rt.cap) {
        rt.res->h2_status = 1;  // not even the retry list takes it: the pass is redone by the node-centric kernels
        return;
    }
    for (int j = 0; j < nparts; ++j) rt.units[first + j] = make_int4(u, (nparts << 16) | j, 0, 0);
}

#ifdef H2_UNIT_TIMES  // diagnostic build (tools/build_variant.sh ut -DH2_UNIT_TIMES): start and duration of every block-class unit
constexpr unsigned H2_UT_CAP = 16384;
__device__ int4 h2_ut[2 * H2_UT_CAP];  // {node, class | partitions << 16 | partition, s_memtime ticks, start tick / 16}, {ticks: clear + seed, sweep A, sweep B, third step}
__device__ int4 h2_ut2[H2_UT_CAP];     // wave 0 of the unit: {ticks in the second sweep's drains, drains, items drained, -}
__device__ unsigned h2_ut_n;
__device__ long long h2_ut_t0;
__global__ void k_h2_ut_mark() { h2_ut_t0 = (long long)__builtin_amdgcn_s_memtime(); h2_ut_n = 0u; }
#endif
#ifdef H2_PROF  // diagnostic build (tools/build_variant.sh prof -DH2_PROF): wave-cycles per section of the wave classes
__device__ unsigned long long h2_prof[32];
#define H2_STAMP(i)                                                      \
    {                                                                    \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime();  \
        if ((threadIdx.x & 63) == 0) s->prof[i] += (unsigned long long)(now_ - t_prof);              \
        t_prof = now_;                                                   \
    }
#else
#define H2_STAMP(i) {}
#endif
// =====================================================================================================================
// wave classes: a node of at most 64 neighbours, by one wave
// =====================================================================================================================
// The LDS of one wave is used in two phases that never overlap (round 4: the arrays of both phases side by side held
// occupancy at 6 / 12 / 20 waves per CU): up to the end of sweep A the first-level bitmap and the row descriptors the
// pieces are addressed with (the pieces themselves are in registers from then on), from sweep B on the exact table, the
// row masks, the queue and the candidate list.  Only the second-level bitmap lives through both.
template <int L1, int EXS, int CLCAP>
struct __attribute__((aligned(16))) H2Small {
    unsigned b2[(1 << (L1 - 2)) / 32];
    union {
        struct {
            unsigned b1[(1 << L1) / 32];   // (directly behind b2: the two are cleared as one range)
            int2 desc[64];
            int poff[66];
        } a;
        struct {
            unsigned exkey[EXS];            // candidate key, or a member of N(u) | H2_NBR
            unsigned exlo[EXS], exhi[EXS];  // candidate: the rows that hold it (64-bit mask); member of N(u): its position in row u
            unsigned adjlo[64], adjhi[64];  // per row i: the rows adjacent to it = triangle partners of edge {u, v_i}
            int pos[64], mx[64], rev[64];
            unsigned qk[H2_QCAP];
            unsigned short cls[CLCAP];     // candidate occurrences: EX slot ...
            unsigned char qr[H2_QCAP];
            unsigned char clr[CLCAP];      // ... and row
        } b;
    };
#ifdef H2_PROF
    unsigned long long prof[8];    // per wave, flushed when the kernel ends
#endif
};

template <int EXS>
__device__ inline unsigned h2s_home(unsigned key) {
    constexpr int BITS = __builtin_ctz(EXS);
    return (h2_mul24b(key) >> (24 - BITS)) & (unsigned)(EXS - 1);
}
// slot of w (a plain id), inserting it if absent; -1: table full
template <int EXS>
__device__ inline int h2s_find_or_insert(unsigned *exkey, unsigned w) {
    unsigned s = h2s_home<EXS>(w);
#pragma unroll 1
    for (int walk = 0; walk < EXS; ++walk) {
        const unsigned e = exkey[s];
        if ((e & ~H2_NBR) == w && e != H2_EMPTY) return (int)s;
        if (e == H2_EMPTY) {
            const unsigned old = atomicCAS(&exkey[s], H2_EMPTY, w);
            if (old == H2_EMPTY || (old & ~H2_NBR) == w) return (int)s;
        }
        s = (s + 1) & (unsigned)(EXS - 1);
    }
    return -1;
}
template <int EXS>
__device__ inline int h2s_insert_nbr(unsigned *exkey, unsigned k) {  // the members of N(u) are distinct
    unsigned s = h2s_home<EXS>(k);
#pragma unroll 1
    for (int walk = 0; walk < EXS; ++walk) {
        if (atomicCAS(&exkey[s], H2_EMPTY, k | H2_NBR) == H2_EMPTY) return (int)s;
        s = (s + 1) & (unsigned)(EXS - 1);
    }
    return -1;
}

// k, rk: the lane's member of N(u) and its row (loaded ahead by the caller); NP: 64-piece steps a node of this class can
// have — all pieces of the node stay in registers and serve both sweeps; mid(): called once between the sweeps (the
// caller's prefetch of the next node)
template <int L1, int EXS, int CLCAP, int NP, typename Mid>
__device__ inline void h2s_node(const View &g, int u, int2 ru, int k, int2 rk, H2Small<L1, EXS, CLCAP> *s, uint4 *rec,
                                const H2Retry rt, Mid mid) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
#ifdef H2_PROF
    long long t_prof = (long long)__builtin_amdgcn_s_memtime();
#endif
    {   // clear: both bitmaps to zero (the second phase's arrays are set up between the sweeps)
        uint4 *z = reinterpret_cast<uint4 *>(s->b2);
        constexpr int NZ = ((1 << L1) / 32 + (1 << (L1 - 2)) / 32) / 4;  // b2, then b1 directly behind it
        using Lds = H2Small<L1, EXS, CLCAP>;
        static_assert(__builtin_offsetof(Lds, a) == sizeof(unsigned) * ((1 << (L1 - 2)) / 32), "b1 directly behind b2");
        for (int i = lane; i < NZ; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    h2_wave_sync();
    H2_STAMP(0)
    // the members of N(u): seeded into both bitmaps (into the exact table between the sweeps)
    bool full = false;
    if (lane >= ru.y || k < 0 || k >= g.n || k == u) {
        k = -1;
        rk = make_int2(0, 0);
    } else {
        if (!row_ok(g, rk, 32, k, u)) rk = make_int2(0, 0);
        h2_seed<L1>(s->a.b1, s->b2, (unsigned)k);
    }
    const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
    int poff_lane;
    const int P = h2_prefix(np, poff_lane);
    s->a.desc[lane] = rk;
    s->a.poff[lane] = poff_lane;
    if (lane == 0) s->a.poff[64] = P;
    h2_wave_sync();
    H2_STAMP(1)
    if (P > 64 * NP) full = true;  // (cannot happen: the class bounds the weight; such a node would be redone elsewhere)
    // every piece of the node is requested at once and kept: meta = valid-entry mask | row << 4 | first slot of the piece << 12
    int4 w[NP];
    unsigned long long meta[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int j = 64 * q + lane;
        w[q] = make_int4(0, 0, 0, 0);
        meta[q] = 0ull;
        const int jf = 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
        if (jf >= P) continue;  // uniform
        const int r = h2_piece_row(s->a.poff, poff_lane, j < P ? j : jl, jf, jl);
        if (j < P) {
            const int2 d = s->a.desc[r];
            const int a = (d.x & ~3) + 4 * (j - s->a.poff[r]);
            w[q] = load_piece(g.col, a);
            meta[q] = (unsigned long long)h2_piece_mask(a, d.x, d.x + d.y) | ((unsigned long long)r << 4) | ((unsigned long long)(unsigned)a << 12);
        }
    }
    H2_STAMP(2)
    // sweep A
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (64 * q >= P) continue;  // uniform
        const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
        const unsigned vm = (unsigned)meta[q] & 0xFu;
        h2_mark4<L1>(s->a.b1, s->b2, kk, vm & ~h2_eq4(kk, (unsigned)u));
    }
    mid();
    h2_wave_sync();  // sweep A is through: b1 and the descriptors are dead, their space becomes the second phase's arrays
    {
        uint4 *k4 = reinterpret_cast<uint4 *>(s->b.exkey);
        for (int i = lane; i < EXS / 4; i += 64) k4[i] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
        uint4 *m4 = reinterpret_cast<uint4 *>(s->b.exlo);  // exlo and exhi are adjacent
        for (int i = lane; i < 2 * EXS / 4; i += 64) m4[i] = make_uint4(0u, 0u, 0u, 0u);
        s->b.adjlo[lane] = 0u;
        s->b.adjhi[lane] = 0u;
        s->b.pos[lane] = 0;
        s->b.mx[lane] = 0;
        s->b.rev[lane] = -1;
    }
    h2_wave_sync();
    if (k >= 0) {  // the members of N(u) into the exact table, flagged, with their position in row u
        const int slot = h2s_insert_nbr<EXS>(s->b.exkey, (unsigned)k);
        if (slot < 0) full = true;
        else s->b.exlo[slot] = (unsigned)lane;
    }
    h2_wave_sync();
    H2_STAMP(3)
    // sweep B: entries whose B2 bit is set are queued and settled against the exact table 64 at a time
    int qn = 0, cln = 0;  // uniform
    auto drain = [&]() {
        h2_wave_sync();
        for (int base = 0; base < qn; base += 64) {
            const int i = base + lane;
            bool cand = false;
            int slot = -1, row = 0;
            if (i < qn) {
                const unsigned w = s->b.qk[i];
                row = s->b.qr[i];
                slot = h2s_find_or_insert<EXS>(s->b.exkey, w);
                if (slot < 0) {
                    full = true;
                } else if (s->b.exkey[slot] & H2_NBR) {  // a member of N(u) in row `row`: rows `row` and exlo[slot] are adjacent
                    const unsigned x = s->b.exlo[slot];
                    atomicOr(x < 32u ? &s->b.adjlo[row] : &s->b.adjhi[row], 1u << (x & 31u));
                } else {
                    atomicOr(row < 32 ? &s->b.exlo[slot] : &s->b.exhi[slot], 1u << (row & 31));
                    cand = true;
                }
            }
            const unsigned long long m = __ballot(cand);
            if (cand) {
                const int idx = cln + __popcll(m & below);
                if (idx < CLCAP) {
                    s->b.cls[idx] = (unsigned short)slot;
                    s->b.clr[idx] = (unsigned char)row;
                } else {
                    full = true;
                }
            }
            cln += __popcll(m);
        }
        h2_wave_sync();
        qn = 0;
    };
    unsigned fl[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        fl[q] = 0u;
        if (64 * q >= P) continue;  // uniform
        const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
        const unsigned vm = (unsigned)meta[q] & 0xFu;
        const unsigned isu = h2_eq4(kk, (unsigned)u) & vm;
        if (isu) s->b.rev[(int)((meta[q] >> 4) & 0xFFull)] = (int)(meta[q] >> 12) + __ffs((int)isu) - 1;  // where u sits in that row
        fl[q] = h2_again4<L1>(s->b2, kk, vm & ~isu);
    }
    // the flagged entries are queued; the drain has ONE call site (inlined per entry position the kernels outgrew the
    // instruction cache)
#pragma unroll 1
    while (true) {
        bool stop = false;  // uniform: the queue is full, the rest waits for the drain
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (64 * q >= P) continue;  // uniform
            const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool p = (fl[q] >> jj) & 1u;
                const unsigned long long m = __ballot(p);
                if (m == 0ull) continue;  // uniform
                if (stop || qn + __popcll(m) > H2_QCAP) {
                    stop = true;
                    continue;
                }
                if (p) {
                    const int idx = qn + __popcll(m & below);
                    s->b.qk[idx] = kk[jj];
                    s->b.qr[idx] = (unsigned char)((meta[q] >> 4) & 0xFFull);
                    fl[q] &= ~(1u << jj);
                }
                qn += __popcll(m);
            }
        }
        if (qn > 0) drain();
        if (!stop) break;
    }
    h2_wave_sync();
    H2_STAMP(4)
    if (__ballot(full) != 0ull) {  // a table or the list filled up: nothing is published, the node is redone elsewhere
        if (lane == 0) h2_retry_push(rt, u, ru.y, L1 == 14 ? 0 : L1 == 15 ? 1 : 2);
        h2_wave_sync();
        return;
    }
    // step C: c(w) for the occurrence of w in row i = the rows that hold w, less row i, less the rows adjacent to row i
    const int cl_n = cln < CLCAP ? cln : CLCAP;
    for (int e = lane; e < cl_n; e += 64) {
        const int slot = s->b.cls[e], row = s->b.clr[e];
        const int c = __popc(s->b.exlo[slot] & ~s->b.adjlo[row]) + __popc(s->b.exhi[slot] & ~s->b.adjhi[row]) - 1;
        if (c > 0) {
            atomicAdd(&s->b.pos[row], 1);
            atomicMax(&s->b.mx[row], c);
        }
    }
    h2_wave_sync();
    if (k >= 0) {
        const int rev = s->b.rev[lane];
        if (rev < 0 || rev >= g.cap_total) {
            row_ok(g, make_int2(-1, rev), 33, u, k);  // adjacency not symmetric: report, never publish
        } else {
            const int T = __popc(s->b.adjlo[lane]) + __popc(s->b.adjhi[lane]);
            rec[(int64_t)ru.x + lane] = make_uint4((unsigned)s->b.pos[lane], (unsigned)s->b.mx[lane], h2_rec_t(T, ru.y), (unsigned)rev);
        }
    }
    h2_wave_sync();  // the arrays are rewritten by the next node
    H2_STAMP(5)
}

// Units are taken grid-stride from a list laid out heaviest first: every wave gets a similar mix.  A node is a chain of
// dependent device-memory reads (unit -> row of u -> rows of its members -> their pieces) with little to do in between,
// so the chain of the NEXT node is started while the current one is worked on: its unit two nodes ahead, its row at the
// start of this node, the rows' descriptors between the sweeps.
// waves per SIMD the three wave-class kernels are compiled for (registers; their LDS allows as many since the two phases
// share it).  Measured one class at a time (us, S100k; profiles/r04_occupancy_ab.txt): lightest class 5 / 6 / 7 / 8 waves:
// 155 / 153 / 144 / 162 (near its vector-issue limit: more waves buy little); middle 3 / 4 / 5: 158 / 150 / 147;
// heaviest 2 / 3: 142 / 106 (latency-bound: a third wave per SIMD is worth its 40 spilled registers).
#ifndef H2_OCC0
#define H2_OCC0 7
#endif
#ifndef H2_OCC1
#define H2_OCC1 4
#endif
#ifndef H2_OCC2
#define H2_OCC2 3
#endif
template <int L1, int EXS, int CLCAP, int WPB, int NP>
__global__ void __launch_bounds__(64 * WPB, (L1 == 14 ? H2_OCC0 : L1 == 15 ? H2_OCC1 : H2_OCC2)) k_h2_small(View g, const int4 *units, const int32_t *count, int64_t unit_cap,
                                                       uint4 *rec, H2Retry rt) {
    __shared__ H2Small<L1, EXS, CLCAP> sm[WPB];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = *count;
    if (total < 0 || total > unit_cap) {
        row_ok(g, make_int2(-1, total), 34, 0, 0);
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * WPB;
    int64_t it = (int64_t)blockIdx.x * WPB + wid;
#ifdef H2_PROF
    if (lane < 8) sm[wid].prof[lane] = 0ull;
#endif
    const int4 none = make_int4(-1, 0, 0, 0);
    auto unit_ok = [&](const int4 un) {
        return un.x >= 0 && un.x < g.n && un.w > 0 && un.w <= H2_SMALL_DEG && un.z >= 0 && (int64_t)un.z + un.w <= g.cap_total;
    };
    // prologue: this node's unit, row and row descriptors; the next node's unit and row
    int4 un = it < total ? units[it] : none;
    int4 un1 = it + stride < total ? units[it + stride] : none;
    int k = (unit_ok(un) && lane < un.w) ? g.col[un.z + lane] : -1;
    int2 rk = (k >= 0 && k < g.n) ? g.rowinfo[k] : make_int2(0, 0);
    int k1 = (unit_ok(un1) && lane < un1.w) ? g.col[un1.z + lane] : -1;
    for (; it < total; it += stride) {
        const int4 un2 = it + 2 * stride < total ? units[it + 2 * stride] : none;  // two nodes ahead
        int2 rk1 = make_int2(0, 0);
        if (unit_ok(un)) {
            // (the plan read the row of u in this same pass: {start, degree} travel with the unit)
            h2s_node<L1, EXS, CLCAP, NP>(g, un.x, make_int2(un.z, un.w), k, rk, &sm[wid], rec, rt, [&]() {
                rk1 = (k1 >= 0 && k1 < g.n) ? g.rowinfo[k1] : make_int2(0, 0);
            });
        } else {
            if (un.x != -1 || it < total) row_ok(g, make_int2(-1, un.x), 35, (int)it, total);
            rk1 = (k1 >= 0 && k1 < g.n) ? g.rowinfo[k1] : make_int2(0, 0);
        }
        un = un1;
        k = k1;
        rk = rk1;
        un1 = un2;
        k1 = (unit_ok(un1) && lane < un1.w) ? g.col[un1.z + lane] : -1;
    }
#ifdef H2_PROF
    h2_wave_sync();
    if (lane < 8) atomicAdd(&h2_prof[lane], sm[wid].prof[lane]);
#endif
}

// =====================================================================================================================
// block classes: a unit (node, key partition) by a workgroup; counting table, third sweep, edge-set probes
// =====================================================================================================================
// ---- the table: open addressing over 4-slot buckets (one 16-byte LDS read settles almost every access) -------------
template <int CAP>
__device__ inline unsigned h2_bucket(unsigned key) {
    constexpr int BITS = __builtin_ctz(CAP / 4);
    return (h2_mul24b(key) >> (24 - BITS)) & ((1u << BITS) - 1);
}
__device__ inline int h2_match(const uint4 e, unsigned w) {
    return e.x == w ? 0 : e.y == w ? 1 : e.z == w ? 2 : e.w == w ? 3 : -1;
}
// slot of w; -1: absent
template <int CAP>
__device__ inline int h2_find(const unsigned *key, unsigned w) {
    const uint4 *tb = reinterpret_cast<const uint4 *>(key);
    unsigned b = h2_bucket<CAP>(w);
#pragma unroll 1
    for (int walk = 0; walk < H2_WALK; ++walk) {  // (insertion never walks further than this)
        const uint4 e = tb[b];
        const int pos = h2_match(e, w);
        if (pos >= 0) return (int)(b * 4) + pos;
        if (e.w == H2_EMPTY) return -1;  // slots of a bucket fill in order: a free last slot means the key never spilled
        b = (b + 1) & (CAP / 4 - 1);
    }
    return -1;
}
// slot of w, inserting it if absent; -1: the table is full (reported by the caller, never loops forever).
// A lane claims the first free slot it saw; slots never empty again, so filled slots always form a prefix of a bucket and a
// key is never stored twice (a racing lane with the same key meets it on its way up).
template <int CAP>
__device__ inline int h2_insert(unsigned *key, unsigned w) {
    const uint4 *tb = reinterpret_cast<const uint4 *>(key);
    unsigned b = h2_bucket<CAP>(w);
#pragma unroll 1
    for (int walk = 0; walk < H2_WALK; ++walk) {  // (a longer walk: too full, treated as full)
        const uint4 e = tb[b];
        const int pos = h2_match(e, w);
        if (pos >= 0) return (int)(b * 4) + pos;
        int ep = e.x == H2_EMPTY ? 0 : e.y == H2_EMPTY ? 1 : e.z == H2_EMPTY ? 2 : e.w == H2_EMPTY ? 3 : 4;
        for (; ep < 4; ++ep) {
            const unsigned old = atomicCAS(&key[b * 4 + ep], H2_EMPTY, w);
            if (old == H2_EMPTY || old == w) return (int)(b * 4) + ep;
        }
        b = (b + 1) & (CAP / 4 - 1);
    }
    return -1;
}
// per-slot state, two 16-bit halves per word: bit 15 = member of N(u), bits 0-14 = occurrences in the neighbours' rows
__device__ inline void h2_cnt_flag(unsigned *cnt, int s) { atomicOr(&cnt[s >> 1], 0x8000u << ((s & 1) * 16)); }
__device__ inline void h2_cnt_add(unsigned *cnt, int s) { atomicAdd(&cnt[s >> 1], 1u << ((s & 1) * 16)); }
__device__ inline unsigned h2_cnt_get(const unsigned *cnt, int s) { return (cnt[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu; }

struct H2Scratch {
    int2 desc[64];   // {start, length} of the rows of the current batch
    int poff[66];    // exclusive prefix of their piece counts; poff[64] = total
    int rowT[64], rowPos[64], rowMx[64], rowRev[64];  // third-sweep accumulators per row of the batch
    unsigned qw[H2_QCAP];       // queued exact-path work: key ...
    unsigned char qr[H2_QCAP];  // ... row of the batch ...
    unsigned char qb[H2_QCAP];  // ... and batch (the second sweep's queue runs across batches)
    unsigned plt[H2_PLCAP];       // third sweep: the members of N(u) met in the rows of the batch (triangle partners) ...
    unsigned char plr[H2_PLCAP];  // ... and the row each was met in
    int pln;                      // how many (counting past the capacity)
    unsigned char trank[64];      // rank of each row of the batch among the rows present (its task is tbase + rank)
#ifdef H2_UNIT_TIMES
    long long ut_drain;           // ticks this wave spent in the second sweep's drains (diagnostic)
    int ut_ndrain, ut_nitems;
#endif
#ifdef H2_PROF
    unsigned long long prof[16];
#endif
};

// the LDS arrays of a unit
struct H2Tab {
    unsigned *b1, *b2;
    unsigned *key;  // [EXS] 4-slot buckets
    unsigned *cnt;  // [EXS / 2] their state, 16 bits each
    int *full;      // set when the table fills up
#ifdef H2_UNIT_TIMES
    long long *ut;  // [4] phase boundaries of the current unit (thread 0)
#endif
};
#ifdef H2_UNIT_TIMES
#define H2_UT_PHASE(i) { if (threadIdx.x == 0) t.ut[i] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define H2_UT_PHASE(i) {}
#endif

// M_u(w) and whether w is a member of N(u); {1, false} for a key this unit keeps no exact state for
template <int L1, int EXS>
__device__ inline int h2_query(const H2Tab t, unsigned w, bool &nbr) {
    nbr = false;
    if (!h2_again(t.b2, h2_bit<L1>(w))) return 1;
    const int s = h2_find<EXS>(t.key, w);
    if (s < 0) return 1;
    const unsigned c16 = h2_cnt_get(t.cnt, s);
    nbr = (c16 & 0x8000u) != 0u;
    return (int)(c16 & 0x7FFFu);
}

// ---- the edge set: every undirected edge as one 64-bit key in an open-addressing table in device memory -------------------
// The triangle step asks it "is w adjacent to t?" a few times per edge with triangles.  Built once (one CAS per edge) and
// then KEPT across passes (round 4): the SDRF loop changes one or two edges between two passes, the edit kernels journal them
// (DevResult::edit_log) and k_h2_eset_apply brings the set up to date — an insertion, or a tombstone in place of a removed
// key (probes walk over tombstones; the bitmap in front of the table keeps the removed edge's bit: a false positive the
// table then answers).  Rebuilt when the journal overflowed, the table was re-sized, or tombstones have piled up.
constexpr unsigned long long H2_ESET_EMPTY = ~0ull;
constexpr unsigned long long H2_ESET_TOMB = ~0ull - 1ull;
__device__ inline unsigned long long h2_edge_key(int a, int b) {
    return a < b ? ((unsigned long long)(unsigned)a << 32) | (unsigned)b : ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
}
__device__ inline unsigned long long h2_eset_slot(unsigned long long key, int bits) {
    return (key * 0x9E3779B97F4A7C15ull) >> (64 - bits);
}
struct H2EdgeSet {
    unsigned long long *tab;
    int bits;
    unsigned *bloom;  // one bit per edge in a bitmap small enough for an XCD's L2: most probes end there
    int bloom_bits;
};
__device__ inline unsigned h2_bloom_bit(unsigned long long key, int bits) {
    return (unsigned)((key * 0xD6E8FEB86659FD93ull) >> (64 - bits));
}
__device__ inline bool h2_eset_has(const H2EdgeSet es, int a, int b) {
    const unsigned long long key = h2_edge_key(a, b), mask = (1ull << es.bits) - 1ull;
    unsigned long long h = h2_eset_slot(key, es.bits);
#pragma unroll 1
    for (unsigned long long walk = 0; walk <= mask; ++walk) {
        const unsigned long long k = es.tab[h];
        if (k == key) return true;
        if (k == H2_ESET_EMPTY) return false;
        h = (h + 1) & mask;
    }
    return false;
}
// the journaled edits, in order, by one thread (a handful per SDRF iteration)
__global__ void k_h2_eset_apply(H2EdgeSet es, DevResult *res, int32_t *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = res->edit_n;
    res->edit_n = 0;
    if (n < 0 || n > EDIT_LOG_CAP) {  // (the host counts the edit launches and rebuilds instead: cannot happen)
        *status = 1;
        return;
    }
    const unsigned long long mask = (1ull << es.bits) - 1ull;
    for (int e = 0; e < n; ++e) {
        const int op = res->edit_log[3 * e], u = res->edit_log[3 * e + 1], v = res->edit_log[3 * e + 2];
        const unsigned long long key = h2_edge_key(u, v);
        unsigned long long h = h2_eset_slot(key, es.bits), free_at = ~0ull;
        bool found = false;
        for (unsigned long long walk = 0; walk <= mask; ++walk) {
            const unsigned long long k = es.tab[h];
            if (k == key) {
                found = true;
                break;
            }
            if (k == H2_ESET_TOMB && free_at == ~0ull) free_at = h;
            if (k == H2_ESET_EMPTY) {
                if (free_at == ~0ull) free_at = h;
                break;
            }
            h = (h + 1) & mask;
        }
        if (op > 0) {
            if (!found) {
                if (free_at == ~0ull) {
                    *status = 1;
                    return;
                }
                es.tab[free_at] = key;
            }
            const unsigned bb = h2_bloom_bit(key, es.bloom_bits);
            es.bloom[bb >> 5] |= 1u << (bb & 31u);
        } else if (found) {
            es.tab[h] = H2_ESET_TOMB;
        }
    }
}
__global__ void __launch_bounds__(256) k_h2_eset_build(View g, H2EdgeSet es, int32_t *status, DevResult *res) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s == 0) res->edit_n = 0;  // the set is built from the rows as they are: every journaled edit is in it
    if (s >= g.cap_total) return;
    const int u = g.slot_row[s];
    if (u < 0 || u >= g.n) return;
    const int2 ru = g.rowinfo[u];
    if (s < ru.x || (int)(s - ru.x) >= ru.y) return;
    const int v = g.col[s];
    if (v <= u || v >= g.n) return;
    const unsigned long long key = h2_edge_key(u, v), mask = (1ull << es.bits) - 1ull;
    const unsigned bb = h2_bloom_bit(key, es.bloom_bits);
    atomicOr(&es.bloom[bb >> 5], 1u << (bb & 31u));
    unsigned long long h = h2_eset_slot(key, es.bits);
#pragma unroll 1
    for (unsigned long long walk = 0; walk <= mask; ++walk) {
        const unsigned long long old = atomicCAS(&es.tab[h], H2_ESET_EMPTY, key);
        if (old == H2_ESET_EMPTY || old == key) return;
        h = (h + 1) & mask;
    }
    *status = 1;
}

// ---- triangle step ------------------------------------------------------------------------------------------------------
// An edge {u,v} with triangles AND candidates (w in row v, outside N(u), M_u(w) >= 2): c(w) = M_u(w) - 1 - |N(w) ∩ Tset|,
// Tset = N(u) ∩ N(v).  Every (candidate, partner) pair is one probe of the edge set, and nearly all the pairs (98 % on the
// bench graph: 13.5 M) belong to a few thousand hub-to-hub edges with hundreds of candidates and partners each, up to
// 57 k pairs for one edge.  Probed on the spot by the wave that owns the row, they were 1.2 ms of a 2.6 ms pass (one
// dependent device-memory read per lane and round, the heavy rows of one hub in the same workgroup).  So the units only
// LIST the work — per such edge a task {record slot, candidates (w, M - 1), partners} in device-memory pools — and one
// dense kernel over all candidates of all tasks (k_h2_triangles) probes and publishes, spread over the whole chip.
struct H2Tasks {
    uint4 *task;   // per row of a block-class unit: {record slot, first partner, partners (0xFFFFFFFF: nothing to correct, its
                   // candidates are skipped), node | 0x80000000 when listed by the retry launch}
    int4 *cand;    // {w, M_u(w) - 1, task, unused}
    int32_t *part;
    int64_t task_cap, cand_cap, part_cap;
    int32_t *n_task, *n_cand, *n_part, *n_done;  // this pool's counters in the result block
    DevResult *res;
    const int32_t *weight;  // bit 31: the node went to the retry list (what its first attempt listed is void)
    unsigned retry_flag;    // 0x80000000 in the retry launch
    unsigned *lists;        // per-wave lists of the third step (H2_LIST_WORDS each, one per wave of the launch; nullptr: none)
};
// Pool space is handed out in chunks per wave (a reservation per edge on three shared counters cost more than the whole
// step: same-address atomics serialise); what a wave leaves of a candidate chunk is marked void.  All members are uniform.
constexpr int H2_CHUNK_C = 256, H2_CHUNK_P = 128, H2_CHUNK_T = 64;
struct H2Alloc {
    int c_cur = 0, c_end = 0, p_cur = 0, p_end = 0, t_cur = 0, t_end = 0;
};
__device__ inline int h2_pool_grab(int32_t *counter, int n) {  // all lanes call this
    int v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(counter, n);
    return __shfl(v, 0);
}
__device__ inline void h2_void_candidates(const H2Tasks tk, int from, int to) {
    for (int64_t i = (int64_t)from + (threadIdx.x & 63); i < to && i < tk.cand_cap; i += 64) tk.cand[i] = make_int4(0, 0, -1, 0);
}
constexpr unsigned H2_TASK_VOID = 0xFFFFFFFFu;
__device__ inline void h2_void_tasks(const H2Tasks tk, int from, int to) {
    for (int64_t i = (int64_t)from + (threadIdx.x & 63); i < to && i < tk.task_cap; i += 64)
        tk.task[i] = make_uint4(0u, 0u, H2_TASK_VOID, 0u);
}

// nt tasks, nc candidates and np partners, each contiguous (all lanes call this; false: the pools are too small)
__device__ inline bool h2_pool_reserve(const H2Tasks tk, H2Alloc &al, int nt, int nc, int np, int &t0, int &c0, int &p0) {
    if (al.t_end - al.t_cur < nt) {
        h2_void_tasks(tk, al.t_cur, al.t_end);
        const int n = nt > H2_CHUNK_T ? nt : H2_CHUNK_T;
        al.t_cur = h2_pool_grab(tk.n_task, n);
        al.t_end = al.t_cur + n;
    }
    if (al.c_end - al.c_cur < nc) {
        h2_void_candidates(tk, al.c_cur, al.c_end);
        const int n = nc > H2_CHUNK_C ? nc : H2_CHUNK_C;
        al.c_cur = h2_pool_grab(tk.n_cand, n);
        al.c_end = al.c_cur + n;
    }
    if (al.p_end - al.p_cur < np) {
        const int n = np > H2_CHUNK_P ? np : H2_CHUNK_P;
        al.p_cur = h2_pool_grab(tk.n_part, n);
        al.p_end = al.p_cur + n;
    }
    t0 = al.t_cur;
    c0 = al.c_cur;
    p0 = al.p_cur;
    al.t_cur += nt;
    al.c_cur += nc;
    al.p_cur += np;
    if (t0 < 0 || (int64_t)t0 + nt > tk.task_cap || c0 < 0 || (int64_t)c0 + nc > tk.cand_cap || p0 < 0 || (int64_t)p0 + np > tk.part_cap) {
        // the pools are too small (or the 31-bit counters wrapped): the counters keep counting, the host grows the pools
        // to what they say and runs the pass again
        if ((threadIdx.x & 63) == 0) {
            if (c0 < 0 || p0 < 0 || t0 < 0) tk.res->h2_status = 1;
            else atomicCAS(&tk.res->h2_status, 0, 2);  // (a table failure elsewhere, 1, stays)
        }
        return false;
    }
    return true;
}

// Fallback when the partner list of a batch did not fit in LDS (rows of hubs: hundreds of partners each): the members of
// N(u) in row v are read again, by one wave (long rows) or one LANE per row (rows of at most H2_SHORT_ROW entries).
template <int L1, int EXS>
__device__ inline void h2_partners_of_row(const View &g, const H2Tasks tk, int u, int2 rv, int p0, int npart, const H2Tab t) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int32_t *rowv = g.col + rv.x;
    int np = 0;  // running total (uniform)
    for (int base = 0; base < rv.y; base += 64) {
        const int i = base + lane;
        const int w = i < rv.y ? rowv[i] : -1;
        bool isT = false;
        if (w >= 0 && w != u) (void)h2_query<L1, EXS>(t, (unsigned)w, isT);
        const unsigned long long mT = __ballot(isT);
        const int it = np + __popcll(mT & below);
        if (isT && it < npart) tk.part[p0 + it] = w;
        np += __popcll(mT);
    }
    if (lane == 0 && np != npart) row_ok(g, make_int2(-1, np), 43, npart, u);  // cannot happen
}
constexpr int H2_SHORT_ROW = 32;
template <int L1, int EXS>
__device__ inline void h2_partners_short(const View &g, const H2Tasks tk, int u, int2 rv, int p0, int npart, const H2Tab t) {
    int pi = p0;
    const int pend = p0 + npart;
    for (int q0 = 0; q0 < rv.y; q0 += 8) {  // (eight loads in flight: a lane walks its row alone)
        int ww[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ww[q] = q0 + q < rv.y ? g.col[rv.x + q0 + q] : -1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int w = ww[q];
            if (w < 0 || w == u) continue;
            bool isT;
            (void)h2_query<L1, EXS>(t, (unsigned)w, isT);
            if (isT) {
                if (pi < pend) tk.part[pi] = w;
                ++pi;
            }
        }
    }
    if (pi != pend) row_ok(g, make_int2(-1, pi - pend), 43, npart, u);  // cannot happen
}

// ---- the third sweep from a list (round 4) ---------------------------------------------------------------------------------
// The third sweep used to stream every row again, test every entry against B2 again, queue and LOOK UP every exact-path entry
// again — for the per-row statistics of the few entries that matter (10 % of them in the middle class).  The second sweep has
// those entries in its hands, table slot included: it now appends {slot, row, batch} to a per-wave list in device memory (and
// where u sits in each row to a per-wave array), and the third step is one pass over that list once the counts are complete.
// A wave whose list would overflow, or with more batches than the array holds, falls back to streaming (the old path).
constexpr int H2_ITEMS = 2048;       // items per wave and unit
constexpr int H2_REV_BATCHES = 16;   // batches of 64 rows per wave whose "position of u" the array holds
constexpr int H2_LIST_WORDS = H2_ITEMS + H2_REV_BATCHES * 64;  // 32-bit words per wave
struct H2List {       // all members uniform over the wave
    unsigned *items;  // slot | row << 13 | batch << 19
    int *revs;        // [batch][row of the batch]: slot of u inside that row, -1: not found
    int n;            // items appended (counts past the capacity)
    bool over;        // the list does not describe the unit: stream the third sweep
};
__device__ inline H2List h2_list_of(unsigned *pool, int waves_per_group) {
    H2List ls;
    unsigned *mine = pool ? pool + ((int64_t)blockIdx.x * waves_per_group + (threadIdx.x >> 6)) * H2_LIST_WORDS : nullptr;
    ls.items = mine;
    ls.revs = reinterpret_cast<int *>(mine ? mine + H2_ITEMS : nullptr);
    ls.n = 0;
    ls.over = mine == nullptr;
    return ls;
}

// ---- the three sweeps: the rows of the neighbours of u, in batches of 64 rows per wave -----------------------------------
// the queued exact-path work of one wave: n is uniform and lives in a register
// PHASE 2 also LISTS what the triangle step needs, on the spot (no row is read again for it): every candidate occurrence
// (w outside N(u), c = M_u(w) - 1 > 0) goes to the device-memory pool with the task of its row (tbase + row: a task
// per row of the batch, filled in when the batch is complete), every member of N(u) met (a triangle partner of its row)
// to the batch's LDS list.
// what the third step does with ONE exact-path entry per lane: `have` (the lane holds one), its table slot, its row of the
// batch, its key.  pln (uniform) counts the batch's triangle partners.
template <int L1, int EXS>
__device__ inline void h2_settle(const H2Tab t, H2Scratch *sc, bool have, int s, int r, unsigned w, int &pln, const H2Tasks tk,
                                 H2Alloc &al, int tbase) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    bool cand = false, partner = false;
    int c = 0;
    if (have && s >= 0) {
        const unsigned c16 = h2_cnt_get(t.cnt, s);
        if (c16 & 0x8000u) {
            atomicAdd(&sc->rowT[r], 1);
            partner = true;
        } else {
            c = (int)(c16 & 0x7FFFu) - 1;
            if (c > 0) {
                atomicAdd(&sc->rowPos[r], 1);
                atomicMax(&sc->rowMx[r], c);
                cand = true;
            }
        }
    }
    const unsigned long long mC = __ballot(cand), mT = __ballot(partner);
    if (mC != 0ull && tbase >= 0) {  // uniform
        const int nc = __popcll(mC);
        if (al.c_end - al.c_cur < nc) {
            h2_void_candidates(tk, al.c_cur, al.c_end);
            al.c_cur = h2_pool_grab(tk.n_cand, H2_CHUNK_C);
            al.c_end = al.c_cur + H2_CHUNK_C;
        }
        const int c0 = al.c_cur;
        al.c_cur += nc;
        if (c0 < 0 || (int64_t)c0 + nc > tk.cand_cap) {
            if (lane == 0) {
                if (c0 < 0) tk.res->h2_status = 1;
                else atomicCAS(&tk.res->h2_status, 0, 2);  // the pool is too small: the host grows it and runs the pass again
            }
        } else if (cand) {
            tk.cand[c0 + __popcll(mC & below)] = make_int4((int)w, c, tbase + (int)sc->trank[r], 0);
        }
    }
    if (mT != 0ull) {  // uniform
        if (partner) {
            const int idx = pln + __popcll(mT & below);
            if (idx < H2_PLCAP) {
                sc->plt[idx] = w;
                sc->plr[idx] = (unsigned char)r;
            }
        }
        pln += __popcll(mT);
    }
}

// PHASE 1: occurrences of the repeated keys (and the list of the third step); PHASE 2 (streaming fallback): per-row statistics.
// PARTS: the unit is one key partition of a split node — only its own keys are counted, but the flagged members of N(u), which
// live in every partition's table, are met (and listed) whatever their partition: T and the partner lists are whole.
template <int L1, int EXS, int PHASE, bool PARTS>
__device__ inline void h2_drain(const H2Tab t, H2Scratch *sc, int &n, const H2Tasks tk, H2Alloc &al, int tbase, H2List &ls, int part,
                                int nparts) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
#ifdef H2_UNIT_TIMES
    const long long ut_d0 = (long long)__builtin_amdgcn_s_memtime();
    if (PHASE == 1 && lane == 0) { sc->ut_ndrain += 1; sc->ut_nitems += n; }
#endif
    h2_wave_sync();
    int pln = PHASE == 2 ? sc->pln : 0;  // uniform
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const unsigned w = i < n ? sc->qw[i] : 0u;
        if (PHASE == 1) {
            int s = -1;
            if (i < n) {
                const bool own = !PARTS || h2_part(w, nparts) == part;
                if (own) {
                    s = h2_insert<EXS>(t.key, w);
                    if (s < 0) *t.full = 1;
                    else h2_cnt_add(t.cnt, s);
                } else {  // another partition's key: of interest only as a member of N(u)
                    s = h2_find<EXS>(t.key, w);
                    if (s >= 0 && !(h2_cnt_get(t.cnt, s) & 0x8000u)) s = -1;
                }
            }
            if (!ls.over) {  // uniform
                const unsigned long long m = __ballot(s >= 0);
                const int cnt = __popcll(m);
                if (ls.n + cnt > H2_ITEMS) {
                    ls.over = true;
                } else if (s >= 0) {
                    ls.items[ls.n + __popcll(m & below)] = (unsigned)s | ((unsigned)sc->qr[i] << 13) | ((unsigned)sc->qb[i] << 19);
                }
                ls.n += cnt;
            }
        } else {
            int s = -1, r = 0;
            if (i < n) {
                s = h2_find<EXS>(t.key, w);
                r = sc->qr[i];
            }
            h2_settle<L1, EXS>(t, sc, i < n, s, r, w, pln, tk, al, tbase);
        }
    }
    if (PHASE == 2 && lane == 0) sc->pln = pln;
    h2_wave_sync();
#ifdef H2_UNIT_TIMES
    if (PHASE == 1 && lane == 0) sc->ut_drain += (long long)__builtin_amdgcn_s_memtime() - ut_d0;
#endif
    n = 0;
}

// the third step from the list: the items of batch `bidx` (they are batch-monotone: the second sweep's queue is first in, first out)
template <int L1, int EXS>
__device__ inline void h2_settle_items(const H2Tab t, H2Scratch *sc, const H2List &ls, int &lcur, int bidx, const H2Tasks tk, H2Alloc &al,
                                       int tbase) {
    const int lane = threadIdx.x & 63;
    int pln = sc->pln;  // uniform
#pragma unroll 1
    while (true) {
        const int i = lcur + lane;
        const unsigned item = i < ls.n ? ls.items[i] : 0xFFFFFFFFu;
        const bool mine = i < ls.n && (int)(item >> 19) == bidx;
        const int cnt = __popcll(__ballot(mine));
        if (cnt == 0) break;  // uniform
        const int s = (int)(item & 0x1FFFu), r = (int)((item >> 13) & 63u);
        const unsigned w = mine ? t.key[s] : 0u;
        h2_settle<L1, EXS>(t, sc, mine, mine ? s : -1, r, w, pln, tk, al, tbase);
        lcur += cnt;
        if (cnt < 64) break;
    }
    if (lane == 0) sc->pln = pln;
    h2_wave_sync();
}

// third sweep, start of a batch of rows: row accumulators, partner list, a task per row present
__device__ inline void h2_batch_begin(const H2Tasks tk, H2Alloc &al, H2Scratch *sc, int k, int &tbase, int &trow, int rev = -1) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    sc->rowT[lane] = 0;
    sc->rowPos[lane] = 0;
    sc->rowMx[lane] = 0;
    sc->rowRev[lane] = rev;  // (-1: found by the sweep that follows; list mode: what the second sweep found)
    if (lane == 0) sc->pln = 0;
    int dummy_c, dummy_p;
    const unsigned long long rows = __ballot(k >= 0);  // a task per row of the batch
    if (!h2_pool_reserve(tk, al, __popcll(rows), 0, 0, tbase, dummy_c, dummy_p)) tbase = -1;
    trow = __popcll(rows & below);  // rank of this lane's row among them (the drain needs rank by row: sc->trank)
    sc->trank[lane] = (unsigned char)trow;
    trow += tbase;
}

// third sweep, end of a batch (the queue is drained): partners to the pool, tasks, records.  i: this lane's position in
// row u; k, rk: its member of N(u) and that one's row
// items / [ifrom, ito): in list mode, the wave's exact-path items of this batch ({slot, row, batch}: h2_settle_items has just walked
// them) — round 5: when the batch has more triangle partners than the LDS list holds, they are taken from these items in a second
// pass instead of reading the rows again (a wave per long row, one after the other: 186 of the 407 us of the longest unit of the
// bench graph, the unit that sets the length of the split class's kernel).  items == nullptr: streaming mode, rows are read again.
template <int L1, int EXS, bool PARTS>
__device__ inline void h2_batch_end(const View &g, const H2Tasks tk, H2Alloc &al, int u, int2 ru, int i, int k, int2 rk, int tbase,
                                    int trow, int part, const H2Tab t, H2Scratch *sc, uint4 *rec, const unsigned *items = nullptr,
                                    int ifrom = 0, int ito = 0) {
    const int lane = threadIdx.x & 63;
    int T = sc->rowT[lane], pos = sc->rowPos[lane], mx = sc->rowMx[lane];
    const int rev = sc->rowRev[lane];
    // Triangles AND positive counts: the counts of this edge still include the triangle partners.  Its candidates
    // are in the pool already (listed by the drain); its partners go there now and its task says where they are:
    // k_h2_triangles publishes its counts, here it contributes none.  The tasks of all other rows are void.
    bool listed = tbase >= 0 && k >= 0 && T > 0 && pos > 0 && rev >= 0 && rev < g.cap_total;
#ifdef H2_NO_STEPC  // timing-only build (results wrong)
    listed = false;
#endif
    int ep;
    const int Pn = h2_prefix(listed ? T : 0, ep);
    int p0 = 0;
    if (Pn > 0) {  // uniform
        int dummy_t, dummy_c;
        if (!h2_pool_reserve(tk, al, 0, 0, Pn, dummy_t, dummy_c, p0)) listed = false;
    }
    if (__ballot(listed) != 0ull) {
        const int pln = sc->pln;
        if (pln <= H2_PLCAP) {
            // the partners were kept: each goes to the range of its row (sc->poff is free now: a cursor per row)
            sc->poff[lane] = listed ? p0 + ep : -1;
            h2_wave_sync();
            for (int j = lane; j < pln; j += 64) {
                const int r = sc->plr[j];
                if (sc->poff[r] >= 0) tk.part[atomicAdd(&sc->poff[r], 1)] = (int32_t)sc->plt[j];
            }
            h2_wave_sync();
        } else if (items) {
            // too many for the LDS list (rows of hubs): a second pass over the batch's items — every flagged one is a partner of its row
            sc->poff[lane] = listed ? p0 + ep : -1;
            h2_wave_sync();
            for (int j = ifrom + lane; j < ito; j += 64) {
                const unsigned item = items[j];
                const int s2 = (int)(item & 0x1FFFu), r = (int)((item >> 13) & 63u);
                if ((h2_cnt_get(t.cnt, s2) & 0x8000u) && sc->poff[r] >= 0) tk.part[atomicAdd(&sc->poff[r], 1)] = (int32_t)t.key[s2];
            }
            h2_wave_sync();
        } else {
            // too many for the list (rows of hubs): the rows are read again for their partners
            const bool shortrow = listed && rk.y <= H2_SHORT_ROW;
            if (shortrow) h2_partners_short<L1, EXS>(g, tk, u, rk, p0 + ep, T, t);
            unsigned long long todo = __ballot(listed && !shortrow);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int2 rv = make_int2(__shfl(rk.x, l), __shfl(rk.y, l));
                h2_partners_of_row<L1, EXS>(g, tk, u, rv, __shfl(p0 + ep, l), __shfl(T, l), t);
            }
        }
    }
    if (tbase >= 0 && k >= 0) {
        tk.task[trow] = listed ? make_uint4((unsigned)(ru.x + i), (unsigned)(p0 + ep), (unsigned)T, (unsigned)u | tk.retry_flag)
                               : make_uint4(0u, 0u, H2_TASK_VOID, 0u);
    }
    if (listed) {
        pos = 0;
        mx = 0;
    }
    if (k >= 0) {
        const int64_t slot = (int64_t)ru.x + i;
        if (rev < 0 || rev >= g.cap_total) {
            row_ok(g, make_int2(-1, rev), 33, u, k);  // adjacency not symmetric: report, never publish
        } else if (!PARTS) {
            rec[slot] = make_uint4((unsigned)pos, (unsigned)mx, h2_rec_t(T, ru.y), (unsigned)rev);
        } else {
            unsigned *r4 = reinterpret_cast<unsigned *>(rec + slot);
            if (pos) atomicAdd(&r4[0], (unsigned)pos);
            if (mx) atomicMax(&r4[1], (unsigned)mx);
            if (part == 0) {  // the flagged neighbours live in every partition's table: T and the slot are whole
                r4[2] = h2_rec_t(T, ru.y);
                r4[3] = (unsigned)rev;
            }
        }
    }
}

// Wave `wid` of NW takes the rows i = wid, wid + NW, ... of row u (strided: a hub's heaviest rows, adjacent at the
// front of its row, spread over the waves); lane l of the batch starting at `base` stands for row base + l * NW + wid.
// PHASE 0: bitmaps; 1: occurrences of the repeated keys; 2: per-row statistics, triangle step, records.
// k0, rk0: this lane's row of the FIRST batch (member of N(u) and its row header, validated), read once by the caller for all
// three sweeps: a unit of the streaming path is a chain of dependent device-memory reads (row of u -> headers of its
// members' rows -> their pieces) with a handful of pieces per lane behind it, and two of its three links do not change
// between the sweeps (round 4: the split class spent most of every sweep waiting for them).
template <int L1, int EXS, int NW, bool PARTS, int PHASE>
__device__ inline void h2_stream(const View &g, const H2Tasks tk, H2Alloc &al, int u, int2 ru, int part, int nparts,
                                 const H2Tab t, H2Scratch *sc, uint4 *rec, int k0, int2 rk0, H2List &ls) {
    const int lane = threadIdx.x & 63;
    const int wid = (int)(threadIdx.x >> 6);
    int qn = 0;  // queued items (uniform)
    int lcur = 0;  // third step from the list: next item (uniform)
    if (PHASE == 0 && ru.y > 64 * NW * H2_REV_BATCHES) ls.over = true;  // more batches than the list's row array holds
#ifdef H2_PROF
    long long t_prof = (long long)__builtin_amdgcn_s_memtime();
    H2Scratch *s = sc;
#endif
    for (int base = 0; base < ru.y; base += 64 * NW) {
        const int i = base + lane * NW + wid;
        int k = k0;
        int2 rk = rk0;
        if (base > 0) {  // uniform
            k = -1;
            rk = make_int2(0, 0);
            if (i < ru.y) {
                k = g.col[ru.x + i];
                if (k >= 0 && k < g.n && k != u) {
                    rk = g.rowinfo[k];
                    if (!row_ok(g, rk, 32, k, u)) rk = make_int2(0, 0);
                } else {
                    k = -1;
                }
            }
        }
        const int np = rk.y > 0 ? ((rk.x + rk.y + 3) >> 2) - (rk.x >> 2) : 0;
        int poff_lane;
        const int P = h2_prefix(np, poff_lane);
        if (P == 0) continue;  // uniform: no row in this wave's share of the batch
        const int bidx = base / (64 * NW);
        int tbase = -1, trow = 0;  // third sweep: the tasks of this batch's rows (tbase uniform; trow: this lane's row)
        if (PHASE == 2 && !ls.over) {
            // the third step from the second sweep's list: no row is streamed, no entry tested or looked up again
            h2_batch_begin(tk, al, sc, k, tbase, trow, ls.revs[bidx * 64 + lane]);
            h2_wave_sync();
            H2_STAMP(8)
            const int lfrom = lcur;
            h2_settle_items<L1, EXS>(t, sc, ls, lcur, bidx, tk, al, tbase);
            H2_STAMP(9)
            h2_batch_end<L1, EXS, PARTS>(g, tk, al, u, ru, i, k, rk, tbase, trow, part, t, sc, rec, ls.items, lfrom, lcur);
            h2_wave_sync();  // the scratch is rewritten by the next batch
            H2_STAMP(11)
            continue;
        }
        if (PHASE == 0 && !ls.over) ls.revs[bidx * 64 + lane] = -1;
        sc->desc[lane] = rk;
        sc->poff[lane] = poff_lane;
        if (lane == 0) sc->poff[64] = P;
        if (PHASE == 2) h2_batch_begin(tk, al, sc, k, tbase, trow);
        h2_wave_sync();
        if (PHASE == 2) H2_STAMP(8)
        const bool listing = PHASE == 1 && !ls.over;  // uniform
        auto flags = [&](const int4 w, unsigned vm, int r, int a) -> unsigned {
            const unsigned kk[4] = {(unsigned)w.x, (unsigned)w.y, (unsigned)w.z, (unsigned)w.w};
            const unsigned isu = h2_eq4(kk, (unsigned)u) & vm;
            if (PHASE == 2 && isu) sc->rowRev[r] = a + __ffs((int)isu) - 1;
            if (listing && isu) ls.revs[bidx * 64 + r] = a + __ffs((int)isu) - 1;  // (where u sits in that row: the third step's)
            unsigned valid = vm & ~isu;
            // (third sweep: no partition test — a key of another partition is simply not in the table, and the flagged
            //  members of N(u), which are in every partition's table, must all be met: T and the partner lists are whole.
            //  The same holds for the second sweep while it lists for the third step: its drain tells the partitions apart.)
            if (PARTS && PHASE != 2 && !listing) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (h2_part(kk[jj], nparts) != part) valid &= ~(1u << jj);
            }
            if (PHASE == 0) {
                h2_mark4<L1>(t.b1, t.b2, kk, valid);
                return 0u;
            }
            return h2_again4<L1>(t.b2, kk, valid);
        };
        h2_for_pieces_queued<(PARTS ? H2_QL : H2_Q)>(g.col, sc->desc, sc->poff, poff_lane, P, qn, flags,
                             [&](int idx, unsigned key, int r) {
                                 sc->qw[idx] = key;
                                 if (PHASE != 0) sc->qr[idx] = (unsigned char)r;
                                 if (PHASE == 1) sc->qb[idx] = (unsigned char)bidx;
                             },
                             [&]() { h2_drain<L1, EXS, PHASE, PARTS>(t, sc, qn, tk, al, tbase, ls, part, nparts); });
        if (PHASE == 2) H2_STAMP(9)
        if (PHASE == 2) {
            h2_drain<L1, EXS, PHASE, PARTS>(t, sc, qn, tk, al, tbase, ls, part, nparts);  // the row totals are read next
            H2_STAMP(10)
            h2_batch_end<L1, EXS, PARTS>(g, tk, al, u, ru, i, k, rk, tbase, trow, part, t, sc, rec);
        }
        h2_wave_sync();  // the scratch is rewritten by the next batch
        if (PHASE == 2) H2_STAMP(11)
    }
    if (qn > 0) h2_drain<L1, EXS, PHASE, PARTS>(t, sc, qn, tk, al, -1, ls, part, nparts);  // (after the loop: a wave's last batches may be empty; PHASE 1 only)
}

// A unit whose rows fit one batch (at most 64 per wave) and whose pieces fit NPB per lane: rows, descriptors and pieces
// are read ONCE and stay in registers for all three sweeps (the streaming version pays the chain unit -> row of u -> rows'
// descriptors -> pieces three times, with a handful of pieces per wave to hide it behind).  false: does not fit (decided
// before anything is written; the caller streams).  `ok` is set like h2_node's return value.
constexpr int H2_NPB = 12;
template <int L1, int EXS, int NW, bool PARTS>
__device__ inline bool h2_node_fast(const View &g, const H2Tasks tk, H2Alloc &al, int u, int2 ru, int part, int nparts,
                                    const H2Tab t, H2Scratch *sc, uint4 *rec, bool &ok, H2List &ls) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NT = 64 * NW, NP = H2_NPB;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ru.y > 64 * NW) return false;  // uniform
#ifdef H2_PROF
    long long t_prof = (long long)__builtin_amdgcn_s_memtime();
    H2Scratch *s = sc;
#endif
    const int i = lane * NW + wid;
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
    int poff_lane;
    const int P = h2_prefix(np, poff_lane);
    if (__syncthreads_or(P > 64 * NP)) return false;  // some wave's share is too long for the registers
    {
        uint4 *z = reinterpret_cast<uint4 *>(t.b1);  // b1 and b2 are adjacent
        constexpr int NZ = ((1 << L1) / 32 + (1 << (L1 - 2)) / 32) / 4;
        for (int j = tid; j < NZ; j += NT) z[j] = make_uint4(0u, 0u, 0u, 0u);
        uint4 *k4 = reinterpret_cast<uint4 *>(t.key);
        for (int j = tid; j < EXS / 4; j += NT) k4[j] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
        uint4 *c4 = reinterpret_cast<uint4 *>(t.cnt);
        for (int j = tid; j < EXS / 8; j += NT) c4[j] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) *t.full = 0;
    }
    sc->desc[lane] = rk;
    sc->poff[lane] = poff_lane;
    if (lane == 0) sc->poff[64] = P;
    h2_wave_sync();
    // every piece of this wave's rows: meta = entries inside the row | those of this unit's key partition << 4 | row << 8 |
    // first slot of the piece << 16
    int4 w[NP];
    unsigned long long meta[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int j = 64 * q + lane;
        w[q] = make_int4(0, 0, 0, 0);
        meta[q] = 0ull;
        const int jf = 64 * q, jl = jf + 63 < P ? jf + 63 : P - 1;
        if (jf >= P) continue;  // uniform
        const int r = h2_piece_row(sc->poff, poff_lane, j < P ? j : jl, jf, jl);
        if (j < P) {
            const int2 d = sc->desc[r];
            const int a = (d.x & ~3) + 4 * (j - sc->poff[r]);
            w[q] = load_piece(g.col, a);
            meta[q] = (unsigned long long)h2_piece_mask(a, d.x, d.x + d.y) | ((unsigned long long)r << 8) | ((unsigned long long)(unsigned)a << 16);
        }
    }
    __syncthreads();  // the tables are cleared
    H2_STAMP(0 + (PARTS ? 4 : 0))
    H2_UT_PHASE(0)
    if (k >= 0) {     // the members of N(u): flagged table entries (in every partition's tables)
        const int sl = h2_insert<EXS>(t.key, (unsigned)k);
        if (sl < 0) *t.full = 1;
        else h2_cnt_flag(t.cnt, sl);
        h2_seed<L1>(t.b1, t.b2, (unsigned)k);
    }
    __syncthreads();
    // sweep A
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (64 * q >= P) continue;  // uniform
        const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
        unsigned valid = (unsigned)meta[q] & 0xFu & ~h2_eq4(kk, (unsigned)u);
        if (PARTS) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if (h2_part(kk[jj], nparts) != part) valid &= ~(1u << jj);
        }
        meta[q] |= (unsigned long long)valid << 4;
        h2_mark4<L1>(t.b1, t.b2, kk, valid);
    }
    __syncthreads();
    H2_STAMP(1 + (PARTS ? 4 : 0))
    H2_UT_PHASE(1)
    // Sweep B: flagged entries are queued, the drain has one call site.  It lists its exact-path entries (slot and row) and
    // where u sits in each row; the third step is then one pass over that list (h2_settle_items) — unless the list overflowed:
    // that wave streams its rows for the third sweep (h2_stream, the path of the units that do not fit the registers).
    int qn = 0, tbase = -1, trow = 0;
    sc->rowRev[lane] = -1;
    h2_wave_sync();
    {
        unsigned fl[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            fl[q] = 0u;
            if (64 * q >= P) continue;  // uniform
            const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
            const unsigned valid = (unsigned)(meta[q] >> 4) & 0xFu;
            // (where u sits in the row: the third step's, found while the pieces are at hand)
            const unsigned isu = h2_eq4(kk, (unsigned)u) & (unsigned)meta[q] & 0xFu;
            if (isu) sc->rowRev[(int)((meta[q] >> 8) & 0xFFull)] = (int)(meta[q] >> 16) + __ffs((int)isu) - 1;
            fl[q] = h2_again4<L1>(t.b2, kk, valid);
        }
#pragma unroll 1
        while (true) {
            bool stop = false;  // uniform: the queue is full, the rest waits for the drain
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (64 * q >= P) continue;  // uniform
                const unsigned kk[4] = {(unsigned)w[q].x, (unsigned)w[q].y, (unsigned)w[q].z, (unsigned)w[q].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const bool p = (fl[q] >> jj) & 1u;
                    const unsigned long long m = __ballot(p);
                    if (m == 0ull) continue;  // uniform
                    if (stop || qn + __popcll(m) > H2_QCAP) {
                        stop = true;
                        continue;
                    }
                    if (p) {
                        const int idx = qn + __popcll(m & below);
                        sc->qw[idx] = kk[jj];
                        sc->qr[idx] = (unsigned char)((meta[q] >> 8) & 0xFFull);
                        sc->qb[idx] = 0;
                        fl[q] &= ~(1u << jj);
                    }
                    qn += __popcll(m);
                }
            }
            if (qn > 0) h2_drain<L1, EXS, 1, PARTS>(t, sc, qn, tk, al, -1, ls, part, nparts);
            if (!stop) break;
        }
    }
    __syncthreads();
    H2_STAMP(2 + (PARTS ? 4 : 0))
    H2_UT_PHASE(2)
    ok = *t.full == 0;  // uniform
    if (ok) {
        if (!ls.over) {  // uniform over the wave
            int lcur = 0;
            const int myrev = sc->rowRev[lane];
            h2_wave_sync();
            h2_batch_begin(tk, al, sc, k, tbase, trow, myrev);
            H2_STAMP(12)
            h2_wave_sync();
            h2_settle_items<L1, EXS>(t, sc, ls, lcur, 0, tk, al, tbase);
            H2_STAMP(13)
            h2_batch_end<L1, EXS, PARTS>(g, tk, al, u, ru, i, k, rk, tbase, trow, part, t, sc, rec, ls.items, 0, lcur);
            H2_STAMP(15)
        } else {
            h2_stream<L1, EXS, NW, PARTS, 2>(g, tk, al, u, ru, part, nparts, t, sc, rec, k, rk, ls);
        }
    }
    __syncthreads();  // the tables are rewritten by the next unit
    return true;
}

// one unit: node u, key partition `part` of `nparts`, by NW waves sharing the tables `t`; false: the table filled up
template <int L1, int EXS, int NW, bool PARTS>
__device__ inline bool h2_node(const View &g, const H2Tasks tk, H2Alloc &al, int u, int2 ru, int part, int nparts,
                               const H2Tab t, H2Scratch *sc, uint4 *rec, H2List &ls) {
    const int tid = (int)threadIdx.x;
    constexpr int NT = 64 * NW;
#ifndef H2_NO_FAST
    if constexpr (!PARTS) {  // (with 16 waves per unit the registers it needs spill: measured slower for the split class)
        bool ok_fast = true;
        if (h2_node_fast<L1, EXS, NW, PARTS>(g, tk, al, u, ru, part, nparts, t, sc, rec, ok_fast, ls)) return ok_fast;
    }
#endif
#ifdef H2_PROF
    long long t_prof = (long long)__builtin_amdgcn_s_memtime();
    H2Scratch *s = sc;
#endif
    // this lane's row of the first batch of every sweep (requested before the tables are cleared)
    int k0 = -1;
    int2 rk0 = make_int2(0, 0);
    {
        const int i0 = (tid & 63) * NW + (tid >> 6);
        if (i0 < ru.y) {
            k0 = g.col[ru.x + i0];
            if (k0 >= 0 && k0 < g.n && k0 != u) {
                rk0 = g.rowinfo[k0];
                if (!row_ok(g, rk0, 32, k0, u)) rk0 = make_int2(0, 0);
            } else {
                k0 = -1;
            }
        }
    }
    {
        uint4 *z = reinterpret_cast<uint4 *>(t.b1);  // b1 and b2 are adjacent
        constexpr int NZ = ((1 << L1) / 32 + (1 << (L1 - 2)) / 32) / 4;
        for (int i = tid; i < NZ; i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
        uint4 *k4 = reinterpret_cast<uint4 *>(t.key);
        for (int i = tid; i < EXS / 4; i += NT) k4[i] = make_uint4(H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY);
        uint4 *c4 = reinterpret_cast<uint4 *>(t.cnt);
        for (int i = tid; i < EXS / 8; i += NT) c4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) *t.full = 0;
    }
    __syncthreads();
    for (int i = tid; i < ru.y; i += NT) {  // the members of N(u): flagged table entries (in every partition's tables)
        const int k = g.col[ru.x + i];
        if (k >= 0 && k < g.n && k != u) {
            const int s = h2_insert<EXS>(t.key, (unsigned)k);
            if (s < 0) *t.full = 1;
            else h2_cnt_flag(t.cnt, s);
            h2_seed<L1>(t.b1, t.b2, (unsigned)k);
        }
    }
    __syncthreads();
    H2_STAMP(8 - 8 + (PARTS ? 4 : 0))
    H2_UT_PHASE(0)
    h2_stream<L1, EXS, NW, PARTS, 0>(g, tk, al, u, ru, part, nparts, t, sc, rec, k0, rk0, ls);
    __syncthreads();
    H2_STAMP(1 + (PARTS ? 4 : 0))
    H2_UT_PHASE(1)
    h2_stream<L1, EXS, NW, PARTS, 1>(g, tk, al, u, ru, part, nparts, t, sc, rec, k0, rk0, ls);
    __syncthreads();
    H2_STAMP(2 + (PARTS ? 4 : 0))
    H2_UT_PHASE(2)
    const bool ok = *t.full == 0;  // uniform
    if (ok) h2_stream<L1, EXS, NW, PARTS, 2>(g, tk, al, u, ru, part, nparts, t, sc, rec, k0, rk0, ls);
    __syncthreads();  // the tables are rewritten by the next unit
    H2_STAMP(3 + (PARTS ? 4 : 0))
    return ok;
}

// RETRY: the units come from the retry list (whose tables must not fill up again: the pass falls back then)
// (at least three waves per SIMD: class M's fast path wanted 232 registers, i.e. two of its 4-wave workgroups per CU where its
//  LDS lets three live; at 168 registers and 136 bytes of spills the pass is 2.6 % faster on S100k, 4.3 % on S1M)
template <int L1, int EXS, int NW, bool PARTS>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3))) k_h2_block(View g, H2Tasks tk, const int4 *units, const int32_t *count, int64_t unit_cap,
                                                       uint4 *rec, H2Retry rt, int is_retry) {
    __shared__ __attribute__((aligned(16))) unsigned bits[(1 << L1) / 32 + (1 << (L1 - 2)) / 32];
    __shared__ __attribute__((aligned(16))) unsigned key[EXS];
    __shared__ __attribute__((aligned(16))) unsigned cnt[EXS / 2];
    __shared__ H2Scratch sc_all[NW];
    __shared__ int full;
    const int wid = threadIdx.x >> 6;
    const int total = *count;
    if (total < 0 || total > unit_cap) {  // uniform
        row_ok(g, make_int2(-1, total), 37, 0, 0);
        return;
    }
#ifdef H2_UNIT_TIMES
    __shared__ long long ut_sh[4];
    const H2Tab t{bits, bits + (1 << L1) / 32, key, cnt, &full, ut_sh};
#else
    const H2Tab t{bits, bits + (1 << L1) / 32, key, cnt, &full};
#endif
    H2Alloc al;
#ifdef H2_PROF
    if ((threadIdx.x & 63) < 16) sc_all[wid].prof[threadIdx.x & 63] = 0ull;
#endif
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {  // every value steering the barriers is uniform
        const int4 un = units[it];
        const int u = un.x;
        const int nparts = PARTS ? (int)((unsigned)un.y >> 16) : 1, part = PARTS ? (un.y & 0xFFFF) : 0;
        bool ok = u >= 0 && u < g.n && nparts >= 1 && part < nparts;
        int2 ru = make_int2(0, 0);
        if (ok) {
            ru = g.rowinfo[u];
            ok = row_ok(g, ru, 38, u, (int)it) && ru.y > 0 && ru.y <= H2_MAXDEG;
        }
        if (!ok) continue;
        H2List ls = h2_list_of(tk.lists, NW);  // (fresh per unit)
#ifdef H2_UNIT_TIMES
        const long long ut0 = (long long)__builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) { sc_all[wid].ut_drain = 0; sc_all[wid].ut_ndrain = 0; sc_all[wid].ut_nitems = 0; }
#endif
        const bool unit_ok = h2_node<L1, EXS, NW, PARTS>(g, tk, al, u, ru, part, nparts, t, &sc_all[wid], rec, ls);
#ifdef H2_UNIT_TIMES
        if (threadIdx.x == 0 && !is_retry) {
            const long long ut1 = (long long)__builtin_amdgcn_s_memtime();
            const unsigned slot = atomicAdd(&h2_ut_n, 1u);
            if (slot < H2_UT_CAP) {
                h2_ut[2 * slot] = make_int4(u, (PARTS ? 0x40000000 : 0) | (nparts << 16) | part, (int)(ut1 - ut0), (int)((ut0 - h2_ut_t0) >> 4));
                h2_ut[2 * slot + 1] = make_int4((int)(ut_sh[0] - ut0), (int)(ut_sh[1] - ut_sh[0]), (int)(ut_sh[2] - ut_sh[1]), (int)(ut1 - ut_sh[2]));
                h2_ut2[slot] = make_int4((int)sc_all[0].ut_drain, sc_all[0].ut_ndrain, sc_all[0].ut_nitems, 0);
            }
        }
#endif
        if (!unit_ok) {
            if (threadIdx.x == 0) {  // every partition of the node is redone (the retry starts from zeroed records)
                if (is_retry) {
                    rt.res->h2_status = 1;
                    atomicAdd(&rt.res->h2_failed[5], 1);
                } else {
                    h2_retry_push(rt, u, ru.y, PARTS ? 4 : 3);
                }
            }
        }
    }
    h2_void_candidates(tk, al.c_cur, al.c_end);
    h2_void_tasks(tk, al.t_cur, al.t_end);
#ifdef H2_PROF
    h2_wave_sync();
    if ((threadIdx.x & 63) < 16) atomicAdd(&h2_prof[8 + (threadIdx.x & 63)], sc_all[wid].prof[threadIdx.x & 63]);
#endif
}

// how many of the partners pt[0..np) are adjacent to w: one probe of the edge set per pair, eight in flight, most of
// them settled by the edge bitmap
__device__ inline int h2_probe_partners(const H2EdgeSet es, int w, const int32_t *pt, int np) {
    int found = 0;
    for (int j = 0; j < np; j += 8) {
        int tt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) tt[q] = j + q < np ? pt[j + q] : -1;
        unsigned long long key[8], k0[8], h[8];
        unsigned bw[8], bb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            key[q] = h2_edge_key(w, tt[q] < 0 ? w : tt[q]);
            bb[q] = h2_bloom_bit(key[q], es.bloom_bits);
            bw[q] = es.bloom[bb[q] >> 5];
        }
        // few pairs pass the bitmap; those that do read their home slot of the edge set together, not one after the other
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool maybe = tt[q] >= 0 && ((bw[q] >> (bb[q] & 31u)) & 1u);
            h[q] = h2_eset_slot(key[q], es.bits);
            k0[q] = maybe ? es.tab[h[q]] : H2_ESET_EMPTY;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bool hit = k0[q] == key[q];
            if (!hit && k0[q] != H2_ESET_EMPTY) hit = h2_eset_has(es, w, tt[q]);  // (rare: the home slot is someone else's)
            found += hit ? 1 : 0;
        }
    }
    return found;
}

// the listed (candidate, partner) pairs: a thread per candidate probes the edge set for each partner of its edge
// and adds its corrected count to the record of the edge; the candidates of an edge are adjacent, so lanes that share a
// record reduce first
// Launched twice: `second` = 0 as soon as the block classes have finished (beside the wave classes, which list nothing),
// for the candidates there are then; `second` = 1 after the retry launch for what that one listed.  (The split is read
// from / left in the result block: h2_ncand_done is only written by the kernel that marks the first launch's end.)
// (Round 4: the first launch of a pool takes the candidates there are and records their number itself — a one-thread kernel did
//  that before, and behind the split class on the main stream it took 20-50 us to get a wave.)
__global__ void __launch_bounds__(256) k_h2_triangles(H2EdgeSet es, H2Tasks tk, uint4 *rec, const int32_t *status, int second) {
    const int cand_now = *tk.n_cand;  // (stable: the kernels that list into this pool are through)
    const int first = second ? *tk.n_done : 0;
    // Round 5 (found by the randomised sweep, one pass in ~1,500 on a graph whose pools had just been grown to what the pass before
    // had counted): the counter advances by whole CHUNKS, so it can pass the end of the pool although every reservation fitted —
    // no status is raised then (h2_pool_reserve fails only when c0 + nc exceeds the pool) and nothing lies beyond the pool.  This
    // kernel used to take "counter beyond the pool" for a failed pass and return: a pass that was NOT run again lost every
    // correction of this launch.  A reservation that did not fit has set the status, which is checked right below.
    const int total = cand_now < tk.cand_cap ? cand_now : (int)tk.cand_cap;
    if (!second && blockIdx.x == 0 && threadIdx.x == 0) *tk.n_done = cand_now;  // read by the retry stage's launch only
    if (*status != 0) return;
    if (total <= first || first < 0) return;
    const int lane = threadIdx.x & 63;
    for (int64_t base = first + (((int64_t)blockIdx.x * 256 + threadIdx.x) & ~63ll); base < total; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + lane;
        int c = 0;
        unsigned slot = 0xFFFFFFFFu;
        if (i < total) {
            const int4 cd = tk.cand[i];
            const uint4 ts = (cd.z >= 0 && cd.z < tk.task_cap) ? tk.task[cd.z] : make_uint4(0u, 0u, H2_TASK_VOID, 0u);  // (void: the tail of a wave's chunk)
            // (a split node some partition of which failed is redone as a whole: what its other partitions listed is void)
            const bool skip = ts.z == H2_TASK_VOID || (!(ts.w & 0x80000000u) && ((unsigned)tk.weight[ts.w & 0x7FFFFFFFu] & 0x80000000u));
            if (!skip) {
                slot = ts.x;
#ifdef H2_NO_PROBE  // timing-only build (results wrong)
                c = cd.y;
#else
                c = cd.y - h2_probe_partners(es, cd.x, tk.part + ts.y, (int)ts.z);
#endif
            }
        }
        // the candidates of an edge are adjacent: one pair of atomics per run of equal record slots (segmented scan)
        const unsigned prev = (unsigned)__shfl_up((int)slot, 1);
        const bool head = lane == 0 || prev != slot;
        const unsigned long long heads = __ballot(head);
        const int seg = __popcll(heads & ((2ull << lane) - 1ull));
        int n = c > 0 ? 1 : 0, mx = c > 0 ? c : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int s2 = __shfl_up(seg, off), n2 = __shfl_up(n, off), m2 = __shfl_up(mx, off);
            if (lane >= off && s2 == seg) {
                n += n2;
                mx = m2 > mx ? m2 : mx;
            }
        }
        const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
        if (tail && n > 0 && slot != 0xFFFFFFFFu) {
            unsigned *r4 = reinterpret_cast<unsigned *>(rec + slot);
            atomicAdd(&r4[0], (unsigned)n);
            atomicMax(&r4[1], (unsigned)mx);
        }
    }
}

// Σ_{k in N(u)} deg(k) for every u: a thread per adjacency slot, run-length sums inside a wave (rows are contiguous),
// one atomic per run
__global__ void __launch_bounds__(256) k_h2_weight(View g, int32_t *weight) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int u = -1, val = 0;
    if (s < g.cap_total) {
        // two rounds of loads, not four: the slot's row and neighbour first (both in bounds for every slot), then the two
        // row headers; what the slot holds is validated afterwards (slack behind a row holds -1)
        const int us = g.slot_row[s];
        const int v = g.col[s];
        const bool uok = us >= 0 && us < g.n, vok = v >= 0 && v < g.n;
        const int2 ru = g.rowinfo[uok ? us : 0];
        const int dv = g.rowinfo[vok ? v : 0].y;
        if (uok) {
            u = us;
            if (vok && s >= ru.x && (int)(s - ru.x) < ru.y) val = dv;
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
    int4 *units[H2_CLASSES];  // {node, partitions << 16 | partition, row start, degree}
    int64_t cap[H2_CLASSES];
};
constexpr int H2_NB = H2_CLASSES * H2_WB;
constexpr int H2_PLAN_THREADS = 1024;

__device__ inline void h2_classify(int d, int S, int &cls, int &wb, int &nparts) {
    nparts = 1;
    int lo = 0;
    if (d <= H2_SMALL_DEG && S <= h2_maxw(2)) {
        cls = S <= h2_maxw(0) ? 0 : S <= h2_maxw(1) ? 1 : 2;
        lo = cls == 0 ? 0 : h2_maxw(cls - 1);
        wb = (int)((int64_t)(S - lo) * H2_WB / (h2_maxw(cls) - lo + 1));
    } else if (S <= h2_maxw(3) && d + S / 4 <= h2_keycap(3)) {
        cls = 3;
        wb = S < 2048 ? 0 : S < 4096 ? 1 : S < 6144 ? 2 : 3;
    } else {
        cls = 4;
        nparts = h2_parts_for(d, S, false);
        wb = S < 16384 ? 0 : S < 32768 ? 1 : S < 65536 ? 2 : 3;
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
    int2 ru = make_int2(0, 0);
    if (u < g.n) {
        ru = g.rowinfo[u];
        const int d = ru.y;
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
        if (tot > (c == 0 ? L.cap[0] : c == 1 ? L.cap[1] : c == 2 ? L.cap[2] : c == 3 ? L.cap[3] : L.cap[4])) {  // the list cannot hold them: nothing of this class runs, the pass is redone elsewhere
            res->h2_status = 1;
            tot = 0;
        }
        res->h2_count[c] = (int)tot;
    }
    __syncthreads();
    if (bkt >= 0) {
        const int64_t first = (int64_t)blk_base[bkt] + my_off;
        // (selects, not L.units[cls]: indexing a kernel argument with a run-time value puts the whole struct into scratch)
        int4 *dst = cls == 0 ? L.units[0] : cls == 1 ? L.units[1] : cls == 2 ? L.units[2] : cls == 3 ? L.units[3] : L.units[4];
        const int64_t cap = cls == 0 ? L.cap[0] : cls == 1 ? L.cap[1] : cls == 2 ? L.cap[2] : cls == 3 ? L.cap[3] : L.cap[4];
        static_assert(H2_CLASSES == 5, "the selects above list the classes");
        if (nparts <= 65535 && first >= 0 && first + nunits <= cap) {
            for (int j = 0; j < nunits; ++j) dst[first + j] = make_int4(u, (nparts << 16) | j, ru.x, ru.y);
        } else if (nparts > 65535) {
            res->h2_status = 1;
        }  // (else: reported through h2_count / h2_status above)
    }
}

// (round 4: also zeroes the weights — a fill launch of its own before)
__global__ void __launch_bounds__(256) k_h2_clear(DevResult *res, int32_t *weight, int64_t n, unsigned *dirty_words, int64_t n_dirty_words) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) weight[i] = 0;
    // (the incremental pass's per-node flags: a full pass recomputes every edge, so they can go now — a fill launch at the
    //  tail of every pass before)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_dirty_words; i += (int64_t)gridDim.x * 256) dirty_words[i] = 0u;
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0 && n_dirty_words > 0) res->touched_n = 0;   // (the list of flagged nodes goes with the flags)
    if (threadIdx.x < 8) res->misc[threadIdx.x] = 0;
    if (threadIdx.x < H2_NB) {
        res->h2_bucket[threadIdx.x] = 0;
        res->h2_fill[threadIdx.x] = 0;
    }
    if (threadIdx.x < H2_CLASSES) res->h2_count[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        res->h2_status = 0;
        res->h2_retry = 0;
        for (int p = 0; p < 2; ++p) res->h2_ntask[p] = res->h2_ncand[p] = res->h2_npart[p] = res->h2_ncand_done[p] = 0;
        for (int c = 0; c < 6; ++c) res->h2_failed[c] = 0;
        res->flag_too_big = 0;
    }
}

// the records of the nodes on the retry list start from zero again (a split node's partitions add into them)
__global__ void __launch_bounds__(256) k_h2_retry_zero(View g, const int4 *units, const int32_t *count, int64_t unit_cap, uint4 *rec) {
    const int total = *count;
    if (total <= 0 || total > unit_cap) return;
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int4 un = units[it];
        if ((un.y & 0xFFFF) != 0 || un.x < 0 || un.x >= g.n) continue;
        const int2 ru = g.rowinfo[un.x];
        if (!row_ok(g, ru, 42, un.x, (int)it)) continue;
        for (int i = threadIdx.x; i < ru.y; i += 256) rec[(int64_t)ru.x + i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---- join the two records of every edge and evaluate the closing expression ------------------------------------------
// Round 4: a fixed grid walks the slots and every WAVE leaves the first minimum and the first maximum (value, slot) of what
// it wrote: the arg-min that follows every pass of the SDRF loop (sdrf_no_cuda.py:27) and the stale arg-max of its removal
// step (:57-61) are then one small reduction each instead of two more sweeps over the slots.  (Per wave, not per workgroup:
// with a barrier at the end a workgroup holds its place until its slowest wave's gathers are back — measured, the kernel
// took 137-268 us instead of 60, growing with the number of workgroups.)  The partner slot's record is only fetched by the
// lanes whose slot holds a value (neighbour id above the row id) — under the execution mask, not from a clamped index:
// half the lanes of every wave asking for element 0 of five arrays is a hot spot of its own.
#ifndef H2_FINAL_Q
#define H2_FINAL_Q 1
#endif
#ifndef H2_FINAL_BLOCKS
#define H2_FINAL_BLOCKS 1024   // (x 4 waves = 4,096 pairs of partial extrema: the reduction that follows is one workgroup)
#endif
// `when` = 0: the pass was launched WITHOUT its retry stage (no pass of this graph has needed it so far: three launches and a
// stream join less on the critical path of every pass); should the retry list turn out not to be empty, nothing is written,
// status 3 is left and the host runs the pass again with the stage.  `when` = 1: behind the retry stage, unconditionally.
__global__ void __launch_bounds__(256) k_h2_final(View g, const uint4 *rec, double *curv, const int32_t *status, Ext *part_min,
                                                  Ext *part_max, const int32_t *retry_count, int when) {
    if (when == 0 && *retry_count > 0) {  // uniform (the count is final once the class kernels are through)
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicCAS(const_cast<int32_t *>(status), 0, 3);
        return;
    }
    double lo_v = 0.0, hi_v = 0.0;
    int lo_s = -1, hi_s = -1;
    const bool live = *status == 0;  // (else some records are missing: the whole pass is redone by the node-centric kernels)
    const int64_t stride = (int64_t)gridDim.x * 256;
    constexpr int Q = H2_FINAL_Q;
    for (int64_t s0 = (int64_t)blockIdx.x * 256 + threadIdx.x; live && s0 < g.cap_total; s0 += Q * stride) {
        // Round one: everything addressed by the slot itself (its row, its neighbour, its record — all in bounds for every
        // slot, slack included).  Round two, for the slots that may hold a value: the two row headers, the partner slot's
        // neighbour, row and record.  Validity is decided afterwards.
        int u[Q], v[Q];
        uint4 a[Q];
        bool want[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t s = s0 + q * stride;
            const bool in = s < g.cap_total;
            u[q] = in ? g.slot_row[s] : -1;
            v[q] = in ? g.col[s] : -1;
            a[q] = in ? rec[s] : make_uint4(0u, 0u, 0u, 0u);  // from u's side: statistics over N(v) \ N(u) \ {u}
        }
        int2 ru[Q];
        int dv[Q], rcol[Q], rrow[Q];
        uint4 b[Q];
        bool rok[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            // the value lives at the slot whose neighbour id exceeds the row id
            want[q] = u[q] >= 0 && u[q] < g.n && v[q] > u[q] && v[q] < g.n;
            const int64_t r = (int64_t)a[q].w;
            rok[q] = want[q] && r >= 0 && r < g.cap_total;
            ru[q] = make_int2(0, 0);
            dv[q] = 0;
            rcol[q] = rrow[q] = -1;
            b[q] = make_uint4(0u, 0u, 0u, 0u);
            if (want[q]) ru[q] = g.rowinfo[u[q]];   // (consecutive slots share it; v's degree comes with the partner record)
            if (rok[q]) {
                // (the partner slot's record points back at this slot — its own sweep found u in v's row where this one found v
                //  in u's: checked below in place of reading the partner slot's row and neighbour, two more random reads per
                //  edge of the four this kernel made; slot 0 alone keeps the old check, a record of zeros would point at it)
                b[q] = rec[r];  // from v's side: statistics over N(u) \ N(v) \ {v}
                dv[q] = (int)(b[q].z >> 16);
                if (s0 + q * stride == 0) {
                    rcol[q] = g.col[r];
                    rrow[q] = g.slot_row[r];
                } else {
                    rcol[q] = (int64_t)b[q].w == s0 + q * stride ? u[q] : -1;
                    rrow[q] = v[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t s = s0 + q * stride;
            if (!want[q]) continue;
            if (s < ru[q].x || (int)(s - ru[q].x) >= ru[q].y) continue;  // slack behind the row
            double val;
            if (ru[q].y == 1 || (rok[q] && dv[q] == 1)) {  // bfc_naive.py:18-19
                val = 0.0;
            } else {
                if (!rok[q] || rcol[q] != u[q] || rrow[q] != v[q]) {
                    row_ok(g, make_int2(-1, (int)a[q].w), 39, u[q], v[q]);
                    continue;
                }
                if ((a[q].z & 0xFFFFu) != (b[q].z & 0xFFFFu) || (int)(a[q].z >> 16) != ru[q].y) {  // both sides count the same triangles
                    row_ok(g, make_int2(-1, (int)a[q].z), 40, u[q], v[q]);
                    continue;
                }
                const int gam = (int)(a[q].y > b[q].y ? a[q].y : b[q].y);
                val = bfc_formula(ru[q].y, dv[q], (int)(a[q].z & 0xFFFFu), (int)b[q].x, (int)a[q].x, gam);
            }
            curv[s] = val;
            ext_take(lo_v, lo_s, val, (int)s, 0);
            ext_take(hi_v, hi_s, val, (int)s, 1);
        }
    }
    ext_wave_reduce(lo_v, lo_s, 0);
    ext_wave_reduce(hi_v, hi_s, 1);
    if ((threadIdx.x & 63) == 0) {
        const int w = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
        part_min[w] = ext_make(lo_v, lo_s);
        part_max[w] = ext_make(hi_v, hi_s);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// the pools of the triangle step were too small for the last pass: remember what it asked for (the counters kept counting)
bool h2_grow_pools(dcr_graph *g) {
    const DevResult &h = *g->hres;  // (per pool; both pools have the same size)
    const int64_t t = std::max(h.h2_ntask[0], h.h2_ntask[1]), c = std::max(h.h2_ncand[0], h.h2_ncand[1]), p = std::max(h.h2_npart[0], h.h2_npart[1]);
    if (h.h2_ntask[0] < 0 || h.h2_ntask[1] < 0 || h.h2_ncand[0] < 0 || h.h2_ncand[1] < 0 || h.h2_npart[0] < 0 || h.h2_npart[1] < 0) return false;
    if (t < 0 || c < 0 || p < 0) return false;  // the 31-bit counters wrapped: not a case for this engine
    // (a pass that ran out of one pool stopped listing into the others too: ask for twice what it counted)
    g->h2_want[0] = std::max<int64_t>(g->h2_want[0], 2 * t + 4096);
    g->h2_want[1] = std::max<int64_t>(g->h2_want[1], 2 * c + 65536);
    g->h2_want[2] = std::max<int64_t>(g->h2_want[2], 2 * p + 65536);
    return true;
}

// Which engine takes a full Balanced Forman pass (DCR_PASS=h2 / nc force one).  Round 4: two fitted cost models instead of two
// thresholds from one graph family.  tools/probe_engine_choice.py times both engines on 31 graphs of four families
// (preferential attachment m = 2 / 5 / 10 / 20 at 2k-500k nodes, uniform random graphs of mean degree 6-20, grids, a dense
// random graph; profiles/r04_engine_choice.txt) and the pass times (ms, MI355X) are, within 11-15 % on average,
//     node-centric:  0.127 + 0.438e-6 E + 1.135e-9 E s + 0.201 min(dmax, 400) / 400
//     two-hop:       0.120 + 1.193e-6 n + 4.498e-9 (sum d^2) (1 + 60 s / n)          (0.190 until round 5)
//     edge by edge:  0.012 + E (5.0e-6 + 4.2e-9 s)                                   (round 5, csrc/dcr_bfc_nc.hip)
// with E edges, n nodes, s = sum d^2 / n (the mean size of a 2-hop neighbourhood), dmax the largest degree: the node-centric
// engine streams about s entries per edge and loses a tenth of a millisecond to the tail of its hub units; the two-hop
// engine reads sum d^2 entries per pass, pays per node, and slows down as neighbourhoods overlap (s / n: the share of the
// graph a 2-hop neighbourhood covers — repeated keys, fuller tables, more partitions).  The two-hop engine is taken when
// its estimate is the lower one, and never for graphs under 3,000 nodes (both engines are launch-bound there and the
// node-centric one has fewer launches), for s / n above 0.045 (measured 1.5-5 x slower there) or hubs beyond its tables.
bool h2_can_take(const dcr_graph *g, int curv_type, bool incremental) {
    if (curv_type != DCR_CURV_BFC || incremental || g->max_deg_bound > H2_MAXDEG || g->cap_total >= (int64_t)1 << 30) return false;
    if (g->pass_impl == 3) return true;
    if (g->pass_impl != 0 || g->n < 3000) return false;
    static const double max_share = getenv("DCR_H2_MAX_SHARE") ? atof(getenv("DCR_H2_MAX_SHARE")) : 0.045;
    const double n = (double)g->n, sd2 = g->sum_deg2, s = sd2 / n, share = s / n;
    if (share > max_share) return false;
    // (round 5: the fixed cost of a two-hop pass went from 0.19 to about 0.10 ms with the three-stream layout — re-fitted, kept a
    //  little above the measurements: on a 500 k-node graph of two edges per node the per-node cost is underestimated — and small
    //  graphs have a third candidate, a workgroup per edge: nc_edges_full_ms)
    static const bool edges_on = !(getenv("DCR_NC_FINE") && atoi(getenv("DCR_NC_FINE")) == 0);
    const double t_nc = nc_class_full_ms(g);
    const double t_h2 = 0.120 + 1.193e-6 * n + 4.498e-9 * sd2 * (1.0 + 60.0 * share);
    if (edges_on && nc_edges_full_ms(g) < t_h2) return false;
    return t_h2 < t_nc;
}

static int ensure_h2(dcr_graph *g) {
    DCR_TRY(dev_regrow(&g->h2_weight, &g->h2_weight_cap, g->n + 64));
    DCR_TRY(dev_regrow(&g->h2_rec, &g->h2_rec_cap, g->cap_total + 64));
    // edge set: at most cap_total / 2 undirected edges, load <= 1/2
    int bits = 10;
    while ((1ll << bits) < g->cap_total) ++bits;
    if (g->h2_eset_bits != bits || !g->h2_eset) {
        if (g->h2_eset) (void)hipFree(g->h2_eset);
        g->h2_eset = nullptr;
        DCR_TRY(dev_alloc(&g->h2_eset, (int64_t)1 << bits));
        g->h2_eset_bits = bits;
        g->h2_eset_valid = false;
    }
    int bbits = 16;
    while ((1ll << bbits) < 4 * g->cap_total) ++bbits;
    if (g->h2_bloom_bits != bbits || !g->h2_bloom) {
        if (g->h2_bloom) (void)hipFree(g->h2_bloom);
        g->h2_bloom = nullptr;
        DCR_TRY(dev_alloc(&g->h2_bloom, ((int64_t)1 << bbits) / 32));
        g->h2_bloom_bits = bbits;
        g->h2_eset_valid = false;
    }
    const int64_t need[H2_CLASSES] = {g->n + 64, g->n + 64, g->n + 64, g->n + 64, g->n + g->cap_total / 4 + 64};
    for (int c = 0; c < H2_CLASSES; ++c) DCR_TRY(dev_regrow(&g->h2_units[c], &g->h2_units_cap[c], need[c]));
    DCR_TRY(dev_regrow(&g->h2_retry, &g->h2_retry_cap, g->n + g->cap_total / 4 + 64));
    // triangle step pools: tasks are edges (with the partitions of split nodes: a few times that), candidates and partners
    // adjacency entries of theirs
    // (sized for power-law graphs; a pass that needs more says how much — h2_grow_pools — and is run again)
    // (two pools of each kind, h2_*_cap counts both: see DevResult)
    DCR_TRY(dev_regrow(&g->h2_task, &g->h2_task_cap, 2 * std::max<int64_t>(g->cap_total + 4096, g->h2_want[0])));
    DCR_TRY(dev_regrow(&g->h2_cand, &g->h2_cand_cap, 2 * std::max<int64_t>(2 * g->cap_total + 65536, g->h2_want[1])));
    DCR_TRY(dev_regrow(&g->h2_part, &g->h2_part_cap, 2 * std::max<int64_t>(g->cap_total + 65536, g->h2_want[2])));
    // the third step's per-wave lists: the split class (one workgroup of 16 waves per CU; the retry launch reuses its part)
    // and class M (three workgroups of 4 waves per CU)
    static const bool use_lists = !(getenv("DCR_H2_LISTS") && atoi(getenv("DCR_H2_LISTS")) == 0);
    if (use_lists) {
        const int cus = g->num_cu > 0 ? g->num_cu : 256;
        DCR_TRY(dev_regrow(&g->h2_lists, &g->h2_lists_cap, (int64_t)cus * (16 + 12) * H2_LIST_WORDS));
    }
    return DCR_OK;
}

// Share of its full-chip grid each class kernel is launched with, in percent: {split class, class M, wave classes 2, 1, 0}.
// All five are persistent kernels that deal their units out up front; each sized for the whole chip, the ones launched first
// hold the LDS and the others' workgroups queue behind them (round 3 timeline: the fifth kernel started 0.7 ms late).
// DCR_H2_SHARE="l,m,s2,s1,s0" overrides (tuning aid, read at every pass).
static int h2_share(int idx, int64_t n_nodes) {
    static const int small_graph[5] = {100, 100, 100, 100, 100};
    static const int large_graph[5] = {100, 100, 100, 100, 100};
    int v = (n_nodes >= 400000 ? large_graph : small_graph)[idx];
    if (const char *e = getenv("DCR_H2_SHARE")) {
        int a[5];
        if (sscanf(e, "%d,%d,%d,%d,%d", &a[0], &a[1], &a[2], &a[3], &a[4]) == 5 && a[idx] > 0 && a[idx] <= 400) v = a[idx];
    }
    return v;
}

template <int C>
static void launch_h2_small(dcr_graph *g, const View &vw, const H2Retry &rt, hipStream_t st) {
    constexpr int H2_WPB = h2_wpb(C);
    constexpr int NP = h2_maxw(C) / 256 + 1;  // 64-piece steps: W / 4 pieces and at most one more per row (alignment)
    auto kern = k_h2_small<h2_l1(C), h2_exs(C), h2_clcap(C), H2_WPB, NP>;
    // workgroups a CU holds at once (LDS and registers): the units are dealt to the workgroups up front (grid-stride), so a
    // workgroup that has to wait for a place still has its whole share ahead of it — launch what is resident, no more
    static int per_cu_res = 0;
    if (per_cu_res == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * H2_WPB, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = (160 * 1024) / (H2_WPB * (int)sizeof(H2Small<h2_l1(C), h2_exs(C), h2_clcap(C)>));
            if (nb > 32 / H2_WPB) nb = 32 / H2_WPB;
        }
        per_cu_res = nb < 1 ? 1 : nb;
    }
    int per_cu = per_cu_res;
    int64_t grid = (int64_t)g->num_cu * per_cu;
    static const int64_t cap = getenv("DCR_H2_GRID") ? atoll(getenv("DCR_H2_GRID")) : 0;  // tuning aid: workgroups per CU
    if (cap > 0) grid = cap * g->num_cu;
    grid = grid * h2_share(4 - C, g->n) / 100;
    const int64_t units = g->h2_last_count[C] >= 0 ? (int64_t)g->h2_last_count[C] + g->h2_last_count[C] / 32 + 8 : g->n;
    if (grid > (units + H2_WPB - 1) / H2_WPB) grid = (units + H2_WPB - 1) / H2_WPB;  // small graphs: no idle workgroups
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * H2_WPB), 0, st, vw,
                       g->h2_units[C], &g->dres->h2_count[C], g->h2_units_cap[C], g->h2_rec, rt);
}

template <int C, bool PARTS>
static void launch_h2_block(dcr_graph *g, const View &vw, const H2Tasks &tk, const H2Retry &rt, const int4 *units,
                            const int32_t *count, int64_t cap, int64_t units_hint, int is_retry, hipStream_t st) {
    int64_t grid = units_hint;
    int64_t most = (int64_t)g->num_cu * (C == 3 ? 3 : 1);
    if (!is_retry) most = most * h2_share(C == 3 ? 1 : 0, g->n) / 100;
    if (grid > most) grid = most;
    if (grid < 1) grid = 1;
    H2Tasks tkl = tk;
    if (grid > (int64_t)g->num_cu * (C == 3 ? 3 : 1)) tkl.lists = nullptr;  // (more workgroups than the list pool has places for)
    hipLaunchKernelGGL((k_h2_block<h2_l1(C), h2_exs(C), h2_waves(C), PARTS>), dim3((unsigned)grid), dim3(64 * h2_waves(C)), 0,
                       st, vw, tkl, units, count, cap, g->h2_rec, rt, is_retry);
}

#ifdef H2_UNIT_TIMES
static void h2_print_unit_times(dcr_graph *g) {
    (void)hipStreamSynchronize(g->stream);
    (void)hipDeviceSynchronize();
    static int4 h[2 * H2_UT_CAP];
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(h2_ut_n), sizeof(n));
    if (n > H2_UT_CAP) n = H2_UT_CAP;
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(h2_ut), sizeof(int4) * 2 * n);
    static int4 h2[H2_UT_CAP];
    (void)hipMemcpyFromSymbol(h2, HIP_SYMBOL(h2_ut2), sizeof(int4) * n);
    std::vector<int2> ri(g->n);
    std::vector<int32_t> wt(g->n);
    (void)hipMemcpy(ri.data(), g->rowinfo, sizeof(int2) * g->n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(wt.data(), g->h2_weight, sizeof(int32_t) * g->n, hipMemcpyDeviceToHost);
    for (int cls = 0; cls < 2; ++cls) {
        std::vector<unsigned> v;
        double sum = 0, ph[4] = {0, 0, 0, 0};
        long long first = -1, last = 0;
        for (unsigned i = 0; i < n; ++i)
            if (((h[2 * i].y >> 30) & 1) == cls) {
                v.push_back(i);
                sum += h[2 * i].z;
                ph[0] += h[2 * i + 1].x; ph[1] += h[2 * i + 1].y; ph[2] += h[2 * i + 1].z; ph[3] += h[2 * i + 1].w;
                const long long st = (long long)h[2 * i].w * 16, en = st + h[2 * i].z;
                if (first < 0 || st < first) first = st;
                if (en > last) last = en;
            }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end(), [&](unsigned a, unsigned b) { return h[2 * a].z > h[2 * b].z; });
        fprintf(stderr, "[h2 units] class %s: %zu units; first start to last end %.0f kticks (= the kernel); sum of unit times %.0f kticks; phases (sum, kticks): clear + seed %.0f, "
                "sweep A %.0f, sweep B %.0f, third step %.0f\n", cls ? "L (split)" : "M", v.size(), (last - first) / 1e3, sum / 1e3, ph[0] / 1e3, ph[1] / 1e3, ph[2] / 1e3, ph[3] / 1e3);
        fprintf(stderr, "    unit time percentiles (kticks): 50%% %.1f  90%% %.1f  99%% %.1f  max %.1f\n", h[2 * v[v.size() / 2]].z / 1e3, h[2 * v[v.size() / 10]].z / 1e3,
                h[2 * v[v.size() / 100]].z / 1e3, h[2 * v[0]].z / 1e3);
        for (size_t i = 0; i < v.size() && i < 8; ++i) {
            const int4 a = h[2 * v[i]], b = h[2 * v[i] + 1];
            fprintf(stderr, "    node %6d deg %5d W %8d part %d/%d: %7.1f kticks (clear + seed %.1f, A %.1f, B %.1f, third %.1f); wave 0 in B: %d drains of %d items, %.1f kticks\n", a.x, ri[a.x].y,
                    wt[a.x] & 0x7FFFFFFF, a.y & 0xFFFF, (a.y >> 16) & 0x3FFF, a.z / 1e3, b.x / 1e3, b.y / 1e3, b.z / 1e3, b.w / 1e3, h2[v[i]].y, h2[v[i]].z, h2[v[i]].x / 1e3);
        }
    }
}
#endif

// DCR_H2_DEBUG: the counters of the pass just enqueued (synchronises)
static int h2_debug_print(dcr_graph *g) {
    DevResult h;
    DCR_HIP(hipStreamSynchronize(g->stream));
    DCR_HIP(hipMemcpy(&h, g->dres, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[h2] units per class %d %d %d %d %d, retry units %d (stage %s), status %d, failed per class %d %d %d %d %d retry %d\n",
            h.h2_count[0], h.h2_count[1], h.h2_count[2], h.h2_count[3], h.h2_count[4], h.h2_retry, g->h2_expect_retry ? "on" : "off", h.h2_status,
            h.h2_failed[0], h.h2_failed[1], h.h2_failed[2], h.h2_failed[3], h.h2_failed[4], h.h2_failed[5]);
    fprintf(stderr, "[h2] triangle step: split class + retry %d tasks, %d candidates, %d partners; class M %d, %d, %d (pool slots, chunk tails "
            "included; pools of %lld %lld %lld each)\n", h.h2_ntask[0], h.h2_ncand[0], h.h2_npart[0], h.h2_ntask[1], h.h2_ncand[1], h.h2_npart[1],
            (long long)g->h2_task_cap / 2, (long long)g->h2_cand_cap / 2, (long long)g->h2_part_cap / 2);
    return DCR_OK;
}

int launch_curvature_pass_h2(dcr_graph *g) {
    if (g->num_cu <= 0) {
        g->num_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0)
            g->num_cu = prop.multiProcessorCount;
    }
    DCR_TRY(ensure_h2(g));
    View vw{g->rowinfo, g->col, g->slot_row, g->cap_total, g->dres->misc, nullptr, (int32_t)g->n, 0, nullptr};
    H2Lists L;
    for (int c = 0; c < H2_CLASSES; ++c) {
        L.units[c] = g->h2_units[c];
        L.cap[c] = g->h2_units_cap[c];
    }
    int32_t *status = &g->dres->h2_status;
    const H2Retry rt{g->h2_retry, g->h2_retry_cap, g->h2_weight, g->dres};
    const int64_t tcap = g->h2_task_cap / 2, ccap = g->h2_cand_cap / 2, pcap = g->h2_part_cap / 2;
    DevResult *dr = g->dres;
    unsigned *lists_L = g->h2_lists, *lists_M = g->h2_lists ? g->h2_lists + (int64_t)g->num_cu * 16 * H2_LIST_WORDS : nullptr;
    H2Tasks tk{g->h2_task, g->h2_cand, g->h2_part, tcap, ccap, pcap, &dr->h2_ntask[0], &dr->h2_ncand[0], &dr->h2_npart[0],
               &dr->h2_ncand_done[0], g->dres, g->h2_weight, 0u, lists_L};  // pool 0: the split class and the retry launch
    const H2Tasks tkM{g->h2_task + tcap, g->h2_cand + ccap, g->h2_part + pcap, tcap, ccap, pcap, &dr->h2_ntask[1], &dr->h2_ncand[1],
                      &dr->h2_npart[1], &dr->h2_ncand_done[1], g->dres, g->h2_weight, 0u, lists_M};  // pool 1: class M
    {
        const int64_t cb = (g->n + 255) / 256;
        // (dirty: n bytes, allocated in whole words? — only the whole words are zeroed here, the caller's fill takes the rest)
        const bool whole = g->n % 4 == 0;
        hipLaunchKernelGGL(k_h2_clear, dim3((unsigned)(cb < 1 ? 1 : cb > 2048 ? 2048 : cb)), dim3(256), 0, g->stream, g->dres, g->h2_weight,
                           g->n, reinterpret_cast<unsigned *>(g->dirty), whole ? g->n / 4 : (int64_t)0);
        g->h2_cleared_dirty = whole;
    }
    static const bool serial = getenv("DCR_SERIAL_BINS") != nullptr;
    const int64_t sblocks = (g->cap_total + 255) / 256;
    const H2EdgeSet es{g->h2_eset, g->h2_eset_bits, g->h2_bloom, g->h2_bloom_bits};
    if (sblocks > 0) hipLaunchKernelGGL(k_h2_weight, dim3((unsigned)sblocks), dim3(256), 0, g->stream, vw, g->h2_weight);
    const int64_t pblocks = (g->n + H2_PLAN_THREADS - 1) / H2_PLAN_THREADS;
    if (pblocks > 0) {
        hipLaunchKernelGGL(k_h2_plan<0>, dim3((unsigned)pblocks), dim3(H2_PLAN_THREADS), 0, g->stream, vw, g->h2_weight, L,
                           g->dres);
        hipLaunchKernelGGL(k_h2_plan<1>, dim3((unsigned)pblocks), dim3(H2_PLAN_THREADS), 0, g->stream, vw, g->h2_weight, L,
                           g->dres);
    }
    // records of split nodes are accumulated with atomics: start from zero
    hipLaunchKernelGGL(k_h2_retry_zero, dim3(1024), dim3(256), 0, g->stream, vw, g->h2_units[4], &g->dres->h2_count[4], g->h2_units_cap[4],
                       g->h2_rec);
    // Round 5: a layout with FEWER streams (DCR_H2_LAYOUT="l,m,s2,s1,s0": the stream, 0 = main, 1-3 = side streams, each class
    // kernel is launched on; kernels on one stream run one after the other).  Five persistent full-chip kernels on five streams
    // do not run five abreast anyway (the runtime maps streams onto four hardware queues, and the split class's workgroups
    // hold 125 KB of a CU's LDS), and every stream the closing kernel has to join costs a barrier packet of 10-15 us on the
    // critical path (the 48-75 us hole in front of k_h2_final in the round-4 timelines).
#ifdef H2_UNIT_TIMES
    hipLaunchKernelGGL(k_h2_ut_mark, dim3(1), dim3(1), 0, g->stream);
#endif
    // Measured (profiles/r05_layouts.txt, interleaved rounds, pass ms): S100k five streams 1.020, "0,1,2,2,1" 0.962, "0,1,1,2,2"
    // 0.962, "0,1,2,2,2" 0.984, "0,1,2,3,3" 0.983, "0,1,1,1,1" 1.094; S1M five streams 10.10, "0,1,2,2,1" 10.70 — the fixed
    // costs do not matter there and five abreast packs the chip better.  Default: three streams below 400k nodes.
    static const char *layout_env = getenv("DCR_H2_LAYOUT");
    const bool layout_default = !layout_env && g->n < 400000;
    if ((layout_default || (layout_env && strcmp(layout_env, "five") != 0)) && !serial) {
        int lay[5] = {0, 1, 2, 2, 1};
        int a[5];
        if (layout_env && sscanf(layout_env, "%d,%d,%d,%d,%d", &a[0], &a[1], &a[2], &a[3], &a[4]) == 5)
            for (int c = 0; c < 5; ++c) lay[c] = a[c] < 0 ? 0 : a[c] > 3 ? 3 : a[c];
        hipStream_t pool[4] = {g->stream, g->side[1], g->low[0], g->low[1]};
        hipEvent_t done[4] = {nullptr, g->ev_join[1], g->ev_join[2], g->ev_join[3]};
        bool used[4] = {true, false, false, false};
        DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
        auto on = [&](int c) {
            const int si = lay[c];
            if (!used[si]) {
                used[si] = true;
                (void)hipStreamWaitEvent(pool[si], g->ev_fork, 0);   // (an error surfaces at hipGetLastError below)
            }
            return pool[si];
        };
        const int64_t hint4 = g->h2_last_count[4] >= 0 ? (int64_t)g->h2_last_count[4] + 8 : g->num_cu;
        const int64_t hint3 = g->h2_last_count[3] >= 0 ? (int64_t)g->h2_last_count[3] + 8 : 3 * (int64_t)g->num_cu;
        static const bool keep_eset2 = !(getenv("DCR_H2_ESET_KEEP") && atoi(getenv("DCR_H2_ESET_KEEP")) == 0);
        const bool patch2 = keep_eset2 && g->h2_eset_valid && g->h2_eset_pending <= EDIT_LOG_CAP &&
                            g->h2_eset_tombs + g->h2_eset_pending <= ((int64_t)1 << g->h2_eset_bits) / 16;
        // the edge set first (probed by the triangle steps only): patched by one thread on the main stream ahead of the split
        // class, or rebuilt on the aux stream beside everything
        bool eset_on_aux = false;
        if (patch2) {
            if (g->h2_eset_pending > 0) {
                hipLaunchKernelGGL(k_h2_eset_apply, dim3(1), dim3(64), 0, g->stream, es, g->dres, status);
                g->h2_eset_tombs += g->h2_eset_pending;
                DCR_HIP(hipEventRecord(g->ev_fork, g->stream));   // (the side streams start behind it: their triangle steps probe the set)
            }
        } else {
            DCR_HIP(hipStreamWaitEvent(g->aux, g->ev_fork, 0));
            DCR_HIP(hipMemsetAsync(g->h2_eset, 0xFF, sizeof(unsigned long long) << g->h2_eset_bits, g->aux));
            DCR_HIP(hipMemsetAsync(g->h2_bloom, 0, sizeof(unsigned) * (((size_t)1 << g->h2_bloom_bits) / 32), g->aux));
            if (sblocks > 0) hipLaunchKernelGGL(k_h2_eset_build, dim3((unsigned)sblocks), dim3(256), 0, g->aux, vw, es, status, g->dres);
            g->h2_eset_tombs = 0;
            DCR_HIP(hipEventRecord(g->ev_aux, g->aux));
            eset_on_aux = true;
        }
        g->h2_eset_valid = true;
        g->h2_eset_pending = 0;
        // launch order = the order of the classes in the layout string's streams: split class, M, then the wave classes
        launch_h2_block<4, true>(g, vw, tk, rt, g->h2_units[4], &g->dres->h2_count[4], g->h2_units_cap[4], hint4, 0, on(0));
        launch_h2_block<3, false>(g, vw, tkM, rt, g->h2_units[3], &g->dres->h2_count[3], g->h2_units_cap[3], hint3, 0, on(1));
        // each block class's triangle step right behind it on its own stream
        if (eset_on_aux) DCR_HIP(hipStreamWaitEvent(pool[lay[0]], g->ev_aux, 0));
        hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 4)), dim3(256), 0, pool[lay[0]], es, tk, g->h2_rec, status, 0);
        if (eset_on_aux && lay[1] != lay[0]) DCR_HIP(hipStreamWaitEvent(pool[lay[1]], g->ev_aux, 0));
        hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 8)), dim3(256), 0, pool[lay[1]], es, tkM, g->h2_rec, status, 0);
        launch_h2_small<2>(g, vw, rt, on(2));
        launch_h2_small<1>(g, vw, rt, on(3));
        launch_h2_small<0>(g, vw, rt, on(4));
        for (int si = 1; si < 4; ++si)
            if (used[si]) {
                DCR_HIP(hipEventRecord(done[si], pool[si]));
                DCR_HIP(hipStreamWaitEvent(g->stream, done[si], 0));
            }
        if (!g->ext_part) {
            Ext *p = nullptr;
            DCR_TRY(dev_alloc(&p, 2 * EXT_PART_BLOCKS));
            g->ext_part = p;
        }
        int64_t fb = (sblocks + H2_FINAL_Q - 1) / H2_FINAL_Q;
        if (fb > H2_FINAL_BLOCKS) fb = H2_FINAL_BLOCKS;
        if (fb > EXT_PART_BLOCKS / 4) fb = EXT_PART_BLOCKS / 4;
        if (fb < 1) fb = 1;
        const bool retry2 = g->h2_expect_retry;
        if (retry2) {
            hipLaunchKernelGGL(k_h2_retry_zero, dim3(1024), dim3(256), 0, g->stream, vw, g->h2_retry, &g->dres->h2_retry, g->h2_retry_cap,
                               g->h2_rec);
            tk.retry_flag = 0x80000000u;
            launch_h2_block<4, true>(g, vw, tk, rt, g->h2_retry, &g->dres->h2_retry, g->h2_retry_cap, 64, 1, g->stream);
            hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 2)), dim3(256), 0, g->stream, es, tk, g->h2_rec, status, 1);
        }
        hipLaunchKernelGGL(k_h2_final, dim3((unsigned)fb), dim3(256), 0, g->stream, vw, g->h2_rec, g->curv, status, (Ext *)g->ext_part,
                           (Ext *)g->ext_part + EXT_PART_BLOCKS, &g->dres->h2_retry, retry2 ? 1 : 0);
        g->ext_part_n = (int)fb * 4;
        g->ext_part_valid = true;
        DCR_HIP(hipGetLastError());
#ifdef H2_UNIT_TIMES
        h2_print_unit_times(g);
#endif
        if (getenv("DCR_H2_DEBUG")) DCR_TRY(h2_debug_print(g));
        return DCR_OK;
    }
    // Streams: the block classes (long units; the triangle step waits for them only) on two high-priority streams, the
    // wave classes on three low-priority ones, the edge set on a stream of its own, the triangle step on a fourth
    // high-priority stream as soon as block classes and edge set are done — beside the wave classes, which list nothing.
    hipStream_t sL = g->stream, sM = g->stream, sS2 = g->stream, sS1 = g->stream, sS0 = g->stream, sT = g->stream, sa = g->stream;
    if (!serial) {
        // (the split class stays on the main stream, launched first: its workgroups need most of a CU's LDS and find no
        //  CU free once the other kernels are resident — measured: started 60 us late, finished last)
        sM = g->side[1];
        sT = g->side[3];
        sS2 = g->low[0];
        sS1 = g->low[1];
        sS0 = g->side[2];
        sa = g->aux;
        DCR_HIP(hipEventRecord(g->ev_fork, g->stream));
    }
    // (each stream's wait is enqueued right before its kernel: the GPU is through the preamble before the host is through
    //  these calls, so every call ahead of a launch is time the chip waits)
    const int64_t hint4 = g->h2_last_count[4] >= 0 ? (int64_t)g->h2_last_count[4] + 8 : g->num_cu;
    const int64_t hint3 = g->h2_last_count[3] >= 0 ? (int64_t)g->h2_last_count[3] + 8 : 3 * (int64_t)g->num_cu;
    launch_h2_block<4, true>(g, vw, tk, rt, g->h2_units[4], &g->dres->h2_count[4], g->h2_units_cap[4], hint4, 0, sL);
    if (!serial) DCR_HIP(hipStreamWaitEvent(sM, g->ev_fork, 0));
    launch_h2_block<3, false>(g, vw, tkM, rt, g->h2_units[3], &g->dres->h2_count[3], g->h2_units_cap[3], hint3, 0, sM);
    if (!serial) DCR_HIP(hipStreamWaitEvent(sS2, g->ev_fork, 0));
    launch_h2_small<2>(g, vw, rt, sS2);
    if (!serial) DCR_HIP(hipStreamWaitEvent(sS1, g->ev_fork, 0));
    launch_h2_small<1>(g, vw, rt, sS1);
    if (!serial) DCR_HIP(hipStreamWaitEvent(sS0, g->ev_fork, 0));
    launch_h2_small<0>(g, vw, rt, sS0);
    if (!serial) DCR_HIP(hipStreamWaitEvent(sa, g->ev_fork, 0));
    // the edge set (probed by k_h2_triangles only): kept from the last pass and patched with the journaled edits, or rebuilt
    // (32 MB of fill and a million atomics beside the class kernels: 0.17 ms of chip time per pass on the bench graph)
    static const bool keep_eset = !(getenv("DCR_H2_ESET_KEEP") && atoi(getenv("DCR_H2_ESET_KEEP")) == 0);
    const bool patch = keep_eset && g->h2_eset_valid && g->h2_eset_pending <= EDIT_LOG_CAP &&
                       g->h2_eset_tombs + g->h2_eset_pending <= ((int64_t)1 << g->h2_eset_bits) / 16;
    if (patch) {
        if (g->h2_eset_pending > 0) {
            hipLaunchKernelGGL(k_h2_eset_apply, dim3(1), dim3(64), 0, sa, es, g->dres, status);
            g->h2_eset_tombs += g->h2_eset_pending;
        }
    } else {
        DCR_HIP(hipMemsetAsync(g->h2_eset, 0xFF, sizeof(unsigned long long) << g->h2_eset_bits, sa));
        DCR_HIP(hipMemsetAsync(g->h2_bloom, 0, sizeof(unsigned) * (((size_t)1 << g->h2_bloom_bits) / 32), sa));
        if (sblocks > 0) hipLaunchKernelGGL(k_h2_eset_build, dim3((unsigned)sblocks), dim3(256), 0, sa, vw, es, status, g->dres);
        g->h2_eset_tombs = 0;
    }
    g->h2_eset_valid = true;
    g->h2_eset_pending = 0;
    // The tail.  Without a retry stage (the usual case) everything below runs on the MAIN stream, behind the split class that is
    // already there: its candidates' triangle step, class M's when M is through, the closing kernel when the wave classes are —
    // the critical path (split class -> triangle steps -> closing kernel) crosses no stream.  (Round 3 had the triangle steps on
    // a stream of their own and the main stream joined five streams before the closing kernel: two hops of 15-30 us each.)
    const bool retry_stage = g->h2_expect_retry;
    if (!serial && !retry_stage) sT = g->stream;
    if (!serial) {
        DCR_HIP(hipEventRecord(g->ev_aux, sa));
        DCR_HIP(hipEventRecord(g->ev_join[1], sM));
        if (sT != sL) {
            DCR_HIP(hipEventRecord(g->ev_join[0], sL));
            DCR_HIP(hipStreamWaitEvent(sT, g->ev_join[0], 0));
        }
        DCR_HIP(hipStreamWaitEvent(sT, g->ev_aux, 0));
    }
    // the split class's candidates as soon as IT is done (beside class M and the wave classes), class M's when M is — on M's own
    // stream: behind the split class's triangle step (292 us beside the other classes, 60 alone) it started 90 us after M's end
    // and sat on the critical path (DCR_H2_TRI_OWN=0: the round-4 order)
    static const bool tri_own = !(getenv("DCR_H2_TRI_OWN") && atoi(getenv("DCR_H2_TRI_OWN")) == 0);
    hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 4)), dim3(256), 0, sT, es, tk, g->h2_rec, status, 0);
    if (!serial && tri_own) {
        DCR_HIP(hipStreamWaitEvent(sM, g->ev_aux, 0));
        hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 8)), dim3(256), 0, sM, es, tkM, g->h2_rec, status, 0);
        DCR_HIP(hipEventRecord(g->ev_join[1], sM));
        DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[1], 0));
    } else {
        if (!serial) DCR_HIP(hipStreamWaitEvent(sT, g->ev_join[1], 0));
        hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 8)), dim3(256), 0, sT, es, tkM, g->h2_rec, status, 0);
    }
    if (!serial) {
        if (sT != g->stream) DCR_HIP(hipEventRecord(g->ev_join[3], sT));
        DCR_HIP(hipEventRecord(g->ev_join[2], sS0));
        DCR_HIP(hipEventRecord(g->ev_aux2, sS1));
        DCR_HIP(hipEventRecord(g->ev_fork, sS2));
        if (sT != g->stream) DCR_HIP(hipStreamWaitEvent(g->stream, g->ev_join[3], 0));
        for (hipEvent_t ev : {g->ev_join[2], g->ev_aux2, g->ev_fork}) DCR_HIP(hipStreamWaitEvent(g->stream, ev, 0));
    }
    if (!g->ext_part) {
        Ext *p = nullptr;
        DCR_TRY(dev_alloc(&p, 2 * EXT_PART_BLOCKS));
        g->ext_part = p;
    }
    int64_t fblocks = (sblocks + H2_FINAL_Q - 1) / H2_FINAL_Q;
    if (fblocks > H2_FINAL_BLOCKS) fblocks = H2_FINAL_BLOCKS;
    if (fblocks > EXT_PART_BLOCKS / 4) fblocks = EXT_PART_BLOCKS / 4;  // (a pair of partial extrema per WAVE)
    if (fblocks < 1) fblocks = 1;
    if (retry_stage) {
        // nodes whose tables filled up in their class: zero their records, redo them with worst-case partitions
        hipLaunchKernelGGL(k_h2_retry_zero, dim3(1024), dim3(256), 0, g->stream, vw, g->h2_retry, &g->dres->h2_retry, g->h2_retry_cap,
                           g->h2_rec);
        tk.retry_flag = 0x80000000u;
        launch_h2_block<4, true>(g, vw, tk, rt, g->h2_retry, &g->dres->h2_retry, g->h2_retry_cap, 64, 1, g->stream);
        hipLaunchKernelGGL(k_h2_triangles, dim3((unsigned)(g->num_cu * 2)), dim3(256), 0, g->stream, es, tk, g->h2_rec, status, 1);
    }
    hipLaunchKernelGGL(k_h2_final, dim3((unsigned)fblocks), dim3(256), 0, g->stream, vw, g->h2_rec, g->curv, status, (Ext *)g->ext_part,
                       (Ext *)g->ext_part + EXT_PART_BLOCKS, &g->dres->h2_retry, retry_stage ? 1 : 0);
    g->ext_part_n = (int)fblocks * 4;
    g->ext_part_valid = true;  // (dropped again by the caller if the pass reports a failure, and by every edit)
    DCR_HIP(hipGetLastError());
    static const bool debug = getenv("DCR_H2_DEBUG") != nullptr;
#ifdef H2_UNIT_TIMES
    h2_print_unit_times(g);
#endif
#ifdef H2_PROF
    {
        unsigned long long h[32];
        DCR_HIP(hipStreamSynchronize(g->stream));
        DCR_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(h2_prof), sizeof(h)));
        fprintf(stderr, "[h2 prof] wave-Mcycles: clear %.1f; seed + prefix %.1f; piece rows + loads %.1f; sweep A (+ wait for the loads) %.1f; "
                "sweep B + drains %.1f; step C + publish %.1f\n", h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6, h[4] / 1e6, h[5] / 1e6);
        fprintf(stderr, "[h2 prof] block classes, wave-Mcycles: M: clear + seed %.1f; sweep A %.1f; sweep B %.1f; sweep C + listing %.1f | "
                "L: clear + seed %.1f; sweep A %.1f; sweep B %.1f; sweep C + listing %.1f\n", h[8] / 1e6, h[9] / 1e6, h[10] / 1e6,
                h[11] / 1e6, h[12] / 1e6, h[13] / 1e6, h[14] / 1e6, h[15] / 1e6);
        fprintf(stderr, "[h2 prof] sweep C of both block classes: batch set-up %.1f; pieces + drains %.1f; last drain %.1f; batch end %.1f\n",
                h[16] / 1e6, h[17] / 1e6, h[18] / 1e6, h[19] / 1e6);
        fprintf(stderr, "[h2 prof] class M (register-resident path), third sweep: batch set-up %.1f; flags + queue + drains %.1f; batch end %.1f\n",
                h[20] / 1e6, h[21] / 1e6, h[23] / 1e6);
        unsigned long long z[32] = {0};
        DCR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(h2_prof), z, sizeof(z)));
    }
#endif
    if (debug) DCR_TRY(h2_debug_print(g));
    return DCR_OK;
}

}  // namespace dcr
