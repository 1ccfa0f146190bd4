// The first GCN layer and what follows it up to the second layer's aggregation, in ONE kernel on the matrix cores:
//     pre = (Â·X)·W1ᵀ + b1          (models/gcn.py:36 — GCNConv of the first layer, with Â·X precomputed: a constant of the run)
//     z_train = dropout(relu(pre))·W2ᵀ,  z_eval = relu(pre)·W2ᵀ   (models/gcn.py:38-42 and the next GCNConv's lin)
// Before: the GEMM library wrote pre (N x H floats), dcr_act_linear_fwd_f32_dev read it back (0.60 + 0.17 ms at
// 1M x 256 -> 128 -> 16).  Here the H columns of a row never leave the registers between the two contractions; pre is
// written once (the backward pass needs it) or not at all (evaluation).
//
// Shape of the work on gfx950:
//   * W1 (H x F floats, 128 KB at 128 x 256) sits in LDS for the life of a workgroup — one workgroup per CU, persistent —
//     XOR-swizzled in 16-byte chunks so that every ds_read_b128 is conflict-free (first_chunk).  W2 (<= 16 x H) and b1 sit
//     beside it.
//   * v_mfma_f32_16x16x4_f32, transposed problem: D[hidden column][row] = W1 tile (16 columns x K) · (Â·X)ᵀ (K x 16 rows).
//     A operand: lane (i = lane & 15, g = lane >> 4) supplies W1[16t + i][k], B operand: (Â·X)[row i][k], where MFMA step
//     (m, q) takes the hardware's k index g to be input column 16m + 4g + q: both operands are the four floats a lane gets
//     from ONE 16-byte load (global for Â·X: 16 rows x 64 bytes per instruction; LDS for W1).  The accumulator register r
//     of lane (i, g) is then pre[row i][16t + 4g + r]: four consecutive columns of one row — a float4 store, and exactly
//     the operand layout k_act_linear_fwd loads from memory (csrc/dcr_gcn.hip), so the epilogue is that kernel's body:
//     same Philox counters, same keep-bit words, same order of operations in the second contraction.
//   * a wave owns a contiguous range of 16-row units and walks it NR units at a time (NR x H/16 accumulator tiles of 4
//     registers); Â·X is read exactly once from HBM, W1 only from LDS (one ds_read_b128 per 4 NR MFMAs).  Two waves per
//     SIMD: one wave's epilogue (Philox on the vector ALU) runs under the other's MFMAs.
// Roofline: MFMA f32 (2·N·F·H + 2·2·N·H·16 flop); HBM traffic N·(4F + 4H + 8·16) bytes is 0.2 ms at 1M rows.
#include "dcr_internal.h"
#include "dcr_philox.h"

namespace dcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DCR_FIRST_NR
#define DCR_FIRST_NR 2   // 16-row units a wave carries at once
#endif
#ifndef DCR_FIRST_WAVES
#define DCR_FIRST_WAVES 8
#endif
constexpr int FIRST_WAVES = DCR_FIRST_WAVES;

struct FirstArgs {
    const float *ax; int64_t ldx;
    float *pre, *z_train, *z_eval; int64_t ldz;
    unsigned long long *bits;
    int64_t n_rows;
    int F, C;
    float scale; uint32_t threshold; uint64_t seed, offset;
    const unsigned long long *dwords;   // the dropout decisions of this call drawn beforehand (dcr_dropout_words_dev; layout of `bits`), or null
};

// the lane's pointer into row (unit·16 + i) of Â·X (rows past the end read row 0: columns of the MFMA are independent, nothing
// of them is stored)
__device__ __forceinline__ const float *first_row_ptr(const FirstArgs &A, int64_t unit, int i, int g) {
    const int64_t row = unit * 16 + i;
    return A.ax + (row < A.n_rows ? row : 0) * A.ldx + 4 * g;
}

template <int CNT>
__device__ __forceinline__ void first_load(const FirstArgs &A, int64_t u, int i, int g, float4 (&a0)[CNT]) {
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) a0[rt] = *reinterpret_cast<const float4 *>(first_row_ptr(A, u + rt, i, g));
}

// The stamp behind the drawn decisions (k_dropout_words): the call they were drawn for.  A call whose (offset, seed, threshold,
// rows) differ — the counter moved because something else drew from it, another seed — draws in line as if it had been
// given none: correctness does not depend on the caller's discipline, only the time does.
template <int HM>
__device__ __forceinline__ bool dropout_words_current(const FirstArgs &A) {
    constexpr int RPW = 64 / (4 * HM);
    const unsigned long long *stamp = A.dwords + (A.n_rows + RPW - 1) / RPW * 4;
    return stamp[0] == A.offset && stamp[1] == A.seed && stamp[2] == (unsigned long long)A.threshold && stamp[3] == (unsigned long long)A.n_rows;
}
// ... and, in the word after the stamp, the offset the NEXT call is expected to have: where the caller points the next
// dcr_dropout_words_dev (its offset_dev), so that drawing ahead reads no counter another stream may be moving.
template <int HM>
__device__ __forceinline__ void dropout_words_next(const FirstArgs &A) {
    constexpr int RPW = 64 / (4 * HM);
    const_cast<unsigned long long *>(A.dwords)[(A.n_rows + RPW - 1) / RPW * 4 + 4] = A.offset + 1;
}

// The epilogue of one 16-row unit: + b1, pre stored, then k_act_linear_fwd's body (csrc/dcr_gcn.hip) on the registers — same
// Philox counters, same keep-bit words, same order of operations in the second contraction.  acc[t] register r of lane (i, g)
// is the contraction for pre[row i of the unit][16t + 4g + r].  Every group of stores sits behind ONE branch (a branch per store
// cuts the epilogue into basic blocks, each with its own LDS read and wait), and the two operands' MFMA chains (4 HM dependent
// instructions each) are interleaved.  Shared by the kernel that keeps W1 in LDS and the K-chunked one for wide inputs.
template <int HM, bool TRAIN, bool EVAL>
__device__ __forceinline__ void first_epilogue(const FirstArgs &A, int64_t unit, const f32x4 (&acc)[HM], const float *w2l, const float *b1g,
                                               int i, int g) {
    constexpr int H = 16 * HM, LPR = H / 4, RPW = 64 / LPR;
    const int gi = g ^ i;
    const bool live = unit * 16 + i < A.n_rows;
    const int64_t row = unit * 16 + i;
    float v[HM][4];
#pragma unroll
    for (int t = 0; t < HM; ++t) {
        const float4 b = *reinterpret_cast<const float4 *>(b1g + 16 * t);
        v[t][0] = acc[t][0] + b.x; v[t][1] = acc[t][1] + b.y; v[t][2] = acc[t][2] + b.z; v[t][3] = acc[t][3] + b.w;
    }
    if (A.pre && live) {
        float *dst = A.pre + row * H + 4 * g;
#pragma unroll
        for (int t = 0; t < HM; ++t) *reinterpret_cast<float4 *>(dst + 16 * t) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
    }
    f32x4 acc_tr = {0.f, 0.f, 0.f, 0.f}, acc_ev = {0.f, 0.f, 0.f, 0.f};
    uint32_t part[4] = {0u, 0u, 0u, 0u};  // keep bits of this lane's elements, bit 4t + g of column q's word
    // The random words (philox_quad16: element-quad e = row·LPR + 4t + g draws call e >> 1, half e & 1).  Lanes g and g ^ 1 hold
    // the two halves of the same calls: each draws HM / 2 of the pair's HM calls and the words change lanes through
    // ds_bpermute (the LDS pipe) — half the Philox instructions, which on this chip are matrix-core time.
    uint32_t rw[HM][4];
    // Round 5: the decisions may come drawn beforehand (A.dwords: one bit per element, packed exactly as `bits` is) by a kernel of
    // their own that runs beside the request-rate-bound aggregations of the epoch before (experiment/training_loop.py): on this
    // chip the Philox instructions of this epilogue add to the matrix core's time.
    const bool drawn = TRAIN && A.dwords != nullptr;   // uniform
    uint32_t fld[4] = {0u, 0u, 0u, 0u};
    if (drawn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long word = live ? A.dwords[(row / RPW) * 4 + q] : 0ull;
            fld[q] = (uint32_t)(word >> ((row % RPW) * LPR));
        }
#pragma unroll
        for (int t = 0; t < HM; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) rw[t][q] = (fld[q] >> (4 * t + g)) & 1u;   // (the decision itself)
    }
    if (TRAIN && !drawn) {
        constexpr int HC = HM / 2;
        const int s = g & 1;
        uint32_t mine[HC][4], theirs[HC][4];
#pragma unroll
        for (int tt = 0; tt < HC; ++tt) {
            const int64_t e = row * LPR + 4 * (HC * s + tt) + (g & ~1);   // (even: the call index is e >> 1)
            philox4x32_10((uint64_t)(e >> 1), A.offset, A.seed, mine[tt]);
#pragma unroll
            for (int q = 0; q < 4; ++q) theirs[tt][q] = (uint32_t)__shfl_xor((int)mine[tt][q], 16);
        }
#pragma unroll
        for (int t = 0; t < HM; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t w = (t / HC == s) ? mine[t % HC][q] : theirs[t % HC][q];
                rw[t][q] = (w >> (16 * s)) & 0xFFFFu;
            }
    }
#pragma unroll
    for (int t = 0; t < HM; ++t) {
        const float4 wv = *reinterpret_cast<const float4 *>(w2l + 4 * ((4 * t) ^ gi));
        const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
        float o[4];
        if (TRAIN) {
            const uint32_t (&r)[4] = rw[t];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool keep = live && v[t][q] > 0.f && (drawn ? r[q] != 0u : r[q] >= A.threshold);
                o[q] = keep ? v[t][q] * A.scale : 0.f;
                part[q] |= keep ? (1u << (4 * t + g)) : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TRAIN) acc_tr = __builtin_amdgcn_mfma_f32_16x16x4f32(o[q], wq[q], acc_tr, 0, 0, 0);
            if (EVAL) acc_ev = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t][q] > 0.f ? v[t][q] : 0.f, wq[q], acc_ev, 0, 0, 0);
        }
    }
    if (TRAIN) {
        unsigned long long words[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t p = part[q];
            p |= (uint32_t)__shfl_xor((int)p, 16);
            p |= (uint32_t)__shfl_xor((int)p, 32);
            unsigned long long word = (unsigned long long)p << ((i % RPW) * LPR);
            uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
#pragma unroll
            for (int d = 1; d < RPW; d <<= 1) {
                lo |= (uint32_t)__shfl_xor((int)lo, d);
                hi |= (uint32_t)__shfl_xor((int)hi, d);
            }
            words[q] = ((unsigned long long)hi << 32) | lo;
        }
        if (g == 0 && (i % RPW) == 0 && live) {
            unsigned long long *dst = A.bits + (row / RPW) * 4;   // (32-byte aligned: two 16-byte stores)
            *reinterpret_cast<ulonglong2 *>(dst) = make_ulonglong2(words[0], words[1]);
            *reinterpret_cast<ulonglong2 *>(dst + 2) = make_ulonglong2(words[2], words[3]);
        }
    }
    // accumulator register r of lane (i, g) is z[row 4g + r of the unit][class i]
    const int64_t orow = unit * 16 + 4 * g;
    if (i < A.C) {
        if (orow + 3 < A.n_rows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (TRAIN) A.z_train[(orow + r) * A.ldz + i] = acc_tr[r];
                if (EVAL) A.z_eval[(orow + r) * A.ldz + i] = acc_ev[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (orow + r < A.n_rows) {
                    if (TRAIN) A.z_train[(orow + r) * A.ldz + i] = acc_tr[r];
                    if (EVAL) A.z_eval[(orow + r) * A.ldz + i] = acc_ev[r];
                }
        }
    }
}

// CNT consecutive 16-row units starting at unit u (all inside the wave's range): main contraction, then the epilogue.
// a0 arrives holding the first piece of the units' rows (first_load) and leaves holding the first piece of the CNT units at
// u_next (u_next >= 0): that load flies under the epilogue.
// W1 in LDS: row R = 16t + i at w1s + R·FS, its 16-byte chunk c at position c ^ (R & 15) (FS a multiple of 64 floats): a
// ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, ... — over 64 banks; with lane (i, g) on chunk
// 4m + g of row i the groups mix two g with complementary i, and the XOR sends the 16 lanes of a group to 16 different
// 16-byte slots (rows 4 floats apart from a multiple of 256 bytes: 2-way conflicts on every read, SQ_LDS_BANK_CONFLICT).
template <int HM, int CNT, bool TRAIN, bool EVAL>
__device__ __forceinline__ void first_chunk(const FirstArgs &A, int64_t u, int64_t u_next, const float *w1l, const float *w2l, const float *b1g,
                                            int FS, int i, int g, float4 (&a0)[CNT]) {
    const int n_m = A.F / 16, gi = g ^ i;
    f32x4 acc[CNT][HM];
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
        for (int t = 0; t < HM; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap[CNT], *apn[CNT];
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) {
        ap[rt] = first_row_ptr(A, u + rt, i, g);
        apn[rt] = u_next >= 0 ? first_row_ptr(A, u_next + rt, i, g) : ap[rt];
    }
    // two register sets: the piece of step m + 1 is in flight under the MFMAs of step m
    auto step = [&](const float4 (&av)[CNT], int m) {
        const float *wm = w1l + 4 * ((4 * m) ^ gi);
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            const float4 w = *reinterpret_cast<const float4 *>(wm + 16 * t * FS);
            const float wq[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < CNT; ++rt) {
                    const float aq = q == 0 ? av[rt].x : q == 1 ? av[rt].y : q == 2 ? av[rt].z : av[rt].w;
                    acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q], aq, acc[rt][t], 0, 0, 0);
                }
        }
    };
    float4 a1[CNT];
    int m = 0;
    for (; m + 1 < n_m; m += 2) {
        // (the scheduling barriers pin each set of loads to the START of the step it flies under; left alone, the scheduler
        //  either sinks them to their first use or hoists the W1 reads of both steps and spills)
#pragma unroll
        for (int rt = 0; rt < CNT; ++rt) a1[rt] = *reinterpret_cast<const float4 *>(ap[rt] + 16 * (m + 1));
        __builtin_amdgcn_sched_barrier(0);
        step(a0, m);
        __builtin_amdgcn_sched_barrier(0);
        const bool last = m + 2 >= n_m;           // then: the first piece of the next chunk
#pragma unroll
        for (int rt = 0; rt < CNT; ++rt) a0[rt] = *reinterpret_cast<const float4 *>(last ? apn[rt] : ap[rt] + 16 * (m + 2));
        __builtin_amdgcn_sched_barrier(0);
        step(a1, m + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (m < n_m) {
        step(a0, m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < CNT; ++rt) a0[rt] = *reinterpret_cast<const float4 *>(apn[rt]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue, 16 rows at a time (first_epilogue)
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) first_epilogue<HM, TRAIN, EVAL>(A, u + rt, acc[rt], w2l, b1g, i, g);
}

template <int HM, int NR, bool TRAIN, bool EVAL>
__global__ void __launch_bounds__(64 * FIRST_WAVES) k_first_layer_fwd(FirstArgs A, const float *__restrict__ w1, const float *__restrict__ b1,
                                                                      const float *__restrict__ w2, const uint64_t *__restrict__ offset_dev,
                                                                      int64_t units_per_wave) {
    constexpr int H = 16 * HM, HS = H;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int F = A.F, FS = (F + 63) / 64 * 64;
    float *w1s = lds;               // [H][FS]
    float *w2s = w1s + H * FS;      // [16][HS], rows >= C zero
    float *b1s = w2s + 16 * HS;     // [H]
    {
        const int f4 = F / 4;
        for (int e = threadIdx.x; e < H * f4; e += 64 * FIRST_WAVES) {
            const int r = e / f4, c4 = e - r * f4;
            *reinterpret_cast<float4 *>(w1s + r * FS + 4 * (c4 ^ (r & 15))) = *reinterpret_cast<const float4 *>(w1 + (int64_t)r * F + 4 * c4);
        }
        for (int e = threadIdx.x; e < 16 * (H / 4); e += 64 * FIRST_WAVES) {
            const int r = e / (H / 4), c4 = e % (H / 4);
            *reinterpret_cast<float4 *>(w2s + r * HS + 4 * (c4 ^ (r & 15))) =
                r < A.C ? *reinterpret_cast<const float4 *>(w2 + (int64_t)r * H + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = threadIdx.x; e < H; e += 64 * FIRST_WAVES) b1s[e] = b1 ? b1[e] : 0.f;
    }
    __syncthreads();
    if (TRAIN && offset_dev) A.offset += *offset_dev;
    if (TRAIN && A.dwords) {
        const bool current = dropout_words_current<HM>(A);
        if (blockIdx.x == 0 && threadIdx.x == 0) dropout_words_next<HM>(A);
        if (!current) A.dwords = nullptr;   // decisions of another call: draw in line
    }

    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t n_units = (A.n_rows + 15) / 16;
    const int64_t gw = (int64_t)blockIdx.x * FIRST_WAVES + wave;
    int64_t u = gw * units_per_wave;
    const int64_t u_end = u + units_per_wave < n_units ? u + units_per_wave : n_units;
    const float *w1l = w1s + i * FS;
    const float *w2l = w2s + i * HS;
    const float *b1g = b1s + 4 * g;
    if (u + NR <= u_end) {
        float4 a[NR];
        first_load<NR>(A, u, i, g, a);
        for (; u + NR <= u_end; u += NR)
            first_chunk<HM, NR, TRAIN, EVAL>(A, u, u + 2 * NR <= u_end ? u + NR : -1, w1l, w2l, b1g, FS, i, g, a);
    }
    if (NR > 1)
        for (; u < u_end; ++u) {
            float4 a[1];
            first_load<1>(A, u, i, g, a);
            first_chunk<HM, 1, TRAIN, EVAL>(A, u, -1, w1l, w2l, b1g, FS, i, g, a);
        }
}

static size_t first_layer_lds_bytes(int F, int H) { return sizeof(float) * ((size_t)H * ((F + 63) / 64 * 64) + 16 * (size_t)H + H); }

template <int HM, int NR>
static int launch_first_layer(bool train, bool eval, const float *ax, int64_t ldx, const float *w1, const float *b1, const float *w2, float *pre,
                              float *z_train, float *z_eval, int64_t ldz, unsigned long long *bits, int64_t n_rows, int F, int C, float scale,
                              uint32_t threshold, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                              const unsigned long long *dwords, hipStream_t st) {
    static int cus_dev[64] = {};   // per device (advisor, round 4: a process may drive several GPUs)
    int dev = 0;
    DCR_HIP(hipGetDevice(&dev));
    int &cus = cus_dev[dev & 63];
    if (!cus) {
        DCR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 1) cus = 1;
    }
    const size_t lds = first_layer_lds_bytes(F, 16 * HM);
    const int64_t n_units = (n_rows + 15) / 16;
    int64_t grid = (n_units + FIRST_WAVES * NR - 1) / (FIRST_WAVES * NR);
    if (grid > cus) grid = cus;
    const int64_t waves = grid * FIRST_WAVES;
    const int64_t upw = (n_units + waves - 1) / waves;
    FirstArgs args{ax, ldx, pre, z_train, z_eval, ldz, bits, n_rows, F, C, scale, threshold, seed, offset, dwords};
#define DCR_FIRST_LAUNCH(TR, EV)                                                                                                       \
    do {                                                                                                                              \
        auto kern = k_first_layer_fwd<HM, NR, TR, EV>;                                                                                \
        static bool raised[64] = {};                                                                                                  \
        if (!raised[dev & 63]) {                                                                                                      \
            DCR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            raised[dev & 63] = true;                                                                                                  \
        }                                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * FIRST_WAVES), lds, st, args, w1, b1, w2, offset_dev, upw);                    \
    } while (0)
    if (train && eval) DCR_FIRST_LAUNCH(true, true);
    else if (train) DCR_FIRST_LAUNCH(true, false);
    else DCR_FIRST_LAUNCH(false, true);
#undef DCR_FIRST_LAUNCH
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide inputs (round 5): W1 does not fit a CU's LDS — the reference's own datasets (Cora 1,433 x 128, Citeseer 3,703 x 64:
// utils/hyperparams.py:2-21, models/gcn.py:16-19).  Same MFMA layout, same epilogue; the contraction is split over K:
//   * workgroup (row group, K chunk): 4 waves, one 16-row unit each; the chunk's KCH = 16384 / H columns of W1 (64 KB) staged
//     once in LDS (XOR-swizzled as above; columns past in_features are zeros, so in_features need not be a multiple of 16 —
//     Â·X is padded to one by the caller) and kept while the workgroup walks its row groups (persistent for large N);
//   * each wave leaves its partial tile [16 rows x H] in the workspace, part[chunk][row][column]; the LAST workgroup of a row
//     group to arrive (a ticket per row group behind an agent-scope fence) adds the chunks' partials IN CHUNK ORDER —
//     deterministic whichever workgroup that is — and runs first_epilogue on the sums: bias, pre, Philox keep bits, both
//     second-layer contractions.  One launch from Â·X to [z_train | z_eval]; the pre-activation is written once.
// Citeseer shape (2,120 x 3,712 -> 64): 34 row groups x 15 chunks = 510 workgroups of 69 KB LDS (two per CU), 256 MFMAs per wave.
// Roofline: MFMA f32 for large N; at the citation sizes the kernel is a few microseconds of latency (staging + one chunk + the
// last arriver's epilogue).
template <int HM>
__host__ __device__ constexpr int wide_kch() { return 16384 / (16 * HM); }   // columns of W1 per chunk: 64 KB in LDS

template <int HM, bool TRAIN, bool EVAL>
__global__ void __launch_bounds__(256, 2) k_first_layer_wide(FirstArgs A, const float *__restrict__ w1, const float *__restrict__ b1,
                                                          const float *__restrict__ w2, const uint64_t *__restrict__ offset_dev,
                                                          float *__restrict__ part, unsigned *__restrict__ tickets, int n_chunks,
                                                          int64_t n_groups, int64_t rows_padded) {
    constexpr int H = 16 * HM, KCH = wide_kch<HM>(), FS = KCH, HS = H, NM = KCH / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int last_sh;
    float *w1s = lds;               // [H][FS]: this chunk's columns of W1
    float *w2s = w1s + H * FS;      // [16][HS], rows >= C zero
    float *b1s = w2s + 16 * HS;     // [H]
    const int F = A.F;              // the true input width; Â·X has (F + 15) / 16 * 16 columns, the pad zeros
    const int F16 = (F + 15) / 16 * 16;
    const int chunk = blockIdx.y, k0 = chunk * KCH;
    const int kw = F16 - k0 < KCH ? F16 - k0 : KCH;   // columns of this chunk (a multiple of 16)
    const int n_m = kw / 16;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, gi = g ^ i;
    const int64_t n_units = (A.n_rows + 15) / 16;
    // At the sizes this kernel exists for (a few thousand rows) every phase is ONE round trip to memory deep, so each phase asks
    // for everything it needs at once: the chunk's pieces of this wave's rows of Â·X first (they fly under the staging of W1) ...
    auto load_rows = [&](int64_t unit, float4 (&a)[NM]) {
        const float *ap = first_row_ptr(A, unit, i, g) + k0;
#pragma unroll
        for (int m = 0; m < NM; ++m) a[m] = m < n_m ? *reinterpret_cast<const float4 *>(ap + 16 * m) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 a[NM];
    {
        const int64_t unit0 = (int64_t)blockIdx.x * 4 + wave;
        load_rows(unit0 < n_units ? unit0 : 0, a);
    }
    {   // ... then W1's chunk: wave w stages rows w, w + 4, ...; lane = 16-byte chunk of the row (KCH / 4 of them); all the loads of a
        // thread are issued before the first store (element by element the loop was a chain of dependent round trips: 25 us)
        constexpr int CPR = KCH / 4, RPI = 256 / CPR, NIT = H / RPI;   // chunks per row, rows per iteration, iterations
        const int c4 = threadIdx.x % CPR, r0 = threadIdx.x / CPR, c = k0 + 4 * c4;
        const bool vec = (F & 3) == 0 && (((uintptr_t)w1) & 15) == 0;
        float4 v[NIT];
        const bool in = 4 * c4 < kw;
        if (vec) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                v[it] = (in && c + 3 < F) ? *reinterpret_cast<const float4 *>(w1 + (int64_t)(r0 + RPI * it) * F + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            // (any alignment: four 4-byte loads from clamped addresses, out-of-range elements replaced afterwards)
            const int c0 = c < F ? c : F - 1, c1 = c + 1 < F ? c + 1 : F - 1, c2 = c + 2 < F ? c + 2 : F - 1, c3 = c + 3 < F ? c + 3 : F - 1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const float *src = w1 + (int64_t)(r0 + RPI * it) * F;
                v[it] = make_float4(src[c0], src[c1], src[c2], src[c3]);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (!in || c >= F) v[it].x = 0.f;
                if (!in || c + 1 >= F) v[it].y = 0.f;
                if (!in || c + 2 >= F) v[it].z = 0.f;
                if (!in || c + 3 >= F) v[it].w = 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = r0 + RPI * it;
            *reinterpret_cast<float4 *>(w1s + r * FS + 4 * (c4 ^ (r & 15))) = v[it];
        }
        for (int e = threadIdx.x; e < 16 * (H / 4); e += 256) {
            const int r = e / (H / 4), cc = e % (H / 4);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < A.C) {
                const float *src = w2 + (int64_t)r * H + 4 * cc;
                t = make_float4(src[0], src[1], src[2], src[3]);
            }
            *reinterpret_cast<float4 *>(w2s + r * HS + 4 * (cc ^ (r & 15))) = t;
        }
        for (int e = threadIdx.x; e < H; e += 256) b1s[e] = b1 ? b1[e] : 0.f;
    }
    __syncthreads();
    if (TRAIN && offset_dev) A.offset += *offset_dev;
    if (TRAIN && A.dwords) {
        const bool current = dropout_words_current<HM>(A);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) dropout_words_next<HM>(A);
        if (!current) A.dwords = nullptr;   // decisions of another call: draw in line
    }

    const float *w1l = w1s + i * FS;
    const float *w2l = w2s + i * HS;
    const float *b1g = b1s + 4 * g;
    for (int64_t rg = blockIdx.x; rg < n_groups; rg += gridDim.x) {
        const int64_t unit = rg * 4 + wave;
        const bool have = unit < n_units;   // (uniform over the wave)
        if (have) {
            f32x4 acc[HM];
#pragma unroll
            for (int t = 0; t < HM; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m < n_m) {   // uniform (a step past the chunk's end would multiply zeros: skipped, not computed)
                    const float *wm = w1l + 4 * ((4 * m) ^ gi);
#pragma unroll
                    for (int t = 0; t < HM; ++t) {
                        const float4 w = *reinterpret_cast<const float4 *>(wm + 16 * t * FS);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a[m].x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a[m].y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a[m].z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a[m].w, acc[t], 0, 0, 0);
                    }
                }
            }
            // the partial tile: register r of acc[t] is column 16t + 4g + r of row i.  Written (and read back below) with
            // agent-scope relaxed atomics, i.e. plain stores / loads that go THROUGH the XCD's L2 (sc1): the tiles are then
            // visible to the other XCDs without an agent-scope release fence, which writes back and invalidates the whole L2 —
            // 510 of them cost 45 of the kernel's 80 us at the Citeseer shape (tools/r05_probe9.sh; timing-only build without
            // fences: 36 us).  What orders tile and ticket is the wait for the stores (workgroup-scope release) ahead of the barrier.
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(part + ((int64_t)chunk * rows_padded + unit * 16 + i) * H + 4 * g);
#pragma unroll
            for (int t = 0; t < HM; ++t) {
                __hip_atomic_store(dst + 8 * t, ((unsigned long long)__float_as_uint(acc[t][1]) << 32) | __float_as_uint(acc[t][0]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 8 * t + 1, ((unsigned long long)__float_as_uint(acc[t][3]) << 32) | __float_as_uint(acc[t][2]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the next row group's pieces (large N: a workgroup keeps its chunk of W1 and walks row groups) fly under the ticket
        if (rg + gridDim.x < n_groups) {
            const int64_t un = (rg + gridDim.x) * 4 + wave;
            load_rows(un < n_units ? un : 0, a);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // every wave's tile stores have completed ...
        __syncthreads();
        if (threadIdx.x == 0) last_sh = atomicAdd(&tickets[rg], 1u) == (unsigned)(n_chunks - 1);   // ... before the ticket is taken
        __syncthreads();
        if (last_sh) {      // uniform over the workgroup: every chunk's tiles of this row group are in memory
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (have) {
                f32x4 acc[HM];
#pragma unroll
                for (int t = 0; t < HM; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(part + (unit * 16 + i) * H + 4 * g);
                constexpr int CU_ = 32 / HM;   // chunks in flight (2 HM 8-byte loads each: 128 registers)
                for (int c = 0; c < n_chunks; c += CU_) {
                    unsigned long long v[CU_][HM][2];
#pragma unroll
                    for (int cc = 0; cc < CU_; ++cc)
#pragma unroll
                        for (int t = 0; t < HM; ++t)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh)
                                v[cc][t][hh] = c + cc < n_chunks ? __hip_atomic_load(src + ((int64_t)(c + cc) * rows_padded * H) / 2 + 8 * t + hh,
                                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                    for (int cc = 0; cc < CU_; ++cc)   // (chunk order: the sum does not depend on which workgroup arrived last;
                        if (c + cc < n_chunks)         //  a chunk past the end adds nothing — not even + 0.0, which would turn -0.0 into +0.0)
#pragma unroll
                            for (int t = 0; t < HM; ++t) {
                                acc[t][0] += __uint_as_float((unsigned)v[cc][t][0]); acc[t][1] += __uint_as_float((unsigned)(v[cc][t][0] >> 32));
                                acc[t][2] += __uint_as_float((unsigned)v[cc][t][1]); acc[t][3] += __uint_as_float((unsigned)(v[cc][t][1] >> 32));
                            }
                }
                first_epilogue<HM, TRAIN, EVAL>(A, unit, acc, w2l, b1g, i, g);
            }
            if (threadIdx.x == 0) tickets[rg] = 0u;   // (for the next call: the tickets are all-zero between launches)
        }
        __syncthreads();    // last_sh is rewritten by the next row group
    }
}

// shapes the K-chunked kernel takes and its workspace: n_chunks x (rows padded to 16) x H floats of partial tiles + a ticket per
// group of 64 rows (as floats: the caller sees one buffer; zero before the first use, left zero by every launch)
static int64_t wide_chunks(int F, int H) { return ((F + 15) / 16 * 16 + 16384 / H - 1) / (16384 / H); }
static int64_t wide_groups(int64_t n_rows) { return ((n_rows + 15) / 16 + 3) / 4; }
static int64_t wide_ws_floats(int64_t n_rows, int F, int H) {
    return wide_chunks(F, H) * ((n_rows + 15) / 16 * 16) * H + wide_groups(n_rows);
}

template <int HM>
static int launch_first_layer_wide(bool train, bool eval, const float *ax, int64_t ldx, const float *w1, const float *b1, const float *w2,
                                   float *pre, float *z_train, float *z_eval, int64_t ldz, unsigned long long *bits, int64_t n_rows, int F,
                                   int C, float scale, uint32_t threshold, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                   const unsigned long long *dwords, float *ws, hipStream_t st) {
    constexpr int H = 16 * HM;
    int dev = 0, cus = 0;
    DCR_HIP(hipGetDevice(&dev));
    DCR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus < 1) cus = 1;
    const int64_t n_chunks = wide_chunks(F, H), n_groups = wide_groups(n_rows), rows_padded = (n_rows + 15) / 16 * 16;
    if (n_chunks > 65535) DCR_FAIL(DCR_ECAPACITY, "first_layer_fwd: input too wide");
    // two workgroups per CU resident; beyond that a workgroup keeps its chunk of W1 and walks row groups
    int64_t gx = (2 * (int64_t)cus + n_chunks - 1) / n_chunks;
    if (gx > n_groups) gx = n_groups;
    if (gx < 1) gx = 1;
    const size_t lds = sizeof(float) * ((size_t)H * wide_kch<HM>() + 16 * (size_t)H + H);
    float *part = ws;
    unsigned *tickets = reinterpret_cast<unsigned *>(ws + n_chunks * rows_padded * H);
    FirstArgs args{ax, ldx, pre, z_train, z_eval, ldz, bits, n_rows, F, C, scale, threshold, seed, offset, dwords};
#define DCR_WIDE_LAUNCH(TR, EV)                                                                                                        \
    do {                                                                                                                              \
        auto kern = k_first_layer_wide<HM, TR, EV>;                                                                                   \
        static bool raised[64] = {};   /* per device (a process may drive several) */                                                \
        if (!raised[dev & 63]) {   /* (the kernel also has a few static bytes: the limit is what it asks for, not the whole LDS) */    \
            DCR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            raised[dev & 63] = true;                                                                                                  \
        }                                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_chunks), dim3(256), lds, st, args, w1, b1, w2, offset_dev, part, tickets, \
                           (int)n_chunks, n_groups, rows_padded);                                                                     \
    } while (0)
    if (train && eval) DCR_WIDE_LAUNCH(true, true);
    else if (train) DCR_WIDE_LAUNCH(true, false);
    else DCR_WIDE_LAUNCH(false, true);
#undef DCR_WIDE_LAUNCH
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

// The dropout decisions of ONE call of the kernels above, drawn by a kernel of their own (round 5): thread per row, the row's
// H / 8 Philox calls (call c = row · H/8 + k holds the element-quads 2k and 2k + 1 of the row in the low and high halves of its
// four words), one keep bit per element packed as first_epilogue packs `bits`: bit e of the row's field of word q is the decision
// for hidden column 4e + q — exactly what the kernels draw in line for the same (seed, offset); tests/test_gcn.py compares the
// two bit for bit.
template <int HM>
__global__ void __launch_bounds__(256) k_dropout_words(unsigned long long *__restrict__ dwords, int64_t n_rows, uint32_t threshold, uint64_t seed,
                                                       uint64_t offset, const uint64_t *__restrict__ offset_dev) {
    constexpr int H = 16 * HM, LPR = H / 4, RPW = 64 / LPR, CALLS = LPR / 2;
    if (offset_dev) offset += *offset_dev;
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the call these decisions belong to (dropout_words_current)
        unsigned long long *stamp = dwords + (n_rows + RPW - 1) / RPW * 4;
        stamp[0] = offset; stamp[1] = seed; stamp[2] = threshold; stamp[3] = (unsigned long long)n_rows;
    }
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n_rows; row += (int64_t)gridDim.x * 256) {
        uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll 4
        for (int k = 0; k < CALLS; ++k) {
            uint32_t r[4];
            philox4x32_10((uint64_t)(row * CALLS + k), offset, seed, r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m[q] |= ((r[q] & 0xFFFFu) >= threshold ? 1u : 0u) << (2 * k);
                m[q] |= ((r[q] >> 16) >= threshold ? 1u : 0u) << (2 * k + 1);
            }
        }
        // the row's LPR-bit field of the four 64-bit words of its group of RPW rows (little-endian halves / quarters of a word)
        if (LPR == 32) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(dwords + (row / RPW) * 4) + (row % RPW);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[2 * q] = m[q];
        } else {
            uint16_t *dst = reinterpret_cast<uint16_t *>(dwords + (row / RPW) * 4) + (row % RPW);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[4 * q] = (uint16_t)m[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the above in ONE kernel: from dz = d loss / d z_train [N x C] to dW1 [H x F], db1 [H] and dW2 [C x H], with the
// gradient of the pre-activation, dpre = keep ? (dz · W2) / (1 - p) : 0  [N x H], living only in registers.  Before:
// dcr_act_linear_bwd_fused_f32_dev wrote dpre (512 MB at 1M x 128), dcr_atb_f32_dev read it back for dW1 = dpreᵀ · (Â·X) with
// every operand fragment fetched from global memory by every wave that needs it (0.21-0.28 + 0.67 ms).  The first layer's
// input needs no gradient (Â·X is a constant), so dpre has no other reader.
//
// A workgroup of 8 waves (one per CU, two waves per SIMD) walks a contiguous range of 16-row units, two per stage;
// v_mfma_f32_16x16x4_f32, lane (i = lane & 15, g = lane >> 4); hidden "tile" (b, q) = the 16 columns 64b + 4i + q (as in
// k_act_linear_bwd_fused); a wave owns one of the H/16 tiles (hidden 64: two waves per tile, half of the features each):
//   dpre tile:  A[row i][k = g] = dz[row i][class 4g + s] (MFMA step s), B[k = g][n = i] = W2[class 4g + s][col(i)] (registers);
//               D register r = dpre[row 4g + r][col(i)], masked with the keep bit of (row, column) and scaled;
//   db1:        column sums of the D registers;
//   dW2 tile:   A[class i][k = g] = dz[row 4g + r][class i], B[k = g][n = i] = h[row 4g + r][col(i)], h = keep ? pre·scale : 0
//               (step r): the same (row, column) and the same keep bit as D register r;
//   dW1 tiles:  A[m = i][k = g] = dpre[row 4g + r][col(i)] = D register r AS IT STANDS (step r: no shuffle, no LDS),
//               B[k = g][n = i] = (Â·X)[row 4g + r][64J + 4i + q'] — one ds_read_b128 of the unit's Â·X rows in LDS gives the
//               four q' of a J; 16 (J, q') feature tiles x 4 accumulator registers stay in registers for the whole range.
// The 32 rows of Â·X of the NEXT stage travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: a plain copy, 1 KiB per
// wave-instruction, conflict-free for the reads above as it lies) while the matrix cores work on this one; dz, keep bits and
// pre of the next stage travel to registers meanwhile.  One barrier per stage.  Per 16 rows and wave: 64 + 4 + 4 MFMAs.
// The workgroups' partial dW1 / dW2 / db1 are added in workgroup order by the slab reduction kernels: deterministic.
// Shape of a workgroup: BWD_WAVES waves, BWD_UR 16-row units per stage, BWD_WGS workgroups per CU.  Round-4 measurements on the
// bench shape: 8 waves x 1 workgroup, 2 units per stage 0.735 ms; 4 waves x 2 workgroups, 1 unit per stage (a wave owns two hidden
// tiles: half the LDS reads per MFMA, and a workgroup at its barrier leaves the CU to the other one): see DESIGN §4.3.
#ifndef DCR_BWD_WAVES
#define DCR_BWD_WAVES 8
#endif
#ifndef DCR_BWD_UR
#define DCR_BWD_UR 2
#endif
constexpr int BWD_WAVES = DCR_BWD_WAVES, BWD_UR = DCR_BWD_UR, BWD_WGS = 8 / BWD_WAVES;
constexpr int BWD_TILE = BWD_UR * 16 * 256;   // floats of one staging buffer (256 feature columns)

// Two staging buffers as two DISTINCT objects: an LDS-DMA load is a pending LDS write on the vector-memory counter, and the
// compiler puts s_waitcnt vmcnt(0) before every ds_read that may alias it — reads of one buffer of a single array would wait
// for the copy into the other, i.e. no overlap at all (measured: 1.0 ms instead of 0.6).  The stage loop is unrolled by two so
// that every access names its buffer statically.  (+ the reach of a partial 64-column group past the last row)
#ifdef DCR_BWD_PROF   // (diagnostic build: wave-cycles per section, tools/probe_first_bwd.py prints them)
__device__ unsigned long long bwd_prof[8];
#define BWD_STAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_acc[slot] += t_ - prof_t; prof_t = t_; } while (0)
#else
#define BWD_STAMP(slot) do { } while (0)
#endif
// (+ 64 / + 16: the reach of a partial 64-column group past the last row of Â·X, of a class index past C past the last row of dz)
constexpr int BWD_PRE = BWD_UR * 16 * 128, BWD_DZ = BWD_UR * 16 * 16 + 16, BWD_BITS = BWD_UR * 16 * 4;   // floats / floats / uint2
__shared__ __attribute__((aligned(16))) float bwd_buf_a[BWD_TILE + 64];
__shared__ __attribute__((aligned(16))) float bwd_buf_b[BWD_TILE + 64];
__shared__ __attribute__((aligned(16))) float bwd_pre_a[BWD_PRE];
__shared__ __attribute__((aligned(16))) float bwd_pre_b[BWD_PRE];
__shared__ __attribute__((aligned(16))) float bwd_dz_a[BWD_DZ];
__shared__ __attribute__((aligned(16))) float bwd_dz_b[BWD_DZ];
__shared__ __attribute__((aligned(16))) uint2 bwd_bits_a[BWD_BITS];
__shared__ __attribute__((aligned(16))) uint2 bwd_bits_b[BWD_BITS];

template <int HM>
__global__ void __launch_bounds__(64 * BWD_WAVES, BWD_WGS) k_first_layer_bwd(const float *__restrict__ dz, const float *__restrict__ w2,
                                                                             const unsigned long long *__restrict__ bits,
                                                                             const float *__restrict__ pre, const float *__restrict__ ax,
                                                                             int64_t ldx, int64_t n_rows, int F, int C, float scale,
                                                                             float *__restrict__ part1, float *__restrict__ part2,
                                                                             int64_t stages_per_wg) {
    // wave w owns the TPW hidden tiles w·TPW ..; with fewer tiles than waves (TPW = 1, NSPLIT waves per tile) the waves of a
    // tile split the feature groups J
    constexpr int H = 16 * HM, LPR = H / 4, RPW = 64 / LPR, SETS = 4 / RPW, UR = BWD_UR, NT = 64 * BWD_WAVES;
    constexpr int TPW = HM >= BWD_WAVES ? HM / BWD_WAVES : 1, NSPLIT = HM >= BWD_WAVES ? 1 : BWD_WAVES / HM, JPW = 4 / NSPLIT;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: what depends on it branches uniformly)
#ifdef DCR_BWD_PROF
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
    const int fb0 = blockIdx.y * 256;
    const int FB = F - fb0 < 256 ? F - fb0 : 256;   // feature columns of this block (a multiple of 16)
    const int jh = NSPLIT == 1 ? 0 : wave / HM;
    int tb[TPW], tq[TPW], col[TPW], bitpos[TPW];
    float w2r[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tt = NSPLIT == 1 ? wave * TPW + j : wave % HM;
        tb[j] = tt >> 2;
        tq[j] = tt & 3;
        col[j] = 64 * tb[j] + 4 * i + tq[j];
        bitpos[j] = 16 * tb[j] + i;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) w2r[j][s4] = 4 * g + s4 < C ? w2[(int64_t)(4 * g + s4) * H + col[j]] : 0.f;
    }
    f32x4 acc1[TPW][4 * JPW], dw2[TPW];
    float cs[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        dw2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs[j] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4 * JPW; ++nt) acc1[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t n_units = (n_rows + 15) / 16, n_stages = (n_units + UR - 1) / UR;
    const int64_t s0 = (int64_t)blockIdx.x * stages_per_wg;
    const int64_t s1 = s0 + stages_per_wg < n_stages ? s0 + stages_per_wg : n_stages;

    // Everything a stage reads travels global -> LDS by LDS-DMA, as plain copies: its UR x 16 rows of Â·X (FB columns, row after
    // row), of pre (H columns), of dz (C columns) and their keep words.  (Round 4, first version: dz in two layouts, pre and the
    // keep words went to registers by per-lane loads — 18-24 dword loads per unit and wave, each touching 4-16 cache lines,
    // issued by every wave for the same rows: the stamped build showed 35-46 % of a wave's life in those "fronts", the texture
    // path saturated by line touches, not by bytes.)  Wave-instruction k of a copy moves the 16-byte chunks 64k .. 64k + 63.
    constexpr int MAXI = UR * 256 / 16 / BWD_WAVES;   // Â·X instructions per wave and stage at 256 columns
    const int cpr = FB / 4, n_inst = UR * FB / 16;
    int coff[MAXI], ccol[MAXI];
#pragma unroll
    for (int n = 0; n < MAXI; ++n) {
        const int c = 64 * (wave + BWD_WAVES * n) + lane, row = c / cpr;
        ccol[n] = 4 * (c - row * cpr);
        coff[n] = row * (int)ldx + ccol[n];
    }
    constexpr int PRE_INST = UR * 16 * H / 256, PRE_PER_WAVE = (PRE_INST + BWD_WAVES - 1) / BWD_WAVES;
    const int dz_chunks = UR * 4 * C, bits_chunks = UR * 32 / RPW;   // 16-byte chunks of the dz rows / of the keep words
    constexpr int N_PIECES = MAXI + PRE_PER_WAVE + 2;   // copy instructions a wave issues per stage, at most
    // piece n of the copies of stage st (n = 0 .. N_PIECES - 1): issued one per pair of sub-steps INSIDE the stage before, between
    // its MFMAs — eight waves issuing their 6-7 copies together right behind the barrier left the matrix cores idle meanwhile
    // (9 % of a wave's life in the stamped build)
    auto stage_piece = [&](int n, int64_t st, float *xdst, float *pdst, float *ddst, uint2 *bdst, bool whole) {
        const int64_t row0 = st * (UR * 16);
        if (whole) {
            if (n < MAXI) {
                const int k = wave + BWD_WAVES * n;
                if (k < n_inst)   // (uniform)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ax + row0 * ldx + fb0 + coff[n]),
                                                     (__attribute__((address_space(3))) void *)(xdst + 256 * k), 16, 0, 0);
            } else if (n < MAXI + PRE_PER_WAVE) {
                const int k = wave + BWD_WAVES * (n - MAXI);
                if (k < PRE_INST)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pre + row0 * H + 256 * k + 4 * lane),
                                                     (__attribute__((address_space(3))) void *)(pdst + 256 * k), 16, 0, 0);
            } else if (n == MAXI + PRE_PER_WAVE) {
                // dz and the keep words: a few hundred bytes each, by the waves that have the fewest pre pieces
                for (int k = BWD_WAVES - 1 - wave; 64 * k < dz_chunks; k += BWD_WAVES)
                    if (64 * k + lane < dz_chunks)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dz + row0 * C + 256 * k + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)(ddst + 256 * k), 16, 0, 0);
            } else if (n == MAXI + PRE_PER_WAVE + 1) {
                if (wave == (BWD_WAVES > 2 ? BWD_WAVES - 3 : 0) && lane < bits_chunks)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const uint2 *>(bits) + (row0 / RPW) * 4 + 2 * lane),
                        (__attribute__((address_space(3))) void *)bdst, 16, 0, 0);
            }
        }
    };
    // the ragged last stage (once per launch, one workgroup): plain loads, rows past the end as zeros
    auto ragged_fill = [&](int64_t st, float *xdst, float *pdst, float *ddst, uint2 *bdst) {
        const int64_t row0 = st * (UR * 16);
        for (int e = threadIdx.x; e < UR * 16 * FB; e += NT) {
            const int row = e / FB, c = e - row * FB;
            xdst[e] = row0 + row < n_rows ? ax[(row0 + row) * ldx + fb0 + c] : 0.f;
        }
        for (int e = threadIdx.x; e < UR * 16 * H; e += NT) pdst[e] = row0 + e / H < n_rows ? pre[row0 * H + e] : 0.f;
        for (int e = threadIdx.x; e < UR * 16 * C; e += NT) ddst[e] = row0 + e / C < n_rows ? dz[row0 * C + e] : 0.f;
        for (int e = threadIdx.x; e < UR * 16 / RPW * 4; e += NT)
            bdst[e] = row0 + (e / 4) * RPW < n_rows ? reinterpret_cast<const uint2 *>(bits)[(row0 / RPW) * 4 + e] : make_uint2(0u, 0u);
    };
    // (a stage is "whole" when a row follows it: no row of it is past the end, and every 16-byte chunk copied lies inside its array)
    auto whole_stage = [&](int64_t st) { return (st + 1) * (UR * 16) < n_rows; };
    const bool cb = i < C;
    const bool c4 = (C & 3) == 0;
    const int xoff = 4 * g * FB + 4 * i + 64 * JPW * jh;   // this lane's float offset into a unit's rows (row 4g, group JPW·jh)
    // the matrix-core work of one stage: per unit the dpre tiles (+ db1) from the unit's dz, keep words and pre in LDS, then 4
    // steps r of JPW reads and 4 JPW TPW MFMAs in parts; the reads of one part are issued before the MFMAs of the part before,
    // and the four dependent MFMAs of dW2 ride in that stream, one per step (scheduling barriers pin the order: left alone,
    // the scheduler hoists every read of the stage to its top and spills)
    constexpr int JH = (TPW > 1 || JPW < 4) ? 1 : 2;   // feature groups per read: each read feeds 4 JH TPW MFMAs
    constexpr int NP = JPW / JH, S = UR * 4 * NP;     // parts per step r; sub-steps per stage
    auto compute = [&](const float *xs, const float *ps, const float *ds, const uint2 *bs, int64_t st, int64_t st_next, float *nx, float *np_,
                       float *nd, uint2 *nb) {
        const bool nwhole = st_next >= 0 && whole_stage(st_next);
        float4 xa[JH], xb[JH];
        float dpre[TPW][4], at[4], hv[TPW][4];
        auto read = [&](float4 (&x)[JH], int k) {   // sub-step k = (4 uu + r) NP + part
            const float *xp = xs + (k / (4 * NP)) * 16 * FB + xoff + ((k / NP) & 3) * FB + (k % NP) * (64 * JH);
            // (every group J of the 256-column block is read and multiplied, also past FB: those bytes are other rows of the
            //  buffer, the products land in accumulator columns that are never stored, and MFMA columns do not mix)
#pragma unroll
            for (int J = 0; J < JH; ++J) x[J] = *reinterpret_cast<const float4 *>(xp + 64 * J);
        };
        auto mfmas = [&](const float4 (&x)[JH], int k) {
            const int r = (k / NP) & 3, part = k % NP;
#pragma unroll
            for (int J = 0; J < JH; ++J) {
                const float xq[4] = {x[J].x, x[J].y, x[J].z, x[J].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < TPW; ++j) {
                        const int nt = 4 * (JH * part + J) + q;
                        acc1[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpre[j][r], xq[q], acc1[j][nt], 0, 0, 0);
                    }
            }
            if (part == 0 && jh == 0)   // (uniform: one wave per hidden tile keeps dW2; one of its four dependent MFMAs per step)
#pragma unroll
                for (int j = 0; j < TPW; ++j) dw2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[r], hv[j][r], dw2[j], 0, 0, 0);
        };
        auto front = [&](int uu) {
            BWD_STAMP(1);
            const int64_t row0 = (st * UR + uu) * 16;
            const int64_t left64 = n_rows - row0;
            const int left = left64 < 0 ? 0 : left64 > 16 ? 16 : (int)left64;   // rows of the unit before the end (uniform)
            const bool live_a = i < left;
            // A[row i][k = g] of step s: dz[row i][class 4g + s] (a class past C reads the next row's numbers or the pad — LDS that
            // nobody wrote, NaN patterns included, and 0 x NaN is NaN: switched off below)
            float za[4];
            const float *zr = ds + (uu * 16 + i) * C + 4 * g;
            if (c4) {
                const float4 t = *reinterpret_cast<const float4 *>(zr);
                za[0] = t.x; za[1] = t.y; za[2] = t.z; za[3] = t.w;
            } else {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) za[s4] = zr[s4];
            }
            float zt[4], pv[TPW][4];
            uint2 bw[TPW][SETS];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zt[r] = ds[(uu * 16 + 4 * g + r) * C + (cb ? i : 0)];
#pragma unroll
                for (int j = 0; j < TPW; ++j) pv[j][r] = ps[(uu * 16 + 4 * g + r) * H + col[j]];
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int h = 0; h < SETS; ++h) bw[j][h] = bs[((uu * 16 + 4 * g) / RPW + h) * 4 + tq[j]];
            f32x4 d[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < TPW; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32((live_a && 4 * g + s4 < C) ? za[s4] : 0.f, w2r[j][s4], d[j], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sh = (r % RPW) * LPR;                      // bit (r % RPW)·LPR + 16 tb + i of the 64-bit word
                at[r] = (4 * g + r < left && cb) ? zt[r] : 0.f;
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const uint32_t half = (sh >> 5) ? bw[j][r / RPW].y : bw[j][r / RPW].x;
                    const bool keep = (half >> ((sh & 31) + bitpos[j])) & 1u;
                    dpre[j][r] = keep ? d[j][r] * scale : 0.f;
                    cs[j] += dpre[j][r];
                    hv[j][r] = keep ? pv[j][r] * scale : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            BWD_STAMP(0);
        };
        read(xa, 0);
#pragma unroll
        for (int k = 0; k < S; k += 2) {
            if (k % (4 * NP) == 0) front(k / (4 * NP));
            read(xb, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xa, k);
            if (nwhole && k / 2 < N_PIECES) stage_piece(k / 2, st_next, nx, np_, nd, nb, true);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < S) read(xa, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(xb, k + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    static_assert(S / 2 >= N_PIECES, "a stage has fewer pairs of sub-steps than a wave has copies to issue");
    if (s0 < s1) {
        if (whole_stage(s0))
            for (int n = 0; n < N_PIECES; ++n) stage_piece(n, s0, bwd_buf_a, bwd_pre_a, bwd_dz_a, bwd_bits_a, true);
        else
            ragged_fill(s0, bwd_buf_a, bwd_pre_a, bwd_dz_a, bwd_bits_a);
    }
    for (int64_t st = s0; st < s1; st += 2) {
        BWD_STAMP(1);
        __syncthreads();   // the copies into a have landed (every wave's), and nobody still reads b
        BWD_STAMP(2);
        compute(bwd_buf_a, bwd_pre_a, bwd_dz_a, bwd_bits_a, st, st + 1 < s1 ? st + 1 : -1, bwd_buf_b, bwd_pre_b, bwd_dz_b, bwd_bits_b);
        if (st + 1 >= s1) break;
        if (!whole_stage(st + 1)) ragged_fill(st + 1, bwd_buf_b, bwd_pre_b, bwd_dz_b, bwd_bits_b);
        BWD_STAMP(1);
        __syncthreads();   // b has landed, nobody still reads a
        BWD_STAMP(2);
        compute(bwd_buf_b, bwd_pre_b, bwd_dz_b, bwd_bits_b, st + 1, st + 2 < s1 ? st + 2 : -1, bwd_buf_a, bwd_pre_a, bwd_dz_a, bwd_bits_a);
        if (st + 2 < s1 && !whole_stage(st + 2)) ragged_fill(st + 2, bwd_buf_a, bwd_pre_a, bwd_dz_a, bwd_bits_a);
    }
#ifdef DCR_BWD_PROF
    BWD_STAMP(1);
    if (lane == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&bwd_prof[k], prof_acc[k]);
#endif

    // the workgroup's parts.  acc1[j][4J + q'] register r = dW1[64 tb + 4 (4g + r) + tq][fb0 + 64 (JPW jh + J) + 4i + q']
    float *o1 = part1 + (int64_t)blockIdx.x * H * F;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int J = 0; J < JPW; ++J) {
            const int f = 64 * (JPW * jh + J) + 4 * i;
            if (f < FB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int hc = 64 * tb[j] + 4 * (4 * g + r) + tq[j];
                    *reinterpret_cast<float4 *>(o1 + (int64_t)hc * F + fb0 + f) =
                        make_float4(acc1[j][4 * J][r], acc1[j][4 * J + 1][r], acc1[j][4 * J + 2][r], acc1[j][4 * J + 3][r]);
                }
            }
        }
    if (blockIdx.y == 0 && jh == 0) {   // [H column sums | dW2, class-major, 16 rows]: register r of dw2[j] = dW2[class 4g + r][col(i)]
        float *o2 = part2 + (int64_t)blockIdx.x * 17 * H;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            float c = cs[j];
            c += __shfl_xor(c, 16, 64);
            c += __shfl_xor(c, 32, 64);
            if (g == 0) o2[col[j]] = c;
#pragma unroll
            for (int r = 0; r < 4; ++r) o2[H + (4 * g + r) * H + col[j]] = dw2[j][r];
        }
    }
}

// workgroups along the rows: one per CU in total (the grid's second dimension tiles the input width by 256 columns; every
// workgroup leaves a partial dW1 tile that the slab reduction reads back, so a wide, short problem — Citeseer: 15 column blocks,
// 67 stages — takes few, longer row ranges)
static int64_t first_layer_bwd_blocks(int64_t n_rows, int F16) {
    static int cus_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int &cus = cus_dev[dev & 63];
    if (!cus) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    const int64_t n_stages = ((n_rows + 15) / 16 + BWD_UR - 1) / BWD_UR;
    const int64_t fblocks = (F16 + 255) / 256;
    int64_t blocks = ((int64_t)cus * BWD_WGS + fblocks - 1) / fblocks;
    if (blocks > n_stages) blocks = n_stages;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace dcr

using namespace dcr;

static bool first_layer_resident(int in_features, int hidden) {   // W1 whole in one CU's LDS (k_first_layer_fwd)
    return in_features >= 16 && (in_features % 16) == 0 && first_layer_lds_bytes(in_features, hidden) <= 160 * 1024;
}

extern "C" int dcr_first_layer_fits(int in_features, int hidden, int classes) {
    if (in_features < 1 || (hidden != 64 && hidden != 128) || classes < 1 || classes > 16) return 0;
    return 1;   // (W1 resident in LDS where it fits, K-chunked otherwise: dcr_first_layer_fwd_workspace says which)
}

extern "C" int dcr_first_layer_fwd_workspace(int64_t n_rows, int in_features, int hidden, int64_t *floats) {
    if (!floats || n_rows < 0 || in_features < 1 || (hidden != 64 && hidden != 128)) DCR_FAIL(DCR_EINVAL, "bad first_layer_fwd_workspace arguments");
    *floats = first_layer_resident(in_features, hidden) ? 0 : dcr::wide_ws_floats(n_rows, in_features, hidden);
    return DCR_OK;
}

extern "C" int dcr_dropout_words_count(int64_t n_rows, int hidden, int64_t *words) {
    if (!words || n_rows < 0 || (hidden != 64 && hidden != 128)) DCR_FAIL(DCR_EINVAL, "bad dropout_words_count arguments (hidden 64 or 128)");
    const int rpw = 64 / (hidden / 4);
    *words = (n_rows + rpw - 1) / rpw * 4 + 8;   // one bit per element, rows in groups of rpw; the stamp of the call; the next call's offset
    return DCR_OK;
}

extern "C" int dcr_dropout_words_dev(uint64_t *dwords, int64_t n_rows, int hidden, double p, uint64_t seed, uint64_t offset,
                                     const uint64_t *offset_dev, void *hip_stream) {
    if (!dwords || n_rows < 0 || (hidden != 64 && hidden != 128) || !(p >= 0.0 && p < 1.0) || ((uintptr_t)dwords & 7))
        DCR_FAIL(DCR_EINVAL, "bad dropout_words arguments (hidden 64 or 128, 0 <= p < 1)");
    if (n_rows == 0) return DCR_OK;
    const uint32_t threshold = dcr::dropout_threshold16(p);
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 256) blocks = 256;   // a workgroup per CU, rows in a grid stride: beside another stream's kernels it takes the fewest wave slots
                                      // (measured in the epoch: 8192 blocks 1.60 ms, 256 blocks 1.585 — profiles/r05_dropout_ahead.txt)
    if (hidden == 128)
        hipLaunchKernelGGL((dcr::k_dropout_words<8>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (unsigned long long *)dwords,
                           n_rows, threshold, seed, offset, offset_dev);
    else
        hipLaunchKernelGGL((dcr::k_dropout_words<4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (unsigned long long *)dwords,
                           n_rows, threshold, seed, offset, offset_dev);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

extern "C" int dcr_first_layer_fwd_ws_f32_dev(const float *ax, int64_t ldx, const float *w1, const float *b1, const float *w2, float *pre,
                                              float *z_train, float *z_eval, int64_t ldz, uint64_t *bits, uint64_t *dwords_in,
                                              int64_t n_rows, int in_features, int hidden, int classes, double p, uint64_t seed,
                                              uint64_t offset, const uint64_t *offset_dev, float *ws, int64_t ws_floats, void *hip_stream) {
    const unsigned long long *dwords = (const unsigned long long *)dwords_in;
    const bool train = z_train != nullptr, eval = z_eval != nullptr;
    const int f16 = (in_features + 15) / 16 * 16;
    if (!ax || !w1 || !w2 || n_rows < 0 || (!train && !eval) || ldx < f16) DCR_FAIL(DCR_EINVAL, "bad first_layer_fwd arguments (ldx >= in_features rounded up to 16)");
    if (train && (!bits || !pre || !(p >= 0.0 && p < 1.0)))
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: the training output needs bits, pre (the backward pass reads both) and 0 <= p < 1");
    if (dwords && (!train || ((uintptr_t)dwords & 7))) DCR_FAIL(DCR_EINVAL, "first_layer_fwd: dropout words go with the training output (8-byte aligned)");
    if (!dcr_first_layer_fits(in_features, hidden, classes) || ldz < classes)
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: hidden 64 or 128, at most 16 classes, ldz >= classes (other shapes take the GEMM library and "
                             "dcr_act_linear_fwd_f32_dev)");
    if (((uintptr_t)ax & 15) || (ldx & 3) || ((uintptr_t)w2 & 15) || (pre && ((uintptr_t)pre & 15)))
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: 16-byte aligned tensors and row stride expected");
    if (n_rows == 0) return DCR_OK;
    const uint32_t threshold = dcr::dropout_threshold16(p);
    const float scale = (float)(1.0 / (1.0 - p));
    hipStream_t st = (hipStream_t)hip_stream;
    if (first_layer_resident(in_features, hidden)) {
        if ((uintptr_t)w1 & 15) DCR_FAIL(DCR_EINVAL, "first_layer_fwd: 16-byte aligned W1 expected");
        if (hidden == 128)
            return launch_first_layer<8, DCR_FIRST_NR>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows,
                                                       in_features, classes, scale, threshold, seed, offset, offset_dev, dwords, st);
        return launch_first_layer<4, DCR_FIRST_NR>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows,
                                                   in_features, classes, scale, threshold, seed, offset, offset_dev, dwords, st);
    }
    if (!ws || ((uintptr_t)ws & 15) || ws_floats < dcr::wide_ws_floats(n_rows, in_features, hidden))
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: this width takes the K-chunked kernel, which needs dcr_first_layer_fwd_workspace floats (16-byte "
                             "aligned; the trailing tickets zero before the first use — every launch leaves them zero)");
    if (hidden == 128)
        return launch_first_layer_wide<8>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows, in_features,
                                          classes, scale, threshold, seed, offset, offset_dev, dwords, ws, st);
    return launch_first_layer_wide<4>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows, in_features,
                                      classes, scale, threshold, seed, offset, offset_dev, dwords, ws, st);
}

extern "C" int dcr_first_layer_fwd_f32_dev(const float *ax, int64_t ldx, const float *w1, const float *b1, const float *w2, float *pre,
                                           float *z_train, float *z_eval, int64_t ldz, uint64_t *bits, int64_t n_rows, int in_features,
                                           int hidden, int classes, double p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                           void *hip_stream) {
    // (the entry point of round 4: shapes whose W1 stays resident in LDS need no workspace)
    return dcr_first_layer_fwd_ws_f32_dev(ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, bits, nullptr, n_rows, in_features, hidden, classes, p, seed,
                                          offset, offset_dev, nullptr, 0, hip_stream);
}

extern "C" int dcr_first_layer_bwd_workspace(int64_t n_rows, int in_features, int hidden, int64_t *floats) {
    if (!floats || n_rows < 0 || in_features < 1 || (hidden != 64 && hidden != 128))
        DCR_FAIL(DCR_EINVAL, "bad first_layer_bwd_workspace arguments");
    const int f16 = (in_features + 15) / 16 * 16;
    *floats = first_layer_bwd_blocks(n_rows, f16) * ((int64_t)hidden * f16 + 17 * hidden);
    return DCR_OK;
}

extern "C" int dcr_first_layer_bwd_f32_dev(const float *dz, const float *w2, const uint64_t *bits, const float *pre, const float *ax, int64_t ldx,
                                           float *dw1, float *db1, float *dw2, float *ws, int64_t ws_floats, int64_t n_rows, int in_features,
                                           int hidden, int classes, double p, void *hip_stream) {
    // in_features: the true width of W1 / dW1; Â·X holds it rounded up to a multiple of 16 columns (the pad finite: its products
    // land in columns of the partial tiles that are never read)
    const int f16 = (in_features + 15) / 16 * 16;
    if (!dz || !w2 || !bits || !pre || !ax || !dw1 || !db1 || !dw2 || !ws || n_rows < 0 || ldx < f16 || !(p >= 0.0 && p < 1.0))
        DCR_FAIL(DCR_EINVAL, "bad first_layer_bwd arguments (ldx >= in_features rounded up to 16)");
    if (in_features < 1 || (hidden != 64 && hidden != 128) || classes < 1 || classes > 16)
        DCR_FAIL(DCR_EINVAL, "first_layer_bwd: hidden 64 or 128, at most 16 classes");
    if (((uintptr_t)ax & 15) || (ldx & 3) || ((uintptr_t)ws & 15) || ((uintptr_t)dz & 15) || ((uintptr_t)pre & 15) || ((uintptr_t)bits & 15))
        DCR_FAIL(DCR_EINVAL, "first_layer_bwd: 16-byte aligned tensors and row stride expected (the stage copies move 16-byte pieces)");
    const int64_t blocks = first_layer_bwd_blocks(n_rows, f16);
    const int64_t per1 = (int64_t)hidden * f16;
    if (ws_floats < blocks * (per1 + 17 * hidden)) DCR_FAIL(DCR_EINVAL, "first_layer_bwd: workspace too small (dcr_first_layer_bwd_workspace)");
    hipStream_t st = (hipStream_t)hip_stream;
    if (n_rows == 0) {
        DCR_HIP(hipMemsetAsync(dw1, 0, sizeof(float) * (size_t)hidden * in_features, st));
        DCR_HIP(hipMemsetAsync(db1, 0, sizeof(float) * hidden, st));
        DCR_HIP(hipMemsetAsync(dw2, 0, sizeof(float) * hidden * classes, st));
        return DCR_OK;
    }
    const int64_t n_stages = ((n_rows + 15) / 16 + dcr::BWD_UR - 1) / dcr::BWD_UR;
    const int64_t upw = (n_stages + blocks - 1) / blocks;
    float *part1 = ws, *part2 = ws + blocks * per1;
    const float scale = (float)(1.0 / (1.0 - p));
    if ((f16 + 255) / 256 > 65535) DCR_FAIL(DCR_ECAPACITY, "first_layer_bwd: input too wide");
    const dim3 grid((unsigned)blocks, (unsigned)((f16 + 255) / 256));
    if (hidden == 128)
        hipLaunchKernelGGL((dcr::k_first_layer_bwd<8>), grid, dim3(64 * dcr::BWD_WAVES), 0, st, dz, w2, (const unsigned long long *)bits, pre, ax, ldx, n_rows,
                           f16, classes, scale, part1, part2, upw);
    else
        hipLaunchKernelGGL((dcr::k_first_layer_bwd<4>), grid, dim3(64 * dcr::BWD_WAVES), 0, st, dz, w2, (const unsigned long long *)bits, pre, ax, ldx, n_rows,
                           f16, classes, scale, part1, part2, upw);
#ifdef DCR_BWD_PROF
    {
        unsigned long long h[8];
        DCR_HIP(hipStreamSynchronize(st));
        DCR_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(dcr::bwd_prof), sizeof(h)));
        fprintf(stderr, "[first_layer_bwd prof] wave-Mcycles: fronts (dpre, keep bits, next operands' loads) %.1f; MFMA sub-steps %.1f; barrier %.1f; "
                "DMA issue %.1f\n", h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6);
        unsigned long long z[8] = {0};
        DCR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dcr::bwd_prof), z, sizeof(z)));
    }
#endif
    // (the partial tiles are f16 columns wide; dW1 takes the true width)
    dcr::launch_slab_reduce_cols(part1, dw1, per1, f16, in_features, in_features, (int)blocks, st);
    dcr::launch_parts_finish(part2, blocks, 17 * hidden, hidden, hidden * (1 + classes), db1, dw2, st);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}
