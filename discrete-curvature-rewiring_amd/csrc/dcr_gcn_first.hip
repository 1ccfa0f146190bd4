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
constexpr int FIRST_WAVES = 8;

struct FirstArgs {
    const float *ax; int64_t ldx;
    float *pre, *z_train, *z_eval; int64_t ldz;
    unsigned long long *bits;
    int64_t n_rows;
    int F, C;
    float scale; uint32_t threshold; uint64_t seed, offset;
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
    constexpr int H = 16 * HM, LPR = H / 4, RPW = 64 / LPR;
    const int n_m = A.F / 16, gi = g ^ i;
    f32x4 acc[CNT][HM];
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt)
#pragma unroll
        for (int t = 0; t < HM; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap[CNT], *apn[CNT];
    bool live[CNT];
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) {
        live[rt] = (u + rt) * 16 + i < A.n_rows;
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

    // epilogue, 16 rows at a time: + b1, pre stored, then k_act_linear_fwd's body on the registers.  Every group of stores
    // sits behind ONE branch (a branch per store cuts the epilogue into basic blocks, each with its own LDS read and wait),
    // and the two operands' MFMA chains (4 HM dependent instructions each) are interleaved.
#pragma unroll
    for (int rt = 0; rt < CNT; ++rt) {
        const int64_t row = (u + rt) * 16 + i;
        float v[HM][4];
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            const float4 b = *reinterpret_cast<const float4 *>(b1g + 16 * t);
            v[t][0] = acc[rt][t][0] + b.x; v[t][1] = acc[rt][t][1] + b.y; v[t][2] = acc[rt][t][2] + b.z; v[t][3] = acc[rt][t][3] + b.w;
        }
        if (A.pre && live[rt]) {
            float *dst = A.pre + row * H + 4 * g;
#pragma unroll
            for (int t = 0; t < HM; ++t) *reinterpret_cast<float4 *>(dst + 16 * t) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
        }
        f32x4 acc_tr = {0.f, 0.f, 0.f, 0.f}, acc_ev = {0.f, 0.f, 0.f, 0.f};
        uint32_t part[4] = {0u, 0u, 0u, 0u};  // keep bits of this lane's elements, bit 4t + g of column q's word
#pragma unroll
        for (int t = 0; t < HM; ++t) {
            const float4 wv = *reinterpret_cast<const float4 *>(w2l + 4 * ((4 * t) ^ gi));
            const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
            float o[4];
            if (TRAIN) {
                const int64_t e = row * LPR + 4 * t + g;   // element-quad index of k_relu_dropout_fwd's numbering
                uint32_t r[4];
                philox4x32_10((uint64_t)e, A.offset, A.seed, r);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool keep = live[rt] && v[t][q] > 0.f && r[q] >= A.threshold;
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
            if (g == 0 && (i % RPW) == 0 && live[rt]) {
                unsigned long long *dst = A.bits + (row / RPW) * 4;   // (32-byte aligned: two 16-byte stores)
                *reinterpret_cast<ulonglong2 *>(dst) = make_ulonglong2(words[0], words[1]);
                *reinterpret_cast<ulonglong2 *>(dst + 2) = make_ulonglong2(words[2], words[3]);
            }
        }
        // accumulator register r of lane (i, g) is z[row 4g + r of the unit][class i]
        const int64_t orow = (u + rt) * 16 + 4 * g;
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
                              uint32_t threshold, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        DCR_HIP(hipGetDevice(&dev));
        DCR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 1) cus = 1;
    }
    const size_t lds = first_layer_lds_bytes(F, 16 * HM);
    const int64_t n_units = (n_rows + 15) / 16;
    int64_t grid = (n_units + FIRST_WAVES * NR - 1) / (FIRST_WAVES * NR);
    if (grid > cus) grid = cus;
    const int64_t waves = grid * FIRST_WAVES;
    const int64_t upw = (n_units + waves - 1) / waves;
    FirstArgs args{ax, ldx, pre, z_train, z_eval, ldz, bits, n_rows, F, C, scale, threshold, seed, offset};
#define DCR_FIRST_LAUNCH(TR, EV)                                                                                                       \
    do {                                                                                                                              \
        auto kern = k_first_layer_fwd<HM, NR, TR, EV>;                                                                                \
        static bool raised = false;                                                                                                   \
        if (!raised) {                                                                                                                \
            DCR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            raised = true;                                                                                                            \
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

}  // namespace dcr

using namespace dcr;

extern "C" int dcr_first_layer_fits(int in_features, int hidden, int classes) {
    if (in_features < 16 || (in_features % 16) != 0 || (hidden != 64 && hidden != 128) || classes < 1 || classes > 16) return 0;
    return first_layer_lds_bytes(in_features, hidden) <= 160 * 1024 ? 1 : 0;
}

extern "C" int dcr_first_layer_fwd_f32_dev(const float *ax, int64_t ldx, const float *w1, const float *b1, const float *w2, float *pre,
                                           float *z_train, float *z_eval, int64_t ldz, uint64_t *bits, int64_t n_rows, int in_features,
                                           int hidden, int classes, double p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                           void *hip_stream) {
    const bool train = z_train != nullptr, eval = z_eval != nullptr;
    if (!ax || !w1 || !w2 || n_rows < 0 || (!train && !eval) || ldx < in_features) DCR_FAIL(DCR_EINVAL, "bad first_layer_fwd arguments");
    if (train && (!bits || !pre || !(p >= 0.0 && p < 1.0)))
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: the training output needs bits, pre (the backward pass reads both) and 0 <= p < 1");
    if (!dcr_first_layer_fits(in_features, hidden, classes) || ldz < classes)
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: in_features a multiple of 16 with W1 fitting the LDS (dcr_first_layer_fits), hidden 64 or 128, "
                             "at most 16 classes, ldz >= classes (other shapes take the GEMM library and dcr_act_linear_fwd_f32_dev)");
    if (((uintptr_t)ax & 15) || (ldx & 3) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15) || (pre && ((uintptr_t)pre & 15)))
        DCR_FAIL(DCR_EINVAL, "first_layer_fwd: 16-byte aligned tensors and row stride expected");
    if (n_rows == 0) return DCR_OK;
    const double th = p * 4294967296.0;
    const uint32_t threshold = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)th;
    const float scale = (float)(1.0 / (1.0 - p));
    hipStream_t st = (hipStream_t)hip_stream;
    if (hidden == 128)
        return launch_first_layer<8, DCR_FIRST_NR>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows,
                                                   in_features, classes, scale, threshold, seed, offset, offset_dev, st);
    return launch_first_layer<4, DCR_FIRST_NR>(train, eval, ax, ldx, w1, b1, w2, pre, z_train, z_eval, ldz, (unsigned long long *)bits, n_rows,
                                               in_features, classes, scale, threshold, seed, offset, offset_dev, st);
}
