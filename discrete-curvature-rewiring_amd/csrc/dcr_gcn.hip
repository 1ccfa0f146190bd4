// GCN sparse aggregation of libdcr_hip.so: C = Â · B (+ bias, ReLU) on CSR, fp32.
//
// Replaces the propagate/scatter-add of torch_geometric GCNConv (third-party; call site models/gcn.py:36).
// HBM / Infinity-Cache gather-bound: every non-zero pulls one row of B (n_feat floats).  A row of Â is owned by a group of LPR lanes,
// each lane holding VEC consecutive features, so one wave-instruction reads (64/LPR) rows of B in 16-byte
// pieces (full 128-B lines for n_feat >= 32) and a wave covers 64/LPR output rows.  MFMA has nothing to do here:
// the dense contraction (X·Wᵀ) is done before this kernel on the matrix cores by the GEMM library.
#include "dcr_internal.h"

namespace dcr {

// accumulate non-zeros e0, e0 + stride, ... < e1 of one row into acc (U gathers of B in flight)
template <int VEC, int U>
__device__ inline void spmm_accumulate(const int32_t *__restrict__ col, const float *__restrict__ val,
                                       const float *__restrict__ B, int64_t ldb, int f0, int64_t e0, int64_t e1,
                                       int64_t stride, float (&acc)[VEC]) {
    int64_t e = e0;
    for (; e + (U - 1) * stride < e1; e += U * stride) {
        int c[U];
        float w[U];
        float b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = col[e + u * stride];
            w[u] = val[e + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float *src = B + (int64_t)c[u] * ldb + f0;
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(src);
                b[u][0] = t.x; b[u][1 % VEC] = t.y; b[u][2 % VEC] = t.z; b[u][3 % VEC] = t.w;
            } else if (VEC == 2) {
                const float2 t = *reinterpret_cast<const float2 *>(src);
                b[u][0] = t.x; b[u][1 % VEC] = t.y;
            } else {
                b[u][0] = src[0];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = fmaf(w[u], b[u][q], acc[q]);
    }
    for (; e < e1; e += stride) {
        const int c = col[e];
        const float w = val[e];
        const float *src = B + (int64_t)c * ldb + f0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = fmaf(w, src[q], acc[q]);
    }
}

// Rows up to SPMM_LONG non-zeros: one LPR-lane group per row, non-zeros in order.  Longer rows (the hubs of a
// power-law graph: thousands of non-zeros, which one group would chew through long after the rest of the grid has
// finished) are parked in LDS and then taken by the whole workgroup: group g accumulates non-zeros g, g + G, ... and
// the partial sums are added in group order, so the result is deterministic.
constexpr int SPMM_LONG = 96;

template <int LPR, int VEC>
__global__ void __launch_bounds__(256) k_spmm_csr(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ val, const float *__restrict__ B,
                                                   float *__restrict__ C, int64_t n_rows, int n_feat, int64_t ldb,
                                                   int64_t ldc, const float *__restrict__ bias, int relu) {
    constexpr int ROWS_PER_BLOCK = 256 / LPR;
    constexpr int G = ROWS_PER_BLOCK;        // lane groups per workgroup
    constexpr int U = LPR <= 8 ? 8 : 4;      // narrow rows of B: more gathers in flight per group
    __shared__ int long_rows[ROWS_PER_BLOCK];
    __shared__ int n_long;
    __shared__ float red[256 * VEC];
    const int sub = threadIdx.x / LPR;       // which row of the block / which group
    const int sl = threadIdx.x % LPR;        // lane inside the group
    // group `sub` of workgroup b owns row sub * gridDim + b: consecutive (hub) rows land in different workgroups
    const int64_t row = (int64_t)sub * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    if (row < n_rows) {
        const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
        if (e1 - e0 > SPMM_LONG) {
            if (sl == 0) long_rows[atomicAdd(&n_long, 1)] = sub;
        } else {
            for (int f0 = sl * VEC; f0 < n_feat; f0 += LPR * VEC) {
                float acc[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
                spmm_accumulate<VEC, U>(col, val, B, ldb, f0, e0, e1, 1, acc);
                float *dst = C + row * ldc + f0;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    float r = acc[q];
                    if (bias) r += bias[f0 + q];
                    if (relu) r = r > 0.f ? r : 0.f;
                    dst[q] = r;
                }
            }
        }
    }
    __syncthreads();
    const int nl = n_long;  // uniform
    for (int li = 0; li < nl; ++li) {
        const int64_t lrow = (int64_t)long_rows[li] * gridDim.x + blockIdx.x;
        const int64_t e0 = rowptr[lrow], e1 = rowptr[lrow + 1];
        for (int fb = 0; fb < n_feat; fb += LPR * VEC) {  // uniform trip count
            const int f0 = fb + sl * VEC;
            float acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
            if (f0 < n_feat) spmm_accumulate<VEC, U>(col, val, B, ldb, f0, e0 + sub, e1, G, acc);
#pragma unroll
            for (int q = 0; q < VEC; ++q) red[threadIdx.x * VEC + q] = acc[q];
            __syncthreads();
            if (sub == 0 && f0 < n_feat) {
                float *dst = C + lrow * ldc + f0;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    float r = 0.f;
                    for (int g = 0; g < G; ++g) r += red[(g * LPR + sl) * VEC + q];
                    if (bias) r += bias[f0 + q];
                    if (relu) r = r > 0.f ? r : 0.f;
                    dst[q] = r;
                }
            }
            __syncthreads();
        }
    }
}

template <int LPR, int VEC>
static void launch_spmm(const int64_t *rowptr, const int32_t *col, const float *val, const float *B, float *C,
                        int64_t n_rows, int n_feat, int64_t ldb, int64_t ldc, const float *bias, int relu,
                        hipStream_t st) {
    constexpr int ROWS_PER_BLOCK = 256 / LPR;
    const int64_t blocks = (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    hipLaunchKernelGGL((k_spmm_csr<LPR, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, val, B, C, n_rows,
                       n_feat, ldb, ldc, bias, relu);
}

}  // namespace dcr

using namespace dcr;

extern "C" int dcr_spmm_csr_f32_dev(const int64_t *rowptr, const int32_t *col, const float *val, const float *B,
                                    float *C, int64_t n_rows, int64_t n_feat, int64_t ldb, int64_t ldc,
                                    const float *bias, int relu, void *hip_stream) {
    if (!rowptr || !B || !C || n_rows < 0 || n_feat <= 0 || ldb < n_feat || ldc < n_feat)
        DCR_FAIL(DCR_EINVAL, "bad SpMM arguments");
    if (n_rows == 0) return DCR_OK;
    if (n_feat > INT32_MAX) DCR_FAIL(DCR_EINVAL, "n_feat too large");
    hipStream_t st = (hipStream_t)hip_stream;
    const int F = (int)n_feat;
    const bool v4 = (F % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && (((uintptr_t)B & 15) == 0);
    const bool v2 = (F % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0) && (((uintptr_t)B & 7) == 0);
#define GO(L, V) launch_spmm<L, V>(rowptr, col, val, B, C, n_rows, F, ldb, ldc, bias, relu, st)
    if (v4) {
        const int lanes = F / 4;
        if (lanes <= 4) GO(4, 4);
        else if (lanes <= 8) GO(8, 4);
        else if (lanes <= 16) GO(16, 4);
        else if (lanes <= 32) GO(32, 4);
        else GO(64, 4);
    } else if (v2) {
        const int lanes = F / 2;
        if (lanes <= 4) GO(4, 2);
        else if (lanes <= 8) GO(8, 2);
        else if (lanes <= 16) GO(16, 2);
        else if (lanes <= 32) GO(32, 2);
        else GO(64, 2);
    } else {
        if (F <= 4) GO(4, 1);
        else if (F <= 8) GO(8, 1);
        else if (F <= 16) GO(16, 1);
        else if (F <= 32) GO(32, 1);
        else GO(64, 1);
    }
#undef GO
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}
