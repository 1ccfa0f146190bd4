// Weight-gradient contraction of the GCN on the matrix cores: C[M x N] = Aᵀ · B with A [K x M], B [K x N] row-major
// and K = number of nodes (10^5 … 10^6), M, N = layer widths (tens to a few thousand).
//
// This is the backward of GCNConv's bias-free Linear (third-party torch_geometric Linear; call site models/gcn.py:36):
// dW = (dZ)ᵀ · X.  The GEMM library splits this tall-skinny reduction poorly (measured on MI355X: 2.0 ms for
// 128 x 256 x 1M and 1.4 ms for 16 x 128 x 1M, 33 and 3 TFLOP/s), so it is written by hand for gfx950:
//   * v_mfma_f32_32x32x2_f32 (exact f32, the result is a k-ordered fmaf chain).  Its A operand is "lane l holds
//     Aᵀ[i = l & 31][k = l >> 5]" = A[k0 + (l >> 5)][m0 + (l & 31)] and B likewise, so both fragments are plain coalesced
//     128-byte row segments of the row-major inputs: registers are filled straight from global memory, no LDS, no
//     transposition;
//   * a workgroup of 4 waves owns an output tile (up to 128 x 256, 8 accumulator tiles = 128 registers per lane) and a
//     slab of K; the waves of a workgroup re-read each other's rows from L1/L2, HBM sees every input byte once;
//   * the K-slabs' partial tiles go to a workspace and a second kernel adds them in slab order: deterministic, no atomics.
#include "dcr_internal.h"

namespace dcr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int ATB_KU = 8;  // MFMA k-steps (of 2 rows each) whose operands are loaded before the first MFMA (4: the narrow
                           // 16 x 128 shape 254 us instead of 195 at K = 1M; the 128 x 256 shape does not care)

// MT x NT accumulator tiles (32 x 32) per wave, waves arranged WGM x WGN inside the workgroup
template <int MT, int NT, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN) k_atb_partial(const float *__restrict__ A, const float *__restrict__ B,
                                                                 float *__restrict__ part, int64_t K, int M, int N,
                                                                 int64_t lda, int64_t ldb, int64_t k_chunk) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m_base = (blockIdx.x * WGM + wm) * MT * 32;
    const int n_base = (blockIdx.y * WGN + wn) * NT * 32;
    const int64_t k_begin = (int64_t)blockIdx.z * k_chunk;
    const int64_t k_end = k_begin + k_chunk < K ? k_begin + k_chunk : K;
    const int h = lane >> 5, c = lane & 31;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool m_ok[MT], n_ok[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) m_ok[i] = m_base + i * 32 + c < M;
#pragma unroll
    for (int j = 0; j < NT; ++j) n_ok[j] = n_base + j * 32 + c < N;

    // operands of the next 2 * ATB_KU rows are in flight while the matrix cores work on the current ones
    // Operands of ATB_KU k-steps are loaded, then their MFMAs issued.  (Two variants were measured slower on MI355X and
    // dropped: register double-buffering across iterations, 0.90 vs 0.88 ms at 128 x 256 x 1M, and unconditional loads
    // from clamped addresses zeroed by a bit mask, 0.98 ms; with two waves per SIMD the partner wave's MFMAs already
    // cover this wave's load latency.)
    for (int64_t k0 = k_begin; k0 < k_end; k0 += 2 * ATB_KU) {
        float a[ATB_KU][MT], b[ATB_KU][NT];
#pragma unroll
        for (int s = 0; s < ATB_KU; ++s) {
            const int64_t k = k0 + 2 * s + h;
            const bool k_ok = k < k_end;
            const float *ap = A + k * lda + m_base + c;
            const float *bp = B + k * ldb + n_base + c;
#pragma unroll
            for (int i = 0; i < MT; ++i) a[s][i] = (k_ok && m_ok[i]) ? ap[i * 32] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) b[s][j] = (k_ok && n_ok[j]) ? bp[j * 32] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < ATB_KU; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }

    // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = part + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_base + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M && n < N) out[(int64_t)m * N + n] = acc[i][j][r];
            }
        }
}

// Sum of the K-slabs' partial tiles in slab order.  64 outputs x 4 slab groups per workgroup: thread (e, zg) adds slabs
// zg, zg + 4, ... (four loads in flight), the four group sums are then added in group order: deterministic, and four
// times the memory-level parallelism of one thread per output.
__global__ void __launch_bounds__(256) k_atb_reduce(const float *__restrict__ part, float *__restrict__ C, int64_t mn,
                                                     int N, int64_t ldc, int splits, int ncols) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < mn) {
        int z = zg;
        for (; z + 12 < splits; z += 16) {
            const float a = part[(int64_t)z * mn + i], b = part[(int64_t)(z + 4) * mn + i],
                        c = part[(int64_t)(z + 8) * mn + i], d = part[(int64_t)(z + 12) * mn + i];
            s += a; s += b; s += c; s += d;
        }
        for (; z < splits; z += 4) s += part[(int64_t)z * mn + i];
    }
    red[zg][e] = s;
    __syncthreads();
    if (zg == 0 && i < mn && (int)(i % N) < ncols) C[(i / N) * ldc + (i % N)] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

struct AtbPlan {
    int cfg;          // 0: 128 x 256 tiles, 1: 64 x 256, 2: 32 x 128
    int tiles_m, tiles_n, splits;
    int64_t k_chunk;
};

static AtbPlan atb_plan(int64_t K, int64_t M, int64_t N) {
    AtbPlan p;
    p.cfg = M > 64 ? 0 : M > 32 ? 1 : 2;
    const int tm = p.cfg == 0 ? 128 : p.cfg == 1 ? 64 : 32;
    const int tn = p.cfg == 2 ? 128 : 256;
    p.tiles_m = (int)((M + tm - 1) / tm);
    p.tiles_n = (int)((N + tn - 1) / tn);
    // about two workgroups per CU in total, at least 512 rows per slab
    int64_t splits = 512 / ((int64_t)p.tiles_m * p.tiles_n);
    if (splits < 1) splits = 1;
    const int64_t max_by_k = (K + 511) / 512;
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
    int64_t chunk = (K + splits - 1) / splits;
    chunk = (chunk + 2 * ATB_KU - 1) / (2 * ATB_KU) * (2 * ATB_KU);
    if (chunk < 2 * ATB_KU) chunk = 2 * ATB_KU;
    p.k_chunk = chunk;
    p.splits = (int)((K + chunk - 1) / chunk);
    if (p.splits < 1) p.splits = 1;
    return p;
}

// C[i / N][i % N] = sum over the splits of part[z][i], in slab order (k_atb_reduce) — for the kernels of other files that leave
// K-slab partial tiles (csrc/dcr_gcn_first.hip)
void launch_slab_reduce(const float *part, float *C, int64_t mn, int N, int64_t ldc, int splits, hipStream_t st) {
    hipLaunchKernelGGL(k_atb_reduce, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, st, part, C, mn, N, ldc, splits, N);
}
// the same, writing only the first ncols columns of every row of the N-column tiles (padded widths: csrc/dcr_gcn_first.hip)
void launch_slab_reduce_cols(const float *part, float *C, int64_t mn, int N, int ncols, int64_t ldc, int splits, hipStream_t st) {
    hipLaunchKernelGGL(k_atb_reduce, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, st, part, C, mn, N, ldc, splits, ncols);
}

}  // namespace dcr

using namespace dcr;

extern "C" int dcr_atb_f32_workspace(int64_t K, int64_t M, int64_t N, int64_t *out_floats) {
    if (!out_floats || K < 0 || M <= 0 || N <= 0) DCR_FAIL(DCR_EINVAL, "bad AtB shape");
    const AtbPlan p = atb_plan(K, M, N);
    *out_floats = (int64_t)p.splits * M * N;
    return DCR_OK;
}

extern "C" int dcr_atb_f32_dev(const float *A, const float *B, float *C, int64_t K, int64_t M, int64_t N, int64_t lda,
                               int64_t ldb, int64_t ldc, float *workspace, int64_t workspace_floats, void *hip_stream) {
    if (((!A || !B) && K > 0) || !C || !workspace || K < 0 || M <= 0 || N <= 0 || lda < M || ldb < N || ldc < N)
        DCR_FAIL(DCR_EINVAL, "bad AtB arguments");
    if (M > INT32_MAX || N > INT32_MAX) DCR_FAIL(DCR_EINVAL, "AtB: M, N too large");
    const AtbPlan p = atb_plan(K, M, N);
    if (workspace_floats < (int64_t)p.splits * M * N) DCR_FAIL(DCR_ECAPACITY, "AtB workspace too small");
    if (p.tiles_n > 65535 || p.splits > 65535) DCR_FAIL(DCR_ECAPACITY, "AtB grid too large");
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((unsigned)p.tiles_m, (unsigned)p.tiles_n, (unsigned)p.splits);
    if (p.cfg == 0)
        hipLaunchKernelGGL((k_atb_partial<2, 4, 2, 2>), grid, dim3(256), 0, st, A, B, workspace, K, (int)M, (int)N, lda, ldb,
                           p.k_chunk);
    else if (p.cfg == 1)
        hipLaunchKernelGGL((k_atb_partial<2, 2, 1, 4>), grid, dim3(256), 0, st, A, B, workspace, K, (int)M, (int)N, lda, ldb,
                           p.k_chunk);
    else
        hipLaunchKernelGGL((k_atb_partial<1, 1, 1, 4>), grid, dim3(256), 0, st, A, B, workspace, K, (int)M, (int)N, lda, ldb,
                           p.k_chunk);
    const int64_t mn = M * N;
    hipLaunchKernelGGL(k_atb_reduce, dim3((unsigned)((mn + 63) / 64)), dim3(256), 0, st, workspace, C, mn, (int)N, ldc,
                       p.splits, (int)N);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}
