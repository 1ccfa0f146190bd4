// Dense float32 Balanced Forman kernels: the numerics of the reference's numba path, for reproducing results obtained
// with it.  rewire('bfc') in the reference runs these formulas (rewiring/rewire.py:8-10 ->
// rewiring/sdrf_cuda_bfc.py -> curvature/bfc_cuda.py), and they differ from curvature/bfc_naive.py.
//
//   dcr_bfc_dense_f32_dev              replaces _balanced_forman_curvature, curvature/bfc_cuda.py:11-48
//   dcr_bfc_dense_post_delta_f32_dev   replaces _balanced_forman_post_delta, curvature/bfc_cuda.py:68-141
//
// Same data layout as the reference (dense row-major N x N float32 A and A2 = A·A, degree vectors), because the callers
// (dense arg-min over C including the zeros of non-edges, sdrf_cuda_bfc.py:40,80) are defined on it; N is bounded by what
// 3·4·N² bytes allow, as in the reference.  What changes is the execution: the reference starts N² threads of which the
// nnz non-zero ones each loop over N; here one 64-lane wave takes one non-zero pair and its lanes share the loop, with
// a wave reduction of the two integers-as-floats (count of positive terms, largest term).
//
// Arithmetic.  numba types these kernels with float32 arrays and int64 literals, which unify to float64: the closing
// expression is evaluated in float64 on float32-valued inputs and rounded to float32 where it is stored (the base
// expression, then `+=` of the 4-cycle term: two roundings).  The same here: double arithmetic in the reference's
// operation order, -ffp-contract=off, two float stores.  Pinned bit for bit by tests/golden/bfc_cuda_curvature.json.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcr_internal.h"

namespace dcr {

__device__ inline float dense_closing(double d_max, double d_min, double a2_xy, double a_xy, int sharp, double lam) {
    double r = 2.0 / d_max;          // ((2 / d_max) + (2 / d_min) - 2 + (2 / d_max + 1 / d_min) * A2 * A), bfc_cuda.py:46
    r = r + 2.0 / d_min;
    r = r - 2.0;
    double m = 2.0 / d_max + 1.0 / d_min;
    m = m * a2_xy;
    m = m * a_xy;
    float c = (float)(r + m);
    if (lam > 0.0) c = (float)((double)c + (double)sharp / (d_max * lam));  // bfc_cuda.py:47-48
    return c;
}

__device__ inline void wave_sum_max(int &sharp, float &lam) {
    for (int off = 32; off > 0; off >>= 1) {
        sharp += __shfl_xor(sharp, off);
        const float o = __shfl_xor(lam, off);
        lam = o > lam ? o : lam;
    }
}

// one wave per non-zero pair (i, j) of A
__global__ void __launch_bounds__(256) k_bfc_dense(const float *A, const float *A2, const float *d_in, const float *d_out,
                                                   int64_t N, const int64_t *pairs, int64_t nnz, float *C) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= nnz) return;
    const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
    const float a_ij = A[i * N + j];
    float d_max, d_min;
    if (d_in[i] > d_out[j]) {  // bfc_cuda.py:20-25
        d_max = d_in[i];
        d_min = d_out[j];
    } else {
        d_max = d_out[j];
        d_min = d_in[i];
    }
    if (d_max * d_min == 0.f) {
        if (lane == 0) C[i * N + j] = 0.f;
        return;
    }
    int sharp = 0;
    float lam = 0.f;
    for (int64_t k = lane; k < N; k += 64) {  // bfc_cuda.py:33-44; every factor is a small integer: exact in float32
        const float a_ik = A[i * N + k], a_kj = A[k * N + j];
        const float t1 = a_kj * (A2[i * N + k] - a_ik) * a_ij;
        const float t2 = a_ik * (A2[k * N + j] - a_kj) * a_ij;
        sharp += (t1 > 0.f) + (t2 > 0.f);
        lam = t1 > lam ? t1 : lam;
        lam = t2 > lam ? t2 : lam;
    }
    wave_sum_max(sharp, lam);
    if (lane == 0) C[i * N + j] = dense_closing((double)d_max, (double)d_min, (double)A2[i * N + j], (double)a_ij, sharp, (double)lam);
}

// one wave per candidate (I, J): curvature of (x, y) in the graph with the edge (i_I, j_J) added
__global__ void __launch_bounds__(256) k_bfc_dense_post_delta(const float *A, const float *A2, float d_in_x0, float d_out_y0,
                                                              int64_t N, float *D, int x, int y, const int32_t *i_nb,
                                                              const int32_t *j_nb, int64_t dim_i, int64_t dim_j) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= dim_i * dim_j) return;
    const int64_t I = p / dim_j, J = p - I * dim_j;
    const int i = i_nb[I], j = j_nb[J];
    if (i < 0 || j < 0 || i >= N || j >= N) return;
    if (i == j || A[(int64_t)i * N + j] != 0.f) {  // bfc_cuda.py:77-79
        if (lane == 0) D[p] = -1000.f;
        return;
    }
    double d_in_x = d_in_x0, d_out_y = d_out_y0;  // (float32 + int64 -> float64 in numba's typing)
    if (j == x) d_in_x += 1.0;
    else if (i == y) d_out_y += 1.0;
    if (d_in_x * d_out_y == 0.0) {
        if (lane == 0) D[p] = 0.f;
        return;
    }
    const double d_max = d_in_x > d_out_y ? d_in_x : d_out_y, d_min = d_in_x > d_out_y ? d_out_y : d_in_x;
    const float a_xy = A[(int64_t)x * N + y], a_jy = A[(int64_t)j * N + y], a_xi = A[(int64_t)x * N + i];
    double a2_xy = A2[(int64_t)x * N + y];
    if (x == i && a_jy != 0.f) a2_xy += a_jy;       // bfc_cuda.py:99-103
    else if (y == j && a_xi != 0.f) a2_xy += a_xi;
    int sharp = 0;
    float lam = 0.f;
    for (int64_t z = lane; z < N; z += 64) {        // bfc_cuda.py:108-137 (small integers: exact in float32)
        float a_zy = A[z * N + y], a_xz = A[(int64_t)x * N + z];
        float a2_zy = A2[z * N + y], a2_xz = A2[(int64_t)x * N + z];
        if (z == i && y == j) a_zy += 1.f;
        if (x == i && z == j) a_xz += 1.f;
        if (z == i && a_jy != 0.f) a2_zy += a_jy;
        if (x == i) {
            const float a_jz = A[(int64_t)j * N + z];
            if (a_jz != 0.f) a2_xz += a_jz;
        }
        if (y == j) {
            const float a_zi = A[z * N + i];
            if (a_zi != 0.f) a2_zy += a_zi;
        }
        if (z == j && a_xi != 0.f) a2_xz += a_xi;
        const float t1 = a_zy * (a2_xz - a_xz) * a_xy;
        const float t2 = a_xz * (a2_zy - a_zy) * a_xy;
        sharp += (t1 > 0.f) + (t2 > 0.f);
        lam = t1 > lam ? t1 : lam;
        lam = t2 > lam ? t2 : lam;
    }
    wave_sum_max(sharp, lam);
    if (lane == 0) D[p] = dense_closing(d_max, d_min, a2_xy, (double)a_xy, sharp, (double)lam);
}

}  // namespace dcr

using namespace dcr;

extern "C" {

int dcr_bfc_dense_f32_dev(const float *A_dev, const float *A2_dev, const float *d_in_dev, const float *d_out_dev, int64_t N,
                          const int64_t *pairs_dev, int64_t nnz, float *C_dev, void *hip_stream) {
    if (!A_dev || !A2_dev || !d_in_dev || !d_out_dev || !C_dev || N < 0 || nnz < 0 || (nnz > 0 && !pairs_dev))
        DCR_FAIL(DCR_EINVAL, "dcr_bfc_dense_f32_dev: bad argument");
    if (nnz == 0) return DCR_OK;
    const int64_t blocks = (nnz + 3) / 4;
    if (blocks > 0x7fffffffll) DCR_FAIL(DCR_ECAPACITY, "too many non-zero pairs for one launch");
    hipLaunchKernelGGL(k_bfc_dense, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, A_dev, A2_dev, d_in_dev,
                       d_out_dev, N, pairs_dev, nnz, C_dev);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

int dcr_bfc_dense_post_delta_f32_dev(const float *A_dev, const float *A2_dev, float d_in_x, float d_out_y, int64_t N,
                                     float *D_dev, int32_t x, int32_t y, const int32_t *i_neighbors_dev,
                                     const int32_t *j_neighbors_dev, int64_t dim_i, int64_t dim_j, void *hip_stream) {
    if (!A_dev || !A2_dev || !D_dev || N <= 0 || x < 0 || y < 0 || x >= N || y >= N || dim_i < 0 || dim_j < 0 ||
        (dim_i * dim_j > 0 && (!i_neighbors_dev || !j_neighbors_dev)))
        DCR_FAIL(DCR_EINVAL, "dcr_bfc_dense_post_delta_f32_dev: bad argument");
    const int64_t total = dim_i * dim_j;
    if (total == 0) return DCR_OK;
    const int64_t blocks = (total + 3) / 4;
    if (blocks > 0x7fffffffll) DCR_FAIL(DCR_ECAPACITY, "too many candidate pairs for one launch");
    hipLaunchKernelGGL(k_bfc_dense_post_delta, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, A_dev, A2_dev,
                       d_in_x, d_out_y, N, D_dev, (int)x, (int)y, i_neighbors_dev, j_neighbors_dev, dim_i, dim_j);
    DCR_HIP(hipGetLastError());
    return DCR_OK;
}

}  // extern "C"
