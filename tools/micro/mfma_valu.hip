// Do vector-ALU instructions issue under f32 MFMAs of the same wave / of the other wave of the SIMD?  A loop of 32 independent
// v_mfma_f32_16x16x4_f32 with NV 32x32->64-bit integer multiplies (the Philox round's instruction) woven in, per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool MFMA>
__global__ void __launch_bounds__(256) k(float *out, const float *in, int iters) {
    f32x4 acc[32];
    for (int a = 0; a < 32; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = in[threadIdx.x + 256 * q]; y[q] = in[threadIdx.x + 256 * (q + 4)]; }
    unsigned c0 = threadIdx.x * 2654435761u, c1 = blockIdx.x * 40503u + 1u, c2 = c0 ^ 0x9E3779B9u, c3 = c1 + 77u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            if (MFMA) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[a & 3], y[(a >> 2) & 3], acc[a], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {   // one Philox half-round: a 32x32->64 multiply and two xors
                const unsigned long long p = (unsigned long long)0xD2511F53u * c0;
                const unsigned n0 = (unsigned)(p >> 32) ^ c1 ^ c3;
                c1 = c2; c2 = (unsigned)p; c3 += 0x9E3779B9u; c0 = n0;
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 32; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(c0 ^ c1 ^ c2 ^ c3);
}

template <typename K>
static float run(K kern, int grid, int iters, float *out, float *in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out, *in;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8); hipMalloc(&in, sizeof(float) * 4096);
    hipMemset(in, 0x3c, sizeof(float) * 4096);
    const int iters = 4000;
    for (int w = 1; w <= 2; ++w) {
        const int grid = cus * w;
        const float m0 = run(k<0, true>, grid, iters, out, in);
        printf("waves/SIMD %d: MFMA only %.3f ms", w, m0);
        printf(" | +1 mul/MFMA %.3f (VALU alone %.3f)", run(k<1, true>, grid, iters, out, in), run(k<1, false>, grid, iters, out, in));
        printf(" | +2 %.3f (alone %.3f)", run(k<2, true>, grid, iters, out, in), run(k<2, false>, grid, iters, out, in));
        printf(" | +4 %.3f (alone %.3f)\n", run(k<4, true>, grid, iters, out, in), run(k<4, false>, grid, iters, out, in));
    }
    return 0;
}
