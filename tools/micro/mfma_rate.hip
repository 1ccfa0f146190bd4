// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 from registers on MI355X: NW waves per SIMD, independent
// accumulators, no memory traffic in the loop.  Prints cycles per MFMA per SIMD (s_memtime) and TFLOP/s (wall).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float *out, const float *in, int iters, unsigned long long *cyc) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = in[threadIdx.x + 256 * q]; y[q] = in[threadIdx.x + 256 * (q + 4)]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[a & 3], y[(a >> 2) & 3], acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void __launch_bounds__(256) k32(float *out, const float *in, int iters, unsigned long long *cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = in[threadIdx.x + 256 * q]; y[q] = in[threadIdx.x + 256 * (q + 4)]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[a & 3], y[(a >> 2) & 3], acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
static void run(const char *name, K kern, int wgs_per_cu, int nacc, int iters, double flop_per_mfma, float *out, float *in, unsigned long long *cyc, int cus) {
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, in, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= grid;
    // s_memtime counts at 100 MHz: cycles of the shader clock = ticks * clock / 100 MHz -> report wall-based numbers
    const double mfma_per_simd = (double)iters * nacc * wgs_per_cu;   // 4 waves per workgroup, one per SIMD
    const double tflops = (double)grid * 4 * iters * nacc * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d  acc %2d  %8.3f ms  %7.1f TFLOP/s  %.1f ns per MFMA per SIMD (memtime ticks per wave %.0f)\n", name, wgs_per_cu, nacc, ms,
           tflops, ms * 1e6 / mfma_per_simd, mean);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8); hipMalloc(&in, sizeof(float) * 4096); hipMalloc(&cyc, 8 * cus * 8);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
    hipMemcpy(in, h.data(), sizeof(float) * 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int w = 1; w <= 2; ++w) {
        run("16x16x4 f32, 8 accumulators", k16<8>, w, 8, iters, 2048.0, out, in, cyc, cus);
        run("16x16x4 f32, 32 accumulators", k16<32>, w, 32, iters / 4, 2048.0, out, in, cyc, cus);
        run("32x32x2 f32, 4 accumulators", k32<4>, w, 4, iters, 4096.0, out, in, cyc, cus);
        run("32x32x2 f32, 8 accumulators", k32<8>, w, 8, iters / 2, 4096.0, out, in, cyc, cus);
    }
    return 0;
}
