// How many LDS cycles go to bank conflicts when the 64 lanes of a wave hit UNIFORMLY RANDOM words of an LDS bitmap — the access
// pattern of the two-hop pass's Bloom bitmaps and hash tables (csrc/dcr_bfc_h2.hip: the word index is a hash of a node id) —
// against consecutive words (conflict-free) and against random words of a bitmap the size of the pass's.  Round-4 verdict:
// "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.45-0.48 in all five class kernels: re-lay the bitmaps (odd stride or XOR swizzle)".
// A swizzle permutes which bank a WORD lives in; it cannot spread 64 independent uniform draws over 64 banks — the expected
// largest bank load of 64 balls in 64 bins is ~3.5, whatever the permutation.  This measures that floor on the chip:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_random.hip -o tools/micro/lds_random
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d out -o p -- tools/micro/lds_random
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: lane l -> word l (+ a rotating base): conflict-free; 1: uniformly random words, returning atomic OR (the first sweep's
// test-and-set); 2: uniformly random words, plain 4-byte reads (the second sweep's test); 3: random 16-byte bucket reads (the
// exact tables' 4-slot buckets)
template <int MODE, int WORDS>
__global__ void __launch_bounds__(256) k_lds(unsigned *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned bm[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) bm[i] = 0u;
    __syncthreads();
    unsigned acc = 0u, key = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    for (int it = 0; it < iters; ++it) {
        key = mix(key + it);
        if (MODE == 0) {
            acc += atomicOr(&bm[(threadIdx.x + 64 * it) & (WORDS - 1)], 1u << (key & 31u));
        } else if (MODE == 1) {
            acc += atomicOr(&bm[key & (WORDS - 1)], 1u << ((key >> 20) & 31u));
        } else if (MODE == 2) {
            acc += bm[key & (WORDS - 1)];
        } else {
            const uint4 e = reinterpret_cast<const uint4 *>(bm)[key & (WORDS / 4 - 1)];
            acc += e.x + e.y + e.z + e.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    unsigned *out;
    hipMalloc(&out, sizeof(unsigned) * 256 * 1024);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_lds<0, 4096>), dim3(1024), dim3(256), 0, 0, out, iters);
        hipLaunchKernelGGL((k_lds<1, 512>), dim3(1024), dim3(256), 0, 0, out, iters);    // 2^14 bits: the lightest wave class
        hipLaunchKernelGGL((k_lds<1, 4096>), dim3(1024), dim3(256), 0, 0, out, iters);   // 2^17 bits: the split class
        hipLaunchKernelGGL((k_lds<2, 4096>), dim3(1024), dim3(256), 0, 0, out, iters);
        hipLaunchKernelGGL((k_lds<3, 8192>), dim3(1024), dim3(256), 0, 0, out, iters);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
