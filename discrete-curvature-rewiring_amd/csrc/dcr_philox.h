// Philox-4x32-10 as the fused ReLU + dropout kernels draw it (csrc/dcr_gcn.hip, csrc/dcr_gcn_first.hip): keyed by
// (seed, call offset), counter = element-quad index >> 1 (philox_quad16 below).  One definition: the kernels must agree on every
// keep bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcr {

__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

__device__ inline void philox4x32_10(uint64_t index, uint64_t offset, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = c[j];
}

// 16 random bits per element (round 4; were 32): element-quad e draws call e >> 1 and takes the (e & 1) half of each of its four
// words; an element is kept when its 16 bits are >= threshold16 = floor(p · 65536), so P(keep) = 1 - threshold16 / 65536: exact at
// p = 0.5, within 2^-16 of 1 - p otherwise (the scale stays 1 / (1 - p), models/gcn.py:40-42 via torch.nn.Dropout).  Half the
// Philox calls per element: on this chip the vector ALU's time is the f32 matrix core's time, and a kernel that
// holds quads e and e ^ 1 in partner lanes draws each call once and exchanges the words (csrc/dcr_gcn_first.hip).
__device__ inline void philox_quad16(uint64_t quad, uint64_t offset, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t r[4];
    philox4x32_10(quad >> 1, offset, seed, r);
    const uint32_t sh = (uint32_t)(quad & 1) * 16u;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (r[j] >> sh) & 0xFFFFu;
}

inline uint32_t dropout_threshold16(double p) {
    const double th = p * 65536.0;
    return th >= 65535.0 ? 65535u : (uint32_t)th;
}

}  // namespace dcr
