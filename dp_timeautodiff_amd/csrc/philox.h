// Philox4x32-10 counter RNG + Box-Muller.  The DP noise of every rank comes from the same
// (seed, offset) stream: element j = normal #(j & 3) of block (offset + j / 4).
// Bit-for-bit the integer pipeline of oracle/opacus_ref.py::philox4x32_10.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t block, uint32_t k0, uint32_t k1) {
    uint32_t c0 = (uint32_t)block, c1 = (uint32_t)(block >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

__device__ __forceinline__ float philox_u01(uint32_t x) {
    return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f;  // 2^-24, u in (0,1)
}

// four standard normals of one Philox block
__device__ __forceinline__ void philox_normal4(uint64_t block, uint32_t k0, uint32_t k1, float out[4]) {
    const Philox4 p = philox4x32_10(block, k0, k1);
    const float two_pi = 6.28318530717958647692f;
    float r = sqrtf(-2.0f * logf(philox_u01(p.x)));
    float s, c;
    sincosf(two_pi * philox_u01(p.y), &s, &c);
    out[0] = r * c;
    out[1] = r * s;
    r = sqrtf(-2.0f * logf(philox_u01(p.z)));
    sincosf(two_pi * philox_u01(p.w), &s, &c);
    out[2] = r * c;
    out[3] = r * s;
}
