// Shared device helpers for the dpk kernels (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpk.h"

#define DPK_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Butterfly sum over the 64 lanes of a wave; every lane ends with the same value and the
// combination order is fixed, so the result is reproducible run to run.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, DPK_WAVE);
    return v;
}

// Sum over a block of NW waves through LDS; result valid in thread 0.  `red` holds NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[i];
    }
    return s;
}

static inline int dpk_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DPK_OK : (int)e;
}

static inline bool dpk_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
