// Lab: the bf16 split's own instruction mix beside a wave that keeps the matrix pipe full (2 waves per SIMD: waves 0-3
// MFMA, waves 4-7 the split of register data).  KIND 0: as the kernels write it (compiler's choice of instructions);
// 1: the same with the SLP vectoriser's packed adds prevented (opaque barriers); 2: without the clamp.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); }
template <int KIND>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& w0, unsigned& w1, unsigned& w2) {
    const float M = __builtin_bit_cast(float, 0x7F7F0000u);
    if (KIND == 2) w0 = pack(x0, x1);
    else w0 = pack(__builtin_amdgcn_fmed3f(x0, -M, M), __builtin_amdgcn_fmed3f(x1, -M, M));
    x0 = x0 - __builtin_bit_cast(float, w0 << 16);
    x1 = x1 - __builtin_bit_cast(float, w0 & 0xffff0000u);
    if (KIND == 1) { asm volatile("" : "+v"(x0)); asm volatile("" : "+v"(x1)); }
    w1 = pack(x0, x1);
    x0 = x0 - __builtin_bit_cast(float, w1 << 16);
    x1 = x1 - __builtin_bit_cast(float, w1 & 0xffff0000u);
    if (KIND == 1) { asm volatile("" : "+v"(x0)); asm volatile("" : "+v"(x1)); }
    w2 = pack(x0, x1);
}
template <int KIND, bool MFMA_ON>
__global__ void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    long long t0 = clock64(), t1;
    float s = 0;
    if (wave < 4) {
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + threadIdx.x - e); }
        if (MFMA_ON)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
            }
        t1 = clock64();
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e) s += acc[c][e];
    } else {
        float x[16];
        for (int e = 0; e < 16; ++e) x[e] = threadIdx.x * 0.37f + e * 1.01f;
        unsigned acc = 0;
        for (int i = 0; i < iters / 2; ++i) {   // 8 column pairs (one task) per iteration
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned w0, w1, w2;
                split_pair<KIND>(x[2 * e], x[2 * e + 1], w0, w1, w2);
                acc ^= w0 ^ w1 ^ w2;
                x[2 * e] += 1.0f;
            }
        }
        t1 = clock64();
        s = __builtin_bit_cast(float, acc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc; long long h[8];
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    const int iters = 4096;
#define RUN(KIND, M)                                                                                   \
    hipLaunchKernelGGL((k<KIND, M>), dim3(1), dim3(512), 0, 0, out, cyc, iters);                       \
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);                                                \
    printf("kind=%d mfma=%d: MFMA wave %.1f cyc/MFMA, split wave %.0f cycles per task (8 column pairs)\n", KIND, M, \
           (double)h[0] / iters / 4, (double)h[4] / (iters / 2));
    RUN(0, false) RUN(0, true) RUN(1, false) RUN(1, true) RUN(2, false) RUN(2, true)
    return 0;
}
