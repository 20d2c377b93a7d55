// Lab: issue rate of the fp32 MFMA shapes on gfx950 (cycles per instruction, one wave, N independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void k16(float* out, long long* cyc, int iters) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k32(float* out, long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 32x32x16 (the K1 split kernels' instruction): CH independent accumulators, every wave of the block runs the loop
template <int CH>
__global__ void kb32(float* out, long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + threadIdx.x - e); }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const int iters = 4096;
#define RUN(K, CH, THREADS)                                                                     \
    hipLaunchKernelGGL((K<CH>), dim3(1), dim3(THREADS), 0, 0, out, cyc, iters);                 \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                               \
    printf(#K " chains=%d waves=%d: %.1f cycles per MFMA per wave\n", CH, THREADS / 64, (double)h / iters / CH);
    RUN(k16, 1, 64) RUN(k16, 2, 64) RUN(k16, 4, 64) RUN(k16, 4, 256) RUN(k16, 4, 512)
    RUN(k32, 1, 64) RUN(k32, 2, 64) RUN(k32, 4, 64) RUN(k32, 4, 256) RUN(k32, 4, 512)
    RUN(kb32, 1, 64) RUN(kb32, 2, 64) RUN(kb32, 4, 64) RUN(kb32, 1, 256) RUN(kb32, 4, 256) RUN(kb32, 1, 512) RUN(kb32, 4, 512) RUN(kb32, 4, 768)
    return 0;
}
