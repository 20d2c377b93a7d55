// Lab: two waves per SIMD, each alternating a block of NM MFMAs (CH accumulator chains) with a block of NV vector
// instructions (two dependent chains, like the bf16 split of one column pair).  SHIFT: waves 4-7 start with the vector
// block.  Reports cycles per (NM MFMAs + NV VALU) iteration of wave 0 and wave 4; the matrix pipe alone needs 2*NM*32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH, int NM, int NV, bool SHIFT>
__global__ void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + threadIdx.x - e); }
    float x0 = threadIdx.x * 0.5f, x1 = threadIdx.x * 0.25f + 1.f;
    auto vblock = [&]() {
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
            asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x0));
            asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x1));
        }
    };
    auto mblock = [&]() {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CH], 0, 0, 0);
    };
    __syncthreads();
    long long t0 = clock64();
    if (SHIFT && wave >= 4) vblock();
    for (int i = 0; i < iters; ++i) {
        mblock();
        __builtin_amdgcn_sched_barrier(0);
        vblock();
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    float s = x0 + x1;
    for (int c = 0; c < CH; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc; long long h[8];
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    const int iters = 2048;
#define RUN(CH, NM, NV, SH, THREADS)                                                                         \
    hipLaunchKernelGGL((k<CH, NM, NV, SH>), dim3(1), dim3(THREADS), 0, 0, out, cyc, iters);                  \
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);                                                      \
    printf("chains=%d NM=%d NV=%d shift=%d waves/SIMD=%d: wave0 %.0f, wave4 %.0f cycles per iteration (pipe alone %d)\n", CH, NM, NV, SH, \
           THREADS / 256, (double)h[0] / iters, (double)h[THREADS > 256 ? 4 : 0] / iters, (THREADS / 256) * NM * 32);
    RUN(1, 6, 22, false, 256) RUN(2, 6, 22, false, 256) RUN(1, 6, 0, false, 512) RUN(2, 6, 0, false, 512)
    RUN(1, 6, 22, false, 512) RUN(1, 6, 22, true, 512) RUN(2, 6, 22, false, 512) RUN(2, 6, 22, true, 512)
    RUN(1, 6, 44, false, 512) RUN(2, 6, 44, true, 512) RUN(1, 48, 176, false, 512) RUN(1, 48, 176, true, 512)
    return 0;
}
