// Lab: how much vector (VALU) work a second wave of a SIMD gets done beside a wave that issues MFMAs back to back.
// Block of 512 threads = 2 waves per SIMD: waves 0-3 run the MFMA loop (CH accumulator chains), waves 4-7 the VALU loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH, bool MFMA_ON, bool VALU_ON, int VKIND>
__global__ void k(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    long long t0 = clock64(), t1;
    float s = 0;
    if (wave < 4) {
        f32x16 acc[CH];
        for (int c = 0; c < CH; ++c)
            for (int e = 0; e < 16; ++e) acc[c][e] = 0;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + threadIdx.x - e); }
        if (MFMA_ON)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
            }
        t1 = clock64();
        for (int c = 0; c < CH; ++c)
            for (int e = 0; e < 16; ++e) s += acc[c][e];
    } else {
        float x[8];
        for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.5f + e;
        if (VALU_ON)
            for (int i = 0; i < iters * CH / 8; ++i) {   // 64 VALU per iteration, 8 independent chains
#pragma unroll
                for (int ee = 0; ee < 64; ++ee) {
                    const int e = ee & 7;
                    if (VKIND == 0) x[e] = __builtin_fmaf(x[e], 1.0001f, 0.5f);
                    if (VKIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[e]));
                    if (VKIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[e]));
                }
            }
        t1 = clock64();
        for (int e = 0; e < 8; ++e) s += x[e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; long long* cyc; long long h[8];
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    const int iters = 4096;
#define RUN(CH, M, V, KIND)                                                                            \
    hipLaunchKernelGGL((k<CH, M, V, KIND>), dim3(1), dim3(512), 0, 0, out, cyc, iters);                \
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);                                                \
    printf("chains=%d mfma=%d valu=%d kind=%d: MFMA wave %.1f cyc/MFMA, VALU wave %.2f cyc/VALU\n", CH, M, V, KIND, \
           (double)h[0] / iters / CH, (double)h[4] / iters / CH / 8);
    RUN(1, true, false, 0) RUN(4, true, false, 0) RUN(4, false, true, 0) RUN(4, false, true, 1) RUN(4, false, true, 2)
    RUN(1, true, true, 0) RUN(2, true, true, 0) RUN(4, true, true, 0) RUN(4, true, true, 1) RUN(4, true, true, 2)
    return 0;
}
