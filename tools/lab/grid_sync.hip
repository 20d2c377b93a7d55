// Lab: cost of a device-scope barrier among a few workgroups on MI355X (for a weight-stationary small-batch recurrence).
// Every spin is bounded: a participant that waits too long sets an error flag and everybody leaves.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/grid_sync tools/lab/grid_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// participants: blocks with (blockIdx.x % stride) == 0; np of them
__global__ void __launch_bounds__(256) sync_kernel(unsigned* counter, float* slots, float* out, int* err, int np, int stride,
                                                   int iters, int payload) {
    if (blockIdx.x % stride) return;
    const int me = blockIdx.x / stride;
    if (me >= np) return;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // publish my slice (payload floats per thread-block, written by the first `payload` threads)
        if ((int)threadIdx.x < payload) slots[(it & 1) * np * 256 + me * 256 + threadIdx.x] = (float)(it + me);
        __threadfence();       // release (device scope)
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)np * (unsigned)(it + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    atomicExch(err, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __threadfence();       // acquire
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        // consume everybody's slice
        for (int p = 0; p < np; ++p)
            if ((int)threadIdx.x < payload)
                acc += __builtin_nontemporal_load(&slots[(it & 1) * np * 256 + p * 256 + threadIdx.x]);
    }
    if (threadIdx.x < 256) out[me * 256 + threadIdx.x] = acc;
}

int main() {
    unsigned* counter; float *slots, *out; int* err;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&slots, 2 * 64 * 256 * 4)); CK(hipMalloc(&out, 64 * 256 * 4)); CK(hipMalloc(&err, 4));
    const int iters = 2000;
    for (int stride : {1, 8}) {
        for (int np : {2, 4, 13, 26, 32}) {
            for (int payload : {0, 64, 256}) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(err, 0, 4));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(sync_kernel, dim3(np * stride), dim3(256), 0, 0, counter, slots, out, err, np, stride, iters, payload);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                float h0; CK(hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost));
                // expected acc for thread 0 (payload > 0): sum_it sum_p (it + p)
                double expect = 0; for (int it = 0; it < iters; ++it) for (int p = 0; p < np; ++p) expect += it + p;
                printf("stride %d (%s) np %2d payload %3d floats/wg: %.2f us per barrier  err=%d  check %s\n", stride,
                       stride == 8 ? "one XCD" : "spread", np, payload, ms * 1e3 / iters, herr,
                       payload == 0 ? "-" : (h0 == (float)expect ? "ok" : "MISMATCH"));
            }
        }
    }
    return 0;
}
