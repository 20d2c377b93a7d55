// K11 — on-device Poisson sampler.  Replaces the host side of Opacus' DPDataLoader as the reference
// reaches it through make_private_with_epsilon (Model_Code/timeautoencoder.py:664-682, consumed at :726-733):
//   mask = rand(N) < sample_rate;  indices = mask.nonzero();  batch = collate([dataset[i] for i in indices]);  H2D copy
// Here sample j is selected iff u_j < q, u_j = lane (j & 3) of Philox block (offset + j / 4) under the sampler's own key,
// and the selected indices are compacted in increasing order into idx[0..n).  The output has a FIXED capacity so the
// whole step stays shape-static (hipGraph replay): idx[n..capacity) is padded with index 0 and counts[0] = n tells K6
// which arena rows are real (the rest get clip factor 0).  counts[1] = number selected before truncation; the caller
// sizes `capacity` so that overflow is astronomically unlikely and reports counts[1] > capacity if it ever happens.
// One workgroup, deterministic: a block scan per 4096-element sweep (N is 10^3..10^6 series windows; microseconds).
#include "dpk_common.h"
#include "philox.h"

namespace {

constexpr int PNT = 1024;

__global__ void __launch_bounds__(PNT) poisson_sample_kernel(uint32_t k0, uint32_t k1, uint64_t offset,
                                                            const uint64_t* __restrict__ offset_dev, float q,
                                                            int64_t N, int capacity, int64_t* __restrict__ idx,
                                                            int* __restrict__ counts) {
    __shared__ int wsum[PNT / 64];
    __shared__ int s_running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t off = offset + (offset_dev ? *offset_dev : 0ull);
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (int64_t base = 0; base < N; base += 4 * PNT) {
        const int64_t j0 = base + (int64_t)tid * 4;
        int flag[4] = {0, 0, 0, 0};
        if (j0 < N) {
            const Philox4 p = philox4x32_10(off + (uint64_t)(j0 >> 2), k0, k1);
            const uint32_t x[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) flag[e] = (j0 + e < N) && (philox_u01(x[e]) < q);
        }
        const int cnt = flag[0] + flag[1] + flag[2] + flag[3];
        int incl = cnt;  // inclusive scan within the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PNT / 64; ++w) {
            const int v = wsum[w];
            if (w < wave) wbase += v;
            total += v;
        }
        const int running = s_running;
        int pos = running + wbase + incl - cnt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (flag[e]) {
                if (pos < capacity) idx[pos] = j0 + e;
                ++pos;
            }
        }
        __syncthreads();  // everyone has read s_running and wsum
        if (tid == 0) s_running = running + total;
        __syncthreads();
    }
    const int n_total = s_running;
    const int n = n_total < capacity ? n_total : capacity;
    for (int i = n + tid; i < capacity; i += PNT) idx[i] = 0;  // padding rows: a valid index, masked out by K6
    if (tid == 0) {
        counts[0] = n;
        counts[1] = n_total;
    }
}

}  // namespace

extern "C" int dpk_poisson_sample(uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float sample_rate,
                                  int64_t N, int capacity, int64_t* idx, int* counts, void* stream) {
    if (!idx || !counts || N <= 0 || capacity <= 0) return DPK_EINVAL;
    if (!(sample_rate >= 0.f && sample_rate <= 1.f)) return DPK_EINVAL;
    if (N > (int64_t)1 << 31) return DPK_ERANGE;
    hipLaunchKernelGGL(poisson_sample_kernel, dim3(1), dim3(PNT), 0, (hipStream_t)stream, (uint32_t)seed,
                       (uint32_t)(seed >> 32), offset, offset_dev, sample_rate, N, capacity, idx, counts);
    return dpk_launch_status();
}
