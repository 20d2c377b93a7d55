// K5 (squared-norm partials of a flat [B,P] matrix), K6 (per-sample norm + clip factor) and
// K7 (clipped weighted sum over the batch + Gaussian noise + scale) — the DPOptimizer.step tail.
//
// K7 is the byte mover of the step: it streams the whole [B, P] per-sample-gradient arena ONCE.
// Each thread owns 4 consecutive columns (one 16-byte load per sample row, a wave reads 1 KiB
// contiguous per row) and walks the batch with 8 independent loads in flight; clip[b] is a
// wave-uniform scalar load.  When the column count alone cannot fill the chip the batch is split
// over gridDim.y and the partial sums are combined in split order by the finalize kernel, so the
// result is reproducible (no float atomics).
#include <stdlib.h>

#include <mutex>

#include "dpk_common.h"
#include "philox.h"

namespace {

constexpr int K5_CHUNK = 8192;  // floats per normsq partial slot

__global__ void __launch_bounds__(256) normsq_kernel(const float* __restrict__ GS, int64_t gs_stride,
                                                     float* __restrict__ part, int64_t part_stride, int slot0,
                                                     int64_t P, int vec) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * K5_CHUNK;
    const float* row = GS + (int64_t)b * gs_stride;
    int64_t end = c0 + K5_CHUNK;
    if (end > P) end = P;
    float sq = 0.f;
    if (vec) {
        for (int64_t j = c0 + 4 * threadIdx.x; j < end; j += 1024) {
            if (j + 4 <= end) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + j);
                sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            } else {
                for (int64_t k = j; k < end; ++k) sq += row[k] * row[k];
            }
        }
    } else {
        for (int64_t j = c0 + threadIdx.x; j < end; j += 256) sq += row[j] * row[j];
    }
    const float tot = block_sum<4>(sq, red);
    if (threadIdx.x == 0) part[(int64_t)b * part_stride + slot0 + blockIdx.x] = tot;
}

// one wave per sample: lane l sums slots l, l+64, ... in order, then a fixed butterfly
__global__ void __launch_bounds__(256) clip_factor_kernel(const float* __restrict__ part, int64_t part_stride,
                                                          int n_slots, float* __restrict__ clip,
                                                          float* __restrict__ norms, int B, float max_grad_norm,
                                                          const int* __restrict__ n_valid_dev) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    if (n_valid_dev && b >= *n_valid_dev) {  // padding row of a fixed-capacity Poisson batch: contributes nothing
        if ((threadIdx.x & 63) == 0) {
            clip[b] = 0.f;
            if (norms) norms[b] = 0.f;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const float* row = part + (int64_t)b * part_stride;
    float s = 0.f;
    for (int i = lane; i < n_slots; i += 64) s += row[i];
    s = wave_sum(s);
    if (lane == 0) {
        const float n = sqrtf(s);
        if (norms) norms[b] = n;
        const float c = max_grad_norm / (n + 1e-6f);
        clip[b] = c > 1.0f ? 1.0f : c;
    }
}

__device__ __forceinline__ f32x4 noise4(const float* __restrict__ noise, int64_t j, uint32_t k0, uint32_t k1,
                                        uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                        float noise_std) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (noise) {
        z = *reinterpret_cast<const f32x4*>(noise + j);
    } else if (noise_std != 0.f) {
        float n[4];
        philox_normal4(offset + (offset_dev ? *offset_dev : 0ull) + (uint64_t)(j >> 2), k0, k1, n);
        z = f32x4{n[0], n[1], n[2], n[3]} * noise_std;
    }
    return z;
}

#ifndef DPK_K7_ROWS
#define DPK_K7_ROWS 8
#endif
constexpr int K7_ROWS = DPK_K7_ROWS;  // independent 16-byte row loads in flight per thread

// MODE 0: write partial sums (split-B); MODE 1: write summed only; MODE 2: summed (optional) + noise + scale
template <int MODE>
__global__ void __launch_bounds__(256) clip_reduce_kernel(const float* __restrict__ GS, int64_t gs_stride,
                                                          const float* __restrict__ clip,
                                                          const float* __restrict__ summed_in,
                                                          float* __restrict__ out0, float* __restrict__ grad,
                                                          const float* __restrict__ noise, uint32_t k0, uint32_t k1,
                                                          uint64_t offset, const uint64_t* __restrict__ offset_dev, float noise_std,
                                                          float inv_scale, int B,
                                                          int64_t P, int b_per_split) {
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= P) return;
    const int b0 = blockIdx.y * b_per_split;
    int b1 = b0 + b_per_split;
    if (b1 > B) b1 = B;
    const float* p = GS + (int64_t)b0 * gs_stride + j;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + K7_ROWS <= b1; b += K7_ROWS) {
        f32x4 v[K7_ROWS];
#pragma unroll
        for (int u = 0; u < K7_ROWS; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (int64_t)u * gs_stride));
#pragma unroll
        for (int u = 0; u < K7_ROWS; ++u) acc += v[u] * clip[b + u];
        p += K7_ROWS * gs_stride;
    }
    for (; b < b1; ++b) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        acc += v * clip[b];
        p += gs_stride;
    }
    if (MODE == 0) {
        *reinterpret_cast<f32x4*>(out0 + (int64_t)blockIdx.y * P + j) = acc;
    } else {
        if (summed_in) acc += *reinterpret_cast<const f32x4*>(summed_in + j);
        if (out0) *reinterpret_cast<f32x4*>(out0 + j) = acc;
        if (MODE == 2) {
            const f32x4 z = noise4(noise, j, k0, k1, offset, offset_dev, noise_std);
            *reinterpret_cast<f32x4*>(grad + j) = (acc + z) * inv_scale;
        }
    }
}

// combine split-B partials in split order (+ summed_in), optionally noise + scale
template <bool FINAL>
__global__ void __launch_bounds__(256) finalize_kernel(const float* __restrict__ partials, int nsplit,
                                                       const float* __restrict__ summed_in,
                                                       float* __restrict__ summed_out, float* __restrict__ grad,
                                                       const float* __restrict__ noise, uint32_t k0, uint32_t k1,
                                                       uint64_t offset, const uint64_t* __restrict__ offset_dev, float noise_std,
                                                       float inv_scale, int64_t P) {
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= P) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) acc += *reinterpret_cast<const f32x4*>(partials + (int64_t)s * P + j);
    if (summed_in) acc += *reinterpret_cast<const f32x4*>(summed_in + j);
    if (summed_out) *reinterpret_cast<f32x4*>(summed_out + j) = acc;
    if (FINAL) {
        const f32x4 z = noise4(noise, j, k0, k1, offset, offset_dev, noise_std);
        *reinterpret_cast<f32x4*>(grad + j) = (acc + z) * inv_scale;
    }
}

// grad = (summed + z) * inv_scale ; scalar tail handled so P need not be a multiple of 4
__global__ void __launch_bounds__(256) noise_scale_kernel(const float* __restrict__ summed,
                                                          const float* __restrict__ noise, uint32_t k0, uint32_t k1,
                                                          uint64_t offset, const uint64_t* __restrict__ offset_dev, float noise_std,
                                                          float inv_scale, float* __restrict__ grad, int64_t P, int vec) {
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= P) return;
    if (vec && j + 4 <= P) {
        const f32x4 s = summed ? *reinterpret_cast<const f32x4*>(summed + j) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 z = noise4(noise, j, k0, k1, offset, offset_dev, noise_std);
        *reinterpret_cast<f32x4*>(grad + j) = (s + z) * inv_scale;
    } else {
        float n[4] = {0.f, 0.f, 0.f, 0.f};
        if (!noise && noise_std != 0.f)
            philox_normal4(offset + (offset_dev ? *offset_dev : 0ull) + (uint64_t)(j >> 2), k0, k1, n);
        for (int k = 0; k < 4 && j + k < P; ++k) {
            const float z = noise ? noise[j + k] : n[k] * noise_std;
            grad[j + k] = ((summed ? summed[j + k] : 0.f) + z) * inv_scale;
        }
    }
}

inline int pick_nsplit(int B, int64_t P) {
#ifdef DPK_LAB_ENV  // lab builds only: sweep the split factor without recompiling
    if (const char* e = getenv("DPK_K7_NSPLIT")) {
        int n = atoi(e);
        if (n > B) n = B;
        return n < 1 ? 1 : n;
    }
#endif
    // Lab sweep (tools/lab/k7_sweep.py, MI355X): one long row stream per block is best once the grid has >= 2 blocks per
    // CU; splitting the batch only pays for a mid-sized P (cfg2: 805 column blocks -> 3 splits, +2.5 %) and hurts a
    // small batch (cfg1, B=64: 27 us unsplit vs 32 us in two halves).  Aim at ~2400 blocks, >= 128 rows per split.
    const int64_t blocks_x = (P / 4 + 255) / 256;
    int64_t n = (2400 + blocks_x / 2) / (blocks_x > 0 ? blocks_x : 1);
    if (n > 32) n = 32;
    const int by_batch = B / 128;
    if (n > by_batch) n = by_batch;
    if (n < 1) n = 1;
    return (int)n;
}

int clip_reduce_impl(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                     const float* noise, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float noise_std,
                     float inv_scale, float* summed_out, float* grad, bool final_, int B, int64_t P, void* workspace,
                     int64_t workspace_bytes, void* stream) {
    if (!GS || !clip) return DPK_EINVAL;
    if (final_ ? !grad : !summed_out) return DPK_EINVAL;
    if (B < 0 || P <= 0) return DPK_EINVAL;
    if ((P & 3) || (gs_stride & 3) || gs_stride < P) return DPK_EALIGN;
    if (!dpk_aligned16(GS) || (summed_in && !dpk_aligned16(summed_in)) || (summed_out && !dpk_aligned16(summed_out)) ||
        (grad && !dpk_aligned16(grad)) || (noise && !dpk_aligned16(noise)))
        return DPK_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t blocks_x = (P / 4 + 255) / 256;
    if (blocks_x > 0x7fffffffLL) return DPK_ERANGE;
    const int nsplit = pick_nsplit(B, P);
    if (nsplit == 1) {
        dim3 grid((unsigned)blocks_x, 1);
        if (final_)
            hipLaunchKernelGGL((clip_reduce_kernel<2>), grid, dim3(256), 0, s, GS, gs_stride, clip, summed_in,
                               summed_out, grad, noise, k0, k1, offset, offset_dev, noise_std, inv_scale, B, P, B > 0 ? B : 1);
        else
            hipLaunchKernelGGL((clip_reduce_kernel<1>), grid, dim3(256), 0, s, GS, gs_stride, clip, summed_in,
                               summed_out, grad, noise, k0, k1, offset, offset_dev, noise_std, inv_scale, B, P, B > 0 ? B : 1);
        return dpk_launch_status();
    }
    const int64_t need = (int64_t)nsplit * P * 4;
    if (!workspace || workspace_bytes < need || !dpk_aligned16(workspace)) return DPK_EINVAL;
    const int bps = (B + nsplit - 1) / nsplit;
    dim3 grid((unsigned)blocks_x, (unsigned)nsplit);
    hipLaunchKernelGGL((clip_reduce_kernel<0>), grid, dim3(256), 0, s, GS, gs_stride, clip, (const float*)nullptr,
                       (float*)workspace, (float*)nullptr, (const float*)nullptr, k0, k1, offset, offset_dev, noise_std,
                       inv_scale, B, P, bps);
    int rc = dpk_launch_status();
    if (rc) return rc;
    if (final_)
        hipLaunchKernelGGL((finalize_kernel<true>), dim3((unsigned)blocks_x), dim3(256), 0, s,
                           (const float*)workspace, nsplit, summed_in, summed_out, grad, noise, k0, k1, offset,
                           offset_dev, noise_std, inv_scale, P);
    else
        hipLaunchKernelGGL((finalize_kernel<false>), dim3((unsigned)blocks_x), dim3(256), 0, s,
                           (const float*)workspace, nsplit, summed_in, summed_out, grad, noise, k0, k1, offset,
                           offset_dev, noise_std, inv_scale, P);
    return dpk_launch_status();
}

}  // namespace

// ---- scratch of dpk_clip_reduce_noise_static ------------------------------------------------------------------
// K7's split-B partial sums [nsplit, P] for callers that may not allocate.  pick_nsplit() never asks for more than
// (2400 + blocks_x / 2) * 1024 floats with blocks_x <= 1600, i.e. 13.1 MB, so 16 MiB covers every (B, P); the size
// check in the entry point guards that bound.  One copy per GPU the library runs on, held until the process ends.
// In host code the array's NAME is only a handle: the device address comes from hipGetSymbolAddress, per device.
static __device__ __align__(256) float scratch[DPK_STATIC_SCRATCH_BYTES / sizeof(float)];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex scratch_mu;
float* scratch_ptr[MAX_DEVICES] = {nullptr};

// Device address of `scratch` on the CURRENT device (looked up once per device, with that device current).  The
// stream the kernels will run on must belong to that device: each GPU has its own copy of the array, and a stream of
// another GPU would be handed this GPU's address.  The first call per device runs hipGetSymbolAddress (a lazy
// code-object lookup): make it outside stream capture; once the address is cached the call is capture-safe
// (tests/test_zz_no_alloc.py captures it into a hipGraph and replays it).
int scratch_for_current_device(float** out, hipStream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return DPK_ERANGE;
    if (stream != nullptr) {  // the null stream is the current device's by definition
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess) return (int)e;
        if ((int)sdev != dev) return DPK_EINVAL;
    }
    std::lock_guard<std::mutex> lock(scratch_mu);
    if (scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (p == nullptr || !dpk_aligned16(p)) return DPK_EALIGN;
        scratch_ptr[dev] = (float*)p;
    }
    *out = scratch_ptr[dev];
    return DPK_OK;
}

}  // namespace

extern "C" int dpk_clip_reduce_noise_static(const float* GS, int64_t gs_stride, const float* clip,
                                            const float* summed_in, const float* noise, uint64_t seed,
                                            uint64_t offset, const uint64_t* offset_dev, float noise_std,
                                            float inv_scale, float* summed_out, float* grad, int B, int64_t P,
                                            void* stream) {
    if (B < 0 || P <= 0) return DPK_EINVAL;
    const int64_t need = dpk_clip_reduce_workspace(B, P);
    if (need > (int64_t)DPK_STATIC_SCRATCH_BYTES) return DPK_ESCRATCH;
    float* ws = nullptr;
    if (need > 0) {  // the partials are fully overwritten before they are read: nothing to clear between calls
        const int rc = scratch_for_current_device(&ws, (hipStream_t)stream);
        if (rc != DPK_OK) return rc;
    }
    return clip_reduce_impl(GS, gs_stride, clip, summed_in, noise, seed, offset, offset_dev, noise_std, inv_scale,
                            summed_out, grad, true, B, P, ws, need, stream);
}

extern "C" int dpk_normsq_slots(int64_t P) {
    if (P <= 0) return DPK_EINVAL;
    const int64_t n = (P + K5_CHUNK - 1) / K5_CHUNK;
    return n > 0x7fffffffLL ? DPK_ERANGE : (int)n;
}

extern "C" int dpk_normsq_partials(const float* GS, int64_t gs_stride, float* normsq_part, int64_t part_stride,
                                   int slot0, int B, int64_t P, void* stream) {
    if (!GS || !normsq_part || B < 0 || P <= 0 || slot0 < 0 || gs_stride < P) return DPK_EINVAL;
    const int64_t nchunk = (P + K5_CHUNK - 1) / K5_CHUNK;
    if (nchunk > 0x7fffffffLL || B > 65535) return DPK_ERANGE;
    if (slot0 + nchunk > part_stride) return DPK_ESLOTS;
    if (B == 0) return DPK_OK;
    const int vec = dpk_aligned16(GS) && (gs_stride % 4 == 0);
    hipLaunchKernelGGL(normsq_kernel, dim3((unsigned)nchunk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, GS,
                       gs_stride, normsq_part, part_stride, slot0, P, vec);
    return dpk_launch_status();
}

extern "C" int dpk_clip_factors(const float* normsq_part, int64_t part_stride, int n_slots, float* clip,
                                float* norms, int B, float max_grad_norm, const int* n_valid_dev, void* stream) {
    if (!normsq_part || !clip || B < 0 || n_slots < 0 || n_slots > part_stride) return DPK_EINVAL;
    if (B == 0) return DPK_OK;
    hipLaunchKernelGGL(clip_factor_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       normsq_part, part_stride, n_slots, clip, norms, B, max_grad_norm, n_valid_dev);
    return dpk_launch_status();
}

extern "C" int64_t dpk_clip_reduce_workspace(int B, int64_t P) {
    if (B < 0 || P <= 0) return 0;
    const int n = pick_nsplit(B, P);
    return n > 1 ? (int64_t)n * P * 4 : 0;
}

extern "C" int dpk_clip_reduce(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                               float* summed_out, int B, int64_t P, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    return clip_reduce_impl(GS, gs_stride, clip, summed_in, nullptr, 0, 0, nullptr, 0.f, 1.f, summed_out, nullptr, false, B,
                            P, workspace, workspace_bytes, stream);
}

extern "C" int dpk_clip_reduce_noise(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                                     const float* noise, uint64_t seed, uint64_t offset,
                                     const uint64_t* offset_dev, float noise_std, float inv_scale,
                                     float* summed_out, float* grad, int B, int64_t P, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    return clip_reduce_impl(GS, gs_stride, clip, summed_in, noise, seed, offset, offset_dev, noise_std, inv_scale, summed_out,
                            grad, true, B, P, workspace, workspace_bytes, stream);
}

extern "C" int dpk_noise_scale(const float* summed, const float* noise, uint64_t seed, uint64_t offset,
                               const uint64_t* offset_dev, float noise_std, float inv_scale, float* grad, int64_t P, void* stream) {
    if (!grad || P <= 0) return DPK_EINVAL;
    const int64_t blocks_x = ((P + 3) / 4 + 255) / 256;
    if (blocks_x > 0x7fffffffLL) return DPK_ERANGE;
    const int vec = dpk_aligned16(grad) && (!summed || dpk_aligned16(summed)) && (!noise || dpk_aligned16(noise));
    hipLaunchKernelGGL(noise_scale_kernel, dim3((unsigned)blocks_x), dim3(256), 0, (hipStream_t)stream, summed,
                       noise, (uint32_t)seed, (uint32_t)(seed >> 32), offset, offset_dev, noise_std, inv_scale, grad, P, vec);
    return dpk_launch_status();
}

extern "C" int dpk_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float std, void* stream) {
    return dpk_noise_scale(nullptr, nullptr, seed, offset, nullptr, std, 1.0f, out, n, stream);
}

extern "C" int dpk_version(void) { return DPK_VERSION; }

extern "C" const char* dpk_error_string(int code) {
    switch (code) {
        case DPK_OK: return "ok";
        case DPK_EINVAL: return "invalid argument (null pointer, non-positive dimension or bad flag)";
        case DPK_ERANGE: return "dimension outside the supported range";
        case DPK_EALIGN: return "pointer or stride not aligned as required (16 bytes / multiple of 4 floats)";
        case DPK_ESLOTS: return "normsq partial row too short for this call's slots";
        case DPK_ESCRATCH: return "input needs more split-B scratch than the library's static buffer holds (DPK_STATIC_SCRATCH_BYTES)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown dpk error";
    }
}
