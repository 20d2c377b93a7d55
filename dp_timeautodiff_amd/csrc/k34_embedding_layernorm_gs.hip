// K3 (nn.Embedding) and K4 (nn.LayerNorm) per-sample gradient kernels.  Both are small,
// HBM/L2-streaming kernels: one workgroup per sample, deterministic summation order (increasing t),
// squared-norm partial emitted in the epilogue (1 slot per sample).
#include "dpk_common.h"

namespace {

constexpr int EMB_LDS_FLOATS = 12288;  // 48 KiB accumulator: covers the reference's V<=48, E=128

// GS[b,v,e] = sum_{t: idx[b,t]==v} g_scale*G[b,t,e].  Thread owns column(s) e -> no conflicts, t-ordered.
__global__ void __launch_bounds__(256) embedding_gs_kernel(const int64_t* __restrict__ idx,
                                                           const float* __restrict__ G, float* __restrict__ GS,
                                                           float* __restrict__ normsq_part, int64_t part_stride,
                                                           int slot0, int T, int V, int E, int64_t gs_stride,
                                                           float g_scale) {
    __shared__ float acc[EMB_LDS_FLOATS];
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int VE = V * E;
    for (int i = threadIdx.x; i < VE; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int64_t* ib = idx + (int64_t)b * T;
    const float* gb = G + (int64_t)b * T * E;
    for (int e = threadIdx.x; e < E; e += 256) {
        for (int t = 0; t < T; ++t) {
            const int64_t v = ib[t];
            if (v >= 0 && v < V) acc[(int)v * E + e] += gb[(int64_t)t * E + e] * g_scale;
        }
    }
    __syncthreads();
    float sq = 0.f;
    float* out = GS + (int64_t)b * gs_stride;
    for (int i = threadIdx.x; i < VE; i += 256) {
        const float v = acc[i];
        out[i] = v;
        sq += v * v;
    }
    const float tot = block_sum<4>(sq, red);
    if (threadIdx.x == 0) normsq_part[(int64_t)b * part_stride + slot0] = tot;
}

constexpr int LN_TCHUNK = 256;

// GSw[b,c] = sum_t xhat[b,t,c] * (g_scale*G[b,t,c]),  GSb[b,c] = sum_t g_scale*G[b,t,c]
// xhat = (x - mean_t) * rsqrt(var_t + eps), biased variance over the last dim (F.layer_norm).
__global__ void __launch_bounds__(256) layernorm_gs_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                           float* __restrict__ GSw, float* __restrict__ GSb,
                                                           float* __restrict__ normsq_part, int64_t part_stride,
                                                           int slot0, int T, int C, float eps, int64_t gs_stride,
                                                           float g_scale) {
    __shared__ float s_mean[LN_TCHUNK];
    __shared__ float s_rstd[LN_TCHUNK];
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float* xb = X + (int64_t)b * T * C;
    const float* gb = G + (int64_t)b * T * C;
    float* ow = GSw + (int64_t)b * gs_stride;
    float* ob = GSb + (int64_t)b * gs_stride;
    const float invC = 1.0f / (float)C;
    float sq = 0.f;

    // columns are processed in passes of 256 when C > 256; statistics are recomputed per T-chunk only
    for (int t0 = 0; t0 < T; t0 += LN_TCHUNK) {
        const int rows = (T - t0 < LN_TCHUNK) ? (T - t0) : LN_TCHUNK;
        // Row statistics, a wave per row.  Up to 512 columns a lane keeps its 8 values of each of FOUR rows in registers:
        // one read per element and 32 independent loads in flight (row after row, each row read twice, a workgroup
        // spent most of its time waiting for one row's load: 326 us at cfg3, 3.98 TB/s).
        if (C <= 512) {
            for (int r0 = wave; r0 < rows; r0 += 16) {
                float xv[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + 4 * i;
                    const float* xr = xb + (int64_t)(t0 + (r < rows ? r : 0)) * C;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int c = lane + 64 * k;
                        xv[i][k] = (r < rows && c < C) ? xr[c] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + 4 * i;
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += xv[i][k];   // (columns >= C hold zeros: the same sum as the loop below)
                    const float mean = wave_sum(s) * invC;
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d = xv[i][k] - mean;
                        if (lane + 64 * k < C) v += d * d;
                    }
                    const float var = wave_sum(v) * invC;
                    if (lane == 0 && r < rows) {
                        s_mean[r] = mean;
                        s_rstd[r] = 1.0f / sqrtf(var + eps);
                    }
                }
            }
        } else {
            for (int r = wave; r < rows; r += 4) {
                const float* xr = xb + (int64_t)(t0 + r) * C;
                float s = 0.f;
                for (int c = lane; c < C; c += 64) s += xr[c];
                const float mean = wave_sum(s) * invC;
                float v = 0.f;
                for (int c = lane; c < C; c += 64) {
                    const float d = xr[c] - mean;
                    v += d * d;
                }
                const float var = wave_sum(v) * invC;
                if (lane == 0) {
                    s_mean[r] = mean;
                    s_rstd[r] = 1.0f / sqrtf(var + eps);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float gw = 0.f, gbias = 0.f;
            int r = 0;
            for (; r + 8 <= rows; r += 8) {   // eight rows' loads in flight; the sums stay in t order
                float gv[8], xw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int64_t o = (int64_t)(t0 + r + i) * C + c;
                    gv[i] = gb[o];
                    xw[i] = xb[o];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float g = gv[i] * g_scale;
                    const float xh = (xw[i] - s_mean[r + i]) * s_rstd[r + i];
                    gw += xh * g;
                    gbias += g;
                }
            }
            for (; r < rows; ++r) {
                const int64_t o = (int64_t)(t0 + r) * C + c;
                const float g = gb[o] * g_scale;
                const float xh = (xb[o] - s_mean[r]) * s_rstd[r];
                gw += xh * g;
                gbias += g;
            }
            if (t0 == 0) {
                ow[c] = gw;
                ob[c] = gbias;
            } else {
                ow[c] += gw;
                ob[c] += gbias;
            }
        }
        __syncthreads();
    }
    // squared norm of the finished rows (re-read: C floats, L2-resident)
    for (int c = threadIdx.x; c < C; c += 256) {
        const float a = ow[c], d = ob[c];
        sq += a * a + d * d;
    }
    const float tot = block_sum<4>(sq, red);
    if (threadIdx.x == 0) normsq_part[(int64_t)b * part_stride + slot0] = tot;
}

}  // namespace

extern "C" int dpk_embedding_slots(int V, int E) { return (V > 0 && E > 0) ? 1 : DPK_EINVAL; }
extern "C" int dpk_layernorm_slots(int C) { return C > 0 ? 1 : DPK_EINVAL; }

extern "C" int dpk_embedding_grad_sample(const int64_t* idx, const float* G, float* GS, float* normsq_part,
                                         int64_t part_stride, int slot0, int B, int T, int V, int E,
                                         int64_t gs_stride, float g_scale, void* stream) {
    if (!idx || !G || !GS || !normsq_part) return DPK_EINVAL;
    if (B < 0 || T <= 0 || V <= 0 || E <= 0 || slot0 < 0) return DPK_EINVAL;
    if ((int64_t)V * E > EMB_LDS_FLOATS) return DPK_ERANGE;
    if (gs_stride < (int64_t)V * E) return DPK_EINVAL;
    if (slot0 + 1 > part_stride) return DPK_ESLOTS;
    if (B == 0) return DPK_OK;
    hipLaunchKernelGGL(embedding_gs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, idx, G, GS, normsq_part,
                       part_stride, slot0, T, V, E, gs_stride, g_scale);
    return dpk_launch_status();
}

extern "C" int dpk_layernorm_grad_sample(const float* X, const float* G, float* GSw, float* GSb,
                                         float* normsq_part, int64_t part_stride, int slot0, int B, int T, int C,
                                         float eps, int64_t gs_stride, float g_scale, void* stream) {
    if (!X || !G || !GSw || !GSb || !normsq_part) return DPK_EINVAL;
    if (B < 0 || T <= 0 || C <= 0 || slot0 < 0) return DPK_EINVAL;
    if (gs_stride < C) return DPK_EINVAL;
    if (slot0 + 1 > part_stride) return DPK_ESLOTS;
    if (B == 0) return DPK_OK;
    hipLaunchKernelGGL(layernorm_gs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, X, G, GSw, GSb,
                       normsq_part, part_stride, slot0, T, C, eps, gs_stride, g_scale);
    return dpk_launch_status();
}
