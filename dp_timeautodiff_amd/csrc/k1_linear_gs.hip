// K1 — per-sample gradient of a Linear layer as a batched fp32 MFMA GEMM with K = T.
//
//   GSw[b,o,i] = sum_t (g_scale*G[b,t,o]) * A[b,t,i]       GSb[b,o] = sum_t g_scale*G[b,t,o]
//
// One workgroup (4 waves) owns a BM x BN tile of ONE sample's [out, in] gradient.  The T x BM
// slice of G and the T x BN slice of A are staged through LDS in chunks of TK time steps with
// 16-byte global loads (scaled by g_scale on the way in, as Opacus scales backprops before its
// einsum); each wave feeds v_mfma_f32_32x32x2_f32 straight from LDS (one ds_read_b32 per operand
// per lane: A-operand rows = o, B-operand cols = i, both unit-stride in LDS -> conflict-free).
// f32-in MFMA is an exact k-ordered fmaf chain, so the result is deterministic.
// The kernel is output-write bound at the reference's T=24 (AI = T/2 flop per byte): the
// accumulator layout (col = lane&31, rows in registers) makes every store instruction write two
// full 128-byte row segments.
// Epilogue also emits sum(GS^2) of the tile (+ the bias slice in column-tile 0) as ONE float per
// (sample, tile) — the norm pass of DPOptimizer.clip_and_accumulate never re-reads the gradient.
//
// Grid: 1-D, XCD-aware: blocks id and id+8 share an XCD (round-robin dispatch), so all tiles of
// sample b are given ids congruent to b mod 8 and consecutive in id/8 — a sample's A/G slices
// are then fetched into ONE XCD's L2 and re-read there by its other tiles.  (Placement is used
// for speed only; results do not depend on it.)
#include "dpk_common.h"

namespace {

constexpr int TK = 32;  // time steps staged per LDS chunk (even)

template <int W>
__device__ __forceinline__ void stage_rows(float* __restrict__ s, const float* __restrict__ g, int t0, int rows,
                                           int T, int c0, int dim, float scale, bool vec) {
    // s[rows][W]  <-  scale * g[t0 + r][c0 + c]   (zero outside [0,T) x [0,dim))
    if (vec) {
        constexpr int W4 = W / 4;
        for (int idx = threadIdx.x; idx < rows * W4; idx += blockDim.x) {
            const int r = idx / W4;
            const int c = (idx - r * W4) * 4;
            const int t = t0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < T && c0 + c < dim) v = *reinterpret_cast<const f32x4*>(g + (int64_t)t * dim + c0 + c);
            v *= scale;
            *reinterpret_cast<f32x4*>(s + r * W + c) = v;
        }
    } else {
        for (int idx = threadIdx.x; idx < rows * W; idx += blockDim.x) {
            const int r = idx / W;
            const int c = idx - r * W;
            const int t = t0 + r;
            float v = 0.f;
            if (t < T && c0 + c < dim) v = g[(int64_t)t * dim + c0 + c];
            s[r * W + c] = v * scale;
        }
    }
}

// WM x WN waves (WM*WN == 4); each wave owns NT 32x32 MFMA tiles side by side along `in`.
template <int WM, int WN, int NT>
__global__ void __launch_bounds__(256) linear_gs_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                        float* __restrict__ GSw, float* __restrict__ GSb,
                                                        float* __restrict__ normsq_part, int64_t part_stride,
                                                        int slot0, int B, int T, int in_f, int out_f,
                                                        int64_t gs_stride, float g_scale, int tiles_m,
                                                        int tiles_n, int vecA, int vecG) {
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * WN * NT;
    __shared__ __attribute__((aligned(16))) float sG[TK * BM];
    __shared__ __attribute__((aligned(16))) float sA[TK * BN];
    __shared__ float red[4];

    // ---- XCD-aware block -> (sample, tile) map -------------------------------------------------
    const int ntile = tiles_m * tiles_n;
    const int64_t id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const int64_t q = id >> 3;
    const int b = (int)(q / ntile) * 8 + xcd;
    const int tile = (int)(q % ntile);
    if (b >= B) return;  // whole block leaves together (uniform)
    const int tm = tile / tiles_n;
    const int tn = tile - tm * tiles_n;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const float* Ab = A + (int64_t)b * T * in_f;
    const float* Gb = G + (int64_t)b * T * out_f;

    // a wave whose whole 32-row strip or whose every 32-col tile is out of range does no MFMA
    const int wrow0 = m0 + wm * 32;
    const int wcol0 = n0 + wn * (32 * NT);
    const bool wave_live = (wrow0 < out_f) && (wcol0 < in_f);

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    float bias_acc = 0.f;
    const bool do_bias = (GSb != nullptr) && (tn == 0) && (threadIdx.x < BM);

    for (int t0 = 0; t0 < T; t0 += TK) {
        int rows = T - t0;
        if (rows > TK) rows = TK;
        const int rows2 = (rows + 1) & ~1;  // MFMA consumes k in pairs; the pad row is zero-filled
        stage_rows<BM>(sG, Gb, t0, rows2, T, m0, out_f, g_scale, vecG != 0);
        stage_rows<BN>(sA, Ab, t0, rows2, T, n0, in_f, 1.0f, vecA != 0);
        __syncthreads();
        if (wave_live) {
            const float* pg = sG + lh * BM + wm * 32 + l31;
            const float* pa = sA + lh * BN + wn * (32 * NT) + l31;
            for (int kk = 0; kk < rows2; kk += 2) {
                const float a = pg[kk * BM];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float bv = pa[kk * BN + j * 32];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
                }
            }
        }
        if (do_bias) {
            for (int r = 0; r < rows; ++r) bias_acc += sG[r * BM + threadIdx.x];
        }
        __syncthreads();
    }

    // ---- epilogue: store the tile, accumulate its squared norm ----------------------------------
    float sq = 0.f;
    if (wave_live) {
        float* out = GSw + (int64_t)b * gs_stride;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wcol0 + j * 32 + l31;
            if (col < in_f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < out_f) {
                        const float v = acc[j][r];
                        out[(int64_t)row * in_f + col] = v;
                        sq += v * v;
                    }
                }
            }
        }
    }
    if (do_bias && m0 + (int)threadIdx.x < out_f) {
        GSb[(int64_t)b * gs_stride + m0 + threadIdx.x] = bias_acc;
        sq += bias_acc * bias_acc;
    }
    const float tot = block_sum<4>(sq, red);
    if (threadIdx.x == 0) normsq_part[(int64_t)b * part_stride + slot0 + tile] = tot;
}

struct Plan {
    int wm, wn, nt;  // waves along out, waves along in, MFMA tiles per wave along in
    int bm, bn;
};

// Tile-shape choice: keep the block tile as large as the layer allows (less operand re-staging
// per output byte) but never wider than the layer (idle waves).
inline Plan pick_plan(int in_f, int out_f) {
    Plan p;
    if (in_f <= 32) {
        p = {4, 1, 1, 128, 32};
    } else if (out_f <= 32) {
        p = (in_f <= 128) ? Plan{1, 4, 1, 32, 128} : Plan{1, 4, 2, 32, 256};
    } else if (in_f <= 64) {
        p = {2, 2, 1, 64, 64};
    } else {
        p = {2, 2, 2, 64, 128};
    }
    return p;
}

}  // namespace

extern "C" int dpk_linear_slots(int in_features, int out_features) {
    if (in_features <= 0 || out_features <= 0) return DPK_EINVAL;
    const Plan p = pick_plan(in_features, out_features);
    const int tiles_m = (out_features + p.bm - 1) / p.bm;
    const int tiles_n = (in_features + p.bn - 1) / p.bn;
    return tiles_m * tiles_n;
}

extern "C" int dpk_linear_grad_sample(const float* A, const float* G, float* GSw, float* GSb, float* normsq_part,
                                      int64_t part_stride, int slot0, int B, int T, int in_f, int out_f,
                                      int64_t gs_stride, float g_scale, void* stream) {
    if (!A || !G || !GSw || !normsq_part) return DPK_EINVAL;
    if (B < 0 || T <= 0 || in_f <= 0 || out_f <= 0 || slot0 < 0) return DPK_EINVAL;
    if (gs_stride < (int64_t)in_f * out_f) return DPK_EINVAL;
    if (B == 0) return DPK_OK;
    const Plan p = pick_plan(in_f, out_f);
    const int tiles_m = (out_f + p.bm - 1) / p.bm;
    const int tiles_n = (in_f + p.bn - 1) / p.bn;
    const int64_t ntile = (int64_t)tiles_m * tiles_n;
    if (slot0 + ntile > part_stride) return DPK_ESLOTS;
    const int64_t nblk = ((int64_t)(B + 7) / 8) * 8 * ntile;
    if (nblk > 0x7fffffffLL) return DPK_ERANGE;
    const int vecA = (in_f % 4 == 0) && dpk_aligned16(A);
    const int vecG = (out_f % 4 == 0) && dpk_aligned16(G);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nblk), block(256);
#define DPK_K1_LAUNCH(WM, WN, NT)                                                                           \
    hipLaunchKernelGGL((linear_gs_kernel<WM, WN, NT>), grid, block, 0, s, A, G, GSw, GSb, normsq_part,       \
                       part_stride, slot0, B, T, in_f, out_f, gs_stride, g_scale, tiles_m, tiles_n, vecA, vecG)
    if (p.wm == 4) {
        DPK_K1_LAUNCH(4, 1, 1);
    } else if (p.wm == 1 && p.nt == 1) {
        DPK_K1_LAUNCH(1, 4, 1);
    } else if (p.wm == 1) {
        DPK_K1_LAUNCH(1, 4, 2);
    } else if (p.nt == 1) {
        DPK_K1_LAUNCH(2, 2, 1);
    } else {
        DPK_K1_LAUNCH(2, 2, 2);
    }
#undef DPK_K1_LAUNCH
    return dpk_launch_status();
}
