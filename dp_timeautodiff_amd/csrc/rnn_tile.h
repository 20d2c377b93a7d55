// Shared device code of the fused recurrences (K8 GRU, K10 LSTM): batch-on-lane MFMA tiles with weights streamed
// from L2 in fragment order, the LDS-only barrier, the gate functions and the one-launch weight packer.
// gfx950 only.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "dpk_common.h"

namespace {

constexpr int GBT = 16;  // batch rows per workgroup

// v_exp_f32 / v_rcp_f32 (about 1 ulp each): the gates are bounded, so absolute error ~1e-7 is the fp32 noise floor
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }

__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for the acks of the
// global stores issued just before it (~1 us each time step); nothing here communicates through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int GNW = 8;  // waves per workgroup (2 per SIMD: one wave's weight loads hide behind the other's MFMAs)
constexpr int GMAX = DPK_GRU_MAX_G;  // GRUs per launch: pointer tables travel by value in the kernel arguments
struct GPtrs { const float* p[GMAX]; };
struct GPtrs3 { const float* p[3 * GMAX]; };

// One launch packs every weight matrix of a call.  Job y: dst[((ot*KG + kg)*64 + l)*4 + m] = W(out, k) with
// out = ot*16 + (l&15), k = kg*16 + m*4 + (l>>4), zero outside H x K, where
//   W(out,k) = !transposed ? src[out*ld + k] : (k < H ? src[k*ld + coff + out] : src2[(k-H)*ld + coff + out])
struct PackJobs {
    const float* src[4 * GMAX];
    const float* src2[4 * GMAX];
    float* dst[4 * GMAX];
    int coff[4 * GMAX];
    int K[4 * GMAX];
    int transposed[4 * GMAX];
};

__global__ void __launch_bounds__(256) gru_pack_kernel(PackJobs jobs, int ld, int H) {
    const int y = blockIdx.y;
    const float* __restrict__ src = jobs.src[y];
    const float* __restrict__ src2 = jobs.src2[y];
    float* __restrict__ dst = jobs.dst[y];
    const int coff = jobs.coff[y], K = jobs.K[y], transposed = jobs.transposed[y];
    const int OT = (H + 15) / 16, KG = (K + 15) / 16;
    const int64_t n = (int64_t)OT * KG * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int m = (int)(idx & 3);
        const int l = (int)((idx >> 2) & 63);
        const int64_t tk = idx >> 8;
        const int kg = (int)(tk % KG);
        const int ot = (int)(tk / KG);
        const int o = ot * 16 + (l & 15);
        const int k = kg * 16 + m * 4 + (l >> 4);
        float v = 0.f;
        if (o < H && k < K) {
            if (!transposed) v = src[(int64_t)o * ld + k];
            else v = k < H ? src[(int64_t)k * ld + coff + o] : src2[(int64_t)(k - H) * ld + coff + o];
        }
        dst[idx] = v;
    }
}

// acc += W-tile x LDS operand, K in groups of 16 (4 MFMAs per 16-byte weight load).  A ring of GPF weight groups is
// kept in flight (L2 latency ~ 3 groups of MFMA issue); the main loop is branch-free so that every ring slot stays in
// fixed registers: a slot is refilled right after the MFMAs that consumed it have issued.
// lab switch: -DDPK_LAB_W_L1 makes every weight load hit the same two K groups (L1-resident), which isolates the cost
// of streaming the weights from L2 (results are then wrong, timing only)
#ifdef DPK_LAB_W_L1
#define DPK_WIDX(x) ((x) & 1)
#else
#define DPK_WIDX(x) (x)
#endif

#define DPK_MFMA4(ACC, W, B0, B1, B2, B3)                                   \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((W).x, B0, ACC, 0, 0, 0);    \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((W).y, B1, ACC, 0, 0, 0);    \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((W).z, B2, ACC, 0, 0, 0);    \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((W).w, B3, ACC, 0, 0, 0)

struct BFrag {  // B operand of one K group: 4 MFMAs x one float per lane
    float b0, b1, b2, b3;
};
__device__ __forceinline__ BFrag ld_bfrag(const float* p) { return BFrag{p[0], p[4 * GBT], p[8 * GBT], p[12 * GBT]}; }

// The steady-state block is scheduled by hand (sched_barrier pins the order): LDS operand of group k+1 is read before
// the MFMAs of group k issue, and slot s is refilled immediately after its MFMAs, so GPF-1 groups of weight loads
// stay in flight.  Left alone, the compiler sinks all refills to the end of the block and exposes the L2 latency.
template <int GPF>
__device__ __forceinline__ void mfma_tile(f32x4& acc, const f32x4* __restrict__ wp, const float* __restrict__ sB, int KG,
                                          int lane) {
    const int q = lane >> 4, j = lane & 15;
    const float* p = sB + q * GBT + j;
    const f32x4* pw = wp + lane;
    f32x4 w[GPF];
#pragma unroll
    for (int s = 0; s < GPF; ++s) w[s] = pw[(s < KG ? s : KG - 1) * 64];
    BFrag bc = ld_bfrag(p);
    int kg = 0;
    for (; kg + 2 * GPF <= KG; kg += GPF) {  // every refill of these blocks is in range
#pragma unroll
        for (int s = 0; s < GPF; ++s) {
            p += 16 * GBT;
            const BFrag bn = ld_bfrag(p);  // kg + s + 1 < KG here
            __builtin_amdgcn_sched_barrier(0);
            DPK_MFMA4(acc, w[s], bc.b0, bc.b1, bc.b2, bc.b3);
            w[s] = pw[DPK_WIDX(kg + s + GPF) * 64];
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
        }
    }
    const int rem = KG - kg;  // < 2*GPF groups left: first block refills only what exists, second block none
#pragma unroll
    for (int s = 0; s < GPF; ++s) {
        if (s < rem) {
            const float* pn = p + (s + 1 < rem ? 16 * GBT : 0);
            const BFrag bn = ld_bfrag(pn);
            DPK_MFMA4(acc, w[s], bc.b0, bc.b1, bc.b2, bc.b3);
            if (s + GPF < rem) w[s] = pw[(kg + s + GPF) * 64];
            p = pn;
            bc = bn;
        }
    }
#pragma unroll
    for (int s = 0; s < GPF; ++s) {
        if (s + GPF < rem) {
            const float* pn = p + (s + GPF + 1 < rem ? 16 * GBT : 0);
            const BFrag bn = ld_bfrag(pn);
            DPK_MFMA4(acc, w[s], bc.b0, bc.b1, bc.b2, bc.b3);
            p = pn;
            bc = bn;
        }
    }
}

// two gates sharing the B operand (independent accumulator chains hide the MFMA dependency latency)
template <int GPF>
__device__ __forceinline__ void mfma_tile2(f32x4& a0, f32x4& a1, const f32x4* __restrict__ w0p,
                                           const f32x4* __restrict__ w1p, const float* __restrict__ sB, int KG, int lane) {
    const int q = lane >> 4, j = lane & 15;
    const float* p = sB + q * GBT + j;
    const f32x4* pw0 = w0p + lane;
    const f32x4* pw1 = w1p + lane;
    f32x4 w0[GPF], w1[GPF];
#pragma unroll
    for (int s = 0; s < GPF; ++s) {
        const int o = (s < KG ? s : KG - 1) * 64;
        w0[s] = pw0[o];
        w1[s] = pw1[o];
    }
    BFrag bc = ld_bfrag(p);
    int kg = 0;
    for (; kg + 2 * GPF <= KG; kg += GPF) {
#pragma unroll
        for (int s = 0; s < GPF; ++s) {
            p += 16 * GBT;
            const BFrag bn = ld_bfrag(p);
            __builtin_amdgcn_sched_barrier(0);
            DPK_MFMA4(a0, w0[s], bc.b0, bc.b1, bc.b2, bc.b3);
            w0[s] = pw0[DPK_WIDX(kg + s + GPF) * 64];
            DPK_MFMA4(a1, w1[s], bc.b0, bc.b1, bc.b2, bc.b3);
            w1[s] = pw1[DPK_WIDX(kg + s + GPF) * 64];
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
        }
    }
    const int rem = KG - kg;
#pragma unroll
    for (int s = 0; s < GPF; ++s) {
        if (s < rem) {
            const float* pn = p + (s + 1 < rem ? 16 * GBT : 0);
            const BFrag bn = ld_bfrag(pn);
            DPK_MFMA4(a0, w0[s], bc.b0, bc.b1, bc.b2, bc.b3);
            DPK_MFMA4(a1, w1[s], bc.b0, bc.b1, bc.b2, bc.b3);
            if (s + GPF < rem) {
                w0[s] = pw0[(kg + s + GPF) * 64];
                w1[s] = pw1[(kg + s + GPF) * 64];
            }
            p = pn;
            bc = bn;
        }
    }
#pragma unroll
    for (int s = 0; s < GPF; ++s) {
        if (s + GPF < rem) {
            const float* pn = p + (s + GPF + 1 < rem ? 16 * GBT : 0);
            const BFrag bn = ld_bfrag(pn);
            DPK_MFMA4(a0, w0[s], bc.b0, bc.b1, bc.b2, bc.b3);
            DPK_MFMA4(a1, w1[s], bc.b0, bc.b1, bc.b2, bc.b3);
            p = pn;
            bc = bn;
        }
    }
}

// one output tile, K split over two independent accumulator chains (even / odd K groups), summed at the end: a single
// chain issues a 16x16x4 MFMA every 44 cycles, two every 32 (tools/lab/mfma_rate.hip)
template <int GPF>
__device__ __forceinline__ void mfma_tile_ks(f32x4& acc, const f32x4* __restrict__ wp, const float* __restrict__ sB,
                                             int KG, int lane) {
    const int q = lane >> 4, j = lane & 15;
    const float* p = sB + q * GBT + j;
    const f32x4* pw = wp + lane;
    const int KP = KG >> 1;  // pairs of K groups
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
    if (KP > 0) {
        f32x4 w0[GPF], w1[GPF];
#pragma unroll
        for (int s = 0; s < GPF; ++s) {
            const int o = (s < KP ? s : KP - 1) * 128;
            w0[s] = pw[o];
            w1[s] = pw[o + 64];
        }
        BFrag b0 = ld_bfrag(p), b1 = ld_bfrag(p + 16 * GBT);
        int kp = 0;
        for (; kp + 2 * GPF <= KP; kp += GPF) {
#pragma unroll
            for (int s = 0; s < GPF; ++s) {
                p += 32 * GBT;
                const BFrag n0 = ld_bfrag(p), n1 = ld_bfrag(p + 16 * GBT);
                __builtin_amdgcn_sched_barrier(0);
                DPK_MFMA4(acc, w0[s], b0.b0, b0.b1, b0.b2, b0.b3);
                w0[s] = pw[DPK_WIDX(kp + s + GPF) * 128];
                DPK_MFMA4(a1, w1[s], b1.b0, b1.b1, b1.b2, b1.b3);
                w1[s] = pw[DPK_WIDX(kp + s + GPF) * 128 + 64];
                __builtin_amdgcn_sched_barrier(0);
                b0 = n0;
                b1 = n1;
            }
        }
        const int rem = KP - kp;
#pragma unroll
        for (int s = 0; s < GPF; ++s) {
            if (s < rem) {
                const float* pn = p + (s + 1 < rem ? 32 * GBT : 0);
                const BFrag n0 = ld_bfrag(pn), n1 = ld_bfrag(pn + 16 * GBT);
                DPK_MFMA4(acc, w0[s], b0.b0, b0.b1, b0.b2, b0.b3);
                DPK_MFMA4(a1, w1[s], b1.b0, b1.b1, b1.b2, b1.b3);
                if (s + GPF < rem) {
                    w0[s] = pw[(kp + s + GPF) * 128];
                    w1[s] = pw[(kp + s + GPF) * 128 + 64];
                }
                p = pn;
                b0 = n0;
                b1 = n1;
            }
        }
#pragma unroll
        for (int s = 0; s < GPF; ++s) {
            if (s + GPF < rem) {
                const float* pn = p + (s + GPF + 1 < rem ? 32 * GBT : 0);
                const BFrag n0 = ld_bfrag(pn), n1 = ld_bfrag(pn + 16 * GBT);
                DPK_MFMA4(acc, w0[s], b0.b0, b0.b1, b0.b2, b0.b3);
                DPK_MFMA4(a1, w1[s], b1.b0, b1.b1, b1.b2, b1.b3);
                p = pn;
                b0 = n0;
                b1 = n1;
            }
        }
    }
    if (KG & 1) {  // odd last group
        const f32x4 w = pw[(KG - 1) * 64];
        const BFrag b = ld_bfrag(sB + (KG - 1) * 16 * GBT + q * GBT + j);
        DPK_MFMA4(a1, w, b.b0, b.b1, b.b2, b.b3);
    }
    acc += a1;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// store 4 consecutive rows (features) of column j into a k-major [K][16] LDS operand
__device__ __forceinline__ void lds_col4(float* s, int o, int j, f32x4 v) {
    s[(o + 0) * GBT + j] = v.x;
    s[(o + 1) * GBT + j] = v.y;
    s[(o + 2) * GBT + j] = v.z;
    s[(o + 3) * GBT + j] = v.w;
}


// launch the packer for `njobs` jobs (grid.y); every job packs an [H x K_job] operand
inline int rnn_launch_pack(const PackJobs& jobs, int njobs, int ld, int H, int Kmax, hipStream_t s) {
    const int64_t n = (int64_t)((H + 15) / 16) * ((Kmax + 15) / 16) * 256;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gru_pack_kernel, dim3((unsigned)blocks, (unsigned)njobs), dim3(256), 0, s, jobs, ld, H);
    return dpk_launch_status();
}

}  // namespace
