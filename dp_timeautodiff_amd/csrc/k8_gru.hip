// K8 — fused Linear-cell GRU recurrence (forward and backward), the layer either side of the per-sample-gradient
// path.  The reference's CustomGRU (Model_Code/timeautoencoder.py:55-96) runs a T-step Python loop
// of three nn.Linear gates + elementwise ops (~45 kernels per time step forward+backward); here ONE launch runs all
// T steps.  The recurrence is independent across batch rows, so a workgroup owns 16 batch rows for the whole
// sequence: no inter-workgroup synchronisation at all.
//
//   comb  = [x_t | h]                 z = sigmoid(Wz comb + bz)      r = sigmoid(Wr comb + br)
//   comb2 = [x_t | r*h]               n = tanh(Wn comb2 + bn)        h' = (1-z) n + z h
//
// MFMA orientation: D[out][batch] = sum_k W[out][k] * comb^T[k][batch] with v_mfma_f32_16x16x4_f32:
//   * B operand (comb^T, k-major [K][16] in LDS): lane reads comb^T[k0 + (l>>4)][l&15] — unit stride, conflict-free;
//   * A operand (weights) is streamed straight from L2 into registers (GEMV-like: no reuse across waves, so no LDS
//     round trip) from a buffer pre-packed in fragment order: one 16-byte load per lane feeds 4 MFMAs, 1 KiB
//     contiguous per wave instruction;
//   * D has batch on the lane (col = l&15) and 4 consecutive `out` indices in the 4 accumulator registers, so
//     z, r, n, h of one (out, batch) element live in the same lane/register in every phase and time step: the
//     elementwise gate math and the carried hidden state / carried dh never leave registers, and every global
//     store is a 16-byte store of 4 consecutive features.
// The backward kernel also emits the gates' pre-activation gradients G_z, G_r, G_n [B,T,H] — exactly the stacked
// backprops K1 contracts with the stacked activations [x_t | h_{t-1}] / [x_t | r*h_{t-1}] (K = T) — so the hooked
// T-call protocol (T stash + T pop + torch.stack) disappears for these layers.
#include "dpk_common.h"

namespace {

constexpr int GBT = 16;  // batch rows per workgroup

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// dst[((ot*KG + kg)*64 + l)*4 + m] = W(out = ot*16 + (l&15), k = kg*16 + m*4 + (l>>4)), zero outside O x K.
// W(out,k) = transposed ? src[k*ld + coff + out] : src[out*ld + coff + k]
__global__ void __launch_bounds__(256) gru_pack_kernel(const float* __restrict__ src, int ld, int coff, int O, int K,
                                                       int transposed, float* __restrict__ dst, int OT, int KG) {
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
        if (o < O && k < K) v = transposed ? src[(int64_t)k * ld + coff + o] : src[(int64_t)o * ld + coff + k];
        dst[idx] = v;
    }
}

// acc[0..n) += W-tile(ot) x LDS operand, K in groups of 16 (4 MFMAs per 16-byte weight load)
__device__ __forceinline__ void mfma_tile(f32x4& acc, const f32x4* __restrict__ wp, const float* __restrict__ sB, int KG,
                                          int lane) {
    const int q = lane >> 4, j = lane & 15;
    const float* p = sB + q * GBT + j;
    const f32x4* pw = wp + lane;
    f32x4 w = pw[0];
    for (int kg = 0; kg < KG; ++kg) {
        pw += (kg + 1 < KG) ? 64 : 0;
        const f32x4 wn = *pw;  // next group's weights are in flight while this group's MFMAs issue
        const float b0 = p[0], b1 = p[4 * GBT], b2 = p[8 * GBT], b3 = p[12 * GBT];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b3, acc, 0, 0, 0);
        w = wn;
        p += 16 * GBT;
    }
}

// two gates sharing the B operand (independent accumulator chains hide the MFMA dependency latency)
__device__ __forceinline__ void mfma_tile2(f32x4& a0, f32x4& a1, const f32x4* __restrict__ w0p,
                                           const f32x4* __restrict__ w1p, const float* __restrict__ sB, int KG, int lane) {
    const int q = lane >> 4, j = lane & 15;
    const float* p = sB + q * GBT + j;
    const f32x4* pw0 = w0p + lane;
    const f32x4* pw1 = w1p + lane;
    f32x4 w0 = pw0[0], w1 = pw1[0];
    for (int kg = 0; kg < KG; ++kg) {
        const int step = (kg + 1 < KG) ? 64 : 0;
        pw0 += step;
        pw1 += step;
        const f32x4 w0n = *pw0, w1n = *pw1;
        const float b0 = p[0], b1 = p[4 * GBT], b2 = p[8 * GBT], b3 = p[12 * GBT];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, b0, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, b1, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, b1, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, b2, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, b2, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, b3, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, b3, a1, 0, 0, 0);
        w0 = w0n;
        w1 = w1n;
        p += 16 * GBT;
    }
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

template <int TPW>
__global__ void __launch_bounds__(256) gru_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ h0,
                                                      const f32x4* __restrict__ Pz, const f32x4* __restrict__ Pr,
                                                      const f32x4* __restrict__ Pn, const float* __restrict__ bz,
                                                      const float* __restrict__ br, const float* __restrict__ bn,
                                                      float* __restrict__ out, float* __restrict__ Z,
                                                      float* __restrict__ R, float* __restrict__ N,
                                                      float* __restrict__ RH, int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KG = (2 * H + 15) / 16;
    const int K16 = KG * 16;
    float* bufA = smem;               // [K16][16]  rows: x_t (H) | h_{t-1} (H) | zero pad
    float* bufB = smem + K16 * GBT;   // [K16][16]  rows: x_t (H) | r*h_{t-1} (H) | zero pad
    const int NT = (H + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * GBT;
    const int b = b0 + j;
    const bool bvalid = b < B;
    const int H4 = H / 4;

    for (int i = threadIdx.x; i < 2 * K16 * GBT; i += 256) smem[i] = 0.f;
    __syncthreads();

    f32x4 hreg[TPW], zreg[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int o = (wave + 4 * i) * 16 + q * 4;
        hreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave + 4 * i < NT && o < H && bvalid && h0) hreg[i] = ld4(h0 + (int64_t)b * H + o);
        if (wave + 4 * i < NT && o < H) lds_col4(bufA, H + o, j, hreg[i]);
    }

    for (int t = 0; t < T; ++t) {
        // ---- x_t -> both operand buffers (transposed into k-major) ------------------------------------------
        for (int idx = threadIdx.x; idx < GBT * H4; idx += 256) {
            const int jj = idx / H4, k4 = idx - jj * H4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + jj < B) v = ld4(xp + ((int64_t)(b0 + jj) * T + t) * H + k4 * 4);
            lds_col4(bufA, k4 * 4, jj, v);
            lds_col4(bufB, k4 * 4, jj, v);
        }
        __syncthreads();
        // ---- phase 1: z, r ; r*h -> bufB -----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + 4 * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 az = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
                if (o < H) { az = ld4(bz + o); ar = ld4(br + o); }
                mfma_tile2(az, ar, Pz + (int64_t)ot * KG * 64, Pr + (int64_t)ot * KG * 64, bufA, KG, lane);
                if (o < H) {
                    f32x4 z, r;
                    z.x = sigmoidf_(az.x); z.y = sigmoidf_(az.y); z.z = sigmoidf_(az.z); z.w = sigmoidf_(az.w);
                    r.x = sigmoidf_(ar.x); r.y = sigmoidf_(ar.y); r.z = sigmoidf_(ar.z); r.w = sigmoidf_(ar.w);
                    const f32x4 rh = r * hreg[i];
                    zreg[i] = z;
                    lds_col4(bufB, H + o, j, rh);
                    if (bvalid) {
                        const int64_t g = ((int64_t)b * T + t) * H + o;
                        st4(Z + g, z);
                        st4(R + g, r);
                        st4(RH + g, rh);
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: n, h' -----------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + 4 * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 an = {0.f, 0.f, 0.f, 0.f};
                if (o < H) an = ld4(bn + o);
                mfma_tile(an, Pn + (int64_t)ot * KG * 64, bufB, KG, lane);
                if (o < H) {
                    f32x4 n;
                    n.x = tanhf(an.x); n.y = tanhf(an.y); n.z = tanhf(an.z); n.w = tanhf(an.w);
                    const f32x4 z = zreg[i];
                    const f32x4 hn = (1.0f - z) * n + z * hreg[i];
                    hreg[i] = hn;
                    lds_col4(bufA, H + o, j, hn);  // phase-1 reads of bufA finished at the barrier above
                    if (bvalid) {
                        const int64_t g = ((int64_t)b * T + t) * H + o;
                        st4(N + g, n);
                        st4(out + g, hn);
                    }
                }
            }
        }
        __syncthreads();  // bufB's x rows are rewritten at the top of the next step
    }
}

template <int TPW>
__global__ void __launch_bounds__(256) gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dhT,
                                                      const float* __restrict__ Z, const float* __restrict__ R,
                                                      const float* __restrict__ N, const float* __restrict__ hall,
                                                      const float* __restrict__ h0, const f32x4* __restrict__ Pnx,
                                                      const f32x4* __restrict__ Pnh, const f32x4* __restrict__ Pzrx,
                                                      const f32x4* __restrict__ Pzrh, float* __restrict__ dxp,
                                                      float* __restrict__ Gz, float* __restrict__ Gr,
                                                      float* __restrict__ Gn, float* __restrict__ dh0, int B, int T,
                                                      int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KGn = (H + 15) / 16;       // K groups of GEMM1 (K = H: dn_pre)
    const int KGzr = (2 * H + 15) / 16;  // K groups of GEMM2 (K = 2H: dz_pre | dr_pre)
    float* sDn = smem;                   // [KGn*16][16]
    float* sDzr = smem + KGn * 16 * GBT; // [KGzr*16][16]
    const int NT = (H + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * GBT + j;
    const bool bvalid = b < B;

    for (int i = threadIdx.x; i < (KGn + KGzr) * 16 * GBT; i += 256) smem[i] = 0.f;
    __syncthreads();

    f32x4 dhc[TPW], rreg[TPW], hpreg[TPW], accx[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int o = (wave + 4 * i) * 16 + q * 4;
        dhc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave + 4 * i < NT && o < H && bvalid && dhT) dhc[i] = ld4(dhT + (int64_t)b * H + o);
    }

    for (int t = T - 1; t >= 0; --t) {
        // ---- elementwise: dn_pre, dz_pre, carry dh*z ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + 4 * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT && o < H) {
                f32x4 dnp = {0.f, 0.f, 0.f, 0.f}, dzp = {0.f, 0.f, 0.f, 0.f};
                rreg[i] = dnp;
                hpreg[i] = dnp;
                if (bvalid) {
                    const int64_t g = ((int64_t)b * T + t) * H + o;
                    const f32x4 dh = ld4(dout + g) + dhc[i];
                    const f32x4 z = ld4(Z + g), n = ld4(N + g);
                    rreg[i] = ld4(R + g);
                    f32x4 hp = {0.f, 0.f, 0.f, 0.f};
                    if (t > 0) hp = ld4(hall + g - H);
                    else if (h0) hp = ld4(h0 + (int64_t)b * H + o);
                    hpreg[i] = hp;
                    dnp = dh * (1.0f - z) * (1.0f - n * n);
                    dzp = dh * (hp - n) * z * (1.0f - z);
                    dhc[i] = dh * z;
                    st4(Gn + g, dnp);
                    st4(Gz + g, dzp);
                }
                lds_col4(sDn, o, j, dnp);
                lds_col4(sDzr, o, j, dzp);
            }
        }
        __syncthreads();
        // ---- GEMM1: d[x | r*h] = Wn^T dn_pre ; dr_pre -> sDzr ------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + 4 * i;
            const int o = ot * 16 + q * 4;
            accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ot < NT) {
                f32x4 ah = {0.f, 0.f, 0.f, 0.f};
                mfma_tile2(accx[i], ah, Pnx + (int64_t)ot * KGn * 64, Pnh + (int64_t)ot * KGn * 64, sDn, KGn, lane);
                if (o < H) {
                    const f32x4 r = rreg[i];
                    const f32x4 drp = ah * hpreg[i] * r * (1.0f - r);
                    dhc[i] += ah * r;
                    lds_col4(sDzr, H + o, j, drp);
                    if (bvalid) st4(Gr + ((int64_t)b * T + t) * H + o, drp);
                }
            }
        }
        __syncthreads();
        // ---- GEMM2: d[x | h] += [Wz;Wr]^T [dz_pre | dr_pre] ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + 4 * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 ah = {0.f, 0.f, 0.f, 0.f};
                mfma_tile2(accx[i], ah, Pzrx + (int64_t)ot * KGzr * 64, Pzrh + (int64_t)ot * KGzr * 64, sDzr, KGzr, lane);
                if (o < H) {
                    dhc[i] += ah;
                    if (bvalid) st4(dxp + ((int64_t)b * T + t) * H + o, accx[i]);
                }
            }
        }
        __syncthreads();  // sDn / sDzr are rewritten by the next (earlier) time step
    }
    if (dh0) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int o = (wave + 4 * i) * 16 + q * 4;
            if (wave + 4 * i < NT && o < H && bvalid) st4(dh0 + (int64_t)b * H + o, dhc[i]);
        }
    }
}

inline int gru_tpw(int H) {
    const int nt = (H + 15) / 16;
    const int per = (nt + 3) / 4;
    return per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 0;
}

}  // namespace

extern "C" int64_t dpk_gru_pack_floats(int O, int K) {
    if (O <= 0 || K <= 0) return 0;
    return (int64_t)((O + 15) / 16) * ((K + 15) / 16) * 256;
}

extern "C" int dpk_gru_pack(const float* src, int ld, int coff, int O, int K, int transposed, float* dst,
                            void* stream) {
    if (!src || !dst || O <= 0 || K <= 0 || ld <= 0 || coff < 0) return DPK_EINVAL;
    if (!dpk_aligned16(dst)) return DPK_EALIGN;
    const int OT = (O + 15) / 16, KG = (K + 15) / 16;
    const int64_t n = (int64_t)OT * KG * 256;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gru_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, coff, O, K,
                       transposed, dst, OT, KG);
    return dpk_launch_status();
}

extern "C" int dpk_gru_forward(const float* xp, const float* h0, const float* Pz, const float* Pr, const float* Pn,
                               const float* bz, const float* br, const float* bn, float* out, float* Z, float* R,
                               float* N, float* RH, int B, int T, int H, void* stream) {
    if (!xp || !Pz || !Pr || !Pn || !bz || !br || !bn || !out || !Z || !R || !N || !RH) return DPK_EINVAL;
    if (B <= 0 || T <= 0 || H <= 0) return DPK_EINVAL;
    if (H % 8) return DPK_ERANGE;  // 2H must be a whole number of 16-wide K groups; rows are stored 4 at a time
    const int tpw = gru_tpw(H);
    if (!tpw) return DPK_ERANGE;
    const void* ptrs[] = {xp, h0, Pz, Pr, Pn, bz, br, bn, out, Z, R, N, RH};
    for (const void* p : ptrs)
        if (p && !dpk_aligned16(p)) return DPK_EALIGN;
    const size_t lds = (size_t)2 * ((2 * H + 15) / 16) * 16 * GBT * sizeof(float);
    if (lds > 150 * 1024) return DPK_ERANGE;
    dim3 grid((unsigned)((B + GBT - 1) / GBT));
    hipStream_t s = (hipStream_t)stream;
#define DPK_GRU_F(TPW)                                                                                              \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)gru_fwd_kernel<TPW>,                                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e_ != hipSuccess) return (int)e_;                                                                   \
        }                                                                                                           \
        hipLaunchKernelGGL((gru_fwd_kernel<TPW>), grid, dim3(256), lds, s, xp, h0, (const f32x4*)Pz,                \
                           (const f32x4*)Pr, (const f32x4*)Pn, bz, br, bn, out, Z, R, N, RH, B, T, H);              \
    } while (0)
    if (tpw == 2) DPK_GRU_F(2);
    else if (tpw == 4) DPK_GRU_F(4);
    else DPK_GRU_F(8);
#undef DPK_GRU_F
    return dpk_launch_status();
}

extern "C" int dpk_gru_backward(const float* dout, const float* dhT, const float* Z, const float* R, const float* N,
                                const float* hall, const float* h0, const float* Pnx, const float* Pnh,
                                const float* Pzrx, const float* Pzrh, float* dxp, float* Gz, float* Gr, float* Gn,
                                float* dh0, int B, int T, int H, void* stream) {
    if (!dout || !Z || !R || !N || !hall || !Pnx || !Pnh || !Pzrx || !Pzrh || !dxp || !Gz || !Gr || !Gn)
        return DPK_EINVAL;
    if (B <= 0 || T <= 0 || H <= 0) return DPK_EINVAL;
    if (H % 8) return DPK_ERANGE;
    const int tpw = gru_tpw(H);
    if (!tpw) return DPK_ERANGE;
    const void* ptrs[] = {dout, dhT, Z, R, N, hall, h0, Pnx, Pnh, Pzrx, Pzrh, dxp, Gz, Gr, Gn, dh0};
    for (const void* p : ptrs)
        if (p && !dpk_aligned16(p)) return DPK_EALIGN;
    const size_t lds = (size_t)(((H + 15) / 16) + ((2 * H + 15) / 16)) * 16 * GBT * sizeof(float);
    if (lds > 150 * 1024) return DPK_ERANGE;
    dim3 grid((unsigned)((B + GBT - 1) / GBT));
    hipStream_t s = (hipStream_t)stream;
#define DPK_GRU_B(TPW)                                                                                              \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)gru_bwd_kernel<TPW>,                                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e_ != hipSuccess) return (int)e_;                                                                   \
        }                                                                                                           \
        hipLaunchKernelGGL((gru_bwd_kernel<TPW>), grid, dim3(256), lds, s, dout, dhT, Z, R, N, hall, h0,            \
                           (const f32x4*)Pnx, (const f32x4*)Pnh, (const f32x4*)Pzrx, (const f32x4*)Pzrh, dxp, Gz,  \
                           Gr, Gn, dh0, B, T, H);                                                                   \
    } while (0)
    if (tpw == 2) DPK_GRU_B(2);
    else if (tpw == 4) DPK_GRU_B(4);
    else DPK_GRU_B(8);
#undef DPK_GRU_B
    return dpk_launch_status();
}
