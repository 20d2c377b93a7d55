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
#include "rnn_tile.h"

namespace {

template <int TPW, bool PRE>
__global__ void __launch_bounds__(GNW * 64) gru_fwd_kernel(GPtrs xps, GPtrs h0s, const f32x4* __restrict__ Pz,
                                                           const f32x4* __restrict__ Pr, const f32x4* __restrict__ Pn,
                                                           GPtrs3 biases, float* __restrict__ out,
                                                           float* __restrict__ Z, float* __restrict__ R,
                                                           float* __restrict__ N, float* __restrict__ Azr,
                                                           float* __restrict__ An, int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTH = GNW * 64;
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
    // blockIdx.y = which of the G independent GRUs of this launch: inputs come through per-GRU pointers, everything
    // this kernel writes (and the packed weights) has a leading [G] dimension
    const int gi = blockIdx.y;
    const float* __restrict__ xp = xps.p[gi];
    const float* __restrict__ h0 = h0s.p[gi];
    const float* __restrict__ bz = biases.p[3 * gi], * __restrict__ br = biases.p[3 * gi + 1],
                             * __restrict__ bn = biases.p[3 * gi + 2];
    {
        const int64_t g = gi, gx = g * B * T * H, gp = g * NT * KG * 64;
        out += gx; Z += gx; R += gx; N += gx; Azr += 2 * gx; An += 2 * gx;
        Pz += gp; Pr += gp; Pn += gp;
    }
    const int nx4 = GBT * H4;  // float4 elements of one x_t tile; <= TPW * NTH by construction of TPW

    for (int i = threadIdx.x; i < 2 * K16 * GBT; i += NTH) smem[i] = 0.f;
    __syncthreads();

    f32x4 hreg[TPW], zreg[TPW], xpre[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int o = (wave + GNW * i) * 16 + q * 4;
        hreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave + GNW * i < NT && o < H && bvalid && h0) hreg[i] = ld4(h0 + (int64_t)b * H + o);
        if (wave + GNW * i < NT && o < H) lds_col4(bufA, H + o, j, hreg[i]);
    }
    auto load_x = [&](int t) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int jj = idx / H4, k4 = idx - jj * H4;
            xpre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < nx4 && b0 + jj < B) xpre[i] = ld4(xp + ((int64_t)(b0 + jj) * T + t) * H + k4 * 4);
        }
    };
    if (PRE) load_x(0);

    for (int t = 0; t < T; ++t) {
        if (!PRE) load_x(t);  // widest tiles: no registers to spare for the prefetch
        // ---- x_t (prefetched during the previous step) -> both operand buffers, transposed to k-major ------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int jj = idx / H4, k4 = idx - jj * H4;
            if (idx < nx4) {
                lds_col4(bufA, k4 * 4, jj, xpre[i]);
                lds_col4(bufB, k4 * 4, jj, xpre[i]);
                if (b0 + jj < B) {  // x half of the gates' stacked activations [x_t | h_{t-1}] and [x_t | r*h_{t-1}]
                    const int64_t ga = ((int64_t)(b0 + jj) * T + t) * 2 * H + k4 * 4;
                    st4(Azr + ga, xpre[i]);
                    st4(An + ga, xpre[i]);
                }
            }
        }
        lds_barrier();
        if (PRE && t + 1 < T) load_x(t + 1);
        // ---- phase 1: z, r ; r*h -> bufB ------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 az = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
                if (o < H) { az = ld4(bz + o); ar = ld4(br + o); }
                mfma_tile2<(TPW < 4 ? 4 : 2)>(az, ar, Pz + (int64_t)ot * KG * 64, Pr + (int64_t)ot * KG * 64, bufA, KG, lane);
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
                        const int64_t ga = ((int64_t)b * T + t) * 2 * H + H + o;
                        st4(Azr + ga, hreg[i]);
                        st4(An + ga, rh);
                    }
                }
            }
        }
        lds_barrier();
        // ---- phase 2: n, h' ---------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 an = {0.f, 0.f, 0.f, 0.f};
                if (o < H) an = ld4(bn + o);
                mfma_tile<(TPW < 4 ? 4 : 2)>(an, Pn + (int64_t)ot * KG * 64, bufB, KG, lane);
                if (o < H) {
                    f32x4 n;
                    n.x = tanhf_(an.x); n.y = tanhf_(an.y); n.z = tanhf_(an.z); n.w = tanhf_(an.w);
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
        lds_barrier();  // bufB's x rows are rewritten at the top of the next step
    }
}

template <int TPW, bool PRE>
__global__ void __launch_bounds__(GNW * 64) gru_bwd_kernel(GPtrs douts, GPtrs dhTs, const float* __restrict__ Z,
                                                           const float* __restrict__ R, const float* __restrict__ N,
                                                           const float* __restrict__ Azr, const f32x4* __restrict__ Pnx,
                                                           const f32x4* __restrict__ Pnh, const f32x4* __restrict__ Pzrx,
                                                           const f32x4* __restrict__ Pzrh, float* __restrict__ dxp,
                                                           float* __restrict__ Gz, float* __restrict__ Gr,
                                                           float* __restrict__ Gn, float* __restrict__ dh0, int B, int T,
                                                           int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTH = GNW * 64;
    const int KGn = (H + 15) / 16;       // K groups of GEMM1 (K = H: dn_pre)
    const int KGzr = (2 * H + 15) / 16;  // K groups of GEMM2 (K = 2H: dz_pre | dr_pre)
    float* sDn = smem;                   // [KGn*16][16]
    float* sDzr = smem + KGn * 16 * GBT; // [KGzr*16][16]
    const int NT = (H + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int b = blockIdx.x * GBT + j;
    const bool bvalid = b < B;
    const float* __restrict__ dout = douts.p[blockIdx.y];  // null: no gradient reached this GRU's output
    const float* __restrict__ dhT = dhTs.p[blockIdx.y];
    {
        const int64_t g = blockIdx.y, gx = g * B * T * H, gh = g * B * H;
        Z += gx; R += gx; N += gx; Azr += 2 * gx; dxp += gx; Gz += gx; Gr += gx; Gn += gx;
        if (dh0) dh0 += gh;
        Pnx += g * NT * KGn * 64; Pnh += g * NT * KGn * 64;
        Pzrx += g * NT * KGzr * 64; Pzrh += g * NT * KGzr * 64;
    }

    for (int i = threadIdx.x; i < (KGn + KGzr) * 16 * GBT; i += NTH) smem[i] = 0.f;
    __syncthreads();

    f32x4 dhc[TPW], rreg[TPW], hpreg[TPW], accx[TPW];
    f32x4 p_do[TPW], p_z[TPW], p_n[TPW], p_r[TPW], p_hp[TPW];  // next step's operands, in flight during the GEMMs
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_step = [&](int t) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            p_do[i] = p_z[i] = p_n[i] = p_r[i] = p_hp[i] = zero4;
            if (ot < NT && o < H && bvalid) {
                const int64_t g = ((int64_t)b * T + t) * H + o;
                if (dout) p_do[i] = ld4(dout + g);
                p_z[i] = ld4(Z + g);
                p_n[i] = ld4(N + g);
                p_r[i] = ld4(R + g);
                p_hp[i] = ld4(Azr + ((int64_t)b * T + t) * 2 * H + H + o);  // h_{t-1} as the forward stacked it
            }
        }
    };
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int o = (wave + GNW * i) * 16 + q * 4;
        dhc[i] = zero4;
        if (wave + GNW * i < NT && o < H && bvalid && dhT) dhc[i] = ld4(dhT + (int64_t)b * H + o);
    }
    if (PRE) load_step(T - 1);

    for (int t = T - 1; t >= 0; --t) {
        if (!PRE) load_step(t);
        // ---- elementwise: dn_pre, dz_pre, carry dh*z --------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT && o < H) {
                const f32x4 dh = p_do[i] + dhc[i];
                const f32x4 z = p_z[i], n = p_n[i], hp = p_hp[i];
                rreg[i] = p_r[i];
                hpreg[i] = hp;
                const f32x4 dnp = dh * (1.0f - z) * (1.0f - n * n);
                const f32x4 dzp = dh * (hp - n) * z * (1.0f - z);
                dhc[i] = dh * z;
                if (bvalid) {
                    const int64_t g = ((int64_t)b * T + t) * H + o;
                    st4(Gn + g, dnp);
                    st4(Gz + g, dzp);
                }
                lds_col4(sDn, o, j, dnp);   // rows of batch columns beyond B are zeros (operands were zero)
                lds_col4(sDzr, o, j, dzp);
            }
        }
        lds_barrier();
        if (PRE && t > 0) load_step(t - 1);
        // ---- GEMM1: d[x | r*h] = Wn^T dn_pre ; dr_pre -> sDzr --------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            accx[i] = zero4;
            if (ot < NT) {
                f32x4 ah = zero4;
                mfma_tile2<(TPW < 4 ? 4 : 2)>(accx[i], ah, Pnx + (int64_t)ot * KGn * 64, Pnh + (int64_t)ot * KGn * 64, sDn, KGn, lane);
                if (o < H) {
                    const f32x4 r = rreg[i];
                    const f32x4 drp = ah * hpreg[i] * r * (1.0f - r);
                    dhc[i] += ah * r;
                    lds_col4(sDzr, H + o, j, drp);
                    if (bvalid) st4(Gr + ((int64_t)b * T + t) * H + o, drp);
                }
            }
        }
        lds_barrier();
        // ---- GEMM2: d[x | h] += [Wz;Wr]^T [dz_pre | dr_pre] ------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT) {
                f32x4 ah = zero4;
                mfma_tile2<(TPW < 4 ? 4 : 2)>(accx[i], ah, Pzrx + (int64_t)ot * KGzr * 64, Pzrh + (int64_t)ot * KGzr * 64, sDzr, KGzr, lane);
                if (o < H) {
                    dhc[i] += ah;
                    if (bvalid) st4(dxp + ((int64_t)b * T + t) * H + o, accx[i]);
                }
            }
        }
        lds_barrier();  // sDn / sDzr are rewritten by the next (earlier) time step
    }
    if (dh0) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int o = (wave + GNW * i) * 16 + q * 4;
            if (wave + GNW * i < NT && o < H && bvalid) st4(dh0 + (int64_t)b * H + o, dhc[i]);
        }
    }
}

inline int gru_tpw(int H) {
    const int nt = (H + 15) / 16;
    const int per = (nt + GNW - 1) / GNW;  // tiles per wave; also bounds the x_t prefetch (16*H/4 <= per*512 float4)
    return per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 0;
}

}  // namespace

extern "C" int64_t dpk_gru_pack_floats(int O, int K) {
    if (O <= 0 || K <= 0) return 0;
    return (int64_t)((O + 15) / 16) * ((K + 15) / 16) * 256;
}

static int gru_check_shape(int B, int T, int H, int G) {
    if (B <= 0 || T <= 0 || H <= 0 || G <= 0) return DPK_EINVAL;
    if (G > GMAX) return DPK_ERANGE;
    if (H % 8) return DPK_ERANGE;  // 2H must be a whole number of 16-wide K groups; rows are stored 4 at a time
    if (!gru_tpw(H)) return DPK_ERANGE;
    return DPK_OK;
}

extern "C" int dpk_gru_pack_forward(const float* const* W, int ld, float* Pz, float* Pr, float* Pn, int H, int G,
                                    void* stream) {
    if (!W || !Pz || !Pr || !Pn || H <= 0 || G <= 0 || ld < 2 * H) return DPK_EINVAL;
    if (G > GMAX) return DPK_ERANGE;
    if (!dpk_aligned16(Pz) || !dpk_aligned16(Pr) || !dpk_aligned16(Pn)) return DPK_EALIGN;
    const int64_t n = dpk_gru_pack_floats(H, 2 * H);
    PackJobs jobs{};
    float* dst[3] = {Pz, Pr, Pn};
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < 3; ++k) {
            const int y = g * 3 + k;
            if (!W[y]) return DPK_EINVAL;
            jobs.src[y] = W[y];
            jobs.src2[y] = W[y];
            jobs.dst[y] = dst[k] + g * n;
            jobs.coff[y] = 0;
            jobs.K[y] = 2 * H;
            jobs.transposed[y] = 0;
        }
    return rnn_launch_pack(jobs, 3 * G, ld, H, 2 * H, (hipStream_t)stream);
}

extern "C" int dpk_gru_pack_backward(const float* const* W, int ld, float* Pnx, float* Pnh, float* Pzrx, float* Pzrh,
                                     int H, int G, void* stream) {
    if (!W || !Pnx || !Pnh || !Pzrx || !Pzrh || H <= 0 || G <= 0 || ld < 2 * H) return DPK_EINVAL;
    if (G > GMAX) return DPK_ERANGE;
    if (!dpk_aligned16(Pnx) || !dpk_aligned16(Pnh) || !dpk_aligned16(Pzrx) || !dpk_aligned16(Pzrh)) return DPK_EALIGN;
    const int64_t n1 = dpk_gru_pack_floats(H, H), n2 = dpk_gru_pack_floats(H, 2 * H);
    PackJobs jobs{};
    for (int g = 0; g < G; ++g) {
        const float *wz = W[3 * g], *wr = W[3 * g + 1], *wn = W[3 * g + 2];
        if (!wz || !wr || !wn) return DPK_EINVAL;
        const float* src[4] = {wn, wn, wz, wz};
        const float* src2[4] = {wn, wn, wr, wr};
        float* dst[4] = {Pnx + g * n1, Pnh + g * n1, Pzrx + g * n2, Pzrh + g * n2};
        const int coff[4] = {0, H, 0, H}, K[4] = {H, H, 2 * H, 2 * H};
        for (int k = 0; k < 4; ++k) {
            const int y = g * 4 + k;
            jobs.src[y] = src[k];
            jobs.src2[y] = src2[k];
            jobs.dst[y] = dst[k];
            jobs.coff[y] = coff[k];
            jobs.K[y] = K[k];
            jobs.transposed[y] = 1;
        }
    }
    return rnn_launch_pack(jobs, 4 * G, ld, H, 2 * H, (hipStream_t)stream);
}

extern "C" int dpk_gru_forward(const float* const* xp, const float* const* h0, const float* Pz, const float* Pr,
                               const float* Pn, const float* const* bias, float* out, float* Z, float* R, float* N,
                               float* Azr, float* An, int B, int T, int H, int G, void* stream) {
    if (!xp || !Pz || !Pr || !Pn || !bias || !out || !Z || !R || !N || !Azr || !An) return DPK_EINVAL;
    if (int rc = gru_check_shape(B, T, H, G)) return rc;
    const void* ptrs[] = {Pz, Pr, Pn, out, Z, R, N, Azr, An};
    for (const void* p : ptrs)
        if (!dpk_aligned16(p)) return DPK_EALIGN;
    GPtrs xps{}, h0s{};
    GPtrs3 bs{};
    for (int g = 0; g < G; ++g) {
        if (!xp[g] || !bias[3 * g] || !bias[3 * g + 1] || !bias[3 * g + 2]) return DPK_EINVAL;
        xps.p[g] = xp[g];
        h0s.p[g] = h0 ? h0[g] : nullptr;
        for (int k = 0; k < 3; ++k) bs.p[3 * g + k] = bias[3 * g + k];
        if (!dpk_aligned16(xps.p[g]) || !dpk_aligned16(h0s.p[g]) || !dpk_aligned16(bs.p[3 * g]) ||
            !dpk_aligned16(bs.p[3 * g + 1]) || !dpk_aligned16(bs.p[3 * g + 2]))
            return DPK_EALIGN;
    }
    const int tpw = gru_tpw(H);
    const size_t lds = (size_t)2 * ((2 * H + 15) / 16) * 16 * GBT * sizeof(float);
    if (lds > 150 * 1024) return DPK_ERANGE;
    dim3 grid((unsigned)((B + GBT - 1) / GBT), (unsigned)G);
    hipStream_t s = (hipStream_t)stream;
#define DPK_GRU_F(TPW)                                                                                              \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)gru_fwd_kernel<TPW, (TPW < 4)>,                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e_ != hipSuccess) return (int)e_;                                                                   \
        }                                                                                                           \
        hipLaunchKernelGGL((gru_fwd_kernel<TPW, (TPW < 4)>), grid, dim3(GNW * 64), lds, s, xps, h0s,                \
                           (const f32x4*)Pz, (const f32x4*)Pr, (const f32x4*)Pn, bs, out, Z, R, N, Azr, An, B, T,  \
                           H);                                                                                      \
    } while (0)
    if (tpw == 1) DPK_GRU_F(1);
    else if (tpw == 2) DPK_GRU_F(2);
    else DPK_GRU_F(4);
#undef DPK_GRU_F
    return dpk_launch_status();
}

extern "C" int dpk_gru_backward(const float* const* dout, const float* const* dhT, const float* Z, const float* R,
                                const float* N, const float* Azr, const float* Pnx, const float* Pnh,
                                const float* Pzrx, const float* Pzrh, float* dxp, float* Gz, float* Gr, float* Gn,
                                float* dh0, int B, int T, int H, int G, void* stream) {
    if (!dout || !Z || !R || !N || !Azr || !Pnx || !Pnh || !Pzrx || !Pzrh || !dxp || !Gz || !Gr || !Gn)
        return DPK_EINVAL;
    if (int rc = gru_check_shape(B, T, H, G)) return rc;
    const void* ptrs[] = {Z, R, N, Azr, Pnx, Pnh, Pzrx, Pzrh, dxp, Gz, Gr, Gn, dh0};
    for (const void* p : ptrs)
        if (p && !dpk_aligned16(p)) return DPK_EALIGN;
    GPtrs ds{}, hs{};
    for (int g = 0; g < G; ++g) {
        ds.p[g] = dout[g];
        hs.p[g] = dhT ? dhT[g] : nullptr;
        if (!dpk_aligned16(ds.p[g]) || !dpk_aligned16(hs.p[g])) return DPK_EALIGN;
    }
    const int tpw = gru_tpw(H);
    const size_t lds = (size_t)(((H + 15) / 16) + ((2 * H + 15) / 16)) * 16 * GBT * sizeof(float);
    if (lds > 150 * 1024) return DPK_ERANGE;
    dim3 grid((unsigned)((B + GBT - 1) / GBT), (unsigned)G);
    hipStream_t s = (hipStream_t)stream;
#define DPK_GRU_B(TPW)                                                                                              \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)gru_bwd_kernel<TPW, (TPW < 4)>,                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e_ != hipSuccess) return (int)e_;                                                                   \
        }                                                                                                           \
        hipLaunchKernelGGL((gru_bwd_kernel<TPW, (TPW < 4)>), grid, dim3(GNW * 64), lds, s, ds, hs, Z, R, N, Azr,    \
                           (const f32x4*)Pnx, (const f32x4*)Pnh, (const f32x4*)Pzrx, (const f32x4*)Pzrh, dxp, Gz,  \
                           Gr, Gn, dh0, B, T, H);                                                                   \
    } while (0)
    if (tpw == 1) DPK_GRU_B(1);
    else if (tpw == 2) DPK_GRU_B(2);
    else DPK_GRU_B(4);
#undef DPK_GRU_B
    return dpk_launch_status();
}
