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

// MFMA helper for one phase over TPW tiles of a wave sharing the LDS operand: pairs of tiles run as two accumulator
// chains (mfma_tile2); a single (or odd last) tile splits K over two chains instead (mfma_tile_ks).
template <int TPW, int GPF>
__device__ __forceinline__ void phase_tiles(f32x4 (&acc)[TPW], const f32x4* __restrict__ P, int64_t tile_stride,
                                            const float* __restrict__ sB, int KG, int NT, int wave, int lane) {
    if constexpr (TPW == 1) {
        if (wave < NT) mfma_tile_ks<GPF>(acc[0], P + (int64_t)wave * tile_stride, sB, KG, lane);
    } else {
#pragma unroll
        for (int i = 0; i < TPW; i += 2) {
            const int ot0 = wave + GNW * i, ot1 = wave + GNW * (i + 1);
            if (ot1 < NT)
                mfma_tile2<GPF>(acc[i], acc[i + 1], P + (int64_t)ot0 * tile_stride, P + (int64_t)ot1 * tile_stride, sB, KG,
                                lane);
            else if (ot0 < NT)
                mfma_tile_ks<GPF>(acc[i], P + (int64_t)ot0 * tile_stride, sB, KG, lane);
        }
    }
}

// Register-resident weights (H <= 128: a wave's whole slice of the three gates is 24 x 16 bytes per lane): loaded once
// before the T-step loop, so no weight leaves L2 inside the latency chain and no phase starts by waiting for one.
template <int R>
__device__ __forceinline__ void res_load(f32x4 (&w)[R], const f32x4* __restrict__ P, bool live, int lane) {
#pragma unroll
    for (int kg = 0; kg < R; ++kg) w[kg] = live ? P[kg * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int R>
__device__ __forceinline__ void res_tile2(f32x4& a0, f32x4& a1, const f32x4 (&w0)[R], const f32x4 (&w1)[R],
                                          const float* __restrict__ sB, int lane) {
    const float* p = sB + (lane >> 4) * GBT + (lane & 15);
#pragma unroll
    for (int kg = 0; kg < R; ++kg) {
        const BFrag b = ld_bfrag(p + kg * 16 * GBT);
        DPK_MFMA4(a0, w0[kg], b.b0, b.b1, b.b2, b.b3);
        DPK_MFMA4(a1, w1[kg], b.b0, b.b1, b.b2, b.b3);
    }
}
template <int R>
__device__ __forceinline__ void res_tile_ks(f32x4& acc, const f32x4 (&w)[R], const float* __restrict__ sB, int lane) {
    const float* p = sB + (lane >> 4) * GBT + (lane & 15);
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kg = 0; kg < R; kg += 2) {
        const BFrag b0 = ld_bfrag(p + kg * 16 * GBT), b1 = ld_bfrag(p + (kg + 1) * 16 * GBT);
        DPK_MFMA4(acc, w[kg], b0.b0, b0.b1, b0.b2, b0.b3);
        DPK_MFMA4(a1, w[kg + 1], b1.b0, b1.b1, b1.b2, b1.b3);
    }
    acc += a1;
}

// Forward.  The x half of every gate (W[:, :H] x_t + b, all T steps) is ONE ordinary GEMM outside the kernel
// (xg [B,T,3H], column blocks z | r | n): only the h half (K = H) is sequential, which halves the MFMA work on the
// T-step latency chain and removes the per-step x_t -> LDS transpose and its barrier.  x_t itself only passes through
// (global -> global) into the x halves of the gates' stacked activations Azr / An that K1 contracts.
template <int TPW, bool PRE, int RKG>  // RKG > 0: KG == RKG K groups of weights live in registers (TPW == 1)
__global__ void __launch_bounds__(GNW * 64) gru_fwd_kernel(GPtrs xps, GPtrs xgs, GPtrs h0s, const f32x4* __restrict__ Pz,
                                                           const f32x4* __restrict__ Pr, const f32x4* __restrict__ Pn,
                                                           float* __restrict__ out, float* __restrict__ Z,
                                                           float* __restrict__ R, float* __restrict__ N,
                                                           float* __restrict__ Azr, float* __restrict__ An, int B, int T,
                                                           int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTH = GNW * 64;
    constexpr int GPF = TPW < 4 ? 4 : 2;
    const int KG = (H + 15) / 16;
    const int K16 = KG * 16;
    float* bufA = smem;               // [K16][16]  h_{t-1}, k-major, zero padded
    float* bufB = smem + K16 * GBT;   // [K16][16]  r * h_{t-1}
    const int NT = KG;
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
    const float* __restrict__ xg = xgs.p[gi];
    const float* __restrict__ h0 = h0s.p[gi];
    {
        const int64_t g = gi, gx = g * B * T * H, gp = g * NT * KG * 64;
        out += gx; Z += gx; R += gx; N += gx; Azr += 2 * gx; An += 2 * gx;
        Pz += gp; Pr += gp; Pn += gp;
    }
    const int64_t tile_stride = (int64_t)KG * 64;
    const int nx4 = GBT * H4;  // float4 elements of one x_t tile; <= TPW * NTH by construction of TPW

    for (int i = threadIdx.x; i < 2 * K16 * GBT; i += NTH) smem[i] = 0.f;
    __syncthreads();

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 hreg[TPW], zreg[TPW], gz[TPW], gr[TPW], gn[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int o = (wave + GNW * i) * 16 + q * 4;
        hreg[i] = zero4;
        if (wave + GNW * i < NT && o < H && bvalid && h0) hreg[i] = ld4(h0 + (int64_t)b * H + o);
        if (wave + GNW * i < NT && o < H) lds_col4(bufA, o, j, hreg[i]);
    }
    auto load_g = [&](int t) {  // this lane's x-side pre-activations of step t (its accumulators' initial values)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int o = (wave + GNW * i) * 16 + q * 4;
            gz[i] = gr[i] = gn[i] = zero4;
            if (wave + GNW * i < NT && o < H && bvalid) {
                const float* src = xg + ((int64_t)b * T + t) * 3 * H + o;
                gz[i] = ld4(src);
                gr[i] = ld4(src + H);
                gn[i] = ld4(src + 2 * H);
            }
        }
    };
    // x_t only passes through, into the x half of the gates' stacked activations [x_t | h_{t-1}] and
    // [x_t | r*h_{t-1}]: loaded before the phase-1 MFMAs, stored after them (the latency hides behind the MFMAs)
    f32x4 xcp[TPW];
    auto load_x = [&](int t) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int jj = idx / H4, k4 = idx - jj * H4;
            xcp[i] = zero4;
            if (idx < nx4 && b0 + jj < B) xcp[i] = ld4(xp + ((int64_t)(b0 + jj) * T + t) * H + k4 * 4);
        }
    };
    auto store_x = [&](int t) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int jj = idx / H4, k4 = idx - jj * H4;
            if (idx < nx4 && b0 + jj < B) {
                const int64_t ga = ((int64_t)(b0 + jj) * T + t) * 2 * H + k4 * 4;
                st4(Azr + ga, xcp[i]);
                st4(An + ga, xcp[i]);
            }
        }
    };
    constexpr int RW = RKG > 0 ? RKG : 1;
    f32x4 wz[RW], wr[RW], wn[RW];
    if constexpr (RKG > 0) {
        res_load<RW>(wz, Pz + (int64_t)wave * tile_stride, wave < NT, lane);
        res_load<RW>(wr, Pr + (int64_t)wave * tile_stride, wave < NT, lane);
        res_load<RW>(wn, Pn + (int64_t)wave * tile_stride, wave < NT, lane);
    }
    load_g(0);
    lds_barrier();

    for (int t = 0; t < T; ++t) {
        // ---- phase 1: z, r ; r*h -> bufB ------------------------------------------------------------------
        f32x4 az[TPW], ar[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) { az[i] = gz[i]; ar[i] = gr[i]; }
        load_x(t);
        if constexpr (RKG > 0) {
            res_tile2<RW>(az[0], ar[0], wz, wr, bufA, lane);
        } else {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int ot = wave + GNW * i;
                if (ot < NT) mfma_tile2<GPF>(az[i], ar[i], Pz + (int64_t)ot * tile_stride, Pr + (int64_t)ot * tile_stride,
                                             bufA, KG, lane);
            }
        }
        store_x(t);
        f32x4 an[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            an[i] = gn[i];
            if (ot < NT && o < H) {
                f32x4 z, r;
                z.x = sigmoidf_(az[i].x); z.y = sigmoidf_(az[i].y); z.z = sigmoidf_(az[i].z); z.w = sigmoidf_(az[i].w);
                r.x = sigmoidf_(ar[i].x); r.y = sigmoidf_(ar[i].y); r.z = sigmoidf_(ar[i].z); r.w = sigmoidf_(ar[i].w);
                const f32x4 rh = r * hreg[i];
                zreg[i] = z;
                lds_col4(bufB, o, j, rh);
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
        if (PRE && t + 1 < T) { load_g(t + 1); }  // gn of this step already moved to an[]
        lds_barrier();
        // ---- phase 2: n, h' ---------------------------------------------------------------------------------
        if constexpr (RKG > 0) res_tile_ks<RW>(an[0], wn, bufB, lane);
        else phase_tiles<TPW, GPF>(an, Pn, tile_stride, bufB, KG, NT, wave, lane);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT && o < H) {
                f32x4 n;
                n.x = tanhf_(an[i].x); n.y = tanhf_(an[i].y); n.z = tanhf_(an[i].z); n.w = tanhf_(an[i].w);
                const f32x4 z = zreg[i];
                const f32x4 hn = (1.0f - z) * n + z * hreg[i];
                hreg[i] = hn;
                lds_col4(bufA, o, j, hn);  // phase-1 reads of bufA finished at the barrier above
                if (bvalid) {
                    const int64_t g = ((int64_t)b * T + t) * H + o;
                    st4(N + g, n);
                    st4(out + g, hn);
                }
            }
        }
        if (!PRE && t + 1 < T) load_g(t + 1);
        lds_barrier();  // h_t complete in bufA; bufB free for the next step's r*h
    }
}

// Backward.  Only the h halves of the transposed products are sequential (d(r*h) = Wn[:, H:]^T dn_pre, then
// dh += [Wz[:, H:]; Wr[:, H:]]^T [dz_pre | dr_pre]); the x halves (dL/dx_t) are one GEMM over all T steps outside,
// fed by the G_z | G_r | G_n rows (G3 [B,T,3H]) this kernel writes anyway for K1 (each gate a column block, ld_g = 3H).
template <int TPW, bool PRE, int RKG>
__global__ void __launch_bounds__(GNW * 64) gru_bwd_kernel(GPtrs douts, GPtrs dhTs, const float* __restrict__ Z,
                                                           const float* __restrict__ R, const float* __restrict__ N,
                                                           const float* __restrict__ Azr, const f32x4* __restrict__ Pnh,
                                                           const f32x4* __restrict__ Pzrh, float* __restrict__ G3,
                                                           float* __restrict__ dh0, int B, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTH = GNW * 64;
    constexpr int GPF = TPW < 4 ? 4 : 2;
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
        Z += gx; R += gx; N += gx; Azr += 2 * gx; G3 += 3 * gx;
        if (dh0) dh0 += gh;
        Pnh += g * NT * KGn * 64;
        Pzrh += g * NT * KGzr * 64;
    }

    for (int i = threadIdx.x; i < (KGn + KGzr) * 16 * GBT; i += NTH) smem[i] = 0.f;
    __syncthreads();

    f32x4 dhc[TPW], rreg[TPW], hpreg[TPW];
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
    constexpr int RW = RKG > 0 ? RKG : 1;
    f32x4 wnh[RW], wzrh[2 * RW];
    if constexpr (RKG > 0) {
        res_load<RW>(wnh, Pnh + (int64_t)wave * KGn * 64, wave < NT, lane);
        res_load<2 * RW>(wzrh, Pzrh + (int64_t)wave * KGzr * 64, wave < NT, lane);
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
                if (bvalid) {  // G3 row = [dz_pre | dr_pre | dn_pre]
                    const int64_t g = ((int64_t)b * T + t) * 3 * H + o;
                    st4(G3 + g + 2 * H, dnp);
                    st4(G3 + g, dzp);
                }
                lds_col4(sDn, o, j, dnp);   // rows of batch columns beyond B are zeros (operands were zero)
                lds_col4(sDzr, o, j, dzp);
            }
        }
        lds_barrier();
        if (PRE && t > 0) load_step(t - 1);
        // ---- GEMM1: d(r*h) = Wn[:, H:]^T dn_pre ; dr_pre -> sDzr ---------------------------------------------
        f32x4 ah[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) ah[i] = zero4;
        if constexpr (RKG > 0) res_tile_ks<RW>(ah[0], wnh, sDn, lane);
        else phase_tiles<TPW, GPF>(ah, Pnh, (int64_t)KGn * 64, sDn, KGn, NT, wave, lane);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ot = wave + GNW * i;
            const int o = ot * 16 + q * 4;
            if (ot < NT && o < H) {
                const f32x4 r = rreg[i];
                const f32x4 drp = ah[i] * hpreg[i] * r * (1.0f - r);
                dhc[i] += ah[i] * r;
                lds_col4(sDzr, H + o, j, drp);
                if (bvalid) st4(G3 + ((int64_t)b * T + t) * 3 * H + H + o, drp);
            }
        }
        lds_barrier();
        // ---- GEMM2: dh += [Wz[:, H:]; Wr[:, H:]]^T [dz_pre | dr_pre] ---------------------------------------
        if constexpr (RKG > 0) res_tile_ks<2 * RW>(dhc[0], wzrh, sDzr, lane);
        else phase_tiles<TPW, GPF>(dhc, Pzrh, (int64_t)KGzr * 64, sDzr, KGzr, NT, wave, lane);
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
    const int64_t n = dpk_gru_pack_floats(H, H);
    PackJobs jobs{};
    float* dst[3] = {Pz, Pr, Pn};
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < 3; ++k) {
            const int y = g * 3 + k;
            if (!W[y]) return DPK_EINVAL;
            jobs.src[y] = W[y] + H;  // the h half of the gate's [H, 2H] weight: columns H .. 2H-1
            jobs.src2[y] = W[y] + H;
            jobs.dst[y] = dst[k] + g * n;
            jobs.coff[y] = 0;
            jobs.K[y] = H;
            jobs.transposed[y] = 0;
        }
    return rnn_launch_pack(jobs, 3 * G, ld, H, H, (hipStream_t)stream);
}

extern "C" int dpk_gru_pack_backward(const float* const* W, int ld, float* Pnh, float* Pzrh, int H, int G,
                                     void* stream) {
    if (!W || !Pnh || !Pzrh || H <= 0 || G <= 0 || ld < 2 * H) return DPK_EINVAL;
    if (G > GMAX) return DPK_ERANGE;
    if (!dpk_aligned16(Pnh) || !dpk_aligned16(Pzrh)) return DPK_EALIGN;
    const int64_t n1 = dpk_gru_pack_floats(H, H), n2 = dpk_gru_pack_floats(H, 2 * H);
    PackJobs jobs{};
    for (int g = 0; g < G; ++g) {
        const float *wz = W[3 * g], *wr = W[3 * g + 1], *wn = W[3 * g + 2];
        if (!wz || !wr || !wn) return DPK_EINVAL;
        // transposed jobs: operand(out = o, k) = src[k][coff + o] for k < H, src2[k - H][coff + o] beyond
        const float* src[2] = {wn, wz};
        const float* src2[2] = {wn, wr};
        float* dst[2] = {Pnh + g * n1, Pzrh + g * n2};
        const int K[2] = {H, 2 * H};
        for (int k = 0; k < 2; ++k) {
            const int y = g * 2 + k;
            jobs.src[y] = src[k];
            jobs.src2[y] = src2[k];
            jobs.dst[y] = dst[k];
            jobs.coff[y] = H;
            jobs.K[y] = K[k];
            jobs.transposed[y] = 1;
        }
    }
    return rnn_launch_pack(jobs, 2 * G, ld, H, 2 * H, (hipStream_t)stream);
}

extern "C" int dpk_gru_forward(const float* const* xp, const float* const* xg, const float* const* h0,
                               const float* Pz, const float* Pr, const float* Pn, float* out, float* Z, float* R,
                               float* N, float* Azr, float* An, int B, int T, int H, int G, void* stream) {
    if (!xp || !xg || !Pz || !Pr || !Pn || !out || !Z || !R || !N || !Azr || !An) return DPK_EINVAL;
    if (int rc = gru_check_shape(B, T, H, G)) return rc;
    const void* ptrs[] = {Pz, Pr, Pn, out, Z, R, N, Azr, An};
    for (const void* p : ptrs)
        if (!dpk_aligned16(p)) return DPK_EALIGN;
    GPtrs xps{}, xgs{}, h0s{};
    for (int g = 0; g < G; ++g) {
        if (!xp[g] || !xg[g]) return DPK_EINVAL;
        xps.p[g] = xp[g];
        xgs.p[g] = xg[g];
        h0s.p[g] = h0 ? h0[g] : nullptr;
        if (!dpk_aligned16(xps.p[g]) || !dpk_aligned16(xgs.p[g]) || !dpk_aligned16(h0s.p[g])) return DPK_EALIGN;
    }
    const int tpw = gru_tpw(H);
    const size_t lds = (size_t)2 * ((H + 15) / 16) * 16 * GBT * sizeof(float);
    dim3 grid((unsigned)((B + GBT - 1) / GBT), (unsigned)G);
    hipStream_t s = (hipStream_t)stream;
#define DPK_GRU_F(TPW, RKG)                                                                                         \
    hipLaunchKernelGGL((gru_fwd_kernel<TPW, (TPW < 4), RKG>), grid, dim3(GNW * 64), lds, s, xps, xgs, h0s,          \
                       (const f32x4*)Pz, (const f32x4*)Pr, (const f32x4*)Pn, out, Z, R, N, Azr, An, B, T, H)
    const int kg = (H + 15) / 16;
    const bool res = H % 16 == 0;  // the backward's [dz | dr] operand must be exactly 2*kg K groups
    if (tpw == 1 && kg == 8 && res) DPK_GRU_F(1, 8);        // H = 128: weights live in registers
    else if (tpw == 1 && kg == 4 && res) DPK_GRU_F(1, 4);   // H = 64
    else if (tpw == 1) DPK_GRU_F(1, 0);
    else if (tpw == 2) DPK_GRU_F(2, 0);
    else DPK_GRU_F(4, 0);
#undef DPK_GRU_F
    return dpk_launch_status();
}

extern "C" int dpk_gru_backward(const float* const* dout, const float* const* dhT, const float* Z, const float* R,
                                const float* N, const float* Azr, const float* Pnh, const float* Pzrh, float* G3,
                                float* dh0, int B, int T, int H, int G, void* stream) {
    if (!dout || !Z || !R || !N || !Azr || !Pnh || !Pzrh || !G3) return DPK_EINVAL;
    if (int rc = gru_check_shape(B, T, H, G)) return rc;
    const void* ptrs[] = {Z, R, N, Azr, Pnh, Pzrh, G3, dh0};
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
#define DPK_GRU_B(TPW, RKG)                                                                                         \
    do {                                                                                                            \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)gru_bwd_kernel<TPW, (TPW < 4), RKG>,                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e_ != hipSuccess) return (int)e_;                                                                   \
        }                                                                                                           \
        hipLaunchKernelGGL((gru_bwd_kernel<TPW, (TPW < 4), RKG>), grid, dim3(GNW * 64), lds, s, ds, hs, Z, R, N,    \
                           Azr,                                                                                     \
                           (const f32x4*)Pnh, (const f32x4*)Pzrh, G3, dh0, B, T, H);                                \
    } while (0)
    const int kg = (H + 15) / 16;
    const bool res = H % 16 == 0;
    if (tpw == 1 && kg == 8 && res) DPK_GRU_B(1, 8);
    else if (tpw == 1 && kg == 4 && res) DPK_GRU_B(1, 4);
    else if (tpw == 1) DPK_GRU_B(1, 0);
    else if (tpw == 2) DPK_GRU_B(2, 0);
    else DPK_GRU_B(4, 0);
#undef DPK_GRU_B
    return dpk_launch_status();
}
