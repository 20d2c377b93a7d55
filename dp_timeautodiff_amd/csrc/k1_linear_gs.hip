// K1 — per-sample gradient of a Linear layer as a batched GEMM with K = T (one GEMM per sample).
//
//   GSw[b,o,i] = sum_t (g_scale*G[b,t,o]) * A[b,t,i]       GSb[b,o] = sum_t g_scale*G[b,t,o]
//
// The output (B x out x in floats, written once, read back once by K7) is what bounds the kernel: AI = T/2 flop per
// byte.  Three kernels, chosen by k1_launch():
//
//   linear_gs_kernel               launch-per-tile, fp32 MFMA; small layers (in <= 128 or out <= 32) and arena rows
//                                  whose stride is not a multiple of 4 floats.  One workgroup stages a [T x RS] slice
//                                  of G and a [T x CS] slice of A of ONE sample in LDS and streams the tile out.
//   linear_gs_persist_wide_kernel  large layers, T <= 32: persistent, one workgroup of 8 waves per CU walks the
//                                  (sample, 128 x 256 tile) items of its XCD; v_mfma_f32_32x32x2_f32 fed by one
//                                  ds_read_b32 + one ds_read_b128 per four MFMAs; rows leave as 16-byte non-temporal
//                                  buffer stores; the next item's operands are prefetched into registers ahead of
//                                  this item's stores.  Store-bound at T = 24 (5.4-5.8 TB/s of 6.0-6.2 achievable).
//   linear_gs_split_kernel         large layers, T > 32 (any length): the fp32 contraction itself was the bound there
//                                  (64 flop/cycle/SIMD on the fp32 matrix pipe), so every fp32 operand is split into
//                                  three bf16 pieces when it is staged and the products run on the bf16 pipe
//                                  (v_mfma_f32_32x32x16_bf16, 16x the rate), 6 of the 9 exact partial products per
//                                  fp32 product (what is dropped is < 2^-26 of the product), fp32 accumulation.  Two workgroups of 4 waves per CU, 128 x 128 tiles,
//                                  chunks of 32 steps, operands prefetched two chunks ahead.
//
// All three: every output element is a fixed-order sum (deterministic, independent of the grid); the epilogue emits
// sum(GS^2) of the tile (+ the bias slice in column-tile 0) as ONE float per (sample, slot), so the norm pass of
// DPOptimizer.clip_and_accumulate never re-reads the gradient; rows of the arena whose stride is a multiple of 32
// floats are written as whole 128-byte lines.  Grids are XCD-aware (blocks id and id+8 share an XCD): a sample's
// A/G slices are fetched into ONE XCD's L2 and re-read there by its other tiles.  Placement is used for speed only.
// Slots: dpk_linear_slots() counts 128-column tiles for the large-layer plan (a 256-column tile owns two and writes 0
// into the second), so that the three kernels agree on a layer's slot range.
#include "dpk_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

constexpr int TK = 32;  // time steps staged per LDS chunk (even)
#ifndef DPK_K1_STREAM_MAX_T
#define DPK_K1_STREAM_MAX_T 32
#endif
constexpr int STREAM_MAX_T = DPK_K1_STREAM_MAX_T;  // longest sequence kept whole in LDS (single-chunk streaming mode)

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // global rows are only 4-byte aligned in general

template <int W, int NTHR>
__device__ __forceinline__ void stage_rows(float* __restrict__ s, const float* __restrict__ g, int t0, int rows,
                                           int T, int c0, int dim, int ld, float scale) {
    // s[rows][W]  <-  scale * g[(t0 + r) * ld + c0 + c]   (zero outside [0,T) x [0,dim)).  One 16-byte global load per 4
    // columns whatever the row alignment (gfx950 global loads need dword alignment only); the last, partial group
    // of a row is read element by element so nothing past the row end is touched.
    constexpr int W4 = W / 4;
    for (int idx = threadIdx.x; idx < rows * W4; idx += NTHR) {
        const int r = idx / W4;
        const int c = (idx - r * W4) * 4;
        const int t = t0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const float* src = g + (int64_t)t * ld + c0 + c;
            const int left = dim - (c0 + c);
            if (left >= 4) {
                v = *reinterpret_cast<const f32x4u*>(src);
            } else if (left > 0) {
                v[0] = src[0];
                if (left > 1) v[1] = src[1];
                if (left > 2) v[2] = src[2];
            }
        }
        v *= scale;
        *reinterpret_cast<f32x4*>(s + r * W + c) = v;
    }
}

// NW waves; block tile RS (out) x CS (in); a wave "unit" is one 32-row strip x NT adjacent 32-col MFMA tiles.
// STREAM (T <= TK, one chunk): a wave computes a unit and stores it at once (16 live accumulator registers per
// MFMA tile, stores of unit u overlap the MFMAs of unit u+1 of other waves); otherwise the accumulators of all
// of a wave's units persist across the K-chunks.
template <int NW, int RS, int CS, int NT, bool STREAM, bool NTS>
__global__ void __launch_bounds__(NW * 64) linear_gs_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                           float* __restrict__ GSw, float* __restrict__ GSb,
                                                           float* __restrict__ normsq_part, int64_t part_stride,
                                                           int slot0, int B, int T, int in_f, int out_f,
                                                           int64_t gs_stride, int ld_w, int ld_g, float g_scale,
                                                           int tiles_m, int tiles_n, int slot_cols, int slot_mul) {
    constexpr int NTHR = NW * 64;
    constexpr int NCG = CS / (32 * NT);        // column groups per tile
    constexpr int UNITS = (RS / 32) * NCG;
    constexpr int U = (UNITS + NW - 1) / NW;   // units per wave
    static_assert(RS % 32 == 0 && CS % (32 * NT) == 0, "tile must be a whole number of MFMA units");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [rows][RS] G slice, [rows][CS] A slice, red[NW]
    const int lds_rows = STREAM ? ((T + 1) & ~1) : TK;
    float* sG = smem;
    float* sA = smem + lds_rows * RS;
    float* red = sA + lds_rows * CS;

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
    const int m0 = tm * RS;
    const int n0 = tn * CS;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const float* Ab = A + (int64_t)b * T * in_f;
    const float* Gb = G + (int64_t)b * T * ld_g;

    float bias_acc = 0.f;
    const bool do_bias = (GSb != nullptr) && (tn == 0) && ((int)threadIdx.x < RS);
    float sq = 0.f;
    float* out = GSw + (int64_t)b * gs_stride;

    auto store_unit = [&](const f32x16 (&acc)[NT], int u) {
        const int mr = u / NCG, cg = u - mr * NCG;
        const int wrow0 = m0 + mr * 32;
        const int wcol0 = n0 + cg * (32 * NT);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wcol0 + j * 32 + l31;
            if (col < ld_w) {  // columns in [in_f, ld_w) are row padding: written as exact zeros -> whole lines
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < out_f) {
                        const float v = acc[j][r];
                        float* p = out + (int64_t)row * ld_w + col;
                        if constexpr (NTS)
                            __builtin_nontemporal_store(v, p);  // whole 128-B lines, read back once much later
                        else
                            *p = v;
                        sq = __builtin_fmaf(v, v, sq);
                    }
                }
            }
        }
    };

    if constexpr (STREAM) {
        const int rows2 = (T + 1) & ~1;  // MFMA consumes k in pairs; the pad row is zero-filled
        stage_rows<RS, NTHR>(sG, Gb, 0, rows2, T, m0, out_f, ld_g, g_scale);
        stage_rows<CS, NTHR>(sA, Ab, 0, rows2, T, n0, in_f, in_f, 1.0f);
        __syncthreads();
        for (int u = wave; u < UNITS; u += NW) {
            const int mr = u / NCG, cg = u - mr * NCG;
            if (m0 + mr * 32 >= out_f || n0 + cg * (32 * NT) >= in_f) continue;
            f32x16 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const float* pg = sG + lh * RS + mr * 32 + l31;
            const float* pa = sA + lh * CS + cg * (32 * NT) + l31;
            for (int kk = 0; kk < rows2; kk += 2) {
                const float a = pg[kk * RS];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pa[kk * CS + j * 32], acc[j], 0, 0, 0);
            }
            store_unit(acc, u);
        }
        if (do_bias) {
            for (int r = 0; r < T; ++r) bias_acc += sG[r * RS + threadIdx.x];
        }
    } else {
        f32x16 acc[U][NT];
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][j][r] = 0.f;
        for (int t0 = 0; t0 < T; t0 += TK) {
            int rows = T - t0;
            if (rows > TK) rows = TK;
            const int rows2 = (rows + 1) & ~1;
            stage_rows<RS, NTHR>(sG, Gb, t0, rows2, T, m0, out_f, ld_g, g_scale);
            stage_rows<CS, NTHR>(sA, Ab, t0, rows2, T, n0, in_f, in_f, 1.0f);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int u = wave + k * NW;
                const int mr = u / NCG, cg = u - mr * NCG;
                const bool live = (u < UNITS) && (m0 + mr * 32 < out_f) && (n0 + cg * (32 * NT) < in_f);
                if (live) {
                    const float* pg = sG + lh * RS + mr * 32 + l31;
                    const float* pa = sA + lh * CS + cg * (32 * NT) + l31;
                    for (int kk = 0; kk < rows2; kk += 2) {
                        const float a = pg[kk * RS];
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[k][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pa[kk * CS + j * 32], acc[k][j], 0, 0, 0);
                    }
                }
            }
            if (do_bias) {
                for (int r = 0; r < rows; ++r) bias_acc += sG[r * RS + threadIdx.x];
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int u = wave + k * NW;
            const int mr = u / NCG, cg = u - mr * NCG;
            if (u < UNITS && m0 + mr * 32 < out_f && n0 + cg * (32 * NT) < in_f) store_unit(acc[k], u);
        }
    }
    if (do_bias && m0 + (int)threadIdx.x < out_f) {
        GSb[(int64_t)b * gs_stride + m0 + threadIdx.x] = bias_acc;
        sq = __builtin_fmaf(bias_acc, bias_acc, sq);
    }
    const float tot = block_sum<NW>(sq, red);
    // a layer's slots are counted in 128-column tiles (dpk_linear_slots); a 256-column tile owns two of them
    if (threadIdx.x == 0) {
        float* pp = normsq_part + (int64_t)b * part_stride + slot0 + tm * slot_cols + tn * slot_mul;
        pp[0] = tot;
        if (slot_mul == 2 && tn * 2 + 1 < slot_cols) pp[1] = 0.f;
    }
}


// ---- persistent kernels (T <= 48) -------------------------------------------------------------------------------
// At the sizes of the real step (arena of several GB, operands beyond the Infinity Cache) the launch-per-tile kernel
// above loses 15-20 %: a workgroup's operand loads are issued after the previous occupant's stores and queue behind
// them in the CU's in-order vector-memory pipe, so staging and storing alternate instead of overlapping.  A persistent
// kernel runs a fixed grid over the (sample, tile) items of its XCD and fetches the operands of item i+1 into registers
// BEFORE item i's MFMAs and stores are issued (the loads sit ahead of the stores in the pipe and land under the
// MFMAs); its barriers wait for LDS only, never for the wave's own stores.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Up to DPK_K1_MAX_JOBS same-shape layers (the six gate Linears of a GRU pair) share one launch: by-value table.
struct K1Jobs {
    const float* A[DPK_K1_MAX_JOBS];
    const float* G[DPK_K1_MAX_JOBS];
    float* W[DPK_K1_MAX_JOBS];
    float* Bi[DPK_K1_MAX_JOBS];  // nullable
    int slot0[DPK_K1_MAX_JOBS];
    int n;
};

// ---- persistent, "wide unit" variant (round 2): the whole-line path of every large layer at T <= 48 -------------
// Same shape as linear_gs_persist_kernel (a fixed grid walks the (sample, tile) items of its XCD; the operands of item
// i+1 are fetched into registers before item i's MFMAs and stores are issued), with what cycle stamps on that kernel
// (tools/lab/k1_stamps.py) showed to cost the most taken out:
//   * a wave's unit is 32 rows x 128 columns with the columns PERMUTED over the lanes: lane c of a half-wave owns the
//     four adjacent columns 4c..4c+3, one per MFMA of the k-step.  One ds_read_b32 (G) and one ds_read_b128 (A: the
//     lane's four columns are contiguous in the [t][col] slice) feed FOUR v_mfma_f32_32x32x2_f32 (1.5 LDS instructions
//     per MFMA before, 0.5 now), issued one k-step ahead of their use; the four accumulators hold four adjacent
//     output elements per register index, so a row leaves as 16-byte non-temporal stores: 16 store instructions of
//     1 KiB per unit instead of 64 of 256 B.  8 units per 128 x 256 tile = one per wave;
//   * every global access is a BUFFER instruction whose guard is the hardware range check (out-of-tile lanes get an
//     out-of-range offset): no branch around a memory instruction, so the wait before staging is a counted vmcnt(N)
//     that leaves the wave's own stores in flight (it was vmcnt(0): every wave drained its stores once per item);
//   * the item walk is a few scalar adds per item (it was 64-bit divisions: 3-6k of a 22.8k-cycle item);
//   * the bias slice is the sum of the G values the wave feeds its MFMAs anyway (it was a serial T-step loop in two
//     waves that the other six waited for at the barrier).
// The k order of every output element is unchanged (t ascending): weights bit-identical to the narrow-unit kernel;
// the bias sums even and odd time steps separately (two half-waves) before adding them.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef DPK_K1_STORE_AUX
#define DPK_K1_STORE_AUX 2  // buffer-store cache policy of the per-sample gradients: nt (bit 1); sc1 = bit 4, sc0 = bit 0
#endif
constexpr unsigned OOB = 0x80000000u;  // beyond any descriptor's range: loads return 0, stores are dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}



// Elements at or beyond `left` (what is left of the row) of a 4-float group are not the tile's: zero them.  Applied
// when the group is staged, not when it is fetched, so that nothing waits on a prefetch load early.
__device__ __forceinline__ f32x4 mask4(f32x4 v, int left) {
    v[1] = left > 1 ? v[1] : 0.f;
    v[2] = left > 2 ? v[2] : 0.f;
    v[3] = left > 3 ? v[3] : 0.f;
    return v;
}

// Products and sums that must round the same way in every kernel that forms them (the bias slice of the bf16-pipe
// kernels): never contracted into an fma, whatever the compiler's fp-contract setting.
// (the pragma takes the `contract` flag off the instructions generated in its scope; an fma needs it on both)
__device__ __forceinline__ f32x4 scale4_rn(f32x4 v, float s) {
#pragma clang fp contract(off)
    return v * s;
}
__device__ __forceinline__ f32x4 add4_rn(f32x4 a, f32x4 b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

template <int NG, int NA>
__global__ void __launch_bounds__(512) linear_gs_persist_wide_kernel(
    const K1Jobs jobs, float* __restrict__ normsq_part, int64_t part_stride, int B, int T, int in_f, int out_f,
    int64_t gs_stride, int ld_w, int ld_g, float g_scale, int tiles_m, int tiles_n, int slot_cols) {
    constexpr int NW = 8, RS = 128, CS = 256, NTHR = NW * 64;
    constexpr int R4 = RS / 4, C4 = CS / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [Tp][RS] G, [Tp][CS] A, 2 pad rows, red[2][NW]
    const int Tp = (T + 1) & ~1;
    float* sG = smem;
    float* sA = smem + Tp * RS;
    float* red = sA + Tp * CS + 2 * CS;  // 2 rows of padding: the one-k-step-ahead operand reads stay inside LDS
    const int totG = Tp * R4, totA = Tp * C4;
    const int ksteps = Tp >> 1;

    // ---- XCD-aware item walk in (job, sample slot, tile row, tile column) form -----------------------------------
    const int ntile = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7;
    const int nbx = gridDim.x >> 3;        // workgroups per XCD (grid is a multiple of 8)
    const int nsx = (B - xcd + 7) / 8;     // samples on this XCD: b = 8 s + xcd
    struct Item {
        int job, s, tm, tn;
    };
    const int d_s = nbx / ntile, d_tm = (nbx % ntile) / tiles_n, d_tn = nbx % tiles_n;
    auto advance = [&](Item x) {
        x.tn += d_tn;
        if (x.tn >= tiles_n) { x.tn -= tiles_n; ++x.tm; }
        x.tm += d_tm;
        if (x.tm >= tiles_m) { x.tm -= tiles_m; ++x.s; }
        x.s += d_s;
        while (x.s >= nsx && x.job < jobs.n) { x.s -= nsx; ++x.job; }  // job == jobs.n: past the last item
        return x;
    };
    Item cur;
    {
        const int w = blockIdx.x >> 3;  // first item of this workgroup (job 0)
        cur.job = 0;
        cur.s = w / ntile;
        cur.tm = (w % ntile) / tiles_n;
        cur.tn = w % tiles_n;
        while (cur.s >= nsx && cur.job < jobs.n) { cur.s -= nsx; ++cur.job; }
    }
    if (nsx <= 0 || cur.job >= jobs.n) return;  // uniform

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, SGPR math
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int mr = wave >> 1, cg = wave & 1;  // this wave's 32 x 128 unit inside the 128 x 256 tile

    f32x4 rg[NG], ra[NA];
    auto prefetch = [&](const Item& x) {
        // the sample's slices: G rows have stride ld_g (out_f valid columns), A rows are dense
        const int64_t b = x.s * 8 + xcd;
        const auto rsG = make_rsrc(jobs.G[x.job] + b * T * ld_g, ((int64_t)(T - 1) * ld_g + out_f) * 4);
        const auto rsA = make_rsrc(jobs.A[x.job] + b * T * in_f, (int64_t)T * in_f * 4);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * NTHR;
            const int t = idx / R4, col = x.tm * RS + (idx - t * R4) * 4;
            const unsigned off = (t < T && col < out_f) ? (unsigned)(t * ld_g + col) * 4u : OOB;
            rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int idx = tid + k * NTHR;
            const int t = idx / C4, col = x.tn * CS + (idx - t * C4) * 4;
            const unsigned off = (t < T && col < in_f) ? (unsigned)(t * in_f + col) * 4u : OOB;
            ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        }
    };
    auto stage = [&](const Item& x) {  // prefetched registers (operands of item x) -> LDS
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * NTHR;
            const int left = out_f - (x.tm * RS + (idx % R4) * 4);
            if (idx < totG) *reinterpret_cast<f32x4*>(sG + idx * 4) = mask4(rg[k], left) * g_scale;
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int idx = tid + k * NTHR;
            const int left = in_f - (x.tn * CS + (idx % C4) * 4);
            if (idx < totA) *reinterpret_cast<f32x4*>(sA + idx * 4) = mask4(ra[k], left);
        }
    };

    prefetch(cur);
    int parity = 0;
    bool more = true;
    // one item: stage its prefetched operands, prefetch the next item's, contract, store, publish the squared norm
    auto do_item = [&]() {
        stage(cur);
        lds_only_barrier();
        const Item nxt = advance(cur);
        const bool has_next = nxt.job < jobs.n;
        if (has_next) prefetch(nxt);  // ahead of this item's stores in the vector-memory pipe

        const int m0 = cur.tm * RS, n0 = cur.tn * CS;
        const int64_t b = cur.s * 8 + xcd;
        const bool live = (m0 + mr * 32 < out_f) && (n0 + cg * 128 < in_f);
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        float bsum = 0.f;
        if (live) {
            const float* pg = sG + lh * RS + mr * 32 + l31;
            const float* pa = sA + lh * CS + cg * 128 + 4 * l31;
            float a_nx = pg[0];
            f32x4 b_nx = *reinterpret_cast<const f32x4*>(pa);
            for (int q = 0; q < ksteps; ++q) {
                const float a = a_nx;
                const f32x4 bv = b_nx;
                pg += 2 * RS;
                pa += 2 * CS;
                a_nx = *pg;  // k-step q+1 (one past the end on the last step: readable padding, never used)
                b_nx = *reinterpret_cast<const f32x4*>(pa);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[3], acc[3], 0, 0, 0);
                bsum += a;
            }
        }
        // stores: rows >= out_f fall outside the descriptor, columns >= ld_w and dead units get the out-of-range offset
        // (rows / columns outside the layer hold exact zeros — zero-filled operands — so the squares need no guard)
        float sq = 0.f;
        {
            const auto rsW = make_rsrc(jobs.W[cur.job] + b * gs_stride, live ? (int64_t)out_f * ld_w * 4 : 0);
            const int col = n0 + cg * 128 + 4 * l31;
            unsigned soff = col < ld_w ? (unsigned)((m0 + mr * 32 + 4 * lh) * ld_w + col) * 4u : OOB;
            const unsigned row_bytes = (unsigned)ld_w * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsW, soff, 0, DPK_K1_STORE_AUX);
                soff += ((r & 3) == 3 ? 5u : 1u) * row_bytes;  // accumulator rows 0 1 2 3 8 9 10 11 ... (OOB stays OOB)
                sq = __builtin_fmaf(v[0], v[0], sq);
                sq = __builtin_fmaf(v[1], v[1], sq);
                sq = __builtin_fmaf(v[2], v[2], sq);
                sq = __builtin_fmaf(v[3], v[3], sq);
            }
        }
        // bias slice (column-tile 0 only): rows m0 + mr*32 + l31, from the waves of column group 0
        {
            bsum += __shfl_xor(bsum, 32, 64);
            const float* GSb = jobs.Bi[cur.job];
            const bool has_bias = live && GSb != nullptr && cur.tn == 0;
            const auto rsB = make_rsrc(GSb + b * gs_stride, has_bias ? (int64_t)out_f * 4 : 0);
            const bool owner = cg == 0 && lh == 0;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rsB,
                                                  owner ? (unsigned)(m0 + mr * 32 + l31) * 4u : OOB, 0, 0);
            if (has_bias && owner) sq = __builtin_fmaf(bsum, bsum, sq);
        }
        sq = wave_sum(sq);
        if (lane == 0) red[parity * NW + wave] = sq;
        lds_only_barrier();  // every wave is done with this item's operands; red[parity] is complete
        {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) tot += red[parity * NW + i];
            const auto rsP = make_rsrc(normsq_part + b * part_stride, (int64_t)part_stride * 4);
            // slots are counted in 128-column tiles: this 256-column tile owns two (the second, if it exists, gets 0)
            const bool second = tid == 1 && cur.tn * 2 + 1 < slot_cols;
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(unsigned, tid == 0 ? tot : 0.f), rsP,
                (tid == 0 || second) ? (unsigned)(jobs.slot0[cur.job] + cur.tm * slot_cols + cur.tn * 2 + tid) * 4u : OOB,
                0, 0);
        }
        parity ^= 1;
        more = has_next;
        cur = nxt;
    };
    // The first item is peeled: the loop is then entered only from code that has issued an item's stores behind its
    // prefetch, which is what lets the compiler count them (vmcnt(N), N >= 17) instead of waiting for everything.
    do_item();
    while (more) do_item();
}

// ---- persistent wide-unit kernel over layers of DIFFERENT shapes (small batches) ---------------------------------------
// At the reference's own batch size (B ~ 50) a layer's launch is a handful of items per CU: 12 launches per step each
// pay a ramp and a tail.  All layers whose arena rows allow 16-byte stores go through ONE launch instead: the job table
// carries each layer's shape, the item walk is flat over (job, sample, tile) with one 32-bit division per item, and the
// per-item constants (T, strides, LDS split) are scalar loads from the table.  Per tile the arithmetic is exactly that
// of linear_gs_persist_wide_kernel (same k order, same bias pairing): results equal per-layer launches bit for bit.
struct K1Shape {
    int T, in_f, out_f, ld_w, ld_g, tiles_m, tiles_n, slot_cols, slot_mul, tile0;  // tile0: tiles of the jobs before
};
struct K1MultiJobs {
    const float* A[DPK_K1_MULTI_MAX_JOBS];
    const float* G[DPK_K1_MULTI_MAX_JOBS];
    float* W[DPK_K1_MULTI_MAX_JOBS];
    float* Bi[DPK_K1_MULTI_MAX_JOBS];  // nullable
    int slot0[DPK_K1_MULTI_MAX_JOBS];
    K1Shape sh[DPK_K1_MULTI_MAX_JOBS];
    int n, tiles_total;
};

template <int NG, int NA>
__global__ void __launch_bounds__(512) linear_gs_multi_kernel(const K1MultiJobs jobs, float* __restrict__ normsq_part,
                                                             int64_t part_stride, int B, int64_t gs_stride,
                                                             float g_scale) {
    constexpr int NW = 8, RS = 128, CS = 256, NTHR = NW * 64;
    constexpr int R4 = RS / 4, C4 = CS / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // red[2][NW], [Tp][RS] G, [Tp][CS] A, 2 pad rows
    float* red = smem;
    float* sG = smem + 2 * NW;

    const int xcd = blockIdx.x & 7;
    const int nbx = gridDim.x >> 3;
    const int nsx = (B - xcd + 7) / 8;               // samples on this XCD: b = 8 s + xcd
    const int total = nsx * jobs.tiles_total;        // items of this XCD, job-major: job j owns nsx * ntile_j of them
    struct Item {
        int it, job, s, tm, tn;
    };
    auto locate = [&](int it, int job) {  // job only moves forward
        Item x;
        x.it = it;
        if (it >= total) { x.job = jobs.n; x.s = x.tm = x.tn = 0; return x; }
        while (job + 1 < jobs.n && it >= nsx * jobs.sh[job + 1].tile0) ++job;
        const K1Shape& h = jobs.sh[job];
        const int local = it - nsx * h.tile0, ntile = h.tiles_m * h.tiles_n;
        x.job = job;
        x.s = local / ntile;
        const int tile = local - x.s * ntile;
        x.tm = tile / h.tiles_n;
        x.tn = tile - x.tm * h.tiles_n;
        return x;
    };
    Item cur = locate(blockIdx.x >> 3, 0);
    if (nsx <= 0 || cur.job >= jobs.n) return;  // uniform

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int mr = wave >> 1, cg = wave & 1;

    f32x4 rg[NG], ra[NA];
    auto prefetch = [&](const Item& x) {
        const K1Shape h = jobs.sh[x.job];
        const int64_t b = x.s * 8 + xcd;
        const auto rsG = make_rsrc(jobs.G[x.job] + b * h.T * h.ld_g, ((int64_t)(h.T - 1) * h.ld_g + h.out_f) * 4);
        const auto rsA = make_rsrc(jobs.A[x.job] + b * h.T * h.in_f, (int64_t)h.T * h.in_f * 4);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * NTHR;
            const int t = idx / R4, col = x.tm * RS + (idx - t * R4) * 4;
            const unsigned off = (t < h.T && col < h.out_f) ? (unsigned)(t * h.ld_g + col) * 4u : OOB;
            rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int idx = tid + k * NTHR;
            const int t = idx / C4, col = x.tn * CS + (idx - t * C4) * 4;
            const unsigned off = (t < h.T && col < h.in_f) ? (unsigned)(t * h.in_f + col) * 4u : OOB;
            ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        }
    };
    auto stage = [&](const Item& x) {
        const K1Shape h = jobs.sh[x.job];
        const int Tp = (h.T + 1) & ~1;
        float* sA = sG + Tp * RS;
        const int totG = Tp * R4, totA = Tp * C4;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * NTHR;
            const int left = h.out_f - (x.tm * RS + (idx % R4) * 4);
            if (idx < totG) *reinterpret_cast<f32x4*>(sG + idx * 4) = mask4(rg[k], left) * g_scale;
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int idx = tid + k * NTHR;
            const int left = h.in_f - (x.tn * CS + (idx % C4) * 4);
            if (idx < totA) *reinterpret_cast<f32x4*>(sA + idx * 4) = mask4(ra[k], left);
        }
    };

    prefetch(cur);
    int parity = 0;
    bool more = true;
    auto do_item = [&]() {
        stage(cur);
        lds_only_barrier();
        const Item nxt = locate(cur.it + nbx, cur.job);
        const bool has_next = nxt.job < jobs.n;
        if (has_next) prefetch(nxt);

        const K1Shape h = jobs.sh[cur.job];
        const int Tp = (h.T + 1) & ~1, ksteps = Tp >> 1;
        const float* sA = sG + Tp * RS;
        const int m0 = cur.tm * RS, n0 = cur.tn * CS;
        const int64_t b = cur.s * 8 + xcd;
        const bool live = (m0 + mr * 32 < h.out_f) && (n0 + cg * 128 < h.in_f);
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        float bsum = 0.f;
        if (live) {
            const float* pg = sG + lh * RS + mr * 32 + l31;
            const float* pa = sA + lh * CS + cg * 128 + 4 * l31;
            float a_nx = pg[0];
            f32x4 b_nx = *reinterpret_cast<const f32x4*>(pa);
            for (int q = 0; q < ksteps; ++q) {
                const float a = a_nx;
                const f32x4 bv = b_nx;
                pg += 2 * RS;
                pa += 2 * CS;
                a_nx = *pg;  // one past the end on the last step: readable padding, never used
                b_nx = *reinterpret_cast<const f32x4*>(pa);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[3], acc[3], 0, 0, 0);
                bsum += a;
            }
        }
        float sq = 0.f;
        {
            const auto rsW = make_rsrc(jobs.W[cur.job] + b * gs_stride, live ? (int64_t)h.out_f * h.ld_w * 4 : 0);
            const int col = n0 + cg * 128 + 4 * l31;
            unsigned soff = col < h.ld_w ? (unsigned)((m0 + mr * 32 + 4 * lh) * h.ld_w + col) * 4u : OOB;
            const unsigned row_bytes = (unsigned)h.ld_w * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsW, soff, 0, DPK_K1_STORE_AUX);
                soff += ((r & 3) == 3 ? 5u : 1u) * row_bytes;
                sq = __builtin_fmaf(v[0], v[0], sq);
                sq = __builtin_fmaf(v[1], v[1], sq);
                sq = __builtin_fmaf(v[2], v[2], sq);
                sq = __builtin_fmaf(v[3], v[3], sq);
            }
        }
        {
            bsum += __shfl_xor(bsum, 32, 64);
            const float* GSb = jobs.Bi[cur.job];
            const bool has_bias = live && GSb != nullptr && cur.tn == 0;
            const auto rsB = make_rsrc(GSb + b * gs_stride, has_bias ? (int64_t)h.out_f * 4 : 0);
            const bool owner = cg == 0 && lh == 0;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rsB,
                                                  owner ? (unsigned)(m0 + mr * 32 + l31) * 4u : OOB, 0, 0);
            if (has_bias && owner) sq = __builtin_fmaf(bsum, bsum, sq);
        }
        sq = wave_sum(sq);
        if (lane == 0) red[parity * NW + wave] = sq;
        lds_only_barrier();
        {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) tot += red[parity * NW + i];
            const auto rsP = make_rsrc(normsq_part + b * part_stride, (int64_t)part_stride * 4);
            // the layer's slots are [tiles_m][slot_cols]; a 256-column tile of the large plan owns two (second: 0)
            const bool second = tid == 1 && h.slot_mul == 2 && cur.tn * 2 + 1 < h.slot_cols;
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(unsigned, tid == 0 ? tot : 0.f), rsP,
                (tid == 0 || second)
                    ? (unsigned)(jobs.slot0[cur.job] + cur.tm * h.slot_cols + cur.tn * h.slot_mul + tid) * 4u
                    : OOB,
                0, 0);
        }
        parity ^= 1;
        more = has_next;
        cur = nxt;
    };
    do_item();
    while (more) do_item();
}

// ---- persistent, split-operand variant (round 2): fp32 products on the bf16 matrix pipe --------------------------
// On gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 flop per cycle per SIMD) is 16x slower than the bf16 one
// (v_mfma_f32_32x32x16_bf16: 1024), and at T = 48 the fp32 contraction alone (12.3k cycles per 128 x 256 item) is longer
// than the item's stores (9.2k cycles at 8 TB/s): K1 was MFMA-issue-bound, not HBM-bound.  Every fp32 number is the
// sum of three bf16 numbers, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to
// nearest; both differences are exact, |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|, and what x3 leaves is below 2^-27 |x|),
// a bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32.  So
//     g * a = sum_{p,q} g_p * a_q                                  (9 exact partial products)
// and the kernel issues TERMS of the 9 bf16 MFMAs per fp32 one: 9 = every partial product; 8 drops only g_3*a_3
// (<= 2^-36 |g a|); 6 drops p+q > 2 (<= 2^-26 |g a|: a quarter of the rounding error of the fp32 product itself,
// 2^-24 |g a|).  Small terms are accumulated first.  The operands are split once, when they are staged into
// LDS (4 VALU ops + 1.5 packs per element), as three k-contiguous bf16 images [piece][row or column][t], so one
// ds_read_b128 is one MFMA operand (8 consecutive t of one row/column); rows are (2*kc+1)*16 bytes: an odd number of
// 16-byte chunks makes both the b128 operand reads and the staging writes conflict-free.
// Structure, item walk, prefetch, buffer addressing and stores are those of linear_gs_persist_wide_kernel; sequences
// longer than 48 steps run as chunks of 48 with the accumulators kept across chunks (MULTI).  The bias slice is
// summed in fp32 from the staged values (per 4-step group by the staging thread, groups in t order by the owner lane).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// {bf16(lo), bf16(hi)}, round to nearest even: one instruction packs a pair
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rn(float lo, float hi) {   // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// First piece of a pair: the operand is clamped to the largest finite bf16 (0x7F7F0000 = 3.3895e38) before it is rounded.
// Round-to-nearest takes a finite fp32 above 3.3961e38 to bf16 infinity; the remainder would then be -inf and every
// product NaN where the fp32 instruction gives a finite number.  Clamped, the piece is the largest finite bf16, the
// remainder (< 2^120) is exact and the split stays exact up to FLT_MAX.  An infinite operand becomes {max, inf, NaN}: its
// products are NaN (the fp32 instruction gives +-inf or NaN); a NaN operand stays NaN through the remainders.
__device__ __forceinline__ unsigned pack_bf16_first(float lo, float hi) {
    const float M = __builtin_bit_cast(float, 0x7F7F0000u);
    return pack_bf16_rn(__builtin_amdgcn_fmed3f(lo, -M, M), __builtin_amdgcn_fmed3f(hi, -M, M));
}
// what the pair's two bf16 values leave of lo / hi: exact differences (the operands are within a factor of two)
__device__ __forceinline__ float rest_lo(float lo, unsigned pk) { return lo - __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float rest_hi(float hi, unsigned pk) { return hi - __builtin_bit_cast(float, pk & 0xffff0000u); }

// ---- hand-counted operand loads --------------------------------------------------------------------------------------
// gfx950 has ONE in-order counter (vmcnt) for a wave's vector-memory loads and stores.  A wait for operands that were
// requested two steps ago must therefore name exactly how many younger operations may stay in flight; the compiler's
// own wait is the minimum over every path it can see through the loop (in this kernel: vmcnt(0) / vmcnt(8), i.e. "my
// last tile's rows have reached memory"), which serialised each wave's stores with its next chunk.  The operand loads
// are therefore issued from inline assembly, which the compiler's counter does not track, and waited for with an
// explicit count.  Contract kept by the code below (checked in the ISA by tools/check_k1_vmcnt.py):
//   * every step issues SPLIT_VM_PER_STEP vector-memory instructions: 8 operand loads, then 18 stores (real rows, or
//     out-of-range 4-byte stand-ins on steps that store nothing; range-checked away by the hardware);
//   * the registers a load writes are not read before split_wait_loads() has named them as its in/out operands.
constexpr int SPLIT_VM_LOADS = 8, SPLIT_VM_STORES = 18, SPLIT_VM_PER_STEP = SPLIT_VM_LOADS + SPLIT_VM_STORES;
// operands of step i were requested in step i-2, ahead of that step's stores: younger = 18 + 26
constexpr int SPLIT_VM_YOUNGER = SPLIT_VM_STORES + SPLIT_VM_PER_STEP;
static_assert(SPLIT_VM_YOUNGER < 64, "vmcnt is a 6-bit counter");

__device__ __forceinline__ u32x4 rsrc_words(const void* base, int64_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
}
// The destination is an in/out operand: the load lands in the registers the variable already lives in, so a value that
// is loaded on one path only (a branch) needs no copy where the paths meet — a copy there would read registers whose
// load is still in flight.
__device__ __forceinline__ void asm_buffer_load_b128(f32x4& v, const u32x4 rs, unsigned off) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen ; DPK_OPLOAD" : "+v"(v) : "v"(off), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void split_wait_loads(f32x4 (&r)[2][4]) {
    asm volatile("s_waitcnt vmcnt(%8) ; DPK_OPWAIT %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[1][0]), "+v"(r[1][1]),
                   "+v"(r[1][2]), "+v"(r[1][3])
                 : "n"(N)
                 : "memory");
}

#ifndef DPK_K1_ABL
#define DPK_K1_ABL 0  // lab ablations (results wrong): 1 = drop the stores, 2 = skip the MFMAs, 4 = skip the staging
#endif
#ifdef DPK_K1_STAMPS  // lab: per-phase cycle stamps of the split kernel (tools/lab/k1_stamps.py)
__device__ unsigned long long dpk_stamps[64 * 12 * 9];
#define STAMP(i)                                                     \
    do {                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_last;                                 \
        st_last = now_;                                              \
    } while (0)
#else
#define STAMP(i)
#endif
constexpr int SPLIT_TC = 32;    // time steps per LDS chunk (2 MFMA k-steps of 16)
constexpr int SPLIT_NQ = 8;     // 4-step groups per chunk
constexpr int SPLIT_CS = 128;   // tile: 128 (out) x 128 (in)

// Two workgroups of 4 waves per CU (one wave of each per SIMD, <= 256 registers, <= 78 KB of LDS each): while one
// workgroup splits and stages a chunk (VALU + LDS writes) the other contracts or stores (matrix pipe, vector memory).
// A wave owns one 32 x 128 unit of the 128 x 128 tile and keeps TWO steps of prefetched operands in registers: a
// step's loads are issued two steps before they are staged, so the chain "loads land -> split and stage -> issue the
// next loads" (load latency behind a CU's own stores is 4-6 us, a chunk's MFMAs take 1) does not pace the kernel.
// A workgroup walks GROUPS (job, sample, column tile) of its XCD and, inside a group, the row tiles: the tiles of one
// group share their A slice.  MODE 2 (32 < T <= 48: the denoiser's T = 48) keeps that slice's images resident in LDS
// for the whole group — A is fetched, split and staged once per group instead of once per tile (44 % of the staging
// work and of the operand fetches at 8 row tiles per group) — and streams G in chunks of 32 steps; MODE 1 (T > 48)
// and MODE 0 (T <= 32) stage both operands per 32-step chunk (the group order still puts a slice's re-reads next to
// each other in time: L2 hits instead of fabric fetches).
template <int TERMS, int MODE>
__global__ void __launch_bounds__(256, 2) linear_gs_split_kernel(
    const K1Jobs jobs, float* __restrict__ normsq_part, int64_t part_stride, int B, int T, int in_f, int out_f,
    int64_t gs_stride, int ld_w, int ld_g, float g_scale, int tiles_m, int tiles_n, int rowch, int rowch_a) {
    constexpr int NW = 4, RS = 128, CS = SPLIT_CS, NTHR = NW * 64, NR = 2;
    constexpr bool ARES = MODE == 2, MULTI = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int ROWB = rowch * 16;                 // bytes per row of one G piece image (an odd number of 16-byte chunks)
    const int ROWA = rowch_a * 16;               // the same for A (ARES: the whole sequence; else = ROWB)
    const int PS = RS * ROWB, PSA = CS * ROWA;   // piece strides
    unsigned char* sG = smem_b;                  // [3][RS][ROWB]
    unsigned char* sA = smem_b + 3 * PS;         // [3][CS][ROWA]
    float* sBias = reinterpret_cast<float*>(sA + 3 * PSA);  // [SPLIT_NQ][RS]
    float* red = sBias + SPLIT_NQ * RS;                     // [2][NW]

    // ---- XCD-aware walk: groups (job, sample slot, column tile) round-robin over the XCD's workgroups; inside a group
    // the row tiles, inside a tile the chunks.  ARES: step c = -1 of a group stages its A slice. --------------------------
    const int xcd = blockIdx.x & 7;
    const int nbx = gridDim.x >> 3;
    const int nsx = (B - xcd + 7) / 8;
    const int nch = MULTI ? (T + SPLIT_TC - 1) / SPLIT_TC : 1;
    struct Cur {
        int job, s, tm, tn, c;
    };
    // ARES: the unit of distribution is the group (stride nbx over (job, s, tn)); otherwise the tile (stride nbx over
    // (job, s, tm, tn): the workgroups of an XCD then work on the same few samples at the same time and meet in L2)
    const int ntile = tiles_m * tiles_n;
    const int d_s = ARES ? nbx / tiles_n : nbx / ntile;
    const int d_tm = ARES ? 0 : (nbx % ntile) / tiles_n;
    const int d_tn = nbx % tiles_n;
    auto next = [&](Cur x) {
        if (ARES && x.c < 0) { x.c = 0; return x; }
        if (MULTI && x.c + 1 < nch) { ++x.c; return x; }
        x.c = 0;
        if constexpr (ARES) {
            if (x.tm + 1 < tiles_m) { ++x.tm; return x; }
            x.tm = 0;
            x.c = -1;
            x.tn += d_tn;
            if (x.tn >= tiles_n) { x.tn -= tiles_n; ++x.s; }
        } else {
            x.tn += d_tn;
            if (x.tn >= tiles_n) { x.tn -= tiles_n; ++x.tm; }
            x.tm += d_tm;
            if (x.tm >= tiles_m) { x.tm -= tiles_m; ++x.s; }
        }
        x.s += d_s;
        while (x.s >= nsx && x.job < jobs.n) { x.s -= nsx; ++x.job; }  // job == jobs.n: past the last item
        return x;
    };
    Cur cur;
    {
        const int w = blockIdx.x >> 3;
        cur.job = 0;
        cur.c = ARES ? -1 : 0;
        if constexpr (ARES) {
            cur.tm = 0;
            cur.s = w / tiles_n;
        } else {
            cur.s = w / ntile;
            cur.tm = (w % ntile) / tiles_n;
        }
        cur.tn = w % tiles_n;
        while (cur.s >= nsx && cur.job < jobs.n) { cur.s -= nsx; ++cur.job; }
    }
    if (nsx <= 0 || cur.job >= jobs.n) return;  // uniform

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int mr = wave;  // this wave's 32 x 128 unit: MFMA row block mr = features m0 + 4*slot + mr

    // chunk c: steps [32 c, ...), kc k-steps of 16, nq staged 4-step groups (MULTI stages whole k-steps: the groups past
    // the sequence end are loaded out of range = zeros; single-chunk mode zeroes them once, below)
    auto chunk_kc = [&](int c) { const int len = min(SPLIT_TC, T - c * SPLIT_TC); return (len + 15) >> 4; };
    auto chunk_nq = [&](int c) { return MULTI ? 4 * chunk_kc(c) : (T + 3) >> 2; };
    const int nq_all = 4 * ((T + 15) >> 4);  // ARES: 4-step groups of the whole (k-step padded) sequence

    // staging tasks of a step: task = (4-step group tq, 4 adjacent columns cgp*4..+3); [0, Gpad) are G's, then A's;
    // a wave's 64 tasks are all of one kind (Gpad is a multiple of 64): the descriptor is chosen with scalar code.
    // ARES: an A step has only A tasks (of the whole sequence), a G step only G tasks.
    auto tasks_g = [&](int c) { return (ARES && c < 0) ? 0 : 32 * chunk_nq(c); };
    auto quads_a = [&](int c) { return ARES ? (c < 0 ? nq_all : 0) : chunk_nq(c); };

    // A cursor past the last group fetches nothing (empty descriptor), so there is no branch around the loads.
    f32x4 rq[2][NR][4] = {};
    auto prefetch = [&](f32x4 (&r)[NR][4], const Cur& x) {
        const bool valid = x.job < jobs.n;
        const int job = valid ? x.job : 0;
        const int64_t b = x.s * 8 + xcd;
        const int nqg = tasks_g(x.c) >> 5, nqa = quads_a(x.c);
        const int t0 = (x.c < 0 ? 0 : x.c) * SPLIT_TC;
        const int Gpad = (32 * nqg + 63) & ~63;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int wbase = wave * 64 + k * NTHR;
            const bool isG = wbase < Gpad;
            const float* base = isG ? jobs.G[job] + b * T * ld_g : jobs.A[job] + b * T * in_f;
            // rows at or past T lie beyond the descriptor: no per-row test
            const int64_t bytes = !valid ? 0 : isG ? ((int64_t)(T - 1) * ld_g + out_f) * 4 : (int64_t)T * in_f * 4;
            const u32x4 rs = rsrc_words(base, bytes);
            const int idx = wbase + lane - (isG ? 0 : Gpad);
            const int tq = idx >> 5;
            const int col = (isG ? x.tm * RS : x.tn * CS) + (idx & 31) * 4;
            const int ld = isG ? ld_g : in_f;
            const int dim = isG ? out_f : in_f;
            unsigned off = (tq < (isG ? nqg : nqa) && col < dim) ? (unsigned)((t0 + 4 * tq) * ld + col) * 4u : OOB;
            const unsigned rowb = (unsigned)ld * 4u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm_buffer_load_b128(r[k][i], rs, off);
                off += rowb;  // an out-of-range offset stays out of range
            }
        }
    };
    auto stage = [&](const f32x4 (&r)[NR][4], const Cur& x) {  // prefetched registers -> three bf16 images in LDS
        const int nqg = tasks_g(x.c) >> 5, nqa = quads_a(x.c);
        const int Gpad = (32 * nqg + 63) & ~63;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int wbase = wave * 64 + k * NTHR;
            const bool isG = wbase < Gpad;
            const int idx = wbase + lane - (isG ? 0 : Gpad);
            const int tq = idx >> 5;
            const int cgp = idx & 31;
            const int left = isG ? out_f - (x.tm * RS + cgp * 4) : in_f - (x.tn * CS + cgp * 4);
            if (tq < (isG ? nqg : nqa)) {
                f32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = mask4(r[k][i], left);
                if (isG) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = scale4_rn(v[i], g_scale);
                    *reinterpret_cast<f32x4*>(sBias + tq * RS + cgp * 4) = add4_rn(add4_rn(v[0], v[1]), add4_rn(v[2], v[3]));
                }
                // column 4*cgp+e of the tile -> position e*32 + cgp (slot cgp of MFMA block e), for G and for A
                const int rowb = isG ? ROWB : ROWA;
                const int pstride = isG ? PS : PSA;
                unsigned char* dst = (isG ? sG : sA) + cgp * rowb + tq * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = v[0][e], x1 = v[1][e], x2 = v[2][e], x3 = v[3][e];
                    unsigned char* d = dst + e * 32 * rowb;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const u32x2 w = p == 0 ? u32x2{pack_bf16_first(x0, x1), pack_bf16_first(x2, x3)}
                                               : u32x2{pack_bf16_rn(x0, x1), pack_bf16_rn(x2, x3)};
                        *reinterpret_cast<u32x2*>(d + p * pstride) = w;
                        if (p < 2) {
                            x0 = rest_lo(x0, w[0]);
                            x1 = rest_hi(x1, w[0]);
                            x2 = rest_lo(x2, w[1]);
                            x3 = rest_hi(x3, w[1]);
                        }
                    }
                }
            }
        }
    };

    if constexpr (!MULTI) {  // k positions between the sequence end and the end of the last k-step: zero, once
        const int nq = (T + 3) >> 2, nqz = 4 * chunk_kc(0) - nq;
        if (nqz > 0) {
            const int tot = 3 * (RS + CS) * nqz;
            for (int idx = tid; idx < tot; idx += NTHR) {
                const int pos = idx / nqz, q = nq + (idx - pos * nqz);  // pos walks [3][RS] then [3][CS]: contiguous rows
                const u32x2 z = {0u, 0u};
                *reinterpret_cast<u32x2*>(smem_b + pos * ROWB + q * 8) = z;
            }
        }
    }

#ifdef DPK_K1_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0, st_items = 0;
#endif
    Cur n1 = next(cur);
    Cur n2 = next(n1);
    auto standin_stores = [&](int n) {  // out-of-range: dropped by the range check, counted by vmcnt
        const auto rsN = make_rsrc(normsq_part, 0);
        for (int q = 0; q < n; ++q) {
            __builtin_amdgcn_raw_buffer_store_b32(0u, rsN, OOB, 0, 0);
            asm volatile("" ::: "memory");
        }
    };
    // the first two steps find the same number of younger operations behind their operands as every later one
    prefetch(rq[0], cur);
    standin_stores(SPLIT_VM_STORES);
    prefetch(rq[1], n1);
    standin_stores(SPLIT_VM_STORES);
    int parity = 0;
    f32x16 acc[4];
    float bsum = 0.f;
    // one step: stage its operands (fetched two steps ago), fetch the step two ahead into the freed registers,
    // contract; on a tile's last chunk store the unit and publish the tile's squared norm
    auto step = [&](f32x4 (&r)[NR][4]) {
        const int m0 = cur.tm * RS, n0 = cur.tn * CS;
        const int64_t b = cur.s * 8 + xcd;
#ifdef DPK_K1_STAMPS
        st_last = __builtin_amdgcn_s_memtime();
        ++st_items;
#if DPK_K1_STAMPS == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7);
#endif
#endif
        // ARES, step c = -1: the group's A slice.  Every wave has passed the previous tile's closing barrier, so nobody
        // reads A any more; the first G step's barrier publishes it.  Same stage / prefetch code, nothing else.
        const bool astep = ARES && cur.c < 0;
        if (!astep && cur.c == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
            bsum = 0.f;
        }
        split_wait_loads<SPLIT_VM_YOUNGER>(r);
#if defined(DPK_K1_STAMPS) && DPK_K1_STAMPS == 1
        STAMP(7);
#endif
        if (!(DPK_K1_ABL & 4)) stage(r, cur);
        STAMP(0);
        if (!astep) lds_only_barrier();
        STAMP(1);
        // the first k-step's operands are requested before the prefetch is issued: their LDS latency passes under the
        // prefetch's address arithmetic, and each later k-step's under the tail of the previous one's MFMAs
        const unsigned char* pg = sG + (mr * 32 + l31) * ROWB + lh * 16;
        const unsigned char* pa = sA + l31 * ROWA + ((ARES ? cur.c * (SPLIT_TC / 8) : 0) + lh) * 16;
        bf16x8 g[3], a[3][4];
        auto load_ops = [&]() {
#pragma unroll
            for (int p = 0; p < 3; ++p) g[p] = *reinterpret_cast<const bf16x8*>(pg + p * PS);
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[q][j] = *reinterpret_cast<const bf16x8*>(pa + q * PSA + j * 32 * ROWA);
            pg += 32;
            pa += 32;
        };
        if (!astep) load_ops();
        prefetch(r, n2);  // ahead of this tile's stores in the vector-memory pipe
        STAMP(2);
        if (!astep && !(DPK_K1_ABL & 2)) {
            const int kc = chunk_kc(cur.c);
            for (int ks = 0; ks < kc; ++ks) {
#pragma unroll
                for (int sum = 4; sum >= 0; --sum)  // smallest partial products first
#pragma unroll
                    for (int p = 2; p >= 0; --p) {
                        const int q = sum - p;
                        if (q < 0 || q > 2) continue;
                        if (TERMS == 8 && sum == 4) continue;
                        if (TERMS == 6 && sum > 2) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[p], a[q][j], acc[j], 0, 0, 0);
                    }
                if (ks + 1 < kc) load_ops();
            }
            if (lh == 0) {  // bias: feature 4*l31 + mr, the chunk's 4-step groups in t order
                const int nq = chunk_nq(cur.c);
                for (int q = 0; q < nq; ++q) bsum = add_rn(bsum, sBias[q * RS + 4 * l31 + mr]);
            }
        }
        STAMP(3);
        if (!astep && (!MULTI || cur.c == nch - 1)) {
            // stores: accumulator row slot m = (q&3) + 8*(q>>2) + 4*lh of MFMA block mr is feature m0 + 4*m + mr; rows
            // >= out_f fall outside the descriptor, columns >= ld_w get the out-of-range offset
            float sq = 0.f;
            const auto rsW = make_rsrc(jobs.W[cur.job] + b * gs_stride, (int64_t)out_f * ld_w * 4);
            const int col = n0 + 4 * l31;
            const unsigned row_bytes = (unsigned)ld_w * 4u;
            unsigned soff = col < ld_w ? (unsigned)((m0 + mr + 16 * lh) * ld_w + col) * 4u : OOB;
            if (DPK_K1_ABL & 1) soff = OOB;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 v = {acc[0][q], acc[1][q], acc[2][q], acc[3][q]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsW, soff, 0, DPK_K1_STORE_AUX);
                soff += ((q & 3) == 3 ? 20u : 4u) * row_bytes;  // slots 0 1 2 3 8 9 10 11 ... -> rows 4*slot
                sq = __builtin_fmaf(v[0], v[0], sq);
                sq = __builtin_fmaf(v[1], v[1], sq);
                sq = __builtin_fmaf(v[2], v[2], sq);
                sq = __builtin_fmaf(v[3], v[3], sq);
            }
            {
                const float* GSb = jobs.Bi[cur.job];
                const bool has_bias = GSb != nullptr && cur.tn == 0;
                const auto rsB = make_rsrc(GSb + b * gs_stride, has_bias ? (int64_t)out_f * 4 : 0);
                const int f = m0 + 4 * l31 + mr;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rsB,
                                                      lh == 0 ? (unsigned)f * 4u : OOB, 0, 0);
                if (has_bias && lh == 0 && f < out_f) sq = __builtin_fmaf(bsum, bsum, sq);
            }
            STAMP(4);
            sq = wave_sum(sq);
            if (lane == 0) red[parity * NW + wave] = sq;
            lds_only_barrier();  // every wave is done with this tile's images; red[parity] is complete
            {
                float tot = 0.f;
#pragma unroll
                for (int i = 0; i < NW; ++i) tot += red[parity * NW + i];
                const auto rsP = make_rsrc(normsq_part + b * part_stride, (int64_t)part_stride * 4);
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(unsigned, tot), rsP,
                    tid == 0 ? (unsigned)(jobs.slot0[cur.job] + cur.tm * tiles_n + cur.tn) * 4u : OOB, 0, 0);
            }
            parity ^= 1;
        } else {
            // a step that stores nothing (an A step, a chunk that is not the tile's last) issues as many out-of-range
            // 4-byte stores as a storing step has stores: the hand-counted wait above needs one count for all paths
            standin_stores(SPLIT_VM_STORES - 1);
            if (!astep) lds_only_barrier();  // all waves done with this chunk's images
            standin_stores(1);
        }
        STAMP(5);
        cur = n1;
        n1 = n2;
        n2 = next(n2);
        STAMP(6);
    };
    for (;;) {
        step(rq[0]);
        if (cur.job >= jobs.n) break;
        step(rq[1]);
        if (cur.job >= jobs.n) break;
    }
#ifdef DPK_K1_STAMPS
    if (blockIdx.x < 64 && lane == 0) {
        for (int i = 0; i < 8; ++i) dpk_stamps[(blockIdx.x * 8 + wave) * 9 + i] = st_acc[i];
        dpk_stamps[(blockIdx.x * 8 + wave) * 9 + 8] = st_items;
    }
#endif
}

// ---- role-split variant (round 3): loaders and contractors are different waves -------------------------------------
// What bounded linear_gs_split_kernel at 32 < T <= 48 (ablations on the MI355X, 250 -> 1000, B = 4096: everything 1146 us;
// MFMAs and stores with NO wait on an operand load 791; the same with the waits 1046): a CU's vector-memory pipe is one
// in-order queue.  A tile's 64 KB of rows take ~5k cycles to leave at the CU's share of the HBM rate, whoever issues a
// vector-memory instruction meanwhile waits for room in the queue, and an operand load sits behind whatever rows were
// queued before it.  A wave that loads, splits, contracts AND stores is therefore a serial chain "rows out -> my loads
// come back -> split -> contract", and two such chains per CU left the queue empty about a third of the time.
//
// Here ONE workgroup of 12 waves owns the CU and the roles are split (3 waves per SIMD, <= 168 registers):
//   * waves 0-3 and 4-7 are two CONTRACTOR groups; a group owns a 128 x 128 tile (wave = 32 x 128 unit, as above) and
//     alternates phase by phase between contracting its next tile (LDS reads + bf16-pipe MFMAs, nothing else) and storing
//     the tile it has just contracted (16 row stores per wave + bias + squared norm).  The groups are one phase apart: in
//     every phase exactly one of them is pushing rows into the memory queue, so the queue always holds a tile's rows;
//   * waves 8-11 are LOADERS: they fetch the operands two phases ahead (they never store a row, so nothing of theirs
//     waits behind rows of their own), split them into the three bf16 images and stage them:
//     during phase n the G slice of tile n+1 goes into the image buffer of the group that is storing (its contraction
//     of tile n-1 is over).  The A slice of a (sample, column tile) group of row tiles stays resident, as in MODE 2;
//     it is restaged in one extra phase per group of tiles (nobody contracts then; the storing group keeps the queue fed).
//   * one LDS-only barrier of all 12 waves per phase.
// Tile n of a workgroup is row tile n % tiles_m of its (n / tiles_m)-th (job, sample, column tile) group; groups are
// dealt round-robin to the workgroups of the XCD that owns the sample (b mod 8).  Arithmetic per output element, k
// order and bias pairing are those of linear_gs_split_kernel: results are bit-identical to it.
constexpr int WS_THREADS = 768;
constexpr int WS_PROD0 = 512;            // first loader thread
constexpr int WS_ROWCH = 7;              // 16-byte chunks per image row: 48 steps of bf16 = 6, +1 (odd: conflict-free)
constexpr int WS_ROWB = WS_ROWCH * 16;   // 112 bytes
constexpr int WS_PS = 128 * WS_ROWB;     // piece stride
constexpr int WS_IMG = 3 * WS_PS;        // one operand slice as three bf16 images: 43,008 bytes
constexpr int WS_KC = 3;                 // MFMA k-steps of 16: 32 < T <= 48
constexpr int WS_NQ = 4 * WS_KC;         // 4-step groups of a 48-step (k-step padded) sequence
constexpr int WS_NH = 2 * WS_NQ;         // 2-step groups
constexpr int WS_LDS = 3 * WS_IMG + 2 * WS_NH * 128 * 4 + 2 * 4 * 4;  // A, G[2], bias partials[2], red[2][4]
// The loaders' operand registers are PRIVATE: the kernel is compiled with amdgpu_num_vgpr(WS_COMPILER_VGPRS), so the
// compiler allocates v0 .. v119 only, and v120 .. v167 (two sets of six 16-byte quads per lane) are named literally in
// inline assembly: loads land there, an explicit `s_waitcnt vmcnt(N)` releases them, and v_mov copies hand their values to
// compiler-visible registers.  (Loads into compiler-allocated registers, tracked or not, did not survive: tracked, the
// compiler's wait is the minimum over every path through the loop and came out as "everything but the last store" on
// one path or another, however the loop was rotated; untracked, register copies at the loop latch read registers whose
// loads were still in flight.  The workgroup's 12 waves get 168 registers each either way.)
// The attribute's value is HALF the compiler's budget on this unified-register-file target (LLVM doubles it, and ignores
// a value whose double exceeds what the launch bounds allow — 120 was silently ignored): 60 = v0 .. v119.
// tools/check_k1_vmcnt.py (run by tests/test_build_checks.py) verifies on the ISA that nothing outside the inline
// assembly names a private register and that no kernel with hand-counted waits spills.
#define WS_COMPILER_VGPRS 60
// (the clobber lists name registers above the compiler's budget on purpose: they are what makes the kernel descriptor
// allocate them; clang's note that they are "reserved" is expected)
#pragma clang diagnostic ignored "-Winline-asm"
#define DPK_WS_QUAD(NAME, A, B, C, D)                                                                           \
    __device__ __forceinline__ void ws_ld_##NAME(const u32x4 rs, unsigned off) {                                \
        asm volatile("buffer_load_dwordx4 v[" #A ":" #D "], %0, %1, 0 offen ; DPK_PRIVLOAD"                      \
                     :                                                                                          \
                     : "v"(off), "s"(rs)                                                                         \
                     : "memory", "v" #A, "v" #B, "v" #C, "v" #D);                                                \
    }                                                                                                           \
    __device__ __forceinline__ f32x4 ws_rd_##NAME() {                                                           \
        float a, b, c, d;                                                                                       \
        asm volatile("v_mov_b32 %0, v" #A "\n\tv_mov_b32 %1, v" #B "\n\tv_mov_b32 %2, v" #C "\n\tv_mov_b32 %3, v" #D \
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                        \
                     :                                                                                          \
                     : "memory");                                                                               \
        return f32x4{a, b, c, d};                                                                               \
    }
DPK_WS_QUAD(s0q0, 120, 121, 122, 123)
DPK_WS_QUAD(s0q1, 124, 125, 126, 127)
DPK_WS_QUAD(s0q2, 128, 129, 130, 131)
DPK_WS_QUAD(s0q3, 132, 133, 134, 135)
DPK_WS_QUAD(s0q4, 136, 137, 138, 139)
DPK_WS_QUAD(s0q5, 140, 141, 142, 143)
DPK_WS_QUAD(s1q0, 144, 145, 146, 147)
DPK_WS_QUAD(s1q1, 148, 149, 150, 151)
DPK_WS_QUAD(s1q2, 152, 153, 154, 155)
DPK_WS_QUAD(s1q3, 156, 157, 158, 159)
DPK_WS_QUAD(s1q4, 160, 161, 162, 163)
DPK_WS_QUAD(s1q5, 164, 165, 166, 167)
#undef DPK_WS_QUAD
template <int SET>
__device__ __forceinline__ void ws_load_set(const u32x4 rs, unsigned off, unsigned rowb) {
    // quad 2k / 2k+1: steps 2*hq and 2*hq + 1 of step pair hq = (ptid >> 5) + 8k
    if constexpr (SET == 0) {
        ws_ld_s0q0(rs, off); ws_ld_s0q1(rs, off + rowb); off += 16u * rowb;
        ws_ld_s0q2(rs, off); ws_ld_s0q3(rs, off + rowb); off += 16u * rowb;
        ws_ld_s0q4(rs, off); ws_ld_s0q5(rs, off + rowb);
    } else {
        ws_ld_s1q0(rs, off); ws_ld_s1q1(rs, off + rowb); off += 16u * rowb;
        ws_ld_s1q2(rs, off); ws_ld_s1q3(rs, off + rowb); off += 16u * rowb;
        ws_ld_s1q4(rs, off); ws_ld_s1q5(rs, off + rowb);
    }
}
template <int SET>
__device__ __forceinline__ void ws_read_set(f32x4 (&r)[3][2]) {
    if constexpr (SET == 0) {
        r[0][0] = ws_rd_s0q0(); r[0][1] = ws_rd_s0q1(); r[1][0] = ws_rd_s0q2();
        r[1][1] = ws_rd_s0q3(); r[2][0] = ws_rd_s0q4(); r[2][1] = ws_rd_s0q5();
    } else {
        r[0][0] = ws_rd_s1q0(); r[0][1] = ws_rd_s1q1(); r[1][0] = ws_rd_s1q2();
        r[1][1] = ws_rd_s1q3(); r[2][0] = ws_rd_s1q4(); r[2][1] = ws_rd_s1q5();
    }
}
// a loader wave's vector-memory instructions per phase: 6 operand loads, then 1 store (the squared-norm slot); a set is
// staged two phases after it was requested: the slot store of its own phase and the whole next phase are younger
constexpr int WS_VM_YOUNGER = 1 + 6 + 1;

template <int TERMS, bool SWAP>
__global__ void __launch_bounds__(WS_THREADS) __attribute__((amdgpu_num_vgpr(WS_COMPILER_VGPRS)))
linear_gs_ws_kernel(
    const K1Jobs jobs, float* __restrict__ normsq_part, int64_t part_stride, int B, int T, int in_f, int out_f,
    int64_t gs_stride, int ld_w, int ld_g, float g_scale, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // A tile group is the tiles that share one RESIDENT slice: the row tiles of a (sample, column tile) with A resident
    // (!SWAP), or — for layers with fewer than 3 row tiles — the column tiles of a (sample, row tile) with G resident
    // (SWAP).  The other operand is STREAMED: one slice per tile.
    unsigned char* sR = smem_b;                           // [3][128][ROWB]   resident slice of the current tile group
    unsigned char* sS = smem_b + WS_IMG;                  // [2][3][128][ROWB] streamed slice of tile n in buffer n & 1
    float* sBias = reinterpret_cast<float*>(smem_b + 3 * WS_IMG);   // [2][WS_NH][128]: sums of 2 steps
    float* red = sBias + 2 * WS_NH * 128;                           // [2][4]

    const int xcd = blockIdx.x & 7;
    const unsigned rank = blockIdx.x >> 3;
    const unsigned nbx = gridDim.x >> 3;
    const unsigned nsx = (unsigned)((B - xcd + 7) / 8);
    const int tiles_in = SWAP ? tiles_n : tiles_m, tiles_out = SWAP ? tiles_m : tiles_n;  // per group / groups per sample
    const unsigned per_job = nsx * (unsigned)tiles_out;
    const unsigned total_sg = per_job * (unsigned)jobs.n;   // (job, sample, column tile) groups of this XCD
    if (rank >= total_sg) return;                            // uniform
    const int N = (int)((total_sg - rank + nbx - 1) / nbx) * tiles_in;   // tiles of this workgroup

    // Tile n of this workgroup, as an iterator: the tile bookkeeping runs on the CU's ONE scalar unit for all 12 waves
    // (with a division-based "tile n -> (job, sample, row tile, column tile)" per use, a phase cost 150-360 scalar
    // instructions per wave: 2.2k cycles per phase with nothing else in it), so stepping costs a few adds and compares
    // and the two divisions run once per tile group.
    struct Tile {
        int n, job, s, tm, tn;
        unsigned U;   // index of the (job, sample, column tile) group among the XCD's
        bool ok;
    };
    auto group_of = [&](Tile& x) {
        x.ok = x.U < total_sg;
        const unsigned U = x.ok ? x.U : 0u;
        x.job = (int)(U / per_job);
        const unsigned r = U - (unsigned)x.job * per_job;
        x.s = (int)(r / (unsigned)tiles_out);
        const int outer = (int)(r - (unsigned)x.s * (unsigned)tiles_out);
        if (SWAP) x.tm = outer;
        else x.tn = outer;
    };
    auto first_tile = [&]() {
        Tile x;
        x.n = 0;
        x.tm = x.tn = 0;
        x.U = rank;
        group_of(x);
        return x;
    };
    auto advance = [&](Tile& x) {
        ++x.n;
        int& inner = SWAP ? x.tn : x.tm;
        if (++inner == tiles_in) {
            inner = 0;
            x.U += nbx;
            group_of(x);
        }
    };
    const Tile no_tile = {-1, 0, 0, 0, 0, 0u, false};

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    if (wave < 8) {
        // ================================ contractors ================================
        const int g = wave >> 2, mr = wave & 3;
        unsigned char* myS = sS + g * WS_IMG;
        const float* myBias = sBias + (SWAP ? 0 : g) * WS_NH * 128;  // the bias partials live with the G slice
        f32x16 acc[4];
        float bsum = 0.f;

        auto contract = [&](const Tile&) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
            // Column block by column block (the 6 products of one accumulator back to back, smallest first: the order of
            // every accumulator's own chain is that of linear_gs_split_kernel), so that only one block's A operands and
            // the next block's are live: 48 operand registers instead of 60 + 60, which is what fits 168.
            const unsigned char* pg = (SWAP ? sR : myS) + (mr * 32 + l31) * WS_ROWB + lh * 16;
            const unsigned char* pa = (SWAP ? myS : sR) + l31 * WS_ROWB + lh * 16;
            bf16x8 gv[2][3], av[2][3];
            auto load_g = [&](bf16x8 (&d)[3], int ks) {
#pragma unroll
                for (int p = 0; p < 3; ++p) d[p] = *reinterpret_cast<const bf16x8*>(pg + p * WS_PS + ks * 32);
            };
            auto load_a = [&](bf16x8 (&d)[3], int ks, int j) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    d[q] = *reinterpret_cast<const bf16x8*>(pa + q * WS_PS + j * 32 * WS_ROWB + ks * 32);
            };
            load_g(gv[0], 0);
            load_a(av[0], 0, 0);
#pragma unroll
            for (int ks = 0; ks < WS_KC; ++ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cur = (ks * 4 + j) & 1;
                    if (j < 3) {
                        load_a(av[cur ^ 1], ks, j + 1);
                    } else if (ks + 1 < WS_KC) {
                        load_g(gv[(ks + 1) & 1], ks + 1);
                        load_a(av[cur ^ 1], ks + 1, 0);
                    }
#pragma unroll
                    for (int sum = 4; sum >= 0; --sum)  // smallest partial products first
#pragma unroll
                        for (int p = 2; p >= 0; --p) {
                            const int q = sum - p;
                            if (q < 0 || q > 2) continue;
                            if (TERMS == 8 && sum == 4) continue;
                            if (TERMS == 6 && sum > 2) continue;
                            if (!(DPK_K1_ABL & 2))
                                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gv[ks & 1][p], av[cur][q], acc[j], 0, 0, 0);
                        }
                }
            }
            bsum = 0.f;
            if (lh == 0) {  // bias: feature 4*l31 + mr, the 4-step groups in t order
#pragma unroll
                for (int q = 0; q < WS_NQ; ++q)  // (steps 4q, 4q+1) + (4q+2, 4q+3), as the 4-step sums of the other kernels
                    bsum = add_rn(bsum, add_rn(myBias[(2 * q) * 128 + 4 * l31 + mr], myBias[(2 * q + 1) * 128 + 4 * l31 + mr]));
            }
        };
        // A tile's rows leave in two halves (accumulator registers 0-7, then 8-15 + bias + squared norm): normally back
        // to back; where an extra phase (A slice of a new tile group) sits before the next contraction, one half in the
        // extra phase and one beside that contraction, so that both phases have rows in the memory queue.
        float sq = 0.f;
        auto store_half = [&](const Tile& x, int half) {
            const int m0 = x.tm * 128, n0 = x.tn * 128;
            const int64_t b = x.s * 8 + xcd;
            const auto rsW = make_rsrc(jobs.W[x.job] + b * gs_stride, (int64_t)out_f * ld_w * 4);
            const int col = n0 + 4 * l31;
            const unsigned row_bytes = (unsigned)ld_w * 4u;
            // slots 0 1 2 3 8 9 10 11 | 16 ... -> rows 4*slot: the second half starts 64 rows further
            unsigned soff = col < ld_w ? (unsigned)((m0 + mr + 16 * lh + 64 * half) * ld_w + col) * 4u : OOB;
            if (DPK_K1_ABL & 1) soff = OOB;
            if (half == 0) sq = 0.f;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                f32x4 v;
                if (half == 0) v = f32x4{acc[0][q8], acc[1][q8], acc[2][q8], acc[3][q8]};
                else v = f32x4{acc[0][q8 + 8], acc[1][q8 + 8], acc[2][q8 + 8], acc[3][q8 + 8]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsW, soff, 0, DPK_K1_STORE_AUX);
                soff += ((q8 & 3) == 3 ? 20u : 4u) * row_bytes;
                sq = __builtin_fmaf(v[0], v[0], sq);
                sq = __builtin_fmaf(v[1], v[1], sq);
                sq = __builtin_fmaf(v[2], v[2], sq);
                sq = __builtin_fmaf(v[3], v[3], sq);
            }
            if (half == 1) {
                const float* GSb = jobs.Bi[x.job];
                const bool has_bias = GSb != nullptr && x.tn == 0;
                const auto rsB = make_rsrc(GSb + b * gs_stride, has_bias ? (int64_t)out_f * 4 : 0);
                const int f = m0 + 4 * l31 + mr;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rsB,
                                                      lh == 0 ? (unsigned)f * 4u : OOB, 0, 0);
                if (has_bias && lh == 0 && f < out_f) sq = __builtin_fmaf(bsum, bsum, sq);
                const float tot = wave_sum(sq);
                if (lane == 0) red[g * 4 + mr] = tot;
            }
        };

        lds_only_barrier();  // the loaders' prologue: A slice of the first tile group and G slice of tile 0 are staged
        Tile tc = first_tile(), ts = no_tile;  // tile n (contracted in phase n) and tile n-1 (stored in phase n)
        for (int n = 0; n <= N + 1; ++n) {
            bool first_half_out = false;
            if (n > 0 && tc.ok && (SWAP ? tc.tn : tc.tm) == 0) {  // a new tile group: its resident slice has a phase of its own
                if (((n - 1) & 1) == g) store_half(ts, 0);
                first_half_out = true;
                lds_only_barrier();
            }
            if ((n & 1) == g) {
                if (tc.ok) contract(tc);
            } else if (ts.ok) {
                if (!first_half_out) store_half(ts, 0);
                store_half(ts, 1);
            }
            lds_only_barrier();
            ts = tc;
            advance(tc);
        }
        return;
    }

    // =================================== loaders ===================================
    // One slice (G of a tile, or A of a tile group) is staged per phase, from private registers (above) requested two
    // phases earlier.  Every phase runs the SAME vector-memory stream — 6 loads, 1 store (the squared-norm slot; out of
    // range for all lanes but one) — whether it stages a G or an A slice or nothing, so one hand-counted wait,
    // vmcnt(WS_VM_YOUNGER), releases a set on every path.  Why the count matters: a load requested in phase n sits in the
    // CU's memory queue behind the rows the storing group has queued and comes back about a phase later; a wait that also
    // covered the set requested one phase ago made the loaders start staging at the END of their phase (measured with such
    // a wait: 1025 us against 742 with the staging skipped).
    const int ptid = tid - WS_PROD0;
    // (s_setprio 2 for the loaders — each SIMD's youngest wave, last in line at equal priority — was measured slower:
    // 956 / 1824 us against 948 / 1765 at 250 -> 1000 / 500 -> 1000, B = 4096)
    const int ngroups = N / tiles_in;
    const int P = N + 2 + (ngroups - 1);  // phases: one per tile, one extra per change of tile group, two to drain
    struct Slice {
        Tile t;
        bool streamed;   // the streamed slice of tile t (staged into buffer t.n & 1), else the resident slice of t's group
    };
    // The loaders' slices in phase order: G of tile 1, 2, ..., G of the first tile of the next tile group, that group's A
    // slice (its own, extra phase), G of the group's second tile, ...
    auto next_slice = [&](Slice& x) {
        if (x.streamed && x.t.ok && (SWAP ? x.t.tn : x.t.tm) == 0) {
            x.streamed = false;  // same tile: its group's resident slice
        } else {
            x.streamed = true;
            advance(x.t);
        }
    };
    // A slice (48 steps x 128 columns) is 768 tasks of 2 steps x 4 columns: three per loader lane, 6 x 16 bytes in flight
    // per lane and set.  Lane l of loader wave w: column group cgp = (l & 7) + 8 w, step pairs hq = (l >> 3) + 8 k,
    // k = 0..2.  The 32 lanes of an LDS-write group then hit 32 different banks (image rows are 28 dwords apart: 8 column
    // groups x 4 step pairs; with 32 column groups per group it was 4 lanes per bank), and a load instruction still reads
    // whole 128-byte lines (8 column groups of a row).
    const int pcg = (ptid & 7) + 8 * (ptid >> 6), phq = (ptid >> 3) & 7;
    auto request = [&](auto set, const Slice& x) {
        const int64_t b = x.t.s * 8 + xcd;
        const int job = x.t.ok ? x.t.job : 0;
        const bool isG = x.streamed != SWAP;
        const float* base = isG ? jobs.G[job] + b * T * ld_g : jobs.A[job] + b * T * in_f;
        // rows at or past T lie beyond the descriptor: zeros; a slice past the last one fetches nothing
        const int64_t bytes = !x.t.ok ? 0 : isG ? ((int64_t)(T - 1) * ld_g + out_f) * 4 : (int64_t)T * in_f * 4;
        const u32x4 rs = rsrc_words(base, bytes);
        const int ld = isG ? ld_g : in_f;
        const int dim = isG ? out_f : in_f;
        const int col = (isG ? x.t.tm : x.t.tn) * 128 + pcg * 4;
        const unsigned off = col < dim ? (unsigned)((2 * phq) * ld + col) * 4u : OOB;
        ws_load_set<decltype(set)::value>(rs, off, (unsigned)ld * 4u);  // an out-of-range offset stays out of range
    };
    auto stage = [&](const f32x4 (&r)[3][2], const Slice& x) {
        const int cgp = pcg;
        const bool isG = x.streamed != SWAP;
        const int left = isG ? out_f - (x.t.tm * 128 + cgp * 4) : in_f - (x.t.tn * 128 + cgp * 4);
        const float scale = isG ? g_scale : 1.0f;
        // column 4*cgp+e of the tile -> position e*32 + cgp (slot cgp of MFMA block e), for G and for A; step pair hq
        // is bytes 4*hq .. 4*hq+3 of the row
        const int gb = x.t.n & 1;
        unsigned char* dst = (x.streamed ? sS + gb * WS_IMG : sR) + cgp * WS_ROWB + phq * 4;
        float* bdst = sBias + (SWAP ? 0 : gb) * WS_NH * 128 + phq * 128 + cgp * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (DPK_K1_ABL & 4) break;
            const f32x4 v0 = scale4_rn(mask4(r[k][0], left), scale), v1 = scale4_rn(mask4(r[k][1], left), scale);
            if (isG) *reinterpret_cast<f32x4*>(bdst + k * 8 * 128) = add4_rn(v0, v1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x0 = v0[e], x1 = v1[e];
                unsigned char* d = dst + k * 32 + e * 32 * WS_ROWB;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned w = p == 0 ? pack_bf16_first(x0, x1) : pack_bf16_rn(x0, x1);
                    *reinterpret_cast<unsigned*>(d + p * WS_PS) = w;
                    if (p < 2) {
                        x0 = rest_lo(x0, w);
                        x1 = rest_hi(x1, w);
                    }
                }
            }
        }
    };
    auto publish = [&](const Tile& x) {  // squared norm of a stored tile: its four waves' sums, in wave order
        const int gsrc = x.n & 1;
        const float tot = ((red[gsrc * 4 + 0] + red[gsrc * 4 + 1]) + red[gsrc * 4 + 2]) + red[gsrc * 4 + 3];
        const int64_t b = x.s * 8 + xcd;
        const auto rsP = make_rsrc(normsq_part + b * part_stride, x.ok ? (int64_t)part_stride * 4 : 0);
        __builtin_amdgcn_raw_buffer_store_b32(  // ONE store instruction per wave and phase on every path (see above)
            __builtin_bit_cast(unsigned, tot), rsP,
            (x.ok && ptid == 0) ? (unsigned)(jobs.slot0[x.job] + x.tm * tiles_n + x.tn) * 4u : OOB, 0, 0);
        asm volatile("" ::: "memory");
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    auto wait_younger = [](auto n) { asm volatile("s_waitcnt vmcnt(%0) ; DPK_PRIVWAIT" ::"n"(decltype(n)::value) : "memory"); };

    Slice st, rq;            // the slice staged in this phase, the slice requested in it (staged two phases on)
    Tile pub1 = no_tile, pub2 = no_tile;   // the tiles contracted one and two contraction phases ago
    {   // prologue: the first tile group's A slice and tile 0's G slice staged; the slices of phases 0 and 1 requested
        Slice a0, g0;
        a0.t = g0.t = first_tile();
        a0.streamed = false;   // (a0: the resident slice, g0: the streamed one — A and G when !SWAP)
        g0.streamed = true;
        f32x4 r[3][2];
        request(S0, g0);
        request(S1, a0);
        wait_younger(std::integral_constant<int, 0>{});
        ws_read_set<1>(r);
        stage(r, a0);
        ws_read_set<0>(r);
        stage(r, g0);
        st = g0;
        st.streamed = false;  // (so that the step below moves on to tile 1 instead of to tile 0's resident slice)
        next_slice(st);  // phase 0 stages G of tile 1
        rq = st;
        request(S0, rq);
        publish(no_tile);  // stand-ins for a phase's slot store: the first two phases then count as every later one
        next_slice(rq);
        request(S1, rq);
        publish(no_tile);
        next_slice(rq);      // phase 0 requests what phase 2 stages
        lds_only_barrier();
    }
    Tile cur = first_tile();  // the tile contracted in the current contraction phase
    auto phase = [&](auto set) {
        f32x4 r[3][2];
        wait_younger(std::integral_constant<int, WS_VM_YOUNGER>{});
        ws_read_set<decltype(set)::value>(r);
        // Unconditional: past the last tile the registers hold zeros (empty descriptor) and go into the buffer of a group
        // that has nothing left to contract.  (Requesting BEFORE staging, for 0.6 us more lead, was slower: 968 / 1970 us
        // against 934 / 1772 at 250 -> 1000 / 500 -> 1000.)
        stage(r, st);
        request(set, rq);
        const bool extra = !st.streamed;   // a resident slice is staged in the extra phase before its group's first tile
        publish(extra ? no_tile : pub2);
        lds_only_barrier();
        if (!extra) {
            pub2 = pub1;
            pub1 = cur;
            advance(cur);
        }
        next_slice(st);
        next_slice(rq);
    };
    for (int ph = 0; ph < P; ph += 2) {
        phase(S0);
        if (ph + 1 < P) phase(S1);
    }
}

// ---- the kernel for LONG sequences (T > 48: cfg5's T = 256) --------------------------------------------------------------
// At T = 256 the contraction has 128 flop per output byte: K1 is bound by the matrix pipe and by the vector pipe that
// splits the operands for it.  What tools/lab/coexec.hip and cycle stamps showed on this chip:
//   * v_mfma_f32_32x32x16_bf16 issues every 32 cycles into independent accumulators, every 44 into the same one;
//   * beside a wave that keeps the matrix pipe full, the SIMD's other waves together get ONE vector instruction per
//     6 - 7 cycles (about five per MFMA), and one wave alone issues one per ~5 cycles: the split's ~7 instructions per
//     operand element are as expensive as the MFMAs they feed unless every staged element is reused widely;
//   * an in-order wave that splits first and contracts afterwards idles one pipe, then the other.
// linear_gs_split_kernel MODE 1 (128 x 128 tiles, every wave splits both slices of its tile, then contracts) kept the
// matrix pipe 32 % busy (2048 x 2048, B = 2048: 32.6 ms).  A 12-wave version with 256 x 128 super-tiles (three staging
// tasks per 48 MFMAs and SIMD: 350 vector instructions where 240 fit) reached 27.6 ms whichever waves did the staging.
// This kernel: one workgroup of 8 waves (2 per SIMD, 256 registers each) owns a 256 x 256 SUPER-TILE — row tiles
// 2 tm2, 2 tm2 + 1 times column tiles 2 tn2, 2 tn2 + 1 — so a phase (one 16-step k-step) stages four slice chunks for
// sixteen tiles' MFMAs: two tasks (2 steps x 4 columns each) per lane and 48 MFMAs per wave.  Wave w owns rows
// 32 (w & 3) .. of row tile w >> 2, all 256 columns (128 accumulator registers).  In a phase every wave
//   contracts k-step c from image buffer c & 1 and splits its two tasks of k-step c + 1 (its own row tile's G rows,
//   column tile w >> 2's A columns; loaded one phase ago into private registers under a hand-counted wait) into the
//   other buffer, in four chunks of 12 MFMAs + half a task, the SIMD partners in opposite order (see the phase body);
//   requests k-step c + 2; stores its rows after the super-tile's last k-step; one barrier.
// Arithmetic, k order and bias pairing are linear_gs_split_kernel's: results are bit-identical.  An odd T / 16 contracts
// one k-step of zeros (phases come in pairs so that the buffer index is the phase's parity).
struct K1Products {   // the partial products (G piece p, A piece q) of one k-step in contraction order: smallest first
    int n;
    int p[9], q[9];
};
constexpr K1Products k1_products(int terms) {
    K1Products L{};
    for (int sum = 4; sum >= 0; --sum)
        for (int p = 2; p >= 0; --p) {
            const int q = sum - p;
            if (q < 0 || q > 2) continue;
            if (terms == 8 && sum == 4) continue;
            if (terms == 6 && sum > 2) continue;
            L.p[L.n] = p;
            L.q[L.n] = q;
            ++L.n;
        }
    return L;
}
constexpr int WL_THREADS = 512;
constexpr int WL_ROWB = 48;                 // image row: 16 steps of bf16 (32 B) + 16 B (3 chunks: odd, conflict-free)
constexpr int WL_PS = 128 * WL_ROWB;        // piece stride
constexpr int WL_IMG = 3 * WL_PS;           // one slice chunk as three bf16 images: 18,432 bytes
constexpr int WL_BUF = 4 * WL_IMG;          // G rows of row tile 0 / 1, A columns of column tile 0 / 1
constexpr int WL_BIAS = 2 * 4 * 128;        // floats per buffer: bias partials [row tile][4-step group][feature]
constexpr int WL_LDS = 2 * WL_BUF + 2 * WL_BIAS * 4 + 2 * 8 * 4;   // + red[column tile][wave]: 155,712 bytes
constexpr int WL_STORE_VM = 33;             // vector-memory instructions of one wave's store(): 2 x 16 rows + the bias
static_assert(WL_STORE_VM + 1 < 64, "vmcnt is a 6-bit counter");

// A lane's in-flight operands: a pair of G rows and a pair of A rows, in v240 .. v255 (the compiler allocates v0 .. v239:
// the attribute's value is half the budget).  They are requested one phase (~1.5 us) before they are taken; a second set
// for a two-phase lead does not fit beside 128 accumulators.
#define WL_COMPILER_VGPRS 120
#define DPK_WL_QUAD(NAME, A, B, C, D)                                                                           \
    __device__ __forceinline__ void wl_ld_##NAME(const u32x4 rs, unsigned off) {                                \
        asm volatile("buffer_load_dwordx4 v[" #A ":" #D "], %0, %1, 0 offen ; DPK_PRIVLOAD"                      \
                     :                                                                                          \
                     : "v"(off), "s"(rs)                                                                         \
                     : "memory", "v" #A, "v" #B, "v" #C, "v" #D);                                                \
    }                                                                                                           \
    __device__ __forceinline__ f32x4 wl_rd_##NAME() {                                                           \
        float a, b, c, d;                                                                                       \
        asm volatile("v_mov_b32 %0, v" #A "\n\tv_mov_b32 %1, v" #B "\n\tv_mov_b32 %2, v" #C "\n\tv_mov_b32 %3, v" #D \
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                        \
                     :                                                                                          \
                     : "memory");                                                                               \
        return f32x4{a, b, c, d};                                                                               \
    }                                                                                                           \
    __device__ __forceinline__ f32x4 wl_rd_scaled_##NAME(float s) { /* the quad times s, rounded once (v_mul_f32) */ \
        float a, b, c, d;                                                                                       \
        asm volatile("v_mul_f32 %0, %4, v" #A "\n\tv_mul_f32 %1, %4, v" #B "\n\tv_mul_f32 %2, %4, v" #C         \
                     "\n\tv_mul_f32 %3, %4, v" #D                                                               \
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                                                    \
                     : "v"(s)                                                                                   \
                     : "memory");                                                                               \
        return f32x4{a, b, c, d};                                                                               \
    }
DPK_WL_QUAD(g0, 240, 241, 242, 243)
DPK_WL_QUAD(g1, 244, 245, 246, 247)
DPK_WL_QUAD(a0, 248, 249, 250, 251)
DPK_WL_QUAD(a1, 252, 253, 254, 255)
#undef DPK_WL_QUAD

// RAGGED: in_f or out_f is not a multiple of 4, so a lane's four columns can straddle the edge and are masked one by
// one; otherwise a column group is inside (loaded) or outside (out-of-range offset: zeros) as a whole.
template <int TERMS, bool RAGGED>
__global__ void __launch_bounds__(WL_THREADS) __attribute__((amdgpu_num_vgpr(WL_COMPILER_VGPRS)))
linear_gs_wsl_kernel(const K1Jobs jobs, float* __restrict__ normsq_part, int64_t part_stride, int B, int T, int in_f,
                     int out_f, int64_t gs_stride, int ld_w, int ld_g, float g_scale, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* sBias = reinterpret_cast<float*>(smem_b + 2 * WL_BUF);   // [2 buffers][2 row tiles][4 groups][128]
    float* red = sBias + 2 * WL_BIAS;                               // [2 column tiles][8 waves]

    const int xcd = blockIdx.x & 7;
    const unsigned rank = blockIdx.x >> 3;
    const unsigned nbx = gridDim.x >> 3;
    const unsigned nsx = (unsigned)((B - xcd + 7) / 8);
    const int tiles_m2 = (tiles_m + 1) >> 1, tiles_n2 = (tiles_n + 1) >> 1;
    const unsigned per_sample = (unsigned)(tiles_m2 * tiles_n2), per_job = nsx * per_sample;
    const unsigned total = per_job * (unsigned)jobs.n;              // super-tiles of this XCD
    if (rank >= total) return;                                      // uniform
    const int ntile = (int)((total - rank + nbx - 1) / nbx);        // super-tiles of this workgroup
    const int nch = (((T + 15) >> 4) + 1) & ~1;                     // phases per super-tile: even, >= 4

    struct Tile {
        int job, s, tm2, tn2;
        unsigned it;
        const float* Gs;   // the sample's G rows and A rows (wave-uniform: scalar registers)
        const float* As;
        int slot0;
    };
    auto locate = [&](unsigned it) {   // once per super-tile (every nch phases)
        Tile x;
        x.it = it;
        const unsigned u = it < total ? it : 0u;
        x.job = (int)(u / per_job);
        unsigned r = u - (unsigned)x.job * per_job;
        x.s = (int)(r / per_sample);
        r -= (unsigned)x.s * per_sample;
        x.tm2 = (int)(r / (unsigned)tiles_n2);
        x.tn2 = (int)(r - (unsigned)x.tm2 * (unsigned)tiles_n2);
        const int64_t b = x.s * 8 + xcd;
        x.Gs = jobs.G[x.job] + b * T * ld_g;
        x.As = jobs.A[x.job] + b * T * in_f;
        x.slot0 = jobs.slot0[x.job];
        return x;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int h = wave >> 2, mr = wave & 3;   // this wave's row tile (and the column tile whose A it stages), MFMA row block
    // staging tasks: lane l of wave w: column group pcg = (l & 7) + 8 (w & 3), step pair phq = (l >> 3) & 7 of row tile /
    // column tile h (conflict-free LDS writes; a load instruction covers whole 128-byte lines)
    const int pcg = (lane & 7) + 8 * mr, phq = (lane >> 3) & 7;

    // k-step c of super-tile x into the private registers: steps 16c + 2 phq (+1); G columns (2 tm2 + h) 128 + 4 pcg ..,
    // A columns (2 tn2 + h) 128 + 4 pcg ..  Rows at or past T, and a super-tile past the last, lie beyond the descriptor.
    auto request_g = [&](const Tile& x, int c) {
        const u32x4 rsG = rsrc_words(x.Gs, x.it < total ? ((int64_t)(T - 1) * ld_g + out_f) * 4 : 0);
        const int colg = (2 * x.tm2 + h) * 128 + pcg * 4;
        const unsigned og = colg < out_f ? (unsigned)((16 * c + 2 * phq) * ld_g + colg) * 4u : OOB;
        wl_ld_g0(rsG, og);
        wl_ld_g1(rsG, og + (unsigned)ld_g * 4u);
    };
    auto request_a = [&](const Tile& x, int c) {
        const u32x4 rsA = rsrc_words(x.As, x.it < total ? (int64_t)T * in_f * 4 : 0);
        const int cola = (2 * x.tn2 + h) * 128 + pcg * 4;
        const unsigned oa = cola < in_f ? (unsigned)((16 * c + 2 * phq) * in_f + cola) * 4u : OOB;
        wl_ld_a0(rsA, oa);
        wl_ld_a1(rsA, oa + (unsigned)in_f * 4u);
    };
    auto request = [&](const Tile& x, int c) {
        request_g(x, c);
        request_a(x, c);
    };
    auto take = [&](f32x4 (&rg)[2], f32x4 (&ra)[2]) {
        rg[0] = wl_rd_g0();
        rg[1] = wl_rd_g1();
        ra[0] = wl_rd_a0();
        ra[1] = wl_rd_a1();
    };
    // the three bf16 pieces of a task's 2 steps x 4 columns (v0: step t0, v1: step t0 + 1), a word per column and piece
    auto split_pair = [](float x0, float x1, unsigned& w0, unsigned& w1, unsigned& w2) {   // one column, steps t0, t0 + 1
        w0 = pack_bf16_first(x0, x1);
        x0 = rest_lo(x0, w0);
        x1 = rest_hi(x1, w0);
        w1 = pack_bf16_rn(x0, x1);
        x0 = rest_lo(x0, w1);
        x1 = rest_hi(x1, w1);
        w2 = pack_bf16_rn(x0, x1);
    };
    auto split = [&](const f32x4& v0, const f32x4& v1, unsigned (&w)[3][4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair(v0[e], v1[e], w[0][e], w[1][e], w[2][e]);
    };
    // column 4*pcg+e of the tile -> position e*32 + pcg (slot pcg of MFMA block e); step pair phq = bytes 4*phq..
    auto write_images = [&](int slice, int pn, const unsigned (&w)[3][4]) {
        unsigned char* dst = smem_b + pn * WL_BUF + slice * WL_IMG + pcg * WL_ROWB + phq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(dst + e * 32 * WL_ROWB + p * WL_PS) = w[p][e];
    };
    // bias partial of a G task: the lane's two steps plus those of the lane 8 up (the other step pair of the 4-step
    // group), written by the lower lane: (g[4q] + g[4q+1]) + (g[4q+2] + g[4q+3]), as linear_gs_split_kernel pairs them
    auto write_bias = [&](int pn, const f32x4& b4) {
        f32x4 s4;
        // (__builtin_amdgcn_update_dpp here came out as ONE v_mov_b32_dpp whose result served all four columns)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(s4[e]) : "v"(b4[e]));
        if ((phq & 1) == 0) *reinterpret_cast<f32x4*>(sBias + pn * WL_BIAS + (h * 4 + (phq >> 1)) * 128 + pcg * 4) = s4;
    };

#ifdef DPK_K1_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    f32x16 acc[8];
    float bsum = 0.f;

    auto store = [&](const Tile& x) {   // WL_STORE_VM vector-memory instructions on every path
        const int tm = 2 * x.tm2 + h;
        const int m0 = tm * 128;
        const int64_t b = x.s * 8 + xcd;
        const bool live = tm < tiles_m;
        const auto rsW = make_rsrc(jobs.W[x.job] + b * gs_stride, live ? (int64_t)out_f * ld_w * 4 : 0);
        const unsigned row_bytes = (unsigned)ld_w * 4u;
        const float* GSb = jobs.Bi[x.job];
        const bool has_bias = live && GSb != nullptr && x.tn2 == 0;
        const auto rsB = make_rsrc(GSb + b * gs_stride, has_bias ? (int64_t)out_f * 4 : 0);
        const int f = m0 + 4 * l31 + mr;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int col = (2 * x.tn2 + jt) * 128 + 4 * l31;
            unsigned soff = col < ld_w ? (unsigned)((m0 + mr + 16 * lh) * ld_w + col) * 4u : OOB;
            if (DPK_K1_ABL & 1) soff = OOB;
            float sq = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 v = {acc[4 * jt + 0][q], acc[4 * jt + 1][q], acc[4 * jt + 2][q], acc[4 * jt + 3][q]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsW, soff, 0, DPK_K1_STORE_AUX);
                soff += ((q & 3) == 3 ? 20u : 4u) * row_bytes;  // slots 0 1 2 3 8 9 10 11 ... -> rows 4*slot
                sq = __builtin_fmaf(v[0], v[0], sq);
                sq = __builtin_fmaf(v[1], v[1], sq);
                sq = __builtin_fmaf(v[2], v[2], sq);
                sq = __builtin_fmaf(v[3], v[3], sq);
            }
            if (jt == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bsum), rsB,
                                                      lh == 0 ? (unsigned)f * 4u : OOB, 0, 0);
                if (has_bias && lh == 0 && f < out_f) sq = __builtin_fmaf(bsum, bsum, sq);
            }
            const float tot = wave_sum(sq);
            if (lane == 0) red[jt * 8 + wave] = tot;
        }
        asm volatile("" ::: "memory");
    };
    // the four squared-norm slots of a stored super-tile: lanes 0 .. 3 of wave 0; ONE store instruction per wave and call
    auto publish = [&](const Tile& x, bool stored) {
        const int hh = (lane >> 1) & 1, jt = lane & 1;
        const float* rp = red + jt * 8 + hh * 4;
        const float tot = ((rp[0] + rp[1]) + rp[2]) + rp[3];
        const int64_t b = x.s * 8 + xcd;
        const int tm = 2 * x.tm2 + hh, tn = 2 * x.tn2 + jt;
        const bool go = stored && x.it < total;
        const auto rsP = make_rsrc(normsq_part + b * part_stride, go ? (int64_t)part_stride * 4 : 0);
        __builtin_amdgcn_raw_buffer_store_b32(
            __builtin_bit_cast(unsigned, tot), rsP,
            (go && tid < 4 && tm < tiles_m && tn < tiles_n) ? (unsigned)(x.slot0 + tm * tiles_n + tn) * 4u : OOB, 0, 0);
        asm volatile("" ::: "memory");
    };
    // the pairs requested in the previous phase; younger: nothing, or (first phase of a later super-tile) the previous
    // super-tile's rows and its slot store
    auto wait_pairs = [](bool stores_between) {
        if (stores_between)
            asm volatile("s_waitcnt vmcnt(%0) ; DPK_GWAIT" ::"n"(WL_STORE_VM + 1) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) ; DPK_GWAIT" ::: "memory");
    };

    Tile cur = locate(rank), nxt = locate(rank + nbx), prev = cur;
    {   // prologue: k-step 0 of the first super-tile staged into buffer 0, k-step 1 requested
        request(cur, 0);
        asm volatile("s_waitcnt vmcnt(0) ; DPK_GWAIT" ::: "memory");
        f32x4 rg[2], ra[2];
        take(rg, ra);
        request(cur, 1);
        const int left_g = out_f - ((2 * cur.tm2 + h) * 128 + pcg * 4), left_a = in_f - ((2 * cur.tn2 + h) * 128 + pcg * 4);
        const f32x4 g0 = scale4_rn(mask4(rg[0], left_g), g_scale), g1 = scale4_rn(mask4(rg[1], left_g), g_scale);
        unsigned wg[3][4], wa[3][4];
        split(g0, g1, wg);
        split(mask4(ra[0], left_a), mask4(ra[1], left_a), wa);
        write_bias(0, add4_rn(g0, g1));
        write_images(h, 0, wg);
        write_images(2 + h, 0, wa);
        lds_only_barrier();
    }
    // Phase c of a super-tile (buffer c & 1).  A wave's vector-memory stream: [slot store of the previous super-tile, c = 0] |
    // wait | 4 loads | [33 stores after the last k-step].
    auto run = [&](auto order) {
    constexpr int ORDER = decltype(order)::value;
    for (int tl = 0; tl < ntile; ++tl) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        bsum = 0.f;
        for (int c = 0; c < nch; ++c) {
            const int pb = c & 1;
            const bool last = c == nch - 1;
            const int tm2n = last ? nxt.tm2 : cur.tm2, tn2n = last ? nxt.tn2 : cur.tn2;   // k-step c + 1's super-tile
            const int left_g = out_f - ((2 * tm2n + h) * 128 + pcg * 4), left_a = in_f - ((2 * tn2n + h) * 128 + pcg * 4);
            if (c == 0 && tl > 0) publish(prev, true);
            {
                // Four chunks per phase.  c_chunk(jp): column blocks jp of the two column tiles, 12 MFMAs alternating between
                // two accumulators (32 cycles each; six in a row into ONE accumulator run at 44 and leave gaps nobody
                // else can use), then the next pair's A pieces read into the same registers.  s_chunk(k): half a task's
                // split (G task columns 0-1, 2-3, A task columns 0-1, 2-3; ~60 vector instructions), the next k-step's
                // loads issued where a task frees its private registers (four loads at once at the top of a phase, from
                // all the waves of the workgroup, cost the issuing wave ~800 cycles).  The waves of row tile 0 contract,
                // then split; their SIMD partners split, then contract: one wave's vector instructions run beside the
                // other's MFMAs, and what is left un-overlapped at the end of a phase is one chunk.  (Whole halves in
                // opposite order: 24.4 ms at 2048 x 2048 x T256 x B2048 against 23.3 in chunks; both partners in the
                // same order, or the older wave splitting first: the same; a block's 6 MFMAs + one column pair's split
                // with everything in one scheduling region: 26-28.)
                constexpr K1Products PL = k1_products(TERMS);
                const int pn = pb ^ 1;
                const Tile& xr = (c + 2 < nch) ? cur : nxt;
                const int cr = (c + 2 < nch) ? c + 2 : c + 2 - nch;
                const unsigned char* buf = smem_b + pb * WL_BUF;
                const unsigned char* pg = buf + h * WL_IMG + (mr * 32 + l31) * WL_ROWB + lh * 16;
                const unsigned char* pa = buf + 2 * WL_IMG + l31 * WL_ROWB + lh * 16;
                bf16x8 gv[3], av[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) gv[p] = *reinterpret_cast<const bf16x8*>(pg + p * WL_PS);
                auto load_a = [&](int jp) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            av[t][q] = *reinterpret_cast<const bf16x8*>(pa + t * WL_IMG + q * WL_PS + jp * 32 * WL_ROWB);
                };
                float bp[4];
                {
                    const float* bq = sBias + pb * WL_BIAS + h * 4 * 128 + 4 * l31 + mr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bp[q] = bq[q * 128];
                }
                load_a(0);
                f32x4 g0, g1, a0, a1, b4;
                unsigned wg[3][4], wa[3][4];
                auto c_chunk = [&](int jp) {
#pragma unroll
                    for (int k = 0; k < PL.n; ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            if (!(DPK_K1_ABL & 2))
                                acc[jp + 4 * t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gv[PL.p[k]], av[t][PL.q[k]],
                                                                                          acc[jp + 4 * t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (jp < 3) load_a(jp + 1);   // into the same registers: the MFMAs above have read them
                };
                auto s_chunk = [&](int k) {
                    if (k == 0) {
                        wait_pairs(c == 0 && tl > 0);
                        if (RAGGED) {
                            g0 = scale4_rn(mask4(wl_rd_g0(), left_g), g_scale);
                            g1 = scale4_rn(mask4(wl_rd_g1(), left_g), g_scale);
                        } else {
                            g0 = wl_rd_scaled_g0(g_scale);
                            g1 = wl_rd_scaled_g1(g_scale);
                        }
                        request_g(xr, cr);
                        b4 = add4_rn(g0, g1);
                    }
                    if (k == 2) {
                        a0 = wl_rd_a0();
                        a1 = wl_rd_a1();
                        request_a(xr, cr);
                        if (RAGGED) {
                            a0 = mask4(a0, left_a);
                            a1 = mask4(a1, left_a);
                        }
                    }
                    if (DPK_K1_ABL & 4) return;
#pragma unroll
                    for (int e = 2 * (k & 1); e < 2 * (k & 1) + 2; ++e) {
                        if (k < 2) split_pair(g0[e], g1[e], wg[0][e], wg[1][e], wg[2][e]);
                        else split_pair(a0[e], a1[e], wa[0][e], wa[1][e], wa[2][e]);
                    }
                    if (k == 1) write_images(h, pn, wg);
                    if (k == 3) write_images(2 + h, pn, wa);
                };
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    if (ORDER == 0) {
                        c_chunk(jp);
                        __builtin_amdgcn_sched_barrier(0);
                        s_chunk(jp);
                    } else {
                        s_chunk(jp);
                        __builtin_amdgcn_sched_barrier(0);
                        c_chunk(jp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(DPK_K1_ABL & 4)) write_bias(pn, b4);
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum = add_rn(bsum, bp[q]);
                STAMP(1);
                if (last) store(cur);
                STAMP(4);
                lds_only_barrier();
                STAMP(2);
            }
        }
        prev = cur;
        cur = nxt;
        nxt = locate(cur.it + nbx);
    }
    };
    if (h == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    publish(prev, true);
#ifdef DPK_K1_STAMPS
    st_acc[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    if (blockIdx.x < 64 && lane == 0) {
        for (int i = 0; i < 8; ++i) dpk_stamps[(blockIdx.x * 12 + wave) * 9 + i] = st_acc[i];
        dpk_stamps[(blockIdx.x * 12 + wave) * 9 + 8] = ntile * nch;
    }
#endif
}


inline int dpk_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

constexpr int PERSIST_MAX_T = 48;

// Which MFMA the large-layer path uses.  DPK_K1_MFMA = "f32" (v_mfma_f32_32x32x2_f32, the wide-unit kernel; T <= 48),
// "split6" | "split8" | "split9" (bf16 pipe, that many exact partial products per fp32 product, any T).  Unset: the
// faster of the two as measured: f32 up to 32 steps, where the kernel is store-bound either way and
// the split's staging work only costs; split6 beyond, where the fp32 contraction is what the stores wait for.
// Returns 0 for f32, else the term count.
struct K1Knobs {
    int mfma;  // -1: default; 0: f32; 6 / 8 / 9: split terms
    int ws;    // role-split kernel for 32 < T <= 48 (default on)
};
inline K1Knobs& k1_knobs() {
    // Process-wide; the initial values come from the environment, read ONCE (first K1 call of the process), and
    // dpk_linear_configure() changes them afterwards.  Every choice meets the parity bar (tests run each).
    static K1Knobs k = [] {
        K1Knobs v = {-1, 1};
        const char* e = getenv("DPK_K1_MFMA");
        if (e && !strcmp(e, "split6")) v.mfma = 6;
        if (e && !strcmp(e, "split8")) v.mfma = 8;
        if (e && !strcmp(e, "split9")) v.mfma = 9;
        if (e && !strcmp(e, "f32")) v.mfma = 0;
        const char* w = getenv("DPK_K1_WS");
        if (w && !strcmp(w, "0")) v.ws = 0;
        return v;
    }();
    return k;
}
inline int k1_mode(int T) {
    const int m = k1_knobs().mfma;
    if (m > 0) return m;
    if (m == 0) return T <= PERSIST_MAX_T ? 0 : 6;
    return T <= 32 ? 0 : 6;
}

struct Plan {
    int id;      // which instantiation
    int rs, cs;  // block tile (out x in)
};

// Tile choice: the widest tile the layer fills (fewest operand bytes per output byte) ...
inline Plan pick_plan(int in_f, int out_f) {
    if (out_f <= 32) return in_f > 64 ? Plan{3, 32, 256} : Plan{2, 128, 64};
    if (in_f <= 32) return Plan{4, 128, 32};
    if (in_f <= 64) return Plan{2, 128, 64};
    if (in_f <= 128) return Plan{1, 128, 128};
#ifdef DPK_K1_NO_PLAN0
    return Plan{1, 128, 128};
#endif
    return Plan{0, 128, 256};
}

}  // namespace

extern "C" int dpk_linear_configure(int arithmetic, int role_split) {
    if (arithmetic != -2 && arithmetic != -1 && arithmetic != 0 && arithmetic != 6 && arithmetic != 8 && arithmetic != 9)
        return DPK_EINVAL;
    if (role_split < -2 || role_split > 1) return DPK_EINVAL;
    K1Knobs& k = k1_knobs();
    if (arithmetic != -2) k.mfma = arithmetic;
    if (role_split != -2) k.ws = role_split == 0 ? 0 : 1;
    return DPK_OK;
}

extern "C" int dpk_linear_slots(int in_features, int out_features) {
    if (in_features <= 0 || out_features <= 0) return DPK_EINVAL;
    const Plan p = pick_plan(in_features, out_features);
    const int tiles_m = (out_features + p.rs - 1) / p.rs;
    const int slot_w = p.id == 0 ? SPLIT_CS : p.cs;  // the large-layer plan counts its slots in 128-column tiles
    return tiles_m * ((in_features + slot_w - 1) / slot_w);
}

namespace {

int k1_launch(const float* const* A, const float* const* G, float* const* GSw, float* const* GSb, const int* slot0,
              int n_jobs, float* normsq_part, int64_t part_stride, int B, int T, int in_f, int out_f,
              int64_t gs_stride, int ld_w, int ld_g, float g_scale, void* stream) {
    if (!A || !G || !GSw || !slot0 || !normsq_part || n_jobs < 1 || n_jobs > DPK_K1_MAX_JOBS) return DPK_EINVAL;
    if (B < 0 || T <= 0 || in_f <= 0 || out_f <= 0) return DPK_EINVAL;
    if (ld_w < in_f || gs_stride < (int64_t)ld_w * out_f || ld_g < out_f) return DPK_EINVAL;
    const Plan p = pick_plan(in_f, out_f);
    const int tiles_m = (out_f + p.rs - 1) / p.rs;
    const int tiles_n = (in_f + p.cs - 1) / p.cs;
    const int64_t ntile = (int64_t)tiles_m * tiles_n;
    const int slot_cols = p.id == 0 ? (in_f + SPLIT_CS - 1) / SPLIT_CS : tiles_n;  // as dpk_linear_slots counts them
    const int slot_mul = p.id == 0 ? 2 : 1;
    int nt_store = (ld_w % 32 == 0) && (gs_stride % 32 == 0);
    for (int j = 0; j < n_jobs; ++j) {
        if (!A[j] || !G[j] || !GSw[j] || slot0[j] < 0) return DPK_EINVAL;
        if (slot0[j] + (int64_t)tiles_m * slot_cols > part_stride) return DPK_ESLOTS;
        // whole 128-byte lines per store segment: row stride and every row start multiples of 32 floats
        nt_store = nt_store && ((((uintptr_t)GSw[j]) & 127u) == 0);
    }
    if (B == 0) return DPK_OK;
    const int64_t nblk = ((int64_t)(B + 7) / 8) * 8 * ntile;
    if (nblk * n_jobs > 0x7fffffffLL) return DPK_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    // Large layers (the 128 x 256 plan) at T <= 48: the persistent wide-unit kernel.  Its 16-byte row stores need a row
    // stride that is a multiple of 4 floats (the arena's is a multiple of 32: whole 128-byte lines); anything else
    // takes the launch-per-tile kernel below.
    if (p.id == 0 && ld_w % 4 == 0 && (int64_t)out_f * ld_w < (1LL << 29) && (int64_t)T * (ld_g > in_f ? ld_g : in_f) < (1LL << 29)) {
        K1Jobs jobs;
        for (int j = 0; j < DPK_K1_MAX_JOBS; ++j) {
            const int q = j < n_jobs ? j : 0;
            jobs.A[j] = A[q]; jobs.G[j] = G[q]; jobs.W[j] = GSw[q]; jobs.Bi[j] = GSb ? GSb[q] : nullptr;
            jobs.slot0[j] = slot0[q];
        }
        jobs.n = n_jobs;
        const int mode = k1_mode(T);
#define DPK_K1_PERSIST_GO(NTHREADS, KFN, LDS_BYTES, ...)                                                         \
    do {                                                                                                         \
        auto kfn = KFN;                                                                                          \
        if ((LDS_BYTES) > 64 * 1024) {                                                                           \
            hipError_t e_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                                (int)(LDS_BYTES));                                               \
            if (e_ != hipSuccess) return (int)e_;                                                                \
        }                                                                                                        \
        hipLaunchKernelGGL(kfn, pgrid, dim3(NTHREADS), (LDS_BYTES), s, jobs, normsq_part, part_stride, B, T, in_f, \
                           out_f, gs_stride, ld_w, ld_g, g_scale, tiles_m, __VA_ARGS__);                         \
    } while (0)
        if (mode != 0) {
            // split-operand kernel: exact fp32 products on the bf16 matrix pipe, any sequence length; 128 x 128 tiles,
            // two workgroups of 4 waves per CU
            // 32 < T <= 48: the A slice of a (sample, column tile) group stays in LDS for all its row tiles, and a
            // workgroup takes whole groups (no more workgroups than groups); otherwise it takes tiles
            const int kmode = T > PERSIST_MAX_T ? 1 : T > SPLIT_TC ? 2 : 0;
            if (kmode == 1 && k1_knobs().ws) {
                // long sequences: 8-wave kernel over 256 x 256 super-tiles, one 16-step k-step per phase
                const int64_t supers = ((int64_t)(B + 7) / 8) * 8 * ((tiles_m + 1) / 2) * ((slot_cols + 1) / 2) * n_jobs;
                int64_t want = dpk_cu_count();
                if (want > supers) want = supers;
                const dim3 pgrid((unsigned)((want + 7) / 8 * 8));
                const bool ragged = (in_f % 4) != 0 || (out_f % 4) != 0;
#define DPK_K1_WSL_GO(TERMSV)                                                                                     \
    do {                                                                                                          \
        auto kfn = ragged ? linear_gs_wsl_kernel<TERMSV, true> : linear_gs_wsl_kernel<TERMSV, false>;             \
        hipError_t e_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS); \
        if (e_ != hipSuccess) return (int)e_;                                                                     \
        hipLaunchKernelGGL(kfn, pgrid, dim3(WL_THREADS), WL_LDS, s, jobs, normsq_part, part_stride, B, T, in_f,   \
                           out_f, gs_stride, ld_w, ld_g, g_scale, tiles_m, slot_cols);                            \
    } while (0)
                if (mode == 6) DPK_K1_WSL_GO(6);
                else if (mode == 9) DPK_K1_WSL_GO(9);
                else DPK_K1_WSL_GO(8);
#undef DPK_K1_WSL_GO
                return dpk_launch_status();
            }
            if (kmode == 2 && k1_knobs().ws) {
                // role-split kernel: one workgroup of 12 waves per CU (loaders + two contractor groups).  A tile group =
                // the row tiles of a column tile (A resident) if there are at least 3, else the column tiles of a row
                // tile (G resident; a large layer has in > 128: at least 2).  Even at 2 x 2 tiles, where every second
                // tile costs an extra phase, it beats the split kernel (250 -> 250, B = 4096: 326 -> 270 us)
                const bool swap = tiles_m < 3;
                int64_t want = dpk_cu_count();
                const int64_t groups = ((int64_t)(B + 7) / 8) * 8 * (swap ? tiles_m : slot_cols) * n_jobs;
                if (want > groups) want = groups;
                const dim3 pgrid((unsigned)((want + 7) / 8 * 8));
#define DPK_K1_WS_GO(TERMSV)                                                                                      \
    do {                                                                                                          \
        auto kfn = swap ? linear_gs_ws_kernel<TERMSV, true> : linear_gs_ws_kernel<TERMSV, false>;                 \
        hipError_t e_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS); \
        if (e_ != hipSuccess) return (int)e_;                                                                     \
        hipLaunchKernelGGL(kfn, pgrid, dim3(WS_THREADS), WS_LDS, s, jobs, normsq_part, part_stride, B, T, in_f,   \
                           out_f, gs_stride, ld_w, ld_g, g_scale, tiles_m, slot_cols);                            \
    } while (0)
                if (mode == 6) DPK_K1_WS_GO(6);
                else if (mode == 9) DPK_K1_WS_GO(9);
                else DPK_K1_WS_GO(8);
#undef DPK_K1_WS_GO
                return dpk_launch_status();
            }
            const int64_t units = ((int64_t)(B + 7) / 8) * 8 * slot_cols * n_jobs * (kmode == 2 ? 1 : tiles_m);
            int64_t want = 2 * (int64_t)dpk_cu_count();
            if (want > units) want = units;
            const dim3 pgrid((unsigned)((want + 7) / 8 * 8));
            const int kcmax = T >= 17 ? 2 : 1;
            const int rowch = 2 * kcmax + 1;
            const int rowch_a = kmode == 2 ? 2 * ((T + 15) / 16) + 1 : rowch;
            const size_t lds = (size_t)3 * 128 * (rowch + rowch_a) * 16 + (SPLIT_NQ * 128 + 2 * 4) * sizeof(float);
#define DPK_K1_SPLIT_TERMS(MODEV)                                                                                 \
    do {                                                                                                          \
        if (mode == 6) DPK_K1_PERSIST_GO(256, (linear_gs_split_kernel<6, MODEV>), lds, slot_cols, rowch, rowch_a); \
        else if (mode == 9) DPK_K1_PERSIST_GO(256, (linear_gs_split_kernel<9, MODEV>), lds, slot_cols, rowch, rowch_a); \
        else DPK_K1_PERSIST_GO(256, (linear_gs_split_kernel<8, MODEV>), lds, slot_cols, rowch, rowch_a);          \
    } while (0)
            if (kmode == 2) DPK_K1_SPLIT_TERMS(2);
            else if (kmode == 1) DPK_K1_SPLIT_TERMS(1);
            else DPK_K1_SPLIT_TERMS(0);
#undef DPK_K1_SPLIT_TERMS
            return dpk_launch_status();
        }
        if (T <= PERSIST_MAX_T) {  // DPK_K1_MFMA=f32: the fp32-MFMA wide-unit kernel (reference point for A/B timing)
            // one workgroup of 8 waves per CU
            int64_t want = dpk_cu_count();
            if (want > nblk * n_jobs) want = nblk * n_jobs;
            const dim3 pgrid((unsigned)((want + 7) / 8 * 8));
            const int Tp = (T + 1) & ~1;
            const size_t lds = ((size_t)Tp * (128 + 256) + 2 * 256 + 2 * 8) * sizeof(float);  // + 2 readable pad rows
            if (T <= 32)
                DPK_K1_PERSIST_GO(512, (linear_gs_persist_wide_kernel<2, 4>), lds, tiles_n, slot_cols);
            else
                DPK_K1_PERSIST_GO(512, (linear_gs_persist_wide_kernel<3, 6>), lds, tiles_n, slot_cols);
            return dpk_launch_status();
        }
#undef DPK_K1_PERSIST_GO
    }
    dim3 grid((unsigned)nblk);
    const bool stream_mode = T <= STREAM_MAX_T;
    const int lds_rows = stream_mode ? ((T + 1) & ~1) : TK;
#define DPK_K1_GO(NW, RS, CS, NT, ST, NTS)                                                                       \
    if (lds > 64 * 1024) {                                                                                       \
        hipError_t e_ = hipFuncSetAttribute((const void*)linear_gs_kernel<NW, RS, CS, NT, ST, NTS>,              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        if (e_ != hipSuccess) return (int)e_;                                                                    \
    }                                                                                                            \
    for (int j = 0; j < n_jobs; ++j)                                                                             \
        hipLaunchKernelGGL((linear_gs_kernel<NW, RS, CS, NT, ST, NTS>), grid, dim3(NW * 64), lds, s, A[j], G[j], \
                           GSw[j], GSb ? GSb[j] : nullptr, normsq_part, part_stride, slot0[j], B, T, in_f,      \
                           out_f, gs_stride, ld_w, ld_g, g_scale, tiles_m, tiles_n, slot_cols, slot_mul)
#define DPK_K1_LAUNCH(NW, RS, CS, NT)                                                                          \
    do {                                                                                                       \
        const size_t lds = ((size_t)lds_rows * (RS + CS) + NW) * sizeof(float);                                \
        if (stream_mode && nt_store) { DPK_K1_GO(NW, RS, CS, NT, true, true); }                                \
        else if (stream_mode) { DPK_K1_GO(NW, RS, CS, NT, true, false); }                                      \
        else if (nt_store) { DPK_K1_GO(NW, RS, CS, NT, false, true); }                                         \
        else { DPK_K1_GO(NW, RS, CS, NT, false, false); }                                                      \
    } while (0)
    switch (p.id) {
        case 0: DPK_K1_LAUNCH(8, 128, 256, 2); break;
        case 1: DPK_K1_LAUNCH(8, 128, 128, 2); break;
        case 2: DPK_K1_LAUNCH(4, 128, 64, 1); break;
        case 3: DPK_K1_LAUNCH(4, 32, 256, 2); break;
        default: DPK_K1_LAUNCH(4, 128, 32, 1); break;
    }
#undef DPK_K1_LAUNCH
#undef DPK_K1_GO
    return dpk_launch_status();
}

}  // namespace

#ifdef DPK_K1_STAMPS
extern "C" int dpk_lab_read_stamps(void* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpk_stamps), sizeof(unsigned long long) * n);
}
#endif

extern "C" int dpk_linear_grad_sample_multi(const dpk_k1_job* jl, int n_jobs, float* normsq_part, int64_t part_stride,
                                            int B, int64_t gs_stride, float g_scale, void* stream) {
    if (!jl || !normsq_part || n_jobs < 1 || n_jobs > DPK_K1_MULTI_MAX_JOBS || B < 0) return DPK_EINVAL;
    K1MultiJobs jobs;
    int tiles = 0, maxT = 0;
    for (int j = 0; j < DPK_K1_MULTI_MAX_JOBS; ++j) {
        const dpk_k1_job& q = jl[j < n_jobs ? j : 0];
        if (j < n_jobs) {
            if (!q.A || !q.G || !q.GSw || q.slot0 < 0 || q.T <= 0 || q.in_features <= 0 || q.out_features <= 0)
                return DPK_EINVAL;
            if (q.ld_w < q.in_features || q.ld_g < q.out_features || gs_stride < (int64_t)q.ld_w * q.out_features)
                return DPK_EINVAL;
            if (q.ld_w % 4 != 0) return DPK_EALIGN;       // 16-byte row stores
            if (q.T > PERSIST_MAX_T) return DPK_ERANGE;   // the whole sequence of a tile is staged in LDS
            if ((int64_t)q.out_features * q.ld_w >= (1LL << 29) ||
                (int64_t)q.T * (q.ld_g > q.in_features ? q.ld_g : q.in_features) >= (1LL << 29))
                return DPK_ERANGE;
        }
        const Plan p = pick_plan(q.in_features, q.out_features);
        K1Shape& h = jobs.sh[j];
        h.T = q.T; h.in_f = q.in_features; h.out_f = q.out_features; h.ld_w = q.ld_w; h.ld_g = q.ld_g;
        h.tiles_m = (q.out_features + 127) / 128;
        h.tiles_n = (q.in_features + 255) / 256;
        const int slot_w = p.id == 0 ? SPLIT_CS : p.cs;
        h.slot_cols = (q.in_features + slot_w - 1) / slot_w;
        h.slot_mul = p.id == 0 ? 2 : 1;
        h.tile0 = tiles;
        if (j < n_jobs) {
            // the wide tiles must cover the layer's slot grid exactly as dpk_linear_slots counts it
            if (h.tiles_m != (q.out_features + p.rs - 1) / p.rs || (h.slot_mul == 1 && h.tiles_n != h.slot_cols))
                return DPK_ERANGE;
            if (q.slot0 + (int64_t)h.tiles_m * h.slot_cols > part_stride) return DPK_ESLOTS;
            tiles += h.tiles_m * h.tiles_n;
            if (q.T > maxT) maxT = q.T;
        }
        jobs.A[j] = q.A; jobs.G[j] = q.G; jobs.W[j] = q.GSw; jobs.Bi[j] = q.GSb; jobs.slot0[j] = q.slot0;
    }
    jobs.n = n_jobs;
    jobs.tiles_total = tiles;
    if (B == 0) return DPK_OK;
    const int64_t items = ((int64_t)(B + 7) / 8) * 8 * tiles;
    if (items > 0x7fffffffLL) return DPK_ERANGE;
    int64_t want = dpk_cu_count();
    if (want > items) want = items;
    const dim3 grid((unsigned)((want + 7) / 8 * 8));
    const int Tp = (maxT + 1) & ~1;
    const size_t lds = ((size_t)Tp * (128 + 256) + 2 * 256 + 2 * 8) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define DPK_K1_MULTI_GO(NG, NA)                                                                                       \
    do {                                                                                                              \
        auto kfn = linear_gs_multi_kernel<NG, NA>;                                                                    \
        if (lds > 64 * 1024) {                                                                                        \
            hipError_t e_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return (int)e_;                                                                     \
        }                                                                                                             \
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, jobs, normsq_part, part_stride, B, gs_stride, g_scale);      \
    } while (0)
    if (maxT <= 32)
        DPK_K1_MULTI_GO(2, 4);
    else
        DPK_K1_MULTI_GO(3, 6);
#undef DPK_K1_MULTI_GO
    return dpk_launch_status();
}

extern "C" int dpk_linear_grad_sample(const float* A, const float* G, float* GSw, float* GSb, float* normsq_part,
                                      int64_t part_stride, int slot0, int B, int T, int in_f, int out_f,
                                      int64_t gs_stride, int ld_w, float g_scale, void* stream) {
    return k1_launch(&A, &G, &GSw, GSb ? &GSb : nullptr, &slot0, 1, normsq_part, part_stride, B, T, in_f, out_f,
                     gs_stride, ld_w, out_f, g_scale, stream);
}

extern "C" int dpk_linear_grad_sample_group(const float* const* A, const float* const* G, float* const* GSw,
                                            float* const* GSb, const int* slot0, int n_jobs, float* normsq_part,
                                            int64_t part_stride, int B, int T, int in_f, int out_f,
                                            int64_t gs_stride, int ld_w, int ld_g, float g_scale, void* stream) {
    return k1_launch(A, G, GSw, GSb, slot0, n_jobs, normsq_part, part_stride, B, T, in_f, out_f, gs_stride, ld_w,
                     ld_g, g_scale, stream);
}
