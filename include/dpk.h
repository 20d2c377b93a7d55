/*
 * dpk.h — C ABI of libdpk_gfx950.so: the MI355X (gfx950 / CDNA4) kernels of the DP-SGD step.
 *
 * The reference (ammanyusuf/dp-timeautodiff) is pure Python and has no FFI of its own: every
 * arithmetic step below is what Opacus dispatches to ATen when the reference executes
 *     loss_Auto.backward()   Model_Code/timeautoencoder.py:749   (per-sample gradients, hooks)
 *     optimizer_ae.step()    Model_Code/timeautoencoder.py:751   (clip -> sum -> noise -> scale)
 * Each entry point cites the Opacus operation and the reference call site
 * it replaces.  The Python host (dp_timeautodiff_amd/) binds these with ctypes; the binding a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; pointers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()),
 *     fp32 row-major, contiguous unless a stride argument says otherwise;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call is asynchronous and stream-ordered and allocates nothing; the only process-wide state is two
 *     per-device caches (CU count; device address of dpk_clip_reduce_noise_static's scratch array);
 *   - return 0 on success, a negative DPK_E* code on an argument error (nothing launched), or a
 *     positive hipError_t if the launch failed.  Nothing throws.
 *   - per-sample squared norms are produced as *partials*: every grad-sample kernel writes, for
 *     sample b, `slots` floats at normsq_part[b*part_stride + slot0 ... ]; dpk_clip_factors sums a
 *     sample's slots in index order.  No float atomics anywhere: results are bitwise reproducible
 *     run to run.
 */
#ifndef DPK_H
#define DPK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPK_VERSION 100 /* 0.1.0 */

#define DPK_OK 0
#define DPK_EINVAL (-1)   /* null pointer / non-positive dimension / bad flag */
#define DPK_ERANGE (-2)   /* dimension beyond what the kernel's grid or LDS plan supports */
#define DPK_EALIGN (-3)   /* pointer or stride not aligned as documented */
#define DPK_ESLOTS (-4)   /* normsq partial row too short for the slots this call writes */
#define DPK_ESCRATCH (-5) /* dpk_clip_reduce_noise_static: input needs more scratch than DPK_STATIC_SCRATCH_BYTES */

int dpk_version(void);
const char* dpk_error_string(int code);

/* Number of normsq partial slots the corresponding *_grad_sample call writes per sample. */
int dpk_linear_slots(int in_features, int out_features);
int dpk_embedding_slots(int num_embeddings, int embedding_dim);
int dpk_layernorm_slots(int channels);
int dpk_normsq_slots(int64_t P);

/*
 * K1 — per-sample gradient of nn.Linear / RNNLinear (batched fp32 MFMA GEMM, K = T).
 * Replaces Opacus' Linear grad sampler  einsum("n...i,n...j->nij", backprops, activations)  and
 * einsum("n...k->nk", backprops), applied once per layer *call* and accumulated T times for the
 * CustomGRU gates (timeautoencoder.py:84-94 via :749).  Here the T calls are stacked by the host
 * and contracted in one launch:
 *     GSw[b,o,i] = sum_t (g_scale*G[b,t,o]) * A[b,t,i]        GSb[b,o] = sum_t g_scale*G[b,t,o]
 * A [B,T,in], G [B,T,out] contiguous.  GSw/GSb rows of sample b start at b*gs_stride floats
 * (gs_stride = row length of the flat per-sample-gradient arena).  GSb may be NULL (no bias).
 * ld_w >= in is the row stride of GSw[b] (the arena pads rows of awkward widths to a multiple of 32 floats so
 * that every store is a whole 128-byte line); columns [in, ld_w) are written as zeros.
 * g_scale = batch size for loss_reduction="mean" (Opacus multiplies backprops by n first; so do we).
 * Writes dpk_linear_slots(in,out) normsq partials per sample, covering weight and bias.
 * Arithmetic: fp32 in, fp32 out, fp32 accumulation in t order.  Small layers and sequences of up to 32 steps use
 * the fp32 matrix instruction; large layers (in > 128, out > 32) at T > 32 split each fp32 operand into three bf16
 * pieces (to 2^-27 of its value) and form 6 of the 9 exact partial products per fp32 product on the bf16 matrix pipe
 * (what is dropped is below 2^-26 of the product: a quarter of the fp32 product's own rounding).
 * dpk_linear_configure() below overrides the choice.
 */
int dpk_linear_grad_sample(const float* A, const float* G, float* GSw, float* GSb,
                           float* normsq_part, int64_t part_stride, int slot0,
                           int B, int T, int in_features, int out_features,
                           int64_t gs_stride, int ld_w, float g_scale, void* stream);

/*
 * Process-wide choice among K1's large-layer kernels (in > 128, out > 32); every choice meets the parity bar and the
 * default is the fastest measured.  No counterpart in Opacus: its einsum has one arithmetic.
 *   arithmetic  -1 default (fp32 matrix instruction up to 32 steps, 6 bf16 partial products beyond) | 0 fp32 matrix
 *               instruction wherever it exists (T <= 48) | 6 | 8 | 9 partial products per fp32 product | -2 unchanged
 *   role_split  1 (default): the round-3 kernels — at 32 < T <= 48 separate loader and
 *               contractor waves, at T > 48 256 x 256 super-tiles with split and contraction halves — both
 *               bit-identical to the round-2 split kernel, which 0 puts back | -1 default | -2 unchanged
 * Initial values: environment variables DPK_K1_MFMA = f32 | split6 | split8 | split9 and DPK_K1_WS = 0, read once at
 * the first K1 call of the process.  Not to be called while launches of another thread are being issued.
 */
int dpk_linear_configure(int arithmetic, int role_split);

/*
 * K1, grouped: n_jobs <= DPK_K1_MAX_JOBS layers of the SAME shape (B, T, in, out, ld_w) in one launch — the update /
 * reset / new gate Linears of CustomGRU (timeautoencoder.py:63-67) for encoder_mu and encoder_logvar (:117-120) reach
 * the sampler back to back, and six thin launches leave six tails.  A, G, GSw, GSb (nullable as a whole or per entry)
 * and slot0 are HOST arrays of n_jobs entries (device pointers / slot indices); every job writes its own
 * dpk_linear_slots(in,out) partials at slot0[j].  Results are identical to n_jobs dpk_linear_grad_sample calls.
 * ld_g >= out is the row stride of every G[j] (the fused GRU backward writes the three gates' backprops side by side
 * in one [B,T,3H] buffer so that dL/dx is a single GEMM: each gate's G is then a column block with ld_g = 3H).
 */
#define DPK_K1_MAX_JOBS 8
int dpk_linear_grad_sample_group(const float* const* A, const float* const* G, float* const* GSw,
                                 float* const* GSb, const int* slot0, int n_jobs,
                                 float* normsq_part, int64_t part_stride,
                                 int B, int T, int in_features, int out_features,
                                 int64_t gs_stride, int ld_w, int ld_g, float g_scale, void* stream);

/*
 * K1, many layers of DIFFERENT shapes in one launch — for the reference's own batch size (B ~ 50, dp_job.sh), where a
 * layer's launch is a handful of tiles per CU and the twelve launches of one backward pass (timeautoencoder.py:749) each
 * pay a ramp and a tail.  `jobs` is a HOST array of n_jobs <= DPK_K1_MULTI_MAX_JOBS descriptors (device pointers inside);
 * every job is what one dpk_linear_grad_sample call would take and writes the same dpk_linear_slots(in,out) partials at
 * its slot0.  Requirements per job: T <= 48, ld_w % 4 == 0 (DPK_EALIGN otherwise: such layers keep their own launch).
 * All jobs share B, gs_stride (one arena), part_stride and g_scale.  Results equal per-layer calls on the fp32 matrix
 * pipe bit for bit for layers of the large plan (in > 128, out > 32); for thinner layers the weights are bit-identical
 * and the bias differs in summation order only.
 */
#define DPK_K1_MULTI_MAX_JOBS 32
typedef struct {
    const float* A;   /* [B,T,in] */
    const float* G;   /* [B,T,out], row stride ld_g */
    float* GSw;       /* window of the weight in arena row 0 */
    float* GSb;       /* window of the bias, or NULL */
    int slot0, T, in_features, out_features, ld_w, ld_g;
} dpk_k1_job;
int dpk_linear_grad_sample_multi(const dpk_k1_job* jobs, int n_jobs, float* normsq_part, int64_t part_stride,
                                 int B, int64_t gs_stride, float g_scale, void* stream);


/*
 * K3 — per-sample gradient of nn.Embedding.  Replaces Opacus' Embedding sampler
 *     zeros(B,V,E).scatter_add_(1, index, backprops)     (Embedding_data, timeautoencoder.py:128-133,160-164)
 * idx [B,T] int64 (values in [0,V)), G [B,T,E]; GS[b,v,:] = sum_{t: idx[b,t]=v} g_scale*G[b,t,:], summed in
 * increasing t.  Out-of-range indices are ignored (the forward would already have raised).
 */
int dpk_embedding_grad_sample(const int64_t* idx, const float* G, float* GS,
                              float* normsq_part, int64_t part_stride, int slot0,
                              int B, int T, int V, int E,
                              int64_t gs_stride, float g_scale, void* stream);

/*
 * K4 — per-sample gradient of nn.LayerNorm over the last dim.  Replaces Opacus' LayerNorm sampler
 *     sum_t(F.layer_norm(x, (C,), eps=eps) * backprops), sum_t(backprops)      (timediffusion.py:107)
 * X, G [B,T,C]; GSw[b,c], GSb[b,c].
 */
int dpk_layernorm_grad_sample(const float* X, const float* G, float* GSw, float* GSb,
                              float* normsq_part, int64_t part_stride, int slot0,
                              int B, int T, int C, float eps,
                              int64_t gs_stride, float g_scale, void* stream);

/*
 * K5 — squared-norm partials of externally produced per-sample gradients (a [B,P] strided matrix),
 * for grads that did not come from K1/K3/K4.  Replaces  g.reshape(len(g),-1).norm(2,dim=-1)  of
 * DPOptimizer.clip_and_accumulate (timeautoencoder.py:751).  Writes dpk_normsq_slots(P) partials.
 */
int dpk_normsq_partials(const float* GS, int64_t gs_stride, float* normsq_part, int64_t part_stride,
                        int slot0, int B, int64_t P, void* stream);

/*
 * K6 — per-sample norm and clip factor.  Replaces
 *     per_sample_norms = stack(per_param_norms).norm(2, dim=1)
 *     per_sample_clip_factor = (max_grad_norm / (per_sample_norms + 1e-6)).clamp(max=1.0)
 * normsq_part [B, part_stride]: slots [0, n_slots) of each row are summed in index order.
 * norms (nullable) receives sqrt(sum); clip receives the factor.
 * n_valid_dev (nullable) points at a DEVICE int: rows b >= *n_valid_dev are the padding of a fixed-capacity Poisson
 * batch (dpk_poisson_sample) and get clip = 0, norm = 0, so they add nothing to the clipped sum.
 */
int dpk_clip_factors(const float* normsq_part, int64_t part_stride, int n_slots,
                     float* clip, float* norms, int B, float max_grad_norm, const int* n_valid_dev, void* stream);

/*
 * K7a — clipped weighted sum over the batch.  Replaces  einsum("i,i...", clip, grad_sample)  per
 * parameter and the `p.summed_grad (+)= grad` accumulation, as ONE launch over the whole flat arena:
 *     summed_out[j] = (summed_in ? summed_in[j] : 0) + sum_b clip[b] * GS[b*gs_stride + j],  j in [0,P)
 * P and gs_stride must be multiples of 4 and all pointers 16-byte aligned (the arena guarantees it).
 * Reads every per-sample gradient element exactly once.  `workspace` (>= dpk_clip_reduce_workspace
 * bytes; may be NULL when that returns 0) holds split-B partial sums; summed_in may alias summed_out.
 */
int64_t dpk_clip_reduce_workspace(int B, int64_t P);
int dpk_clip_reduce(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                    float* summed_out, int B, int64_t P, void* workspace, int64_t workspace_bytes,
                    void* stream);

/*
 * K7b — add_noise + scale_grad.  Replaces
 *     p.grad = p.summed_grad + normal(0, noise_multiplier*max_grad_norm)      (add_noise)
 *     p.grad /= expected_batch_size * accumulated_iterations                  (scale_grad)
 * grad[j] = (summed[j] + z[j]) * inv_scale.  z = noise[j] if `noise` is non-NULL (injected tensor,
 * already scaled by std — the parity path), else noise_std * Philox4x32-10 normal #j of the stream
 * (seed, offset): element j uses Philox block (offset + j/4), lane j%4 — every rank of a data-parallel
 * job draws the same z from the same (seed, offset).  noise_std == 0 and noise == NULL -> z = 0.
 * `seed` is the Philox KEY as is.  dpk_poisson_sample runs the same generator from block 0 too: a caller that seeds both
 * from one user seed must hand each its own key (the Python host derives key = splitmix64(seed ^ purpose tag),
 * dp_timeautodiff_amd/rng.py), or sampling and noise would share their uniforms.
 * offset_dev (nullable) points at a DEVICE uint64 added to `offset` when the kernel runs: a step captured into
 * a hipGraph keeps drawing fresh noise on every replay (the host, or a node of the graph, advances the word).
 */
int dpk_noise_scale(const float* summed, const float* noise, uint64_t seed, uint64_t offset,
                    const uint64_t* offset_dev, float noise_std, float inv_scale, float* grad, int64_t P, void* stream);

/* K7 fused = K7a followed by K7b without the intermediate P-sized round trip (single-GPU path).
 * summed_out (nullable) additionally receives the un-noised clipped sum. */
int dpk_clip_reduce_noise(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                          const float* noise, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                          float noise_std, float inv_scale, float* summed_out, float* grad, int B, int64_t P,
                          void* workspace, int64_t workspace_bytes, void* stream);

/*
 * K7 fused, for callers that cannot allocate.  No counterpart in the reference: it exists for hosts that may not call
 * hipMalloc around the step — code under hipGraph capture, tight serving/training loops, harnesses that forbid it.
 * Same arguments, kernels and results (bit for bit) as dpk_clip_reduce_noise, minus the workspace: the split-B
 * partial sums live in a file-scope __device__ array of DPK_STATIC_SCRATCH_BYTES inside the library (one copy per
 * GPU the library runs on, held until the process ends; its device address is looked up once per device on the
 * first call there).  The call allocates and frees nothing.  Inputs whose split needs more than that return
 * DPK_ESCRATCH and launch nothing (with today's split rule the need never exceeds 13.1 MB).  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU (two streams, two threads) race — serialise them.  Nothing
 * has to be cleared between calls: every partial a call reads it has written itself.
 * The device whose array is used is the CURRENT device, and `stream` must be one of its streams (another GPU's
 * stream: DPK_EINVAL, nothing launched).  The FIRST call on a device performs the address lookup (hipGetSymbolAddress):
 * make it outside stream capture, with that device current; every later call on that device is capture-safe.
 */
#define DPK_STATIC_SCRATCH_BYTES (16 * 1024 * 1024)
int dpk_clip_reduce_noise_static(const float* GS, int64_t gs_stride, const float* clip, const float* summed_in,
                                 const float* noise, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                 float noise_std, float inv_scale, float* summed_out, float* grad, int B, int64_t P,
                                 void* stream);

/* Philox normal stream on its own (tests; also lets host code pre-draw the shared noise). */
int dpk_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float std, void* stream);

/*
 * K8 — fused Linear-cell GRU recurrence: the caller either side of K1.  Replaces the T-step
 * Python loop of CustomGRU.forward (Model_Code/timeautoencoder.py:74-96: cat -> update_gate/reset_gate ->
 * cat -> new_gate -> blend, per time step) and its autograd backward with ONE launch each.  x has already been
 * through input_layer (xp [B,T,H]).  H must be a multiple of 8 and at most 512 (DPK_ERANGE otherwise).
 *
 * One launch runs G <= DPK_GRU_MAX_G independent GRUs of equal shape (e.g. DeapStack_DP's encoder_mu and
 * encoder_logvar, timeautoencoder.py:117-120, which read the same input): a recurrence has only B/16 workgroups
 * per GRU, so batching independent ones is what fills the chip.  Per-GRU INPUTS arrive as host arrays of device
 * pointers (`const float* const*`, read during the call; the tensors may live anywhere); everything the kernels
 * WRITE, and the packed weights, are single buffers with a leading contiguous [G] dimension.
 *
 * Only the HIDDEN half of every gate is sequential.  The input half, W[:, :H] x_t + b for all T steps, is one
 * ordinary GEMM the caller runs first (xg [B,T,3H], column blocks update | reset | new), and dL/dxp is one GEMM over
 * all T steps afterwards (dxp = Gz W_u[:, :H] + Gr W_r[:, :H] + Gn W_n[:, :H]): that halves the MFMA work on the T-step
 * latency chain and takes the x_t -> LDS transposes off it.
 *
 * Gate weights are nn.Linear(2H, H) weights ([H, 2H] row-major, row stride ld), given as W[3g+0..2] =
 * {update, reset, new} of GRU g; their hidden halves W[:, H:] are re-ordered into MFMA fragment order by the pack
 * calls (one launch each):
 *   pack_forward : Pz, Pr, Pn          [G, dpk_gru_pack_floats(H, H)]
 *   pack_backward: Pnh                 [G, dpk_gru_pack_floats(H, H)]     W_new[:, H:]^T
 *                  Pzrh                [G, dpk_gru_pack_floats(H, 2H)]    [W_update[:, H:]; W_reset[:, H:]]^T
 *
 * forward:  xp[g] [B,T,H] (copied through into Azr / An), xg[g] [B,T,3H] (biases included); h0 NULL or h0[g] [B,H]
 *   (NULL entry = zeros).
 *   writes out = h_t, the gate activations Z, R, N ([G,B,T,H]) and the gates' stacked input activations
 *   Azr = [xp_t | h_{t-1}] (update, reset) and An = [xp_t | r_t*h_{t-1}] (new), both [G,B,T,2H] — exactly the `A`
 *   operand dpk_linear_grad_sample contracts with K = T.
 * backward: dout[g] [B,T,H] = dL/dout (NULL entry = zeros); dhT NULL or dhT[g] [B,H] added at t = T-1.
 *   writes the gates' pre-activation gradients side by side, G3 [G,B,T,3H] = [Gz | Gr | Gn] (= dL/dxg; each column
 *   block is a `G` operand of dpk_linear_grad_sample_group with ld_g = 3H) and, if non-NULL, dh0 [G,B,H].
 */
#define DPK_GRU_MAX_G 8
int64_t dpk_gru_pack_floats(int O, int K);
int dpk_gru_pack_forward(const float* const* W, int ld, float* Pz, float* Pr, float* Pn, int H, int G, void* stream);
int dpk_gru_pack_backward(const float* const* W, int ld, float* Pnh, float* Pzrh, int H, int G, void* stream);
int dpk_gru_forward(const float* const* xp, const float* const* xg, const float* const* h0, const float* Pz,
                    const float* Pr, const float* Pn, float* out, float* Z, float* R, float* N,
                    float* Azr, float* An, int B, int T, int H, int G, void* stream);
int dpk_gru_backward(const float* const* dout, const float* const* dhT, const float* Z, const float* R,
                     const float* N, const float* Azr, const float* Pnh, const float* Pzrh,
                     float* G3, float* dh0, int B, int T, int H, int G, void* stream);

/*
 * K11 — on-device Poisson sampler.  Replaces the host side of Opacus' DPDataLoader /
 * UniformWithReplacementSampler as the reference reaches it through make_private_with_epsilon
 * (timeautoencoder.py:664-682, consumed at :726-733): `mask = rand(N) < sample_rate; idx = mask.nonzero()`.
 * Sample j is selected iff u_j < sample_rate, u_j = ((x >> 8) + 0.5) * 2^-24 with x = word (j & 3) of Philox4x32-10
 * block (offset + *offset_dev + j / 4) under key `seed`; the selected indices are written in increasing order to
 * idx[0..n), idx[n..capacity) is padded with 0, counts[0] = min(n_selected, capacity), counts[1] = n_selected.
 * A call consumes ceil(N / 4) Philox blocks.  idx: device int64[capacity]; counts: device int[2].
 */
int dpk_poisson_sample(uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float sample_rate,
                       int64_t N, int capacity, int64_t* idx, int* counts, void* stream);

/*
 * K10 — fused DPLSTM recurrence (denoiser).  Replaces, per layer, the T-step loop of Opacus'
 * DPLSTM cells (what ModuleValidator.fix makes of the nn.LSTM at Model_Code/timediffusion.py:106; cell math as in
 * nn.LSTM, gate order i,f,g,o) and its autograd backward with ONE launch each for G <= DPK_GRU_MAX_G independent
 * recurrences of equal shape (the two directions of a bidirectional layer): bit g of reverse_mask makes recurrence g
 * walk t = T-1..0.  The input projection is done outside for all steps at once: xg[g] = ih(x) [B,T,4H] (bias_ih
 * included); Whh[g] is the hh weight [4H,H] row-major, bhh[g] its bias [4H] (bhh NULL or NULL entry = none).
 *   pack_forward : P  [G, 4, dpk_gru_pack_floats(H, H)]      pack_backward: Pt [G, dpk_gru_pack_floats(H, 4H)]
 * forward writes out = h_t [B,T,G*H] (recurrence g in the column block [gH, (g+1)H): the concatenated output of a
 *   bidirectional layer, no torch.cat), the gate activations `gates` [G,B,T,4H], the cell state C [G,B,T,H] and
 *   Hprev [G,B,T,H] = the hidden state each step consumed (the `A` operand of hh's per-sample gradient).
 *   h0 / c0: NULL or per-recurrence [B,H] (NULL entry = zeros).
 * backward writes dpre [G,B,T,4H] = dL/d(pre-activation) — both dL/dxg and the `G` operand of the ih and hh
 *   per-sample-gradient contractions — and, if non-NULL, dh0 / dc0 [G,B,H].  dout[g]: [B,T,H] rows of stride ld_dout
 *   (H for a dense tensor, G*H for column block g of the concatenated output's gradient; NULL entry = zeros).
 *   dcT: NULL or per-recurrence [B,H] = dL/dc_n, the gradient of the final cell state (a loss that reads c_n, e.g. a
 *   state handed to the next segment); NULL entry = zeros.  (dL/dh_n arrives through dout: h_n is a row of out.)
 * H <= 256, any H >= 1 (rows that are not 16-byte aligned are accessed in smaller pieces).
 */
int dpk_lstm_pack_forward(const float* const* Whh, float* P, int H, int G, void* stream);
int dpk_lstm_pack_backward(const float* const* Whh, float* Pt, int H, int G, void* stream);
int dpk_lstm_forward(const float* const* xg, const float* const* h0, const float* const* c0,
                     const float* const* bhh, const float* P, float* out, float* gates, float* C,
                     float* Hprev, int reverse_mask, int B, int T, int H, int G, void* stream);
int dpk_lstm_backward(const float* const* dout, const float* const* dcT, const float* const* c0, const float* gates,
                      const float* C, const float* Pt, float* dpre, float* dh0, float* dc0,
                      int reverse_mask, int B, int T, int H, int G, int ld_dout, void* stream);

/*
 * K9 — Adam with L2 weight decay on the arena-layout gradient.  Replaces
 * torch.optim.Adam.step() as called by DPOptimizer.step() (optimizer built at timeautoencoder.py:658, stepped at
 * :751) with one launch:  g = grad + wd*p;  m += (g-m)(1-b1);  v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = *step_dev + 1  (the caller then advances *step_dev).
 * grad / exp_avg / exp_avg_sq are [P] vectors laid out like an arena row.  win_table: n_windows rows of 5 int64
 * {param device pointer (dense row-major [rows,cols]), arena offset, rows, cols, ld}, ordered by offset, offsets
 * multiples of 4, ld a multiple of 4 wherever ld != cols.  hyper: n_windows rows of 8 floats
 * {lr, beta1, beta2, eps, weight_decay, 1-beta1, 1-beta2, 0} (the complements rounded from double, as torch
 * passes them).  Both tables live in device memory.
 */
int dpk_adam_step(const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* win_table,
                  const float* hyper, int n_windows, const float* step_dev, int64_t P, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPK_H */
