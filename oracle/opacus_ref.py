"""ORACLE — test infrastructure only.  Never imported by the product package.

CPU restatement (torch CPU ops, float32 unless told otherwise) of the DP-SGD step the
reference delegates to Opacus at these call sites of
/root/reference/TimeAutoDiff-main/Model_Code/timeautoencoder.py:

    :674  make_private_with_epsilon  -> GradSampleModule(hooks) + DPOptimizer
    :734  ae(batch_input)            -> forward hooks stash layer inputs
    :749  loss_Auto.backward()       -> backward hooks: per-sample grads (Linear/Embedding/LayerNorm)
    :751  optimizer_ae.step()        -> clip_and_accumulate -> add_noise -> scale_grad
    :732  optimizer_ae.zero_grad()

Opacus itself (unpinned in requirements.txt:6; 1.4.x-1.5.x by date) is absent from
/root/reference and from this image, so the algorithm is restated from its published
behaviour as recorded in SURVEY.md Appendix A (A.1 hooks, A.2 samplers, A.3 optimizer).
The *op order* of the reference is kept on purpose: one einsum per layer call followed by
an accumulate into a [B, out, in] buffer, a two-pass norm-then-einsum clip, per-parameter
noise.  That makes this module both the parity checker and the honest CPU baseline
("port") that bench.py times next to the GPU path.

Pinning (tests/test_oracle_golden.py): every function here is checked against
tests/golden/*.npz, whose expected per-sample gradients come from a micro-batch (B=1)
autograd loop over the *reference's own model classes* run in the build container
(tests/golden/make_golden.py), and whose clip/sum/noise/scale tail is the A.3 formula set
evaluated in float64.  The reference has no tests/golden vectors of its own
(SURVEY.md §4), so that is the strongest pin available.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# A.2 grad samplers
# --------------------------------------------------------------------------------------


def linear_grad_sample(A: torch.Tensor, G: torch.Tensor, has_bias: bool = True):
    """nn.Linear / RNNLinear sampler: weight einsum('n...i,n...j->nij', G, A); bias einsum('n...k->nk', G)."""
    gw = torch.einsum("n...i,n...j->nij", G, A)
    gb = torch.einsum("n...k->nk", G) if has_bias else None
    return gw, gb


def embedding_grad_sample(idx: torch.Tensor, G: torch.Tensor, num_embeddings: int):
    """nn.Embedding sampler: zeros(B,V,E).scatter_add_(1, index, G.reshape(B,-1,E))."""
    B = idx.shape[0]
    E = G.shape[-1]
    if B == 0:
        return torch.zeros(1, num_embeddings, E, dtype=G.dtype)
    index = idx.unsqueeze(-1).expand(*idx.shape, E).reshape(B, -1, E)
    gs = torch.zeros(B, num_embeddings, E, dtype=G.dtype)
    gs.scatter_add_(1, index, G.reshape(B, -1, E))
    return gs


def layernorm_grad_sample(x: torch.Tensor, G: torch.Tensor, normalized_shape, eps: float):
    """nn.LayerNorm sampler: sum over all-but-batch-and-last-n dims of layer_norm(x)*G, and of G."""
    n = len(normalized_shape)
    xhat = F.layer_norm(x, normalized_shape, eps=eps)
    dims = list(range(1, x.dim() - n))
    gw = (xhat * G).sum(dim=dims) if dims else xhat * G
    gb = G.sum(dim=dims) if dims else G
    return gw, gb


# --------------------------------------------------------------------------------------
# A.1 hook bookkeeping over a list of layer calls (what the backward hook does per call)
# --------------------------------------------------------------------------------------


def accumulate_linear_calls(As: Sequence[torch.Tensor], Gs: Sequence[torch.Tensor], g_scale: float,
                            has_bias: bool = True):
    """Opacus order: for calls in *backward* order, G <- grad_out * n; gs = sampler; _current_grad_sample += gs."""
    gw_acc = None
    gb_acc = None
    for A, G in zip(reversed(list(As)), reversed(list(Gs))):
        Gn = G * g_scale
        gw, gb = linear_grad_sample(A, Gn, has_bias)
        if gw_acc is None:
            gw_acc = gw.clone()
            gb_acc = gb.clone() if gb is not None else None
        else:
            gw_acc += gw
            if gb is not None:
                gb_acc += gb
    return gw_acc, gb_acc


# --------------------------------------------------------------------------------------
# A.3 DPOptimizer arithmetic
# --------------------------------------------------------------------------------------


def clip_and_accumulate(grad_samples: Sequence[torch.Tensor], max_grad_norm: float):
    """Returns (per_sample_norms, per_sample_clip_factor, [summed grad per param])."""
    if len(grad_samples[0]) == 0:
        clip = torch.zeros(0, dtype=grad_samples[0].dtype)
        norms = torch.zeros(0, dtype=grad_samples[0].dtype)
    else:
        per_param_norms = [g.reshape(len(g), -1).norm(2, dim=-1) for g in grad_samples]
        norms = torch.stack(per_param_norms, dim=1).norm(2, dim=1)
        clip = (max_grad_norm / (norms + 1e-6)).clamp(max=1.0)
    summed = [torch.einsum("i,i...", clip, g) for g in grad_samples]
    return norms, clip, summed


def add_noise_and_scale(summed: Sequence[torch.Tensor], noises: Optional[Sequence[torch.Tensor]],
                        noise_multiplier: float, max_grad_norm: float, expected_batch_size: float,
                        accumulated_iterations: int = 1, loss_reduction: str = "mean",
                        generator: Optional[torch.Generator] = None):
    """add_noise + scale_grad.  `noises` injects the N(0, sigma*C) draw (already scaled by std)."""
    out = []
    std = noise_multiplier * max_grad_norm
    for k, s in enumerate(summed):
        if noises is not None:
            z = noises[k].reshape(s.shape)
        elif std == 0:
            z = torch.zeros_like(s)
        else:
            z = torch.normal(mean=0.0, std=std, size=s.shape, generator=generator)
        g = s + z
        if loss_reduction == "mean":
            g = g / (expected_batch_size * accumulated_iterations)
        out.append(g)
    return out


def dp_step_flat(gs_flat: torch.Tensor, C: float, noise_flat: Optional[torch.Tensor], expected_bs: float,
                 summed_in: Optional[torch.Tensor] = None):
    """Whole tail on a flat [B, P] matrix (any float dtype): norms, clip, summed, grad."""
    norms = gs_flat.norm(2, dim=1)
    clip = (C / (norms + 1e-6)).clamp(max=1.0)
    summed = torch.einsum("i,ij->j", clip, gs_flat)
    if summed_in is not None:
        summed = summed + summed_in
    grad = summed if noise_flat is None else summed + noise_flat
    grad = grad / expected_bs
    return norms, clip, summed, grad


# --------------------------------------------------------------------------------------
# Philox4x32-10 counter RNG + Box-Muller, the shared-stream noise of the product kernels
# (north_star: "noise is drawn once from a shared seeded Philox stream").  Integer part is
# bit-exact; the float transform uses float32 numpy log/sin/cos.
# --------------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter [N,4] uint32, key [2] uint32 -> [N,4] uint32."""
    c = counter.astype(np.uint32).copy()
    k0 = np.uint32(key[0])
    k1 = np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c[:, 0].astype(np.uint64)
            p1 = _PHILOX_M1 * c[:, 2].astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask).astype(np.uint32)
            n0 = hi1 ^ c[:, 1] ^ k0
            n1 = lo1
            n2 = hi0 ^ c[:, 3] ^ k1
            n3 = lo0
            c = np.stack([n0, n1, n2, n3], axis=1)
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c


def philox_normal(n: int, seed: int, offset: int, std: float) -> np.ndarray:
    """Element j of the stream = normal #(j & 3) of Philox block (offset + j // 4); key = seed.

    Block counter = (lo32(blk), hi32(blk), 0, 0).  u = ((x >> 8) + 0.5) * 2^-24;
    z0 = r cos(2 pi u1), z1 = r sin(2 pi u1) from (x0, x1); z2, z3 from (x2, x3); r = sqrt(-2 ln u0).
    """
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64) + np.uint64(offset)
    ctr = np.zeros((nblk, 4), np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (blk >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    x = philox4x32_10(ctr, key)
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    out = np.empty((nblk, 4), np.float32)
    two_pi = np.float32(2.0 * math.pi)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, a]))
        th = two_pi * u[:, a + 1]
        out[:, a] = r * np.cos(th)
        out[:, a + 1] = r * np.sin(th)
    return (out.reshape(-1)[:n] * np.float32(std)).astype(np.float32)


# --------------------------------------------------------------------------------------
# Hooked module + optimizer restatement (A.1 + A.3), used as CPU baseline and as the CPU
# stand-in backend for host-logic tests.
# --------------------------------------------------------------------------------------

_SAMPLED = (nn.Linear, nn.Embedding, nn.LayerNorm)


class OracleGradSampleModule:
    """Per-call einsum + accumulate, exactly the hook protocol of A.1 (hooks mode, batch_first, mean)."""

    def __init__(self, module: nn.Module, loss_reduction: str = "mean", sampled_types=_SAMPLED):
        self.module = module
        self.loss_reduction = loss_reduction
        self.handles = []
        for p in module.parameters():
            p.grad_sample = None
            p._forward_counter = 0
        for m in module.modules():
            if isinstance(m, sampled_types) and any(p.requires_grad for p in m.parameters(recurse=False)):
                m._oracle_acts = []
                self.handles.append(m.register_forward_hook(self._fwd))
                self.handles.append(m.register_full_backward_hook(self._bwd))

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def _fwd(self, m, inp, out):
        if not m.training or not torch.is_grad_enabled():
            return
        m._oracle_acts.append(inp[0].detach())
        for p in m.parameters(recurse=False):
            p._forward_counter += 1

    def _bwd(self, m, gin, gout):
        G = gout[0].detach()
        A = m._oracle_acts.pop()
        n = A.shape[0]
        if self.loss_reduction == "mean":
            G = G * n
        if isinstance(m, nn.Linear):
            gw, gb = linear_grad_sample(A, G, m.bias is not None)
            pairs = [(m.weight, gw)] + ([(m.bias, gb)] if m.bias is not None else [])
        elif isinstance(m, nn.Embedding):
            pairs = [(m.weight, embedding_grad_sample(A, G, m.num_embeddings))]
        else:
            gw, gb = layernorm_grad_sample(A, G, m.normalized_shape, m.eps)
            pairs = [(m.weight, gw), (m.bias, gb)]
        for p, g in pairs:
            if hasattr(p, "_current_grad_sample"):
                p._current_grad_sample[: g.shape[0]] += g
            else:
                p._current_grad_sample = g.clone()
            p._forward_counter -= 1
            if p._forward_counter == 0:
                p.grad_sample = p._current_grad_sample
                del p._current_grad_sample

    def remove(self):
        for h in self.handles:
            h.remove()


class OracleDPOptimizer:
    def __init__(self, params: List[nn.Parameter], noise_multiplier: float, max_grad_norm: float,
                 expected_batch_size: float, generator: Optional[torch.Generator] = None):
        self.params = [p for p in params if p.requires_grad]
        self.noise_multiplier = noise_multiplier
        self.max_grad_norm = max_grad_norm
        self.expected_batch_size = expected_batch_size
        self.generator = generator
        self.last_norms = None
        self.last_clip = None

    def zero_grad(self):
        for p in self.params:
            p.grad_sample = None
            p.summed_grad = None
            p.grad = None

    def pre_step(self, noises: Optional[Dict[int, torch.Tensor]] = None, skip: bool = False):
        gs = [p.grad_sample for p in self.params]
        norms, clip, summed = clip_and_accumulate(gs, self.max_grad_norm)
        self.last_norms, self.last_clip = norms, clip
        for p, s in zip(self.params, summed):
            if getattr(p, "summed_grad", None) is not None:
                p.summed_grad = p.summed_grad + s
            else:
                p.summed_grad = s
        if skip:
            return False
        grads = add_noise_and_scale([p.summed_grad for p in self.params],
                                    None if noises is None else noises, self.noise_multiplier,
                                    self.max_grad_norm, self.expected_batch_size, generator=self.generator)
        for p, g in zip(self.params, grads):
            p.grad = g.view_as(p)
        return True


def stream_key(seed: int, purpose: str) -> int:
    """Philox key of one purpose's stream under a user seed: splitmix64(seed ^ tag).  The engine never keys the noise
    kernel and the Poisson sampler with the same word (independence of sampling and noise is an assumption of the
    DP analysis); this restates dp_timeautodiff_amd/rng.py for the checkers."""
    m = (1 << 64) - 1
    tag = {"noise": int.from_bytes(b"noise_dp", "big"), "sampler": int.from_bytes(b"poisson_", "big")}[purpose]
    x = ((int(seed) & m) ^ tag) & m
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def poisson_select(N: int, sample_rate: float, seed: int, offset: int) -> np.ndarray:
    """Indices a Poisson draw selects: j is in iff u_j < sample_rate (float32 compare), u_j = word (j & 3) of Philox
    block (offset + j // 4) mapped to (0,1) as in philox_normal.  Restates Opacus' UniformWithReplacementSampler
    (`(rand(N) < sample_rate).nonzero()`) on the counter-based stream of dpk_poisson_sample."""
    nblk = (N + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64) + np.uint64(offset)
    ctr = np.zeros((nblk, 4), np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (blk >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    x = philox4x32_10(ctr, key)
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    mask = (u.reshape(-1)[:N] < np.float32(sample_rate))
    return np.nonzero(mask)[0].astype(np.int64)
