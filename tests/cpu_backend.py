"""TEST INFRASTRUCTURE: a stand-in for dp_timeautodiff_amd._native built from the CPU oracle.

The product has no CPU path.  To exercise its *host logic* (hook bookkeeping, K=T stacking, arena
windows, skip-step protocol, distributed exchange) in the CPU-only container, tests monkeypatch the
ctypes wrappers of `_native` with these oracle-backed functions.  Nothing outside tests/ imports this.
"""
import numpy as np
import torch

from oracle import opacus_ref as O


def _win(v, B):
    return v.base[:B, v.offset:v.offset + v.numel]


def linear_slots(in_f, out_f):
    return 2


def normsq_slots(P):
    return 1


def linear_grad_sample(A, G, gs_w, gs_b, part, slot0, g_scale, ld_w=None):
    B = A.shape[0]
    if B == 0:
        return
    assert A.is_contiguous() and G.is_contiguous() and A.dim() == 3 and G.dim() == 3
    gw, gb = O.linear_grad_sample(A, G * g_scale, gs_b is not None)
    out_f, in_f = gw.shape[1:]
    ld = in_f if ld_w is None else ld_w
    w = _win(gs_w, B).view(B, out_f, ld)
    w.zero_()
    w[:, :, :in_f] = gw
    part[:B, slot0] = (gw.reshape(B, -1) ** 2).sum(1)
    part[:B, slot0 + 1] = 0.0
    if gs_b is not None:
        _win(gs_b, B).copy_(gb)
        part[:B, slot0 + 1] = (gb ** 2).sum(1)


def embedding_grad_sample(idx, G, gs, V, part, slot0, g_scale):
    B = idx.shape[0]
    if B == 0:
        return
    g = O.embedding_grad_sample(idx, G * g_scale, V)
    _win(gs, B).copy_(g.reshape(B, -1))
    part[:B, slot0] = (g.reshape(B, -1) ** 2).sum(1)


def layernorm_grad_sample(X, G, gs_w, gs_b, part, slot0, eps, g_scale):
    B = X.shape[0]
    if B == 0:
        return
    gw, gb = O.layernorm_grad_sample(X, G * g_scale, (X.shape[-1],), eps)
    _win(gs_w, B).copy_(gw)
    _win(gs_b, B).copy_(gb)
    part[:B, slot0] = (gw ** 2).sum(1) + (gb ** 2).sum(1)


def clip_factors(part, n_slots, clip, norms, B, max_grad_norm, n_valid_dev=None):
    if B == 0:
        return
    n = part[:B, :n_slots].sum(1).sqrt()
    c = (max_grad_norm / (n + 1e-6)).clamp(max=1.0)
    if n_valid_dev is not None:
        pad = torch.arange(B) >= int(n_valid_dev)
        n = n.masked_fill(pad, 0.0)
        c = c.masked_fill(pad, 0.0)
    if norms is not None:
        norms[:B] = n
    clip[:B] = c


def clip_reduce_workspace(B, P):
    return 0


def clip_reduce(gs, clip, summed_in, summed_out, B, workspace=None):
    P = gs.numel
    s = torch.einsum("i,ij->j", clip[:B], _win(gs, B)) if B > 0 else torch.zeros(P)
    if summed_in is not None:
        s = s + summed_in[:P]
    summed_out[:P] = s


def _noise(noise, seed, offset, std, P):
    if noise is not None:
        return noise[:P]
    if std == 0:
        return torch.zeros(P)
    return torch.from_numpy(O.philox_normal(P, seed, offset, std))


def noise_scale(summed, noise, seed, offset, noise_std, inv_scale, grad, P, offset_dev=None):
    if offset_dev is not None:
        offset = offset + int(offset_dev.item())
    s = summed[:P] if summed is not None else torch.zeros(P)
    grad[:P] = (s + _noise(noise, seed, offset, noise_std, P)) * inv_scale


def clip_reduce_noise(gs, clip, summed_in, noise, seed, offset, noise_std, inv_scale, summed_out, grad, B,
                      workspace=None, offset_dev=None):
    P = gs.numel
    tmp = torch.zeros(P)
    clip_reduce(gs, clip, summed_in, tmp, B)
    if summed_out is not None:
        summed_out[:P] = tmp
    noise_scale(tmp, noise, seed, offset, noise_std, inv_scale, grad, P, offset_dev)


def philox_normal(out, seed, offset, std):
    out.copy_(torch.from_numpy(O.philox_normal(out.numel(), seed, offset, std)))


def multi_ok(A, G, ld_w):
    return A.dim() == 3 and G.dim() == 3 and A.shape[1] <= 48 and ld_w % 4 == 0


def linear_grad_sample_multi(jobs, part, g_scale):
    for A, G, gs_w, gs_b, slot0, ld_w in jobs:  # same host protocol, one layer at a time
        linear_grad_sample(A, G, gs_w, gs_b, part, slot0, g_scale, ld_w=ld_w)


PATCHED = ["linear_slots", "normsq_slots", "linear_grad_sample", "linear_grad_sample_multi", "multi_ok", "embedding_grad_sample", "layernorm_grad_sample",
           "clip_factors", "clip_reduce_workspace", "clip_reduce", "noise_scale", "clip_reduce_noise",
           "philox_normal"]


def install(monkeypatch=None):
    """Patch dp_timeautodiff_amd._native with the oracle stand-ins (CPU tests only)."""
    import sys
    from dp_timeautodiff_amd import _native
    me = sys.modules[__name__]
    for name in PATCHED:
        if monkeypatch is not None:
            monkeypatch.setattr(_native, name, getattr(me, name))
        else:
            setattr(_native, name, getattr(me, name))
    if monkeypatch is not None:
        monkeypatch.setattr(_native, "_ALLOW_NON_CUDA", True, raising=False)
    else:
        _native._ALLOW_NON_CUDA = True
