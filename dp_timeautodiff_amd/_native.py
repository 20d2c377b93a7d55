"""ctypes binding of libdpk_gfx950.so (include/dpk.h) — the only compute backend of this package.

There is deliberately no CPU or PyTorch fallback: if the shared library is missing, or a tensor
is not a CUDA(HIP) fp32 tensor, the call raises.  The wrappers check on the host that operand
shapes match what each kernel's grid assumes before anything is launched.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libdpk_gfx950.so"
LIB_PATH = os.environ.get("DPK_LIB_PATH") or os.path.join(_HERE, LIB_NAME)  # override: lab builds only

EXPORTS = {
    # name: (restype, argtypes)
    "dpk_version": (c_int, []),
    "dpk_error_string": (c_char_p, [c_int]),
    "dpk_linear_slots": (c_int, [c_int, c_int]),
    "dpk_embedding_slots": (c_int, [c_int, c_int]),
    "dpk_layernorm_slots": (c_int, [c_int]),
    "dpk_normsq_slots": (c_int, [c_int64]),
    "dpk_linear_configure": (c_int, [c_int, c_int]),
    "dpk_linear_grad_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                       c_int, c_int, c_int, c_int, c_int64, c_int, c_float, c_void_p]),
    "dpk_linear_grad_sample_group": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64,
                                             c_int, c_int, c_int, c_int, c_int64, c_int, c_int, c_float, c_void_p]),
    "dpk_linear_grad_sample_multi": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_float, c_void_p]),
    "dpk_embedding_grad_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_int, c_int, c_int, c_int, c_int64, c_float, c_void_p]),
    "dpk_layernorm_grad_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_int, c_int, c_int, c_float, c_int64, c_float, c_void_p]),
    "dpk_normsq_partials": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "dpk_clip_factors": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "dpk_poisson_sample": (c_int, [c_uint64, c_uint64, c_void_p, c_float, c_int64, c_int, c_void_p, c_void_p,
                                   c_void_p]),
    "dpk_clip_reduce_workspace": (c_int64, [c_int, c_int64]),
    "dpk_clip_reduce": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p,
                                c_int64, c_void_p]),
    "dpk_noise_scale": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, c_float, c_float, c_void_p,
                                c_int64, c_void_p]),
    "dpk_clip_reduce_noise": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64,
                                      c_void_p, c_float, c_float, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int64,
                                      c_void_p]),
    "dpk_clip_reduce_noise_static": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64,
                                             c_void_p, c_float, c_float, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "dpk_philox_normal": (c_int, [c_void_p, c_int64, c_uint64, c_uint64, c_float, c_void_p]),
    "dpk_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "dpk_lstm_pack_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dpk_lstm_pack_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dpk_lstm_forward": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p]),
    "dpk_lstm_backward": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    "dpk_gru_pack_floats": (c_int64, [c_int, c_int]),
    "dpk_gru_pack_forward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dpk_gru_pack_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dpk_gru_forward": (c_int, [c_void_p] * 12 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dpk_gru_backward": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_void_p]),
}


class DpkError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP extension; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DpkError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().dpk_error_string(rc).decode()
        raise DpkError(f"{what} failed: {msg} (code {rc})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DpkError(f"{name}: expected a CUDA/HIP tensor (this package has no CPU path)")
    if t.dtype != dtype:
        raise DpkError(f"{name}: expected dtype {dtype}, got {t.dtype}")


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatView:
    """A [B, n] strided window into the per-sample-gradient arena: base tensor + column offset."""

    __slots__ = ("base", "offset", "numel")

    def __init__(self, base: torch.Tensor, offset: int, numel: int):
        assert base.dim() == 2 and base.stride(1) == 1
        assert 0 <= offset and offset + numel <= base.shape[1]
        self.base, self.offset, self.numel = base, offset, numel

    @property
    def ptr(self):
        return c_void_p(self.base.data_ptr() + 4 * self.offset)

    @property
    def stride(self):
        return self.base.stride(0)


# ---- thin, shape-checked wrappers ---------------------------------------------------------------
_K1_ARITHMETIC = {None: -1, "default": -1, "f32": 0, "split6": 6, "split8": 8, "split9": 9}


def linear_configure(arithmetic="unchanged", role_split="unchanged"):
    """Process-wide choice among K1's large-layer kernels (include/dpk.h: dpk_linear_configure).
    arithmetic: None/"default" | "f32" | "split6" | "split8" | "split9"; role_split: None (default) | True | False."""
    a = -2 if arithmetic == "unchanged" else _K1_ARITHMETIC[arithmetic]
    r = -2 if role_split == "unchanged" else (-1 if role_split is None else int(bool(role_split)))
    _check(load().dpk_linear_configure(a, r), "dpk_linear_configure")




def linear_slots(in_f: int, out_f: int) -> int:
    return load().dpk_linear_slots(in_f, out_f)


def normsq_slots(P: int) -> int:
    return load().dpk_normsq_slots(P)


def linear_grad_sample(A, G, gs_w: FlatView, gs_b: Optional[FlatView], part: torch.Tensor, slot0: int,
                       g_scale: float, ld_w: Optional[int] = None):
    _req(A, "A"); _req(G, "G"); _req(part, "normsq_part"); _req(gs_w.base, "arena")
    if G.dim() == 3 and not G.is_contiguous() and _rows_ld(G) is not None:  # column block of a wider buffer
        return linear_grad_sample_group([(A, G, gs_w, gs_b, slot0)], part, g_scale, ld_w)
    if A.dim() != 3 or G.dim() != 3 or not A.is_contiguous() or not G.is_contiguous():
        raise DpkError("A and G must be contiguous [B,T,features]")
    B, T, in_f = A.shape
    if G.shape[0] != B or G.shape[1] != T:
        raise DpkError(f"A {tuple(A.shape)} and G {tuple(G.shape)} disagree on [B,T]")
    out_f = G.shape[2]
    if B == 0:
        return
    ld_w = in_f if ld_w is None else int(ld_w)
    if ld_w < in_f or gs_w.numel != ld_w * out_f or (gs_b is not None and gs_b.numel != out_f):
        raise DpkError("grad-sample window does not match the layer shape")
    if gs_w.base.shape[0] < B or part.shape[0] < B or part.dim() != 2 or part.stride(1) != 1:
        raise DpkError("arena / partial buffer has fewer rows than the batch")
    if gs_b is not None and gs_b.base.data_ptr() != gs_w.base.data_ptr():
        raise DpkError("weight and bias windows must live in the same arena")
    rc = load().dpk_linear_grad_sample(_ptr(A), _ptr(G), gs_w.ptr, None if gs_b is None else gs_b.ptr,
                                       _ptr(part), part.stride(0), slot0, B, T, in_f, out_f, gs_w.stride, ld_w,
                                       float(g_scale), _stream())
    _check(rc, "dpk_linear_grad_sample")


K1_MAX_JOBS = 8


def _rows_ld(G):
    """Row stride of a [B,T,out] operand whose rows may be column blocks of a wider buffer; None if unsupported."""
    if G.dim() != 3 or G.stride(2) != 1 or G.stride(0) != G.shape[1] * G.stride(1) or G.stride(1) < G.shape[2]:
        return None
    return G.stride(1)


def linear_grad_sample_group(jobs, part: torch.Tensor, g_scale: float, ld_w: Optional[int] = None):
    """jobs: list of (A, G, gs_w, gs_b or None, slot0) of ONE shape; one launch (dpk_linear_grad_sample_group).
    G may be a column block of a wider [B,T,W] buffer (row stride ld_g = W, the same for every job)."""
    if not 1 <= len(jobs) <= K1_MAX_JOBS:
        raise DpkError(f"1..{K1_MAX_JOBS} jobs per group, got {len(jobs)}")
    A0, G0, w0, _, _ = jobs[0]
    _req(part, "normsq_part")
    if A0.dim() != 3 or G0.dim() != 3:
        raise DpkError("A and G must be [B,T,features]")
    B, T, in_f = A0.shape
    out_f = G0.shape[2]
    ld_g = _rows_ld(G0)
    ld_w = in_f if ld_w is None else int(ld_w)
    if part.shape[0] < B or part.dim() != 2 or part.stride(1) != 1:
        raise DpkError("partial buffer has fewer rows than the batch")
    for A, G, gs_w, gs_b, slot0 in jobs:
        _req(A, "A"); _req(G, "G"); _req(gs_w.base, "arena")
        if tuple(A.shape) != (B, T, in_f) or tuple(G.shape) != (B, T, out_f) or not A.is_contiguous():
            raise DpkError("all jobs of a group must share one [B,T,in] / [B,T,out] shape, A contiguous")
        if ld_g is None or _rows_ld(G) != ld_g:
            raise DpkError("G must be [B,T,out] rows of one common stride (contiguous, or a column block)")
        if ld_w < in_f or gs_w.numel != ld_w * out_f or (gs_b is not None and gs_b.numel != out_f):
            raise DpkError("grad-sample window does not match the layer shape")
        if gs_w.base.shape[0] < B or gs_w.stride != w0.stride:
            raise DpkError("arena has fewer rows than the batch / jobs live in different arenas")
        if gs_b is not None and gs_b.base.data_ptr() != gs_w.base.data_ptr():
            raise DpkError("weight and bias windows must live in the same arena")
    if B == 0:
        return
    n = len(jobs)
    arr = lambda vals: (c_void_p * n)(*vals)
    pa = arr([j[0].data_ptr() for j in jobs]); pg = arr([j[1].data_ptr() for j in jobs])
    pw = arr([j[2].ptr.value for j in jobs])
    pb = arr([None if j[3] is None else j[3].ptr.value for j in jobs])
    slots = (c_int * n)(*[int(j[4]) for j in jobs])
    rc = load().dpk_linear_grad_sample_group(pa, pg, pw, pb, slots, n, _ptr(part), part.stride(0), B, T, in_f, out_f,
                                             w0.stride, ld_w, ld_g, float(g_scale), _stream())
    _check(rc, "dpk_linear_grad_sample_group")


K1_MULTI_MAX_JOBS = 32
K1_MULTI_MAX_T = 48


class _K1Job(ctypes.Structure):  # dpk_k1_job (include/dpk.h)
    _fields_ = [("A", c_void_p), ("G", c_void_p), ("GSw", c_void_p), ("GSb", c_void_p), ("slot0", c_int), ("T", c_int),
                ("in_features", c_int), ("out_features", c_int), ("ld_w", c_int), ("ld_g", c_int)]


def multi_ok(A, G, ld_w: int) -> bool:
    """Can this layer's contraction join a dpk_linear_grad_sample_multi launch?"""
    return (A.dim() == 3 and G.dim() == 3 and A.is_contiguous() and _rows_ld(G) is not None
            and A.shape[1] <= K1_MULTI_MAX_T and ld_w % 4 == 0)


def linear_grad_sample_multi(jobs, part: torch.Tensor, g_scale: float):
    """jobs: list of (A, G, gs_w, gs_b or None, slot0, ld_w) of ANY shapes sharing B and the arena; launches of up to
    K1_MULTI_MAX_JOBS layers each (dpk_linear_grad_sample_multi)."""
    if not jobs:
        return
    _req(part, "normsq_part")
    B = jobs[0][0].shape[0]
    w0 = jobs[0][2]
    if part.shape[0] < B or part.dim() != 2 or part.stride(1) != 1:
        raise DpkError("partial buffer has fewer rows than the batch")
    recs = []
    for A, G, gs_w, gs_b, slot0, ld_w in jobs:
        _req(A, "A"); _req(G, "G"); _req(gs_w.base, "arena")
        if A.dim() != 3 or G.dim() != 3 or not A.is_contiguous() or A.shape[0] != B or G.shape[:2] != A.shape[:2]:
            raise DpkError("every job needs A [B,T,in] contiguous and G [B,T,out] with one common B")
        ld_g = _rows_ld(G)
        in_f, out_f = A.shape[2], G.shape[2]
        if ld_g is None:
            raise DpkError("G must be [B,T,out] rows of one stride (contiguous, or a column block)")
        if ld_w < in_f or gs_w.numel != ld_w * out_f or (gs_b is not None and gs_b.numel != out_f):
            raise DpkError("grad-sample window does not match the layer shape")
        if gs_w.base.shape[0] < B or gs_w.stride != w0.stride or gs_w.base.data_ptr() != w0.base.data_ptr():
            raise DpkError("arena has fewer rows than the batch / jobs live in different arenas")
        if gs_b is not None and gs_b.base.data_ptr() != gs_w.base.data_ptr():
            raise DpkError("weight and bias windows must live in the same arena")
        recs.append(_K1Job(_ptr(A), _ptr(G), gs_w.ptr, None if gs_b is None else gs_b.ptr, int(slot0), A.shape[1],
                           in_f, out_f, int(ld_w), int(ld_g)))
    if B == 0:
        return
    lib = load()
    for i in range(0, len(recs), K1_MULTI_MAX_JOBS):
        chunk = recs[i:i + K1_MULTI_MAX_JOBS]
        arr = (_K1Job * len(chunk))(*chunk)
        rc = lib.dpk_linear_grad_sample_multi(ctypes.cast(arr, c_void_p), len(chunk), _ptr(part), part.stride(0), B,
                                              w0.stride, float(g_scale), _stream())
        _check(rc, "dpk_linear_grad_sample_multi")


def embedding_grad_sample(idx, G, gs: FlatView, V: int, part: torch.Tensor, slot0: int, g_scale: float):
    _req(idx, "idx", torch.int64); _req(G, "G"); _req(part, "normsq_part")
    if idx.dim() != 2 or G.dim() != 3 or not idx.is_contiguous() or not G.is_contiguous():
        raise DpkError("idx must be contiguous [B,T] and G contiguous [B,T,E]")
    B, T = idx.shape
    E = G.shape[2]
    if B == 0:
        return
    if G.shape[0] != B or G.shape[1] != T or gs.numel != V * E:
        raise DpkError("embedding grad-sample shapes disagree")
    if gs.base.shape[0] < B or part.shape[0] < B:
        raise DpkError("arena / partial buffer has fewer rows than the batch")
    rc = load().dpk_embedding_grad_sample(_ptr(idx), _ptr(G), gs.ptr, _ptr(part), part.stride(0), slot0, B, T, V, E,
                                          gs.stride, float(g_scale), _stream())
    _check(rc, "dpk_embedding_grad_sample")


def layernorm_grad_sample(X, G, gs_w: FlatView, gs_b: FlatView, part: torch.Tensor, slot0: int, eps: float,
                          g_scale: float):
    _req(X, "X"); _req(G, "G"); _req(part, "normsq_part")
    if X.dim() != 3 or X.shape != G.shape or not X.is_contiguous() or not G.is_contiguous():
        raise DpkError("X and G must be contiguous [B,T,C] of equal shape")
    B, T, C = X.shape
    if B == 0:
        return
    if gs_w.numel != C or gs_b.numel != C or gs_w.base.shape[0] < B or part.shape[0] < B:
        raise DpkError("layernorm grad-sample shapes disagree")
    rc = load().dpk_layernorm_grad_sample(_ptr(X), _ptr(G), gs_w.ptr, gs_b.ptr, _ptr(part), part.stride(0), slot0,
                                          B, T, C, float(eps), gs_w.stride, float(g_scale), _stream())
    _check(rc, "dpk_layernorm_grad_sample")


def normsq_partials(gs: FlatView, B: int, part: torch.Tensor, slot0: int):
    _req(gs.base, "GS"); _req(part, "normsq_part")
    if gs.base.shape[0] < B or part.shape[0] < B:
        raise DpkError("arena / partial buffer has fewer rows than the batch")
    rc = load().dpk_normsq_partials(gs.ptr, gs.stride, _ptr(part), part.stride(0), slot0, B, gs.numel, _stream())
    _check(rc, "dpk_normsq_partials")


def clip_factors(part: torch.Tensor, n_slots: int, clip: torch.Tensor, norms: Optional[torch.Tensor], B: int,
                 max_grad_norm: float, n_valid_dev: Optional[torch.Tensor] = None):
    _req(part, "normsq_part"); _req(clip, "clip")
    if n_valid_dev is not None:
        _req(n_valid_dev, "n_valid", torch.int32)
    if part.shape[0] < B or clip.numel() < B or (norms is not None and norms.numel() < B) or n_slots > part.shape[1]:
        raise DpkError("clip_factors: buffers shorter than the batch / slots")
    rc = load().dpk_clip_factors(_ptr(part), part.stride(0), n_slots, _ptr(clip), _ptr(norms), B,
                                 float(max_grad_norm), _ptr(n_valid_dev), _stream())
    _check(rc, "dpk_clip_factors")


def clip_reduce_workspace(B: int, P: int) -> int:
    return int(load().dpk_clip_reduce_workspace(B, P))


def _ws_args(workspace: Optional[torch.Tensor]):
    if workspace is None:
        return None, 0
    return _ptr(workspace), workspace.numel() * workspace.element_size()


def clip_reduce(gs: FlatView, clip, summed_in, summed_out, B: int, workspace=None):
    _req(gs.base, "GS"); _req(clip, "clip"); _req(summed_out, "summed_out")
    P = gs.numel
    if clip.numel() < B or gs.base.shape[0] < B or summed_out.numel() < P or (summed_in is not None and summed_in.numel() < P):
        raise DpkError("clip_reduce: buffers shorter than [B] / [P]")
    wp, wb = _ws_args(workspace)
    rc = load().dpk_clip_reduce(gs.ptr, gs.stride, _ptr(clip), _ptr(summed_in), _ptr(summed_out), B, P, wp, wb,
                                _stream())
    _check(rc, "dpk_clip_reduce")


def noise_scale(summed, noise, seed: int, offset: int, noise_std: float, inv_scale: float, grad, P: int,
                offset_dev=None):
    _req(grad, "grad")
    if grad.numel() < P or (summed is not None and summed.numel() < P) or (noise is not None and noise.numel() < P):
        raise DpkError("noise_scale: buffers shorter than P")
    rc = load().dpk_noise_scale(_ptr(summed), _ptr(noise), seed & (2**64 - 1), offset & (2**64 - 1),
                                _ptr(offset_dev), float(noise_std), float(inv_scale), _ptr(grad), P, _stream())
    _check(rc, "dpk_noise_scale")


def clip_reduce_noise(gs: FlatView, clip, summed_in, noise, seed: int, offset: int, noise_std: float,
                      inv_scale: float, summed_out, grad, B: int, workspace=None, offset_dev=None):
    _req(gs.base, "GS"); _req(clip, "clip"); _req(grad, "grad")
    P = gs.numel
    for t, n in ((summed_in, "summed_in"), (noise, "noise"), (summed_out, "summed_out"), (grad, "grad")):
        if t is not None and t.numel() < P:
            raise DpkError(f"clip_reduce_noise: {n} shorter than P")
    if clip.numel() < B or gs.base.shape[0] < B:
        raise DpkError("clip_reduce_noise: buffers shorter than the batch")
    wp, wb = _ws_args(workspace)
    rc = load().dpk_clip_reduce_noise(gs.ptr, gs.stride, _ptr(clip), _ptr(summed_in), _ptr(noise),
                                      seed & (2**64 - 1), offset & (2**64 - 1), _ptr(offset_dev), float(noise_std),
                                      float(inv_scale), _ptr(summed_out), _ptr(grad), B, P, wp, wb, _stream())
    _check(rc, "dpk_clip_reduce_noise")


STATIC_SCRATCH_BYTES = 16 * 1024 * 1024


def clip_reduce_noise_static(gs: FlatView, clip, summed_in, noise, seed: int, offset: int, noise_std: float,
                             inv_scale: float, summed_out, grad, B: int, offset_dev=None):
    """dpk_clip_reduce_noise without a workspace argument: the split-B partials live in the library's own static
    device array (no allocation on the call path).  Calls on one GPU must not overlap."""
    _req(gs.base, "GS"); _req(clip, "clip"); _req(grad, "grad")
    P = gs.numel
    for t, n in ((summed_in, "summed_in"), (noise, "noise"), (summed_out, "summed_out"), (grad, "grad")):
        if t is not None and t.numel() < P:
            raise DpkError(f"clip_reduce_noise_static: {n} shorter than P")
    if clip.numel() < B or gs.base.shape[0] < B:
        raise DpkError("clip_reduce_noise_static: buffers shorter than the batch")
    rc = load().dpk_clip_reduce_noise_static(gs.ptr, gs.stride, _ptr(clip), _ptr(summed_in), _ptr(noise),
                                             seed & (2**64 - 1), offset & (2**64 - 1), _ptr(offset_dev),
                                             float(noise_std), float(inv_scale), _ptr(summed_out), _ptr(grad), B, P,
                                             _stream())
    _check(rc, "dpk_clip_reduce_noise_static")


def philox_normal(out: torch.Tensor, seed: int, offset: int, std: float):
    _req(out, "out")
    rc = load().dpk_philox_normal(_ptr(out), out.numel(), seed & (2**64 - 1), offset & (2**64 - 1), float(std),
                                  _stream())
    _check(rc, "dpk_philox_normal")


# ---- K8: fused Linear-cell GRU recurrence -------------------------------------------------------------------------
# Per-GRU inputs travel as host arrays of device pointers; outputs and packed weights carry a leading [G] dimension.

GRU_MAX_G = 8


def gru_pack_floats(O: int, K: int) -> int:
    return int(load().dpk_gru_pack_floats(O, K))


def _ptr_array(tensors, name, shape=None, allow_none=False):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is None:
            if not allow_none:
                raise DpkError(f"{name}[{i}] is None")
            arr[i] = None
            continue
        _req(t, f"{name}[{i}]")
        if not t.is_contiguous() or (shape is not None and tuple(t.shape) != tuple(shape)):
            raise DpkError(f"{name}[{i}] must be contiguous {tuple(shape) if shape else ''}, got {tuple(t.shape)}")
        arr[i] = t.data_ptr()
    return arr


def _gru_weights(weights, H):
    """weights: 3 per GRU (update, reset, new), each nn.Linear(2H, H).weight -> pointer array, G, device."""
    if len(weights) == 0 or len(weights) % 3:
        raise DpkError("gru_pack: expected 3 weights (update, reset, new) per GRU")
    G = len(weights) // 3
    if G > GRU_MAX_G:
        raise DpkError(f"at most {GRU_MAX_G} GRUs per launch")
    return _ptr_array(weights, "W", (H, 2 * H)), G, weights[0].device


def gru_pack_forward(weights, H: int):
    """Hidden halves W[:, H:] of the gates in MFMA fragment order: Pz, Pr, Pn [G, packed(H,H)]."""
    arr, G, dev = _gru_weights(weights, H)
    n = gru_pack_floats(H, H)
    Pz, Pr, Pn = (torch.empty(G, n, device=dev, dtype=torch.float32) for _ in range(3))
    _check(load().dpk_gru_pack_forward(arr, 2 * H, _ptr(Pz), _ptr(Pr), _ptr(Pn), H, G, _stream()),
           "dpk_gru_pack_forward")
    return Pz, Pr, Pn


def gru_pack_backward(weights, H: int):
    """Transposed hidden halves: Pnh [G, packed(H,H)], Pzrh [G, packed(H,2H)]."""
    arr, G, dev = _gru_weights(weights, H)
    Pnh = torch.empty(G, gru_pack_floats(H, H), device=dev, dtype=torch.float32)
    Pzrh = torch.empty(G, gru_pack_floats(H, 2 * H), device=dev, dtype=torch.float32)
    _check(load().dpk_gru_pack_backward(arr, 2 * H, _ptr(Pnh), _ptr(Pzrh), H, G, _stream()), "dpk_gru_pack_backward")
    return Pnh, Pzrh


def gru_forward(xps, xgs, h0s, Pz, Pr, Pn):
    """xps: G tensors [B,T,H]; xgs: G tensors [B,T,3H] = x-side pre-activations (update | reset | new, biases
    included); h0s: None or G tensors [B,H].
    Returns out, Z, R, N [G,B,T,H] and the stacked gate inputs Azr, An [G,B,T,2H]."""
    G = len(xps)
    if G == 0 or G > GRU_MAX_G or len(xgs) != G or (h0s is not None and len(h0s) != G):
        raise DpkError("gru_forward: inconsistent number of GRUs")
    _req(xps[0], "xp[0]")
    if xps[0].dim() != 3:
        raise DpkError("gru_forward: xp must be [B,T,H]")
    B, T, H = xps[0].shape
    if B == 0 or T == 0:
        raise DpkError("gru_forward: empty batch / sequence")
    xa = _ptr_array(xps, "xp", (B, T, H))
    ga = _ptr_array(xgs, "xg", (B, T, 3 * H))
    ha = None if h0s is None else _ptr_array(h0s, "h0", (B, H))
    n_pack = gru_pack_floats(H, H)
    for t, n in ((Pz, "Pz"), (Pr, "Pr"), (Pn, "Pn")):
        _req(t, n)
        if tuple(t.shape) != (G, n_pack) or not t.is_contiguous():
            raise DpkError(f"gru_forward: {n} is not [G, packed(H,H)]")
    dev = xps[0].device
    out, Z, R, N = (torch.empty(G, B, T, H, device=dev, dtype=torch.float32) for _ in range(4))
    Azr, An = (torch.empty(G, B, T, 2 * H, device=dev, dtype=torch.float32) for _ in range(2))
    rc = load().dpk_gru_forward(xa, ga, ha, _ptr(Pz), _ptr(Pr), _ptr(Pn), _ptr(out), _ptr(Z), _ptr(R), _ptr(N),
                                _ptr(Azr), _ptr(An), B, T, H, G, _stream())
    _check(rc, "dpk_gru_forward")
    return out, Z, R, N, Azr, An


def gru_backward(douts, dhTs, Z, R, N, Azr, Pnh, Pzrh, want_dh0: bool):
    """douts: G tensors [B,T,H] (None entries = zeros).  Returns G3 [G,B,T,3H] = [Gz | Gr | Gn] (the gates'
    pre-activation gradients = dL/dxg), dh0 [G,B,H] or None."""
    _req(Z, "Z")
    if Z.dim() != 4 or not Z.is_contiguous():
        raise DpkError("gru_backward: Z must be contiguous [G,B,T,H]")
    G, B, T, H = Z.shape
    if len(douts) != G or (dhTs is not None and len(dhTs) != G):
        raise DpkError("gru_backward: inconsistent number of GRUs")
    for t, n in ((R, "R"), (N, "N")):
        _req(t, n)
        if tuple(t.shape) != (G, B, T, H) or not t.is_contiguous():
            raise DpkError(f"gru_backward: {n} does not match Z")
    _req(Azr, "Azr")
    if tuple(Azr.shape) != (G, B, T, 2 * H) or not Azr.is_contiguous():
        raise DpkError("gru_backward: Azr must be contiguous [G,B,T,2H]")
    for t, n, cnt in ((Pnh, "Pnh", gru_pack_floats(H, H)), (Pzrh, "Pzrh", gru_pack_floats(H, 2 * H))):
        _req(t, n)
        if tuple(t.shape) != (G, cnt) or not t.is_contiguous():
            raise DpkError(f"gru_backward: {n} has the wrong packed size")
    da = _ptr_array(douts, "dout", (B, T, H), allow_none=True)
    ha = None if dhTs is None else _ptr_array(dhTs, "dhT", (B, H), allow_none=True)
    G3 = torch.empty(G, B, T, 3 * H, device=Z.device, dtype=torch.float32)
    dh0 = torch.empty(G, B, H, device=Z.device, dtype=torch.float32) if want_dh0 else None
    rc = load().dpk_gru_backward(da, ha, _ptr(Z), _ptr(R), _ptr(N), _ptr(Azr), _ptr(Pnh), _ptr(Pzrh), _ptr(G3),
                                 _ptr(dh0), B, T, H, G, _stream())
    _check(rc, "dpk_gru_backward")
    return G3, dh0


# ---- K10: fused DPLSTM recurrence -----------------------------------------------------------------------------------

LSTM_MAX_HIDDEN = 256


def lstm_pack_forward(whh, H: int):
    G = len(whh)
    arr = _ptr_array(whh, "Whh", (4 * H, H))
    P = torch.empty(G, 4, gru_pack_floats(H, H), device=whh[0].device, dtype=torch.float32)
    _check(load().dpk_lstm_pack_forward(arr, _ptr(P), H, G, _stream()), "dpk_lstm_pack_forward")
    return P


def lstm_pack_backward(whh, H: int):
    G = len(whh)
    arr = _ptr_array(whh, "Whh", (4 * H, H))
    Pt = torch.empty(G, gru_pack_floats(H, 4 * H), device=whh[0].device, dtype=torch.float32)
    _check(load().dpk_lstm_pack_backward(arr, _ptr(Pt), H, G, _stream()), "dpk_lstm_pack_backward")
    return Pt


def lstm_forward(xgs, h0s, c0s, bhhs, P, reverse_mask: int):
    """xgs: G tensors [B,T,4H]; h0s/c0s: None or G tensors [B,H]; bhhs: None or G tensors [4H].
    Returns out [B,T,G*H] (recurrence g in column block g), gates [G,B,T,4H], C, Hprev [G,B,T,H]."""
    G = len(xgs)
    _req(xgs[0], "xg[0]")
    if G == 0 or G > GRU_MAX_G or xgs[0].dim() != 3 or xgs[0].shape[2] % 4:
        raise DpkError("lstm_forward: xg must be G <= 8 tensors [B,T,4H]")
    B, T, H4 = xgs[0].shape
    H = H4 // 4
    if B == 0 or T == 0:
        raise DpkError("lstm_forward: empty batch / sequence")
    xa = _ptr_array(xgs, "xg", (B, T, 4 * H))
    ha = None if h0s is None else _ptr_array(h0s, "h0", (B, H))
    ca = None if c0s is None else _ptr_array(c0s, "c0", (B, H))
    ba = None if bhhs is None else _ptr_array(bhhs, "bhh", (4 * H,))
    _req(P, "P")
    if tuple(P.shape) != (G, 4, gru_pack_floats(H, H)) or not P.is_contiguous():
        raise DpkError("lstm_forward: P is not [G, 4, packed(H,H)]")
    dev = xgs[0].device
    C, Hprev = (torch.empty(G, B, T, H, device=dev, dtype=torch.float32) for _ in range(2))
    out = torch.empty(B, T, G * H, device=dev, dtype=torch.float32)
    gates = torch.empty(G, B, T, 4 * H, device=dev, dtype=torch.float32)
    rc = load().dpk_lstm_forward(xa, ha, ca, ba, _ptr(P), _ptr(out), _ptr(gates), _ptr(C), _ptr(Hprev),
                                 int(reverse_mask), B, T, H, G, _stream())
    _check(rc, "dpk_lstm_forward")
    return out, gates, C, Hprev


def lstm_backward(douts, c0s, gates, C, Pt, reverse_mask: int, want_d0: bool, dcTs=None):
    """douts: G tensors [B,T,H] (None = zeros), all dense or all column blocks of one [B,T,W] gradient buffer.
    dcTs: None or G tensors [B,H] = gradient of each recurrence's final cell state (None entries = zeros).
    Returns dpre [G,B,T,4H] and (dh0, dc0) [G,B,H] or (None, None)."""
    _req(C, "C"); _req(gates, "gates"); _req(Pt, "Pt")
    if C.dim() != 4 or not C.is_contiguous():
        raise DpkError("lstm_backward: C must be contiguous [G,B,T,H]")
    G, B, T, H = C.shape
    if tuple(gates.shape) != (G, B, T, 4 * H) or not gates.is_contiguous():
        raise DpkError("lstm_backward: gates must be contiguous [G,B,T,4H]")
    if tuple(Pt.shape) != (G, gru_pack_floats(H, 4 * H)) or not Pt.is_contiguous():
        raise DpkError("lstm_backward: Pt has the wrong packed size")
    if len(douts) != G or (c0s is not None and len(c0s) != G):
        raise DpkError("lstm_backward: inconsistent number of recurrences")
    lds = {_rows_ld(d) for d in douts if d is not None}
    if None in lds or len(lds) > 1 or any(d is not None and tuple(d.shape) != (B, T, H) for d in douts):
        raise DpkError("lstm_backward: dout entries must be [B,T,H] rows of one common stride")
    ld_dout = lds.pop() if lds else H
    da = (c_void_p * G)()
    for i, d in enumerate(douts):
        if d is not None:
            _req(d, f"dout[{i}]")
            da[i] = d.data_ptr()
    ca = None if c0s is None else _ptr_array(c0s, "c0", (B, H))
    if dcTs is not None and len(dcTs) != G:
        raise DpkError("lstm_backward: dcT must have one entry per recurrence")
    dca = None if dcTs is None or all(d is None for d in dcTs) else _ptr_array(dcTs, "dcT", (B, H), allow_none=True)
    dpre = torch.empty_like(gates)
    dh0 = torch.empty(G, B, H, device=C.device, dtype=torch.float32) if want_d0 else None
    dc0 = torch.empty(G, B, H, device=C.device, dtype=torch.float32) if want_d0 else None
    rc = load().dpk_lstm_backward(da, dca, ca, _ptr(gates), _ptr(C), _ptr(Pt), _ptr(dpre), _ptr(dh0), _ptr(dc0),
                                  int(reverse_mask), B, T, H, G, int(ld_dout), _stream())
    _check(rc, "dpk_lstm_backward")
    return dpre, dh0, dc0


# ---- K9: Adam on the arena-layout gradient ------------------------------------------------------------------------


def adam_step(grad, exp_avg, exp_avg_sq, win_table, hyper, step_dev, P: int):
    for t, n in ((grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (hyper, "hyper"),
                 (step_dev, "step")):
        _req(t, n)
    _req(win_table, "win_table", torch.int64)
    n = win_table.shape[0]
    if win_table.dim() != 2 or win_table.shape[1] != 5 or tuple(hyper.shape) != (n, 8) or n == 0:
        raise DpkError("adam_step: tables must be [n,5] int64 and [n,8] float32")
    if not (win_table.is_contiguous() and hyper.is_contiguous()):
        raise DpkError("adam_step: tables must be contiguous")
    if min(grad.numel(), exp_avg.numel(), exp_avg_sq.numel()) < P or step_dev.numel() != 1:
        raise DpkError("adam_step: vectors shorter than P")
    rc = load().dpk_adam_step(_ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(win_table), _ptr(hyper), n,
                              _ptr(step_dev), P, _stream())
    _check(rc, "dpk_adam_step")


# ---- K11: on-device Poisson sampler -------------------------------------------------------------------------------


def poisson_sample(seed: int, offset: int, sample_rate: float, N: int, idx: torch.Tensor, counts: torch.Tensor,
                   offset_dev: Optional[torch.Tensor] = None):
    """idx: int64 [capacity] (selected indices, ascending, padded with 0); counts: int32 [2] = (n kept, n selected)."""
    _req(idx, "idx", torch.int64); _req(counts, "counts", torch.int32)
    if not idx.is_contiguous() or counts.numel() != 2 or idx.numel() == 0:
        raise DpkError("poisson_sample: idx must be a contiguous int64 vector, counts int32[2]")
    rc = load().dpk_poisson_sample(seed & (2**64 - 1), offset & (2**64 - 1), _ptr(offset_dev), float(sample_rate),
                                   int(N), idx.numel(), _ptr(idx), _ptr(counts), _stream())
    _check(rc, "dpk_poisson_sample")
