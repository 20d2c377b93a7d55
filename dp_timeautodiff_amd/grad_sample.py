"""GradSampleModule (hooks mode) on top of the HIP per-sample-gradient kernels.

Drop-in for the object `PrivacyEngine.make_private_with_epsilon` returns as `ae` at
Model_Code/timeautoencoder.py:674 (Opacus' `GradSampleModule`):

  * forward hooks stash the (detached) input of every call of every Linear / Embedding / LayerNorm
    (fired by `ae(batch_input)`, :734) and bump the parameters' `_forward_counter`;
  * backward hooks (fired by `loss_Auto.backward()`, :749) pair each stashed input with its
    grad_output; when a layer's counter returns to 0 its per-sample gradient is materialised.

MI355X-first differences (observable state after backward is the same: `p.grad_sample` of shape
[B, *p.shape], fp32, on the parameter's device):

  * a layer invoked T times per forward (the CustomGRU gates, timeautoencoder.py:84-94; DPLSTM cells)
    is NOT contracted T times into a [B, out, in] buffer.  Its T (input, grad_output) pairs are stacked
    and contracted by ONE K=T MFMA GEMM launch (dpk_linear_grad_sample);
  * all per-sample gradients live in one persistent flat arena [B_capacity, P] (one allocation, reused
    every step; `p.grad_sample` is a strided view into it), so the optimizer's clip/sum/noise pass is a
    single launch over the whole parameter vector;
  * every kernel also emits the squared norm of what it wrote, so the norm pass never re-reads HBM.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _native
from .layers import RNNLinear, SequenceBias
from .layers.fused_gru import FusedLinearCellGRU

ARENA_ALIGN = 64  # floats: every parameter window starts on a 256-byte boundary


def _row_stride(p: nn.Parameter, pad: bool = True) -> int:
    """Arena row stride of a 2-D weight: rows of awkward width (>= 64 and not a multiple of 32 floats) are padded
    to the next multiple of 32 so that K1 writes whole 128-byte lines (non-temporal) — e.g. 250 -> 256, 400 -> 416;
    narrower rows are padded to a multiple of 4 floats (13 -> 16, 7 -> 8) so that every Linear layer's rows can be
    stored 16 bytes at a time and join the small-batch multi-layer launch.  K1 writes exact zeros into the padding."""
    if p.dim() != 2:
        return p.shape[-1] if p.dim() else 1
    n = p.shape[1]
    if not pad:
        return n
    if n >= 64:
        return (n + 31) // 32 * 32 if n % 32 else n
    return (n + 3) // 4 * 4


class _ParamWindow:
    """Where one parameter's per-sample gradient lives inside an arena row (and inside summed/grad)."""
    __slots__ = ("name", "param", "offset", "numel", "ld")

    def __init__(self, name, param, offset, pad_rows=True):
        self.name, self.param, self.offset = name, param, offset
        self.ld = _row_stride(param, pad_rows)
        self.numel = param.shape[0] * self.ld if param.dim() == 2 else param.numel()

    def view(self, storage: torch.Tensor, lead: Tuple[int, ...] = (), lead_stride: Tuple[int, ...] = (),
             base: int = 0) -> torch.Tensor:
        p = self.param
        if p.dim() == 2:
            size, stride = tuple(p.shape), (self.ld, 1)
        else:
            size = tuple(p.shape)
            stride, acc = [], 1
            for d in reversed(size):
                stride.insert(0, acc)
                acc *= d
            stride = tuple(stride)
        return storage.as_strided(lead + size, lead_stride + stride, base + self.offset)


class GradSampleArena:
    """Flat device storage of one step's per-sample gradients + the DP tail's small buffers.

    buf     [capacity, P]   per-sample gradients (row = sample); P padded so rows are 256-B aligned
    part    [capacity, S]   squared-norm partials, S = sum of every layer's slots
    clip / norms [capacity]
    summed, grad [P]        clipped sum (p.summed_grad views) and final gradient (p.grad views)
    """

    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]], device: torch.device, dense=()):
        """`dense`: ids of 2-D parameters whose rows must stay unpadded (everything but Linear weights: K1 is the
        only kernel that writes rows with a stride)."""
        self.device = device
        self.windows: Dict[int, _ParamWindow] = {}
        off = 0
        for name, p in named_params:
            if not p.requires_grad or id(p) in self.windows:
                continue
            if p.dtype != torch.float32:
                raise ValueError(f"{name}: only fp32 parameters are supported (got {p.dtype})")
            self.windows[id(p)] = _ParamWindow(name, p, off, pad_rows=id(p) not in dense)
            off += (self.windows[id(p)].numel + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
        self.P = max(off, ARENA_ALIGN)
        self.n_slots = 0
        self.capacity = 0
        self.buf = None
        self.part = None
        self.clip = None
        self.norms = None
        self.summed = torch.zeros(self.P, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.P, device=device, dtype=torch.float32)
        self.workspace = None
        self.summed_valid = False
        self.pass_B: List[int] = []  # batch size of every un-stepped backward pass
        self.consumed = False

    # -- allocation -----------------------------------------------------------------------------------
    def reserve_slots(self, n: int) -> int:
        s0 = self.n_slots
        self.n_slots += n
        return s0

    def ensure(self, rows: int):
        if rows <= self.capacity:
            return
        new_cap = max(rows, int(self.capacity * 1.25) + 8)
        new_cap = (new_cap + 7) // 8 * 8
        old = self.buf
        live = sum(self.pass_B[:-1]) if len(self.pass_B) > 1 else 0
        S = max((self.n_slots + 3) // 4 * 4, 4)
        buf = torch.zeros(new_cap, self.P, device=self.device, dtype=torch.float32)
        part = torch.zeros(new_cap, S, device=self.device, dtype=torch.float32)
        if old is not None and live > 0:  # keep earlier passes of an accumulating step
            buf[:live].copy_(old[:live])
            part[:live].copy_(self.part[:live])
        self.buf, self.part = buf, part
        self.clip = torch.zeros(new_cap, device=self.device, dtype=torch.float32)
        self.norms = torch.zeros(new_cap, device=self.device, dtype=torch.float32)
        self.capacity = new_cap
        del old

    def ensure_workspace(self, B: int):
        need = _native.clip_reduce_workspace(B, self.P)
        if need and (self.workspace is None or self.workspace.numel() * 4 < need):
            self.workspace = torch.empty((need + 3) // 4, device=self.device, dtype=torch.float32)
        return self.workspace if need else None

    # -- views ----------------------------------------------------------------------------------------
    def window(self, p: nn.Parameter) -> _ParamWindow:
        return self.windows[id(p)]

    def flat(self, p: nn.Parameter, row0: int = 0) -> _native.FlatView:
        w = self.windows[id(p)]
        return _native.FlatView(self.buf[row0:] if row0 else self.buf, w.offset, w.numel)

    def sample_view(self, p: nn.Parameter, row0: int, rows: int) -> torch.Tensor:
        """[rows, *p.shape] strided view of the per-sample gradients of `p` (rows row0 .. row0+rows)."""
        return self.windows[id(p)].view(self.buf, (rows,), (self.P,), base=row0 * self.P)

    def vec_view(self, vec: torch.Tensor, p: nn.Parameter) -> torch.Tensor:
        """p-shaped (possibly row-padded) view into a [P] vector laid out like an arena row."""
        return self.windows[id(p)].view(vec)

    @property
    def rows(self) -> int:
        return sum(self.pass_B)

    def reset_pass(self):
        self.pass_B = []
        self.consumed = False


class _Layer:
    """Hook state of one sampled leaf module."""

    def __init__(self, name: str, module: nn.Module, kind: str, slot0: int, nslots: int):
        self.name, self.module, self.kind, self.slot0, self.nslots = name, module, kind, slot0, nslots
        self.params: List[nn.Parameter] = [p for p in module.parameters(recurse=False)]
        self.frozen_for_autograd = False
        self.acts: List[torch.Tensor] = []
        self.pairs: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.counter = 0
        self.passes_written = 0


def _kind_of(m: nn.Module) -> Optional[str]:
    t = type(m)
    if t in (nn.Linear, RNNLinear):
        return "linear"
    if t is nn.Embedding:
        return "embedding"
    if t is nn.LayerNorm:
        return "layernorm"
    if t is SequenceBias:
        return "seqbias"
    return None


class _GradTap(torch.autograd.Function):
    """Identity that puts `y` into the autograd graph without any upstream work: backward returns nothing, so a
    layer whose input is data and whose parameters are frozen for autograd costs no dW GEMM and no reduction."""

    @staticmethod
    def forward(ctx, y, anchor):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return None, None


class GradSampleModule(nn.Module):
    """Wraps `m`; after backward every trainable parameter has `.grad_sample` ([B, *shape] view)."""

    def __init__(self, m: nn.Module, *, batch_first: bool = True, loss_reduction: str = "mean",
                 strict: bool = True, force_functorch: bool = False, skip_autograd_param_grads: bool = True,
                 overlap_grad_sample: bool = False):
        super().__init__()
        # Autograd's own (summed) parameter gradients are never used by DP-SGD: DPOptimizer overwrites p.grad.
        # While a sampled layer runs forward on an input that already requires grad, its parameters are
        # switched to requires_grad=False, so backward computes only dL/dx (which the hooks need) and skips the
        # dW GEMM, the bias reduction and the .grad accumulation of every one of the T calls.
        self.skip_autograd_param_grads = skip_autograd_param_grads
        if loss_reduction not in ("mean", "sum"):
            raise ValueError(f"loss_reduction = {loss_reduction}. Only 'sum' and 'mean' losses are supported")
        self._module = m
        self.batch_first = batch_first
        self.loss_reduction = loss_reduction
        self.hooks_enabled = True
        self.forbid_grad_accumulation = False  # set by PrivacyEngine when Poisson sampling is on
        self._layers: List[_Layer] = []
        self._handles = []

        params = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        device = params[0][1].device
        if device.type != "cuda" and not getattr(_native, "_ALLOW_NON_CUDA", False):
            raise _native.DpkError(
                "GradSampleModule needs the module on a HIP device (cuda:N): the per-sample-gradient path "
                "runs only in hand-written gfx950 kernels; there is no CPU fallback")
        errors = self.validate(m, strict=strict)
        if errors and strict:
            raise NotImplementedError(f"Model contains layers without a per-sample-gradient kernel: {errors}")
        linear_w = {id(sub.weight) for sub in m.modules() if _kind_of(sub) == "linear"}
        self.arena = GradSampleArena(params, device, dense={id(p) for _, p in params if p.dim() == 2 and id(p) not in linear_w})
        # overlap_grad_sample=True issues the per-sample-gradient launches on a side stream, joined when backward
        # ends.  Measured on cfg2 (MI355X): 3.14 ms/step with overlap vs 2.97 ms without — the store-bound K1
        # launches slow the dX GEMMs / recurrence they run beside by more than the overlap hides.  Off by default.
        object.__setattr__(self, "_side", torch.cuda.Stream(device) if (overlap_grad_sample and device.type == "cuda")
                           else None)
        object.__setattr__(self, "_inflight", [])  # (A, G) kept alive until the join
        # anchor of _GradTap (not a parameter: invisible to state_dict / optimizers)
        object.__setattr__(self, "_tap", torch.zeros((), device=device, requires_grad=True))

        covered = set()
        for name, sub in m.named_modules():
            own = [p for p in sub.parameters(recurse=False) if p.requires_grad]
            if not own:
                continue
            kind = _kind_of(sub)
            if kind is None:
                continue
            if kind == "linear":
                n = _native.linear_slots(sub.in_features, sub.out_features)
            elif kind == "seqbias":
                n = _native.linear_slots(1, sub.bias.numel())
            elif kind == "layernorm":
                if len(sub.normalized_shape) != 1 or not sub.elementwise_affine:
                    raise NotImplementedError(f"{name}: only affine LayerNorm over the last dim is supported")
                n = 1
            else:
                if sub.padding_idx is not None:
                    raise NotImplementedError(f"{name}: nn.Embedding with padding_idx is not supported")
                n = 1
            if any(not p.requires_grad for p in sub.parameters(recurse=False)):
                raise NotImplementedError(f"{name}: partially frozen layers are not supported")
            layer = _Layer(name, sub, kind, self.arena.reserve_slots(n), n)
            self._layers.append(layer)
            for p in own:
                covered.add(id(p))
            self._handles.append(sub.register_forward_pre_hook(self._make_pre(layer)))
            self._handles.append(sub.register_forward_hook(self._make_fwd(layer)))
            self._handles.append(sub.register_full_backward_hook(self._make_bwd(layer)))
        # fused recurrences hand their gates' stacked (activations, backprops) over directly (no per-call hooks)
        self._by_module = {id(l.module): l for l in self._layers}
        self._deferred = None  # list while _fused_sink_many collects launches
        # Small batches (the reference's own B ~ 50): a layer's contraction is a handful of tiles per CU, and a dozen
        # launches per backward pass each pay a ramp and a tail.  Layers whose arena rows allow 16-byte stores are
        # then queued and contracted by ONE multi-shape launch when the backward pass ends (autograd engine callback;
        # also flushed by check_complete()), instead of one launch per layer as its backward hook fires.
        self.multi_launch_max_batch = int(os.environ.get("DPK_MULTI_LAUNCH_MAX_BATCH", "256"))
        object.__setattr__(self, "_multi_pending", [])
        # autograd graph-task id of the backward pass whose end-of-pass callback is already queued (one per pass: a
        # pass that raises never runs its callbacks, so "the list is non-empty" cannot stand in for "a flush is queued")
        object.__setattr__(self, "_multi_task", None)
        object.__setattr__(self, "_side_task", None)
        for sub in m.modules():
            if isinstance(sub, FusedLinearCellGRU) or "_dpk_sink" in sub.__dict__:  # fused GRU modules, DPLSTM cells
                object.__setattr__(sub, "_dpk_sink", self._fused_sink)
                object.__setattr__(sub, "_dpk_sink_active", lambda: self.hooks_enabled)
        for n_, p in params:
            p.grad_sample = None
            p._forward_counter = 0
            p._dpk_arena = self.arena
            if id(p) not in covered and strict:
                raise NotImplementedError(f"parameter {n_} is owned by a module without a per-sample-gradient kernel")

    # ---- nn.Module plumbing -------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        return self._module(*args, **kwargs)

    def __getattr__(self, item):
        try:
            return super().__getattr__(item)
        except AttributeError as e:
            submodules = dict(self._module.named_modules())
            if item and item in submodules:
                return submodules[item]
            if "_module" in self.__dict__.get("_modules", {}):
                return getattr(self._modules["_module"], item)
            raise e

    def state_dict(self, *args, **kwargs):
        return self._module.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        return self._module.load_state_dict(state_dict, strict=strict, **kwargs)

    def zero_grad(self, set_to_none: bool = False):
        if set_to_none is False:
            pass  # grad_sample is always dropped, as upstream
        for p in self.parameters():
            p.grad_sample = None
        self._reset_step_state()
        super().zero_grad(set_to_none)

    def _reset_step_state(self):
        """Forget every un-stepped backward pass: also what a pass that RAISED left behind (queued multi-shape jobs and
        half-collected (A, G) pairs would otherwise join the next pass's launch and write the same arena windows)."""
        self.arena.reset_pass()
        for layer in self._layers:
            layer.passes_written = 0
            layer.pairs = []
        self._multi_pending.clear()
        object.__setattr__(self, "_multi_task", None)
        if self._inflight:
            self._join_side()
        object.__setattr__(self, "_side_task", None)

    def del_grad_sample(self):
        for p in self.parameters():
            p.grad_sample = None
        self._reset_step_state()

    def to_standard_module(self) -> nn.Module:
        self._close()
        return self._module

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def disable_hooks(self):
        self.hooks_enabled = False

    def enable_hooks(self):
        self.hooks_enabled = True

    def _close(self):
        self.remove_hooks()
        for sub in self._module.modules():
            if "_dpk_sink" in sub.__dict__:
                object.__setattr__(sub, "_dpk_sink", None)
                object.__setattr__(sub, "_dpk_sink_active", None)
        for p in self._module.parameters():
            for attr in ("grad_sample", "_forward_counter", "_dpk_arena", "summed_grad"):
                if hasattr(p, attr):
                    delattr(p, attr)

    def __repr__(self):
        return f"GradSampleModule({self._module.__repr__()})"

    @classmethod
    def validate(cls, module: nn.Module, *, strict: bool = False) -> List[NotImplementedError]:
        errors = []
        for name, sub in module.named_modules():
            own = [p for p in sub.parameters(recurse=False) if p.requires_grad]
            if own and _kind_of(sub) is None:
                errors.append(NotImplementedError(
                    f"{name or 'Main'} ({type(sub).__name__}) owns trainable parameters but has no "
                    "per-sample-gradient kernel"))
        if strict and errors:
            raise NotImplementedError(errors)
        return errors

    # ---- hooks ----------------------------------------------------------------------------------------
    def _make_pre(self, layer: _Layer):
        def hook(module, inputs):
            if (self.skip_autograd_param_grads and self.hooks_enabled and module.training
                    and torch.is_grad_enabled() and isinstance(inputs[0], torch.Tensor)):
                for p in layer.params:
                    p.requires_grad_(False)
                layer.frozen_for_autograd = True

        return hook

    def _make_fwd(self, layer: _Layer):
        def hook(module, inputs, output):
            tapped = None
            if layer.frozen_for_autograd:
                for p in layer.params:
                    p.requires_grad_(True)
                layer.frozen_for_autograd = False
                if isinstance(output, torch.Tensor) and not output.requires_grad:
                    # first layer of the model (its input is data): nothing upstream needs a gradient, but the
                    # backward hook must still see grad_output -> keep the output in the graph at zero cost
                    if self._tap.device != output.device:
                        object.__setattr__(self, "_tap", torch.zeros((), device=output.device, requires_grad=True))
                    tapped = _GradTap.apply(output, self._tap)
            if not self.hooks_enabled or not module.training or not torch.is_grad_enabled():
                return tapped
            layer.acts.append(inputs[0].detach())
            layer.counter += 1
            for p in layer.params:
                p._forward_counter += 1
            return tapped

        return hook

    def _make_bwd(self, layer: _Layer):
        def hook(module, grad_input, grad_output):
            if not self.hooks_enabled or not layer.acts:
                return
            g = grad_output[0]
            if g is None:
                raise RuntimeError(f"{layer.name}: no gradient reached this layer's output")
            a = layer.acts.pop()  # LIFO: backward visits calls in reverse
            layer.pairs.append((a, g.detach()))
            layer.counter -= 1
            for p in layer.params:
                p._forward_counter -= 1
            if layer.counter == 0:
                self._materialise(layer)

        return hook

    # ---- the deferred contraction ---------------------------------------------------------------------
    def _stack(self, tensors: List[torch.Tensor], feat_dims: int, batch_first: Optional[bool] = None) -> torch.Tensor:
        """calls x [B, *, F]  ->  contiguous [B, T_total, F] (feat_dims=1) or [B, T_total] (indices)."""
        batch_first = self.batch_first if batch_first is None else batch_first
        flat = []
        for t in tensors:
            if not batch_first and t.dim() >= 2 + feat_dims:
                t = t.transpose(0, 1)
            # explicit sizes: reshape(-1) is ambiguous for an empty Poisson batch (0 samples)
            if feat_dims == 0:
                flat.append(t.reshape(t.shape[0], math.prod(t.shape[1:])))
            elif t.dim() == 2:
                flat.append(t.unsqueeze(1))
            else:
                flat.append(t.reshape(t.shape[0], math.prod(t.shape[1:-1]), t.shape[-1]))
        if len(flat) == 1:
            return flat[0].contiguous()
        B = flat[0].shape[0]
        if any(f.shape[0] != B for f in flat):
            raise NotImplementedError("a layer was called with different batch sizes within one forward pass")
        return torch.cat(flat, dim=1)

    def _materialise(self, layer: _Layer):
        arena = self.arena
        pairs = layer.pairs[::-1]  # back to forward order (t ascending)
        layer.pairs = []
        m = layer.module
        bf = True if isinstance(m, RNNLinear) else None  # DP RNN cells are always called batch-major
        if layer.kind == "seqbias":
            # the bias is the LAST token of the output: its per-sample gradient is that token's backprop.  Written as
            # a Linear contraction with a unit activation (in = 1), so K1 does the copy, the scaling and the norm.
            G = torch.stack([(g[:, -1] if m.batch_first else g[-1]) for _, g in pairs], 1).contiguous()
            self._emit(layer, G.new_ones(G.shape[0], G.shape[1], 1), G)
            return
        if layer.kind == "embedding":
            A = self._stack([p[0] for p in pairs], 0).to(torch.int64)
        else:
            A = self._stack([p[0] for p in pairs], 1, bf)
        G = self._stack([p[1] for p in pairs], 1, bf)
        self._emit(layer, A, G)

    def _fused_sink(self, module: nn.Module, A: torch.Tensor, G: torch.Tensor):
        self._emit(self._by_module[id(module)], A, G)

    def _fused_sink_many(self, jobs):
        """jobs: [(module, A, G)] produced together (the gate Linears of the GRUs of one K8 launch).  Same
        bookkeeping as one _fused_sink call each; Linear jobs of one shape share a grouped K1 launch."""
        self._deferred = []
        try:
            for module, A, G in jobs:
                self._emit(self._by_module[id(module)], A, G)
            pending, self._deferred = self._deferred, None
            groups = {}
            for job in pending:
                layer, m, A, G, arena, row0, part, g_scale = job
                ld = arena.window(m.weight).ld if layer.kind == "linear" else None
                key = (layer.kind, tuple(A.shape), tuple(G.shape), row0, ld, g_scale) if layer.kind == "linear" else id(job)
                groups.setdefault(key, []).append(job)
            for key, members in groups.items():
                if len(members) == 1 or members[0][0].kind != "linear":
                    for job in members:
                        self._launch(*job)
                    continue
                for i in range(0, len(members), _native.K1_MAX_JOBS):
                    chunk = members[i:i + _native.K1_MAX_JOBS]
                    _, _, _, _, arena, row0, part, g_scale = chunk[0]
                    _native.linear_grad_sample_group(
                        [(A, G, arena.flat(m.weight, row0), arena.flat(m.bias, row0) if m.bias is not None else None,
                          layer.slot0) for layer, m, A, G, _, _, _, _ in chunk],
                        part, g_scale, ld_w=key[4])
        finally:
            self._deferred = None

    def _emit(self, layer: _Layer, A: torch.Tensor, G: torch.Tensor):
        """Contract one layer's stacked activations A and backprops G into its arena window."""
        arena = self.arena
        m = layer.module
        if G.dtype != torch.float32 or (layer.kind != "embedding" and A.dtype != torch.float32):
            raise _native.DpkError(f"{layer.name}: only fp32 activations/backprops are supported")
        B = A.shape[0]
        k = layer.passes_written  # index of the un-stepped backward pass this belongs to
        if k >= 1 and self.forbid_grad_accumulation:
            raise ValueError("Poisson sampling is not compatible with grad accumulation. You need to call "
                             "optimizer.step() after every forward/backward pass or consider using "
                             "BatchMemoryManager")
        if k == len(arena.pass_B):
            arena.pass_B.append(B)
        elif k < len(arena.pass_B):
            if arena.pass_B[k] != B:
                raise ValueError(f"{layer.name}: batch size {B} differs from {arena.pass_B[k]} seen by other layers")
        else:
            raise RuntimeError("per-sample-gradient bookkeeping out of sync (missing zero_grad?)")
        row0 = sum(arena.pass_B[:k])
        arena.ensure(max(row0 + B, 1))
        layer.passes_written += 1
        g_scale = float(B) if self.loss_reduction == "mean" else 1.0
        part = arena.part[row0:] if row0 else arena.part
        if B > 0 and self._side is not None:
            task = self._graph_task()
            if task != -1 and task != self._side_task:
                # first launch of this backward pass: join when the autograd engine finishes the pass
                torch.autograd.Variable._execution_engine.queue_callback(self._join_side)
                object.__setattr__(self, "_side_task", task)
            self._inflight.append((A, G))
            self._side.wait_stream(torch.cuda.current_stream(arena.device))
            with torch.cuda.stream(self._side):
                self._launch(layer, m, A, G, arena, row0, part, g_scale)
        elif B > 0 and self._multi_eligible(layer, m, A, G, arena, B):
            task = self._graph_task()
            if task != self._multi_task:
                # first eligible layer of THIS backward pass.  Anything still queued belongs to a pass that raised
                # before its callback ran (a completed pass flushes): its rows are about to be rewritten, drop it.
                self._multi_pending.clear()
                torch.autograd.Variable._execution_engine.queue_callback(self._flush_multi)
                object.__setattr__(self, "_multi_task", task)
            self._multi_pending.append((layer, m, A, G, arena, row0, part, g_scale))
        elif B > 0 and self._deferred is not None:
            self._deferred.append((layer, m, A, G, arena, row0, part, g_scale))
        elif B > 0:
            self._launch(layer, m, A, G, arena, row0, part, g_scale)
        self._publish(m, arena, row0, B, k)

    def _multi_eligible(self, layer, m, A, G, arena, B) -> bool:
        if layer.kind != "linear" or B > self.multi_launch_max_batch or self._side is not None:
            return False
        return _native.multi_ok(A, G, arena.window(m.weight).ld) and self._in_backward()

    @staticmethod
    def _graph_task() -> int:
        try:
            return torch._C._current_graph_task_id()
        except Exception:
            return -1

    @classmethod
    def _in_backward(cls) -> bool:
        # queue_callback is only legal while the autograd engine runs; outside (direct _fused_sink calls in tests)
        # launches stay immediate
        return cls._graph_task() != -1

    def _flush_multi(self):
        """One multi-shape K1 launch for everything queued during this backward pass."""
        pending = self._multi_pending
        object.__setattr__(self, "_multi_task", None)
        if not pending:
            return
        object.__setattr__(self, "_multi_pending", [])
        by_row = {}
        for job in pending:
            by_row.setdefault((job[5], job[7]), []).append(job)  # (row0, g_scale): one pass = one group
        for (row0, g_scale), jobs in by_row.items():
            arena, part = jobs[0][4], jobs[0][6]
            recs = [(A, G, arena.flat(m.weight, row0), arena.flat(m.bias, row0) if m.bias is not None else None,
                     layer.slot0, arena.window(m.weight).ld) for layer, m, A, G, _, _, _, _ in jobs]
            _native.linear_grad_sample_multi(recs, part, g_scale)

    def _join_side(self):
        object.__setattr__(self, "_side_task", None)
        if self._inflight:
            torch.cuda.current_stream(self.arena.device).wait_stream(self._side)
            self._inflight.clear()

    @staticmethod
    def _launch(layer, m, A, G, arena, row0, part, g_scale):
        if layer.kind == "linear":
            vb = arena.flat(m.bias, row0) if m.bias is not None else None
            _native.linear_grad_sample(A, G, arena.flat(m.weight, row0), vb, part, layer.slot0, g_scale,
                                       ld_w=arena.window(m.weight).ld)
        elif layer.kind == "seqbias":
            _native.linear_grad_sample(A, G, arena.flat(m.bias, row0), None, part, layer.slot0, g_scale, ld_w=1)
        elif layer.kind == "embedding":
            _native.embedding_grad_sample(A, G, arena.flat(m.weight, row0), m.num_embeddings, part, layer.slot0,
                                          g_scale)
        else:
            _native.layernorm_grad_sample(A, G, arena.flat(m.weight, row0), arena.flat(m.bias, row0), part,
                                          layer.slot0, m.eps, g_scale)

    @staticmethod
    def _publish(m, arena, row0, B, k):
        for p in m.parameters(recurse=False):
            v = arena.sample_view(p, row0, B)
            if k == 0:
                p.grad_sample = v
            elif isinstance(p.grad_sample, list):
                p.grad_sample.append(v)
            else:
                p.grad_sample = [p.grad_sample, v]

    # ---- used by the optimizer ------------------------------------------------------------------------
    def check_complete(self):
        self._join_side()  # no-op unless per-sample gradients were produced outside an autograd backward pass
        self._flush_multi()
        n = len(self.arena.pass_B)
        for layer in self._layers:
            if layer.passes_written != n:
                raise ValueError(
                    f"Per sample gradient is not initialized for layer '{layer.name}' "
                    f"({layer.passes_written} of {n} backward passes reached it). "
                    "Not updated in backward pass?")
