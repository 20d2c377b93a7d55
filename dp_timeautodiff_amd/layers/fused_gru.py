"""FusedLinearCellGRU — the reference's CustomGRU (Model_Code/timeautoencoder.py:55-96) with its T-step Python
loop replaced by ONE forward and ONE backward launch of the K8 recurrence kernels (include/dpk.h).

Same constructor, same submodules (`input_layer`, `update_gate`, `reset_gate`, `new_gate` — so the same
state_dict keys and the same nn.Parameter objects), same return value `(output [B,T,H], h_T [1,B,H])`.

Under a GradSampleModule the backward kernel hands the gates' stacked pre-activation gradients G_z, G_r, G_n
[B,T,H] and the stacked activations [x_t | h_{t-1}] / [x_t | r_t*h_{t-1}] [B,T,2H] straight to the K = T
per-sample-gradient GEMM (dpk_linear_grad_sample): no per-call hooks, no T-way torch.stack, and no autograd
weight gradients (DP-SGD never uses them).  Without a GradSampleModule the weight gradients are produced the
ordinary way, so the module also trains non-privately.

`input_layer` is still an nn.Linear called as a module (one [B,T,in] call), hooked like any other layer.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import _native

MAX_HIDDEN = 512


def fusable_hidden(h: int) -> bool:
    return h % 8 == 0 and 8 <= h <= MAX_HIDDEN


class _Recurrence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, h0, wz, bz, wr, br, wn, bn, owner, feed):
        H = xp.shape[-1]
        xp = xp.contiguous()
        h0c = None if h0 is None else h0.contiguous()
        pz = _native.gru_pack(wz, H, 2 * H, False)
        pr = _native.gru_pack(wr, H, 2 * H, False)
        pn = _native.gru_pack(wn, H, 2 * H, False)
        out, Z, R, N, RH = _native.gru_forward(xp, h0c, pz, pr, pn, bz.contiguous(), br.contiguous(), bn.contiguous())
        ctx.owner = owner
        ctx.feed = feed
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(xp, h0c, wz, wr, wn, out, Z, R, N, RH)
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, h0, wz, wr, wn, out, Z, R, N, RH = ctx.saved_tensors
        B, T, H = xp.shape
        wzr = torch.cat([wz, wr], dim=0)
        pnx = _native.gru_pack(wn, H, H, True, 0)
        pnh = _native.gru_pack(wn, H, H, True, H)
        pzrx = _native.gru_pack(wzr, H, 2 * H, True, 0)
        pzrh = _native.gru_pack(wzr, H, 2 * H, True, H)
        want_dh0 = ctx.has_h0 and ctx.needs_input_grad[1]
        dxp, Gz, Gr, Gn, dh0 = _native.gru_backward(dout.contiguous(), None, Z, R, N, out, h0, pnx, pnh, pzrx, pzrh,
                                                    want_dh0)
        owner = ctx.owner
        need_w = any(ctx.needs_input_grad[2:8])
        sink = owner._dpk_sink
        feed = ctx.feed and sink is not None
        grads = [None] * 6
        if feed or need_w:
            first = h0.unsqueeze(1) if h0 is not None else out.new_zeros(B, 1, H)
            hprev = torch.cat([first, out[:, :-1]], dim=1) if T > 1 else first
            a_zr = torch.cat([xp, hprev], dim=2)
            a_n = torch.cat([xp, RH], dim=2)
            if feed:
                sink(owner.update_gate, a_zr, Gz)
                sink(owner.reset_gate, a_zr, Gr)
                sink(owner.new_gate, a_n, Gn)
            if need_w:
                for k, (a, g) in enumerate(((a_zr, Gz), (a_zr, Gr), (a_n, Gn))):
                    if ctx.needs_input_grad[2 + 2 * k]:
                        grads[2 * k] = torch.einsum("bto,bti->oi", g, a)
                    if ctx.needs_input_grad[3 + 2 * k]:
                        grads[2 * k + 1] = g.sum(dim=(0, 1))
        return (dxp, dh0, *grads, None, None)


class FusedLinearCellGRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.batch_first = num_layers, batch_first
        self.input_layer = nn.Linear(input_size, hidden_size)
        self.update_gate = nn.Linear(2 * hidden_size, hidden_size)
        self.reset_gate = nn.Linear(2 * hidden_size, hidden_size)
        self.new_gate = nn.Linear(2 * hidden_size, hidden_size)
        self._init_runtime()

    def _init_runtime(self):
        # set by GradSampleModule: sink(linear_module, A [B,T,2H], G [B,T,H]) and a predicate "hooks are on"
        object.__setattr__(self, "_dpk_sink", None)
        object.__setattr__(self, "_dpk_sink_active", None)

    @classmethod
    def from_module(cls, src: nn.Module) -> "FusedLinearCellGRU":
        """Re-house the four nn.Linear submodules of a CustomGRU-shaped module (the Linear objects, their
        parameters and any hooks on them are shared, not copied)."""
        new = cls.__new__(cls)
        nn.Module.__init__(new)
        new.input_size, new.hidden_size = src.input_size, src.hidden_size
        new.num_layers, new.batch_first = getattr(src, "num_layers", 1), getattr(src, "batch_first", True)
        for n in ("input_layer", "update_gate", "reset_gate", "new_gate"):
            setattr(new, n, getattr(src, n))
        new._init_runtime()
        new.train(src.training)
        return new

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        nn.Module.__init__(new)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k.startswith("_dpk_sink"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._init_runtime()
        return new

    # the reference's loop, used where the kernels do not apply (CPU tensors, empty batch, odd widths)
    def _forward_loop(self, x, h):
        B, T, _ = x.shape
        if h is None:
            h = x.new_zeros(B, self.hidden_size)
        x = self.input_layer(x)
        outs = []
        for t in range(T):
            x_t = x[:, t]
            comb = torch.cat([x_t, h], dim=1)
            z = torch.sigmoid(self.update_gate(comb))
            r = torch.sigmoid(self.reset_gate(comb))
            n = torch.tanh(self.new_gate(torch.cat([x_t, r * h], dim=1)))
            h = (1 - z) * n + z * h
            outs.append(h)
        out = torch.stack(outs, dim=1) if outs else x.new_zeros(B, 0, self.hidden_size)
        return out, h.unsqueeze(0)

    def forward(self, x, h=None):
        if not self.batch_first:
            x = x.transpose(0, 1)
        B, T, _ = x.shape
        fused = (x.is_cuda and x.dtype == torch.float32 and B > 0 and T > 0 and fusable_hidden(self.hidden_size)
                 and self.update_gate.bias is not None)
        if not fused:
            return self._forward_loop(x, h)
        xp = self.input_layer(x)
        gates = (self.update_gate, self.reset_gate, self.new_gate)
        dp = (self._dpk_sink is not None and self.training and torch.is_grad_enabled()
              and (self._dpk_sink_active is None or self._dpk_sink_active()))
        if dp:
            # per-sample gradients come from the sink; autograd must not also build the summed ones
            wb = [t.detach() for g in gates for t in (g.weight, g.bias)]
            if not xp.requires_grad:
                raise RuntimeError("FusedLinearCellGRU under DP needs a differentiable input projection")
        else:
            wb = [t for g in gates for t in (g.weight, g.bias)]
        out = _Recurrence.apply(xp, h, *wb, self, dp)
        return out, out[:, -1].unsqueeze(0)


def looks_like_linear_cell_gru(m: nn.Module) -> bool:
    """Duck-typed CustomGRU (timeautoencoder.py:55-67): four nn.Linear gates of the right shapes."""
    if isinstance(m, FusedLinearCellGRU) or type(m).__name__ not in ("CustomGRU", "LinearCellGRU"):
        return False
    names = ("input_layer", "update_gate", "reset_gate", "new_gate")
    if not all(type(getattr(m, n, None)) is nn.Linear for n in names):
        return False
    H = getattr(m, "hidden_size", None)
    if not isinstance(H, int) or m.input_layer.out_features != H:
        return False
    return all(g.in_features == 2 * H and g.out_features == H and g.bias is not None
               for g in (m.update_gate, m.reset_gate, m.new_gate))


def fuse_recurrences(root: nn.Module, predicate: Optional[Callable[[nn.Module], bool]] = None) -> nn.Module:
    """Replace, in place, every CustomGRU-shaped submodule whose hidden size the K8 kernels support."""
    todo = [(n, m) for n, m in root.named_modules()
            if looks_like_linear_cell_gru(m) and fusable_hidden(m.hidden_size) and (predicate is None or predicate(m))]
    for name, sub in todo:
        new = FusedLinearCellGRU.from_module(sub)
        if name == "":
            return new
        parent = root
        parts = name.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], new)
    return root
