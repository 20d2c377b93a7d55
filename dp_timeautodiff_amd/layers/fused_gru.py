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
    """G independent GRUs of equal shape in one launch each way.  Tensor arguments, per GRU g:
    xp_g, h0_g (or None for all), Wz_g, bz_g, Wr_g, br_g, Wn_g, bn_g;  returns out_0 .. out_{G-1}."""

    @staticmethod
    def forward(ctx, owners, feed, *args):
        G = len(owners)
        per = [args[8 * g:8 * g + 8] for g in range(G)]
        xps = [p[0].contiguous() for p in per]
        H = xps[0].shape[-1]
        has_h0 = per[0][1] is not None
        h0s = [p[1].contiguous() for p in per] if has_h0 else None
        weights = [p[2 + 2 * k].contiguous() for p in per for k in range(3)]
        # input halves of all three gates for all T steps: one GEMM per GRU (the recurrence only runs the hidden halves)
        wx = torch.cat([w[:, :H] for w in weights], 0)                      # [G*3H, H]: update | reset | new per GRU
        bx = torch.cat([p[3 + 2 * k] for p in per for k in range(3)], 0)    # [G*3H]
        xgs = [torch.addmm(bx[3 * H * g:3 * H * (g + 1)], xps[g].view(-1, H), wx[3 * H * g:3 * H * (g + 1)].t())
               .view(*xps[g].shape[:2], 3 * H) for g in range(G)]
        packs = _native.gru_pack_forward(weights, H)
        out, Z, R, N, Azr, An = _native.gru_forward(xps, xgs, h0s, *packs)
        ctx.owners, ctx.feed, ctx.has_h0 = owners, feed, has_h0
        ctx.save_for_backward(Z, R, N, Azr, An, wx, *weights)
        return tuple(out[g] for g in range(G))

    @staticmethod
    def backward(ctx, *douts):
        Z, R, N, Azr, An, wx, *weights = ctx.saved_tensors
        G, B, T, H = Z.shape
        owners = ctx.owners
        packs = _native.gru_pack_backward(weights, H)
        need = ctx.needs_input_grad[2:]
        want_dh0 = ctx.has_h0 and any(need[8 * g + 1] for g in range(G))
        douts = [None if d is None else d.contiguous() for d in douts]
        G3, dh0 = _native.gru_backward(douts, None, Z, R, N, Azr, *packs, want_dh0)
        Gz, Gr, Gn = G3[..., :H], G3[..., H:2 * H], G3[..., 2 * H:]  # column blocks: K1 reads them with ld_g = 3H
        dxp = [None] * G
        for g in range(G):
            if need[8 * g]:  # dL/dxp: the input halves' transposed products, all T steps and all gates in one GEMM
                dxp[g] = torch.mm(G3[g].view(-1, 3 * H), wx[3 * H * g:3 * H * (g + 1)]).view(B, T, H)
        grads = []
        if ctx.feed:  # per-sample gradients: K1 straight from the kernels' buffers
            jobs, hosts = [], set()
            for g in range(G):
                owner = owners[g]
                if owner._dpk_sink is not None:
                    hosts.add(id(getattr(owner._dpk_sink, "__self__", owner._dpk_sink)))
                    jobs.append((owner._dpk_sink, owner.update_gate, Azr[g], Gz[g]))
                    jobs.append((owner._dpk_sink, owner.reset_gate, Azr[g], Gr[g]))
                    jobs.append((owner._dpk_sink, owner.new_gate, An[g], Gn[g]))
            many = getattr(getattr(jobs[0][0], "__self__", None), "_fused_sink_many", None) if jobs else None
            if many is not None and len(hosts) == 1:
                many([j[1:] for j in jobs])  # same-shape gates of all G GRUs share one grouped K1 launch
            else:
                for sink, mod, a, gg in jobs:
                    sink(mod, a, gg)
        for g in range(G):
            owner = owners[g]
            gw = [None] * 6
            for k, (a, gg) in enumerate(((Azr[g], Gz[g]), (Azr[g], Gr[g]), (An[g], Gn[g]))):
                if need[8 * g + 2 + 2 * k]:
                    gw[2 * k] = torch.einsum("bto,bti->oi", gg, a)
                if need[8 * g + 3 + 2 * k]:
                    gw[2 * k + 1] = gg.sum(dim=(0, 1))
            grads += [dxp[g], dh0[g] if (dh0 is not None and need[8 * g + 1]) else None, *gw]
        return (None, None, *grads)


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

    def _fusable_call(self, x, h) -> bool:
        return (x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 and x.shape[0] > 0 and x.shape[1] > 0
                and fusable_hidden(self.hidden_size) and self.update_gate.bias is not None
                and (h is None or (h.dim() == 2 and h.is_cuda and h.dtype == torch.float32)))

    def _dp_active(self) -> bool:
        return (self._dpk_sink is not None and self.training and torch.is_grad_enabled()
                and (self._dpk_sink_active is None or self._dpk_sink_active()))

    def _prepare(self, x, h):
        """input projection + the 8 tensor arguments of this GRU for _Recurrence; also whether the DP sink is on."""
        xp = self.input_layer(x)
        gates = (self.update_gate, self.reset_gate, self.new_gate)
        dp = self._dp_active()
        if dp:
            # per-sample gradients come from the sink; autograd must not also build the summed ones
            wb = [t.detach() for g in gates for t in (g.weight, g.bias)]
            if not xp.requires_grad:
                raise RuntimeError("FusedLinearCellGRU under DP needs a differentiable input projection")
        else:
            wb = [t for g in gates for t in (g.weight, g.bias)]
        return [xp, h, *wb], dp

    def forward(self, x, h=None):
        if not self.batch_first:
            x = x.transpose(0, 1)
        if not self._fusable_call(x, h):
            return self._forward_loop(x, h)
        args, dp = self._prepare(x, h)
        (out,) = _Recurrence.apply((self,), dp, *args)
        return out, out[:, -1].unsqueeze(0)


def run_together(modules, xs, hs=None):
    """Call several recurrent modules that do not depend on each other (DeapStack_DP's encoder_mu / encoder_logvar,
    timeautoencoder.py:117-120).  When all are FusedLinearCellGRU of one shape they share ONE forward and ONE
    backward launch — a recurrence has only B/16 workgroups, so this is what fills the chip; otherwise they are
    simply called in turn.  Returns [module(x, h) for ...]."""
    hs = hs if hs is not None else [None] * len(modules)
    ok = (len(modules) > 1 and all(isinstance(m, FusedLinearCellGRU) and m.batch_first for m in modules)
          and len({m.hidden_size for m in modules}) == 1
          and len({tuple(x.shape[:2]) for x in xs}) == 1
          and (all(h is None for h in hs) or all(h is not None for h in hs))
          and all(m._fusable_call(x, h) for m, x, h in zip(modules, xs, hs))
          and len({m._dp_active() for m in modules}) == 1)
    if not ok:
        return [m(x, h) if h is not None else m(x) for m, x, h in zip(modules, xs, hs)]
    args = []
    for m, x, h in zip(modules, xs, hs):
        args += m._prepare(x, h)[0]
    outs = _Recurrence.apply(tuple(modules), modules[0]._dp_active(), *args)
    return [(o, o[:, -1].unsqueeze(0)) for o in outs]


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
