"""DP-compatible recurrent layers built only from Linear cells, so every trainable tensor has a
per-sample-gradient kernel (K1).

`ModuleValidator.fix` swaps `nn.LSTM` / `nn.GRU` for these: the reference's
denoiser `BiRNN_score` holds an `nn.LSTM(H, H, 2, batch_first=True, bidirectional=True)`
(Model_Code/timediffusion.py:106) that must become a DPLSTM before it can be DP-trained.  Parameters keep
the `nn.LSTM` names (`weight_ih_l0`, `weight_hh_l0_reverse`, ...) so state_dicts interchange and the
per-sample gradients are reported under the names the reference's optimizer sees.

Each cell linear (`ih`, `hh`) is called once per time step; the GradSampleModule stacks those T calls
into ONE K=T contraction instead of T rank-B updates of a [B, 4H, in] buffer.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn


class RNNLinear(nn.Linear):
    """nn.Linear used inside DP RNN cells.  A distinct type so the grad-sample registry can tell a
    cell projection (called T times per forward, batch dim always first) from a user Linear."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__(in_features, out_features, bias)


class _Cell(nn.Module):
    gates: int = 1

    def __init__(self, input_size: int, hidden_size: int, bias: bool):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.ih = RNNLinear(input_size, self.gates * hidden_size, bias)
        self.hh = RNNLinear(hidden_size, self.gates * hidden_size, bias)
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class DPLSTMCell(_Cell):
    gates = 4

    def forward(self, x, hc):
        h, c = hc
        g = self.ih(x) + self.hh(h)
        i, f, gg, o = torch.split(g, self.hidden_size, 1)  # gate order i, f, g, o (nn.LSTM layout)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        return h, c


class DPGRUCell(_Cell):
    gates = 3

    def forward(self, x, h):
        gi = self.ih(x)
        gh = self.hh(h)
        i_r, i_z, i_n = torch.split(gi, self.hidden_size, 1)  # gate order r, z, n (nn.GRU layout)
        h_r, h_z, h_n = torch.split(gh, self.hidden_size, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        return (1 - z) * n + z * h


class DPRNNBase(nn.Module):
    cell_type = None
    has_cell_state = False

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 batch_first: bool = False, dropout: float = 0.0, bidirectional: bool = False,
                 proj_size: int = 0):
        super().__init__()
        if proj_size != 0:
            raise NotImplementedError("proj_size is not supported by the DP RNN layers")
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bias = bias
        self.batch_first = batch_first
        self.dropout = dropout
        self.bidirectional = bidirectional
        self.num_directions = 2 if bidirectional else 1
        self.dropout_layer = nn.Dropout(dropout) if dropout > 0 else None
        self.cells = nn.ModuleList()
        for layer in range(num_layers):
            in_sz = input_size if layer == 0 else hidden_size * self.num_directions
            for _ in range(self.num_directions):
                self.cells.append(self.cell_type(in_sz, hidden_size, bias))
        # expose the cell parameters under the nn.LSTM / nn.GRU names (state_dict interchange)
        self._rename = {}    # lstm-style name -> (module, attr)
        self._cell_key = {}  # lstm-style name -> "cells.k.ih.weight"
        for layer in range(num_layers):
            for d in range(self.num_directions):
                k = layer * self.num_directions + d
                cell = self.cells[k]
                suffix = f"_l{layer}" + ("_reverse" if d == 1 else "")
                names = [("weight_ih", "ih", "weight"), ("weight_hh", "hh", "weight")]
                if bias:
                    names += [("bias_ih", "ih", "bias"), ("bias_hh", "hh", "bias")]
                for pre, sub, attr in names:
                    self._rename[pre + suffix] = (getattr(cell, sub), attr)
                    self._cell_key[pre + suffix] = f"cells.{k}.{sub}.{attr}"
        self._register_load_state_dict_pre_hook(self._rename_on_load)
        self._register_state_dict_hook(self._rename_on_save)

    def _rename_on_load(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for new, cell_key in self._cell_key.items():
            if prefix + new in state_dict:
                state_dict[prefix + cell_key] = state_dict.pop(prefix + new)

    @staticmethod
    def _rename_on_save(module, state_dict, prefix, local_metadata):
        # this module's keys are the most recently inserted ones: re-insert them under the nn.LSTM names
        for new, cell_key in module._cell_key.items():
            if prefix + cell_key in state_dict:
                state_dict[prefix + new] = state_dict.pop(prefix + cell_key)
        return state_dict

    def __getattr__(self, name):
        rn = self.__dict__.get("_rename")
        if rn is not None and name in rn:
            mod, attr = rn[name]
            return getattr(mod, attr)
        return super().__getattr__(name)

    # --- forward --------------------------------------------------------------------------------------
    def _run_direction(self, cell, x_steps: List[torch.Tensor], state, reverse: bool):
        T = len(x_steps)
        outs: List[Optional[torch.Tensor]] = [None] * T
        order = range(T - 1, -1, -1) if reverse else range(T)
        for t in order:
            if self.has_cell_state:
                state = cell(x_steps[t], state)
                outs[t] = state[0]
            else:
                state = cell(x_steps[t], state)
                outs[t] = state
        return outs, state

    def forward(self, x: torch.Tensor, state=None):
        if isinstance(x, torch.nn.utils.rnn.PackedSequence):
            raise NotImplementedError("PackedSequence input is not supported")
        if self.batch_first:
            x = x.transpose(0, 1)  # -> [T, B, D]
        T, B, _ = x.shape
        L, D, H = self.num_layers, self.num_directions, self.hidden_size
        if state is None:
            h0 = x.new_zeros(L * D, B, H)
            c0 = x.new_zeros(L * D, B, H) if self.has_cell_state else None
        elif self.has_cell_state:
            h0, c0 = state
        else:
            h0, c0 = state, None
        steps = list(x.unbind(0))
        hn, cn = [], []
        for layer in range(L):
            dir_outs = []
            for d in range(D):
                k = layer * D + d
                st = (h0[k], c0[k]) if self.has_cell_state else h0[k]
                outs, st = self._run_direction(self.cells[k], steps, st, reverse=(d == 1))
                dir_outs.append(outs)
                if self.has_cell_state:
                    hn.append(st[0]); cn.append(st[1])
                else:
                    hn.append(st)
            if D == 1:
                steps = dir_outs[0]
            else:
                steps = [torch.cat([dir_outs[0][t], dir_outs[1][t]], dim=1) for t in range(T)]
            if self.dropout_layer is not None and layer < L - 1:
                steps = [self.dropout_layer(s) for s in steps]
        out = torch.stack(steps, dim=0)
        if self.batch_first:
            out = out.transpose(0, 1)
        h_n = torch.stack(hn, 0)
        if self.has_cell_state:
            return out, (h_n, torch.stack(cn, 0))
        return out, h_n


class _LSTMRecurrence(torch.autograd.Function):
    """The recurrences of one DPLSTM layer (one per direction) in one K10 launch each way.
    Tensor arguments per direction d: xg_d [B,T,4H] (= ih_d(x), bias_ih included), h0_d, c0_d ([B,H] or None for
    all), Whh_d [4H,H], bhh_d ([4H] or None).  Returns out [B,T,G*H] (the directions' outputs already concatenated:
    the kernel writes direction d into column block d), then c_n per direction."""

    @staticmethod
    def forward(ctx, cells, feed, reverse_mask, *args):
        from .. import _native
        G = len(cells)
        per = [args[5 * g:5 * g + 5] for g in range(G)]
        xgs = [p[0].contiguous() for p in per]
        H = xgs[0].shape[-1] // 4
        has_state = per[0][1] is not None
        h0s = [p[1].contiguous() for p in per] if has_state else None
        c0s = [p[2].contiguous() for p in per] if has_state else None
        whh = [p[3].contiguous() for p in per]
        bhhs = [p[4].contiguous() for p in per] if per[0][4] is not None else None
        P = _native.lstm_pack_forward(whh, H)
        out, gates, C, Hprev = _native.lstm_forward(xgs, h0s, c0s, bhhs, P, reverse_mask)
        ctx.cells, ctx.feed, ctx.reverse_mask, ctx.has_state = cells, feed, reverse_mask, has_state
        ctx.has_bias = bhhs is not None
        ctx.save_for_backward(gates, C, Hprev, *whh, *(c0s or []))
        T = out.shape[1]
        # final cell states: returned as their own (differentiable) outputs — a loss that reads c_n (a state handed on
        # to the next segment) gets its gradient back through the kernel's dcT input
        c_n = tuple(C[g, :, 0 if (reverse_mask >> g) & 1 else T - 1].clone() for g in range(G))
        return (out,) + c_n

    @staticmethod
    def backward(ctx, *grads_out):
        from .. import _native
        G = len(ctx.cells)
        gates, C, Hprev, *rest = ctx.saved_tensors
        whh, c0s = rest[:G], (rest[G:] if ctx.has_state else None)
        H = C.shape[-1]
        need = ctx.needs_input_grad[3:]
        want_d0 = ctx.has_state and any(need[5 * g + 1] or need[5 * g + 2] for g in range(G))
        dcat = grads_out[0]
        if dcat is not None and not dcat.is_contiguous():
            dcat = dcat.contiguous()
        douts = [None if dcat is None else dcat[..., g * H:(g + 1) * H] for g in range(G)]  # column blocks, ld = G*H
        Pt = _native.lstm_pack_backward(list(whh), H)
        dcTs = [None if d is None else d.contiguous() for d in grads_out[1:1 + G]]
        dpre, dh0, dc0 = _native.lstm_backward(douts, list(c0s) if c0s else None, gates, C, Pt, ctx.reverse_mask,
                                               want_d0, dcTs=dcTs)
        grads = []
        for g in range(G):
            cell = ctx.cells[g]
            sink = cell._dpk_sink
            if ctx.feed and sink is not None:  # hh's per-sample gradient: A = h_{t-1}, G = dpre (ih gets its own
                sink(cell.hh, Hprev[g], dpre[g])  # through the ordinary Linear hook on xg's gradient)
            gw = gb = None
            if need[5 * g + 3]:
                gw = torch.einsum("bto,bti->oi", dpre[g], Hprev[g])
            if ctx.has_bias and need[5 * g + 4]:
                gb = dpre[g].sum(dim=(0, 1))
            grads += [dpre[g], dh0[g] if (want_d0 and need[5 * g + 1]) else None,
                      dc0[g] if (want_d0 and need[5 * g + 2]) else None, gw, gb]
        return (None, None, None, *grads)


class DPLSTM(DPRNNBase):
    cell_type = DPLSTMCell
    has_cell_state = True
    fuse = True  # run each layer's recurrences as K10 launches when the tensors are on a HIP device

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for cell in self.cells:
            object.__setattr__(cell, "_dpk_sink", None)
            object.__setattr__(cell, "_dpk_sink_active", None)

    def _fusable(self, x) -> bool:
        from .. import _native
        return (self.fuse and x.is_cuda and x.dtype == torch.float32 and x.shape[0] > 0 and x.shape[1] > 0
                and self.hidden_size <= _native.LSTM_MAX_HIDDEN
                and not (self.dropout_layer is not None and self.training and self.num_layers > 1 and False))

    def forward(self, x: torch.Tensor, state=None):
        if isinstance(x, torch.nn.utils.rnn.PackedSequence) or not self._fusable(x):
            return super().forward(x, state)
        if not self.batch_first:
            x = x.transpose(0, 1)  # the kernels are batch-major: [B, T, D]
        B, T, _ = x.shape
        L, D, H = self.num_layers, self.num_directions, self.hidden_size
        h0, c0 = state if state is not None else (None, None)
        hn, cn = [], []
        inp = x
        for layer in range(L):
            cells = [self.cells[layer * D + d] for d in range(D)]
            sink_on = [c._dpk_sink is not None and self.training and torch.is_grad_enabled()
                       and (c._dpk_sink_active is None or c._dpk_sink_active()) for c in cells]
            feed = all(sink_on)
            args = []
            for d, cell in enumerate(cells):
                k = layer * D + d
                xg = cell.ih(inp)  # ONE [B,T,in] -> [B,T,4H] call per direction (hooked like any Linear)
                w, bias = cell.hh.weight, cell.hh.bias
                if feed:  # per-sample gradients come from the sink; autograd must not also build the summed ones
                    w, bias = w.detach(), (None if bias is None else bias.detach())
                args += [xg, None if h0 is None else h0[k], None if c0 is None else c0[k], w, bias]
            res = _LSTMRecurrence.apply(tuple(cells), feed, 2 if D == 2 else 0, *args)
            inp, c_last = res[0], res[1:]  # [B,T,D*H]: both directions, concatenated by the kernel itself
            Hh = self.hidden_size
            for d in range(D):
                hn.append(inp[:, 0, Hh:2 * Hh] if d == 1 else inp[:, T - 1, :Hh])
                cn.append(c_last[d])
            if self.dropout_layer is not None and layer < L - 1:
                inp = self.dropout_layer(inp)
        out = inp if self.batch_first else inp.transpose(0, 1)
        return out, (torch.stack(hn, 0), torch.stack(cn, 0))


class DPGRU(DPRNNBase):
    cell_type = DPGRUCell
    has_cell_state = False
