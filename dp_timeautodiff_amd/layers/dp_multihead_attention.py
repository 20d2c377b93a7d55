"""DPMultiheadAttention: nn.MultiheadAttention rebuilt from four nn.Linear projections so that its
parameters are served by the Linear per-sample-gradient kernel (K1).

The reference has no live attention layer (its Transformer_Block is commented out,
Model_Code/timeautoencoder.py:179-212); north_star's "attention-style layers" therefore reduce to this
module, which `ModuleValidator.fix` substitutes for `nn.MultiheadAttention`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dp_rnn import RNNLinear


class SequenceBias(nn.Module):
    """Adds one learned bias token to a [T, B, E] sequence (add_bias_kv of nn.MultiheadAttention)."""

    def __init__(self, embed_dim: int, batch_first: bool = False):
        super().__init__()
        self.batch_first = batch_first
        self.bias = nn.Parameter(torch.empty(embed_dim))
        nn.init.normal_(self.bias)

    def forward(self, x):
        if self.batch_first:
            bsz = x.shape[0]
            return torch.cat([x, self.bias.repeat(bsz, 1, 1)], 1)
        bsz = x.shape[1]
        return torch.cat([x, self.bias.repeat(1, bsz, 1)])


class DPMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None, batch_first=False, device=None, dtype=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.batch_first = batch_first
        self.head_dim = embed_dim // num_heads
        if self.head_dim * num_heads != embed_dim:
            raise ValueError("embed_dim must be divisible by num_heads")
        # RNNLinear: "always called batch-major" for the grad-sample hooks, whatever layout the wrapped model uses
        # (forward transposes [T,B,E] inputs first, so these four never see a sequence-first tensor)
        self.qlinear = RNNLinear(embed_dim, embed_dim, bias=bias)
        self.klinear = RNNLinear(self.kdim, embed_dim, bias=bias)
        self.vlinear = RNNLinear(self.vdim, embed_dim, bias=bias)
        self.out_proj = RNNLinear(embed_dim, embed_dim, bias=bias)
        self.dropout_layer = nn.Dropout(dropout)
        self.add_bias_kv = add_bias_kv
        self.add_zero_attn = add_zero_attn
        if add_bias_kv:  # nn.MultiheadAttention's bias_k / bias_v: one learned token appended to k and to v
            self.seq_bias_k = SequenceBias(embed_dim, batch_first=True)
            self.seq_bias_v = SequenceBias(embed_dim, batch_first=True)

    def load_from_mha(self, mha: nn.MultiheadAttention):
        """Copy weights from an nn.MultiheadAttention (packed in_proj or separate q/k/v)."""
        E = self.embed_dim
        with torch.no_grad():
            if mha.in_proj_weight is not None:
                self.qlinear.weight.copy_(mha.in_proj_weight[:E])
                self.klinear.weight.copy_(mha.in_proj_weight[E:2 * E])
                self.vlinear.weight.copy_(mha.in_proj_weight[2 * E:])
            else:
                self.qlinear.weight.copy_(mha.q_proj_weight)
                self.klinear.weight.copy_(mha.k_proj_weight)
                self.vlinear.weight.copy_(mha.v_proj_weight)
            if mha.in_proj_bias is not None:
                self.qlinear.bias.copy_(mha.in_proj_bias[:E])
                self.klinear.bias.copy_(mha.in_proj_bias[E:2 * E])
                self.vlinear.bias.copy_(mha.in_proj_bias[2 * E:])
            self.out_proj.weight.copy_(mha.out_proj.weight)
            if mha.out_proj.bias is not None:
                self.out_proj.bias.copy_(mha.out_proj.bias)
            if self.add_bias_kv:
                self.seq_bias_k.bias.copy_(mha.bias_k.reshape(-1))
                self.seq_bias_v.bias.copy_(mha.bias_v.reshape(-1))

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None,
                is_causal=False):
        if not self.batch_first:  # -> [B, T, E]: the Linear projections then see batch-first inputs
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        B, Tq, E = query.shape
        Tk = key.shape[1]
        H, Dh = self.num_heads, self.head_dim
        q = self.qlinear(query).view(B, Tq, H, Dh).transpose(1, 2)
        k, v = self.klinear(key), self.vlinear(value)
        extra = 0
        if self.add_bias_kv:  # appended after the projections, as nn.MultiheadAttention does
            k, v = self.seq_bias_k(k), self.seq_bias_v(v)
            extra += 1
        if self.add_zero_attn:
            k, v = (torch.cat([t, t.new_zeros(B, 1, E)], 1) for t in (k, v))
            extra += 1
        if extra:  # the appended tokens are visible to every query and never padding
            Tk += extra
            if attn_mask is not None:
                attn_mask = F.pad(attn_mask, (0, extra))
            if key_padding_mask is not None:
                key_padding_mask = F.pad(key_padding_mask, (0, extra))
        k = k.view(B, Tk, H, Dh).transpose(1, 2)
        v = v.view(B, Tk, H, Dh).transpose(1, 2)
        scores = torch.matmul(q * (Dh ** -0.5), k.transpose(-2, -1))
        if attn_mask is not None:
            if attn_mask.dtype == torch.bool:
                scores = scores.masked_fill(attn_mask, float("-inf"))
            else:
                scores = scores + attn_mask
        if key_padding_mask is not None:
            scores = scores.masked_fill(key_padding_mask.view(B, 1, 1, Tk).to(torch.bool), float("-inf"))
        attn = self.dropout_layer(F.softmax(scores, dim=-1))
        out = torch.matmul(attn, v).transpose(1, 2).reshape(B, Tq, E)
        out = self.out_proj(out)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return (out, attn.mean(dim=1)) if need_weights else (out, None)
