"""ModuleValidator: check / repair a module for DP-SGD with per-sample-gradient kernels.

Behavioural restatement of `opacus.validators.ModuleValidator` as the reference
calls it at Model_Code/timeautoencoder.py:653-656:

    ae = ModuleValidator.fix(ae)
    if not ModuleValidator.is_valid(ae): print(ModuleValidator.validate(ae, strict=False)); raise ValueError

`fix` returns a deep clone in which nn.LSTM / nn.GRU / nn.MultiheadAttention are replaced by their
Linear-cell DP versions (weights copied) and BatchNorm by GroupNorm.  `DeapStack_DP` contains only
Linear / Embedding / activations, so for it `fix` is a plain clone.
"""
from __future__ import annotations

import copy
from typing import List

import torch
import torch.nn as nn

from .layers import DPGRU, DPLSTM, DPMultiheadAttention, RNNLinear, SequenceBias


class UnsupportedModuleError(ValueError):
    pass


class IllegalModuleConfigurationError(ValueError):
    pass


class ShouldReplaceModuleError(UnsupportedModuleError):
    pass


# leaf types whose parameters are served by a HIP per-sample-gradient kernel
SUPPORTED_PARAM_MODULES = (nn.Linear, RNNLinear, nn.Embedding, nn.LayerNorm, SequenceBias)
_BATCHNORM = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)
_INSTANCENORM = (nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)


def clone_module(module: nn.Module) -> nn.Module:
    """Deep clone (parameters, buffers and submodule structure); the original is left untouched."""
    return copy.deepcopy(module)


def _replace(root: nn.Module, name: str, new: nn.Module) -> nn.Module:
    if name == "":
        return new
    parent = root
    parts = name.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)
    return root


class ModuleValidator:
    @classmethod
    def validate(cls, module: nn.Module, *, strict: bool = False) -> List[UnsupportedModuleError]:
        errors: List[UnsupportedModuleError] = []
        if strict and not module.training:
            errors.append(IllegalModuleConfigurationError("Model needs to be in training mode"))
        for name, sub in module.named_modules():
            if isinstance(sub, (nn.LSTM, nn.GRU, nn.RNN)):
                errors.append(ShouldReplaceModuleError(
                    f"{name or 'Main'}: {type(sub).__name__} has no per-sample-gradient kernel; use "
                    "ModuleValidator.fix() to replace it with its DP version (DPLSTM/DPGRU)."))
            elif isinstance(sub, nn.MultiheadAttention):
                errors.append(ShouldReplaceModuleError(
                    f"{name or 'Main'}: nn.MultiheadAttention must be replaced by DPMultiheadAttention (fix())."))
            elif isinstance(sub, _BATCHNORM):
                errors.append(ShouldReplaceModuleError(
                    f"{name or 'Main'}: BatchNorm cannot support training with differential privacy. The reason "
                    "for it is that BatchNorm makes each sample's normalized value depend on its peers in a "
                    "batch, ie the same sample x will get normalized to a different value depending on who else "
                    "is in the batch. Replace with GroupNorm (fix())."))
            elif isinstance(sub, _INSTANCENORM) and sub.track_running_stats:
                errors.append(IllegalModuleConfigurationError(
                    f"{name or 'Main'}: InstanceNorm with track_running_stats=True is not supported."))
            else:
                own = [p for p in sub.parameters(recurse=False) if p.requires_grad]
                if own and not isinstance(sub, SUPPORTED_PARAM_MODULES) and not isinstance(sub, (DPLSTM, DPGRU)):
                    errors.append(UnsupportedModuleError(
                        f"{name or 'Main'}: {type(sub).__name__} owns trainable parameters but has no "
                        "per-sample-gradient kernel in this build (supported: Linear, Embedding, LayerNorm, "
                        "DPLSTM/DPGRU cells, DPMultiheadAttention, SequenceBias)."))
        if strict and len(errors) > 0:
            raise UnsupportedModuleError(errors)
        return errors

    @classmethod
    def is_valid(cls, module: nn.Module) -> bool:
        return len(cls.validate(module, strict=False)) == 0

    @classmethod
    def fix(cls, module: nn.Module, **kwargs) -> nn.Module:
        module = clone_module(module)
        # collect first, then replace (named_modules would otherwise walk into the new children)
        todo = [(n, m) for n, m in module.named_modules()
                if isinstance(m, (nn.LSTM, nn.GRU, nn.MultiheadAttention) + _BATCHNORM + _INSTANCENORM)]
        for name, sub in todo:
            new = cls._fixed(sub, **kwargs)
            if new is not sub:
                new.train(sub.training)
                dev = next(sub.parameters(), torch.empty(0)).device
                module = _replace(module, name, new.to(dev))
        return module

    @classmethod
    def fix_and_validate(cls, module: nn.Module, **kwargs) -> nn.Module:
        module = cls.fix(module, **kwargs)
        cls.validate(module, strict=True)
        return module

    @staticmethod
    def _fixed(sub: nn.Module, **kwargs) -> nn.Module:
        if isinstance(sub, (nn.LSTM, nn.GRU)):
            klass = DPLSTM if isinstance(sub, nn.LSTM) else DPGRU
            new = klass(sub.input_size, sub.hidden_size, num_layers=sub.num_layers, bias=sub.bias,
                        batch_first=sub.batch_first, dropout=sub.dropout, bidirectional=sub.bidirectional,
                        proj_size=getattr(sub, "proj_size", 0))
            new.load_state_dict(sub.state_dict())
            return new
        if isinstance(sub, nn.MultiheadAttention):
            new = DPMultiheadAttention(sub.embed_dim, sub.num_heads, dropout=sub.dropout,
                                       bias=sub.in_proj_bias is not None, add_bias_kv=sub.bias_k is not None,
                                       add_zero_attn=sub.add_zero_attn, kdim=sub.kdim, vdim=sub.vdim,
                                       batch_first=sub.batch_first)
            new.load_from_mha(sub)
            return new
        if isinstance(sub, _BATCHNORM):
            return nn.GroupNorm(min(32, sub.num_features), sub.num_features, affine=sub.affine)
        if isinstance(sub, _INSTANCENORM) and sub.track_running_stats:
            return type(sub)(sub.num_features, affine=sub.affine, track_running_stats=False)
        return sub
