"""DifferentiallyPrivateDistributedDataParallel (DPDDP) — the marker wrapper of the data-parallel path.

North_star: "the batch is sharded data-parallel across the 8 GPUs of one node: each GPU clips its own
examples locally, clipped sums are all-reduced with RCCL over xGMI".  As in Opacus (SURVEY.md Appendix
A.7) the wrapper itself only broadcasts parameters at construction so every rank starts from rank 0's
weights; the gradient exchange is ONE flat all-reduce inside DistributedDPOptimizer.step().
One process per GPU; the process group must already be initialised (backend "nccl" = RCCL on ROCm,
"gloo" in CPU tests).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class DifferentiallyPrivateDistributedDataParallel(nn.Module):
    def __init__(self, model: nn.Module, process_group=None):
        super().__init__()
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        params = list(model.parameters())
        buffers = list(model.buffers())
        with torch.no_grad():
            if params:  # one flat broadcast instead of one per tensor
                flat = torch.cat([p.detach().reshape(-1) for p in params])
                dist.broadcast(flat, src=0, group=process_group)
                off = 0
                for p in params:
                    n = p.numel()
                    p.copy_(flat[off:off + n].view_as(p))
                    off += n
            for b in buffers:
                dist.broadcast(b, src=0, group=process_group)
        self.module = model

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
