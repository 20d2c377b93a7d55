"""FusedAdam: torch.optim.Adam's update as ONE launch (K9) over the arena-layout gradient that K7 wrote.

The reference builds `Adam(ae.parameters(), lr=lr, weight_decay=weight_decay)` (Model_Code/timeautoencoder.py:658)
and Opacus' DPOptimizer calls its `.step()` after noise+scale (:751).  DPOptimizer substitutes this class for that
call when the wrapped optimizer is a plain `torch.optim.Adam` (no amsgrad / maximize / decoupled decay) on a HIP
device.  Hyper-parameters are read from the wrapped optimizer's `param_groups` every step (lr schedulers keep
working), and its `state` is kept populated (`exp_avg`, `exp_avg_sq`, `step` per parameter: views into the flat
moment buffers), so `state_dict()` / `load_state_dict()` round-trip with stock torch.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _native


def supported(opt: torch.optim.Optimizer) -> bool:
    if type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("decoupled_weight_decay"):
            return False
        if isinstance(g["lr"], torch.Tensor):
            return False
    return True


class FusedAdam:
    def __init__(self, opt: torch.optim.Adam, arena):
        self.opt, self.arena = opt, arena
        dev = arena.device
        self.exp_avg = torch.zeros(arena.P, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(arena.P, device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros((), device=dev, dtype=torch.float32)
        self.params: List[torch.nn.Parameter] = []
        self.group_of: List[int] = []
        for gi, g in enumerate(opt.param_groups):
            for p in g["params"]:
                if p.requires_grad:
                    self.params.append(p)
                    self.group_of.append(gi)
        order = sorted(range(len(self.params)), key=lambda i: arena.window(self.params[i]).offset)
        self.params = [self.params[i] for i in order]
        self.group_of = [self.group_of[i] for i in order]
        rows = []
        for p in self.params:
            if not p.is_contiguous():
                raise ValueError("FusedAdam needs contiguous parameters")
            w = arena.window(p)
            r, c = (p.shape[0], p.shape[1]) if p.dim() == 2 else (1, p.numel())
            ld = w.ld if p.dim() == 2 else c
            rows.append([p.data_ptr(), w.offset, r, c, ld])
        self._ptrs = [r[0] for r in rows]
        self.win_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._hyper_host: Optional[list] = None
        self.hyper = torch.zeros(len(rows), 8, device=dev, dtype=torch.float32)
        self._import_state()
        self._publish_state()

    # -- wrapped optimizer's state <-> flat buffers -------------------------------------------------------
    def _import_state(self):
        """Adopt moments/step already present in the wrapped optimizer (e.g. after load_state_dict)."""
        steps = set()
        for p in self.params:
            st = self.opt.state.get(p)
            if not st:
                continue
            self.arena.vec_view(self.exp_avg, p).copy_(st["exp_avg"])
            self.arena.vec_view(self.exp_avg_sq, p).copy_(st["exp_avg_sq"])
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError("FusedAdam: parameters with different step counts are not supported")
        if steps:
            self.step_dev.fill_(steps.pop())

    def _publish_state(self):
        for p in self.params:
            self.opt.state[p] = {"step": self.step_dev, "exp_avg": self.arena.vec_view(self.exp_avg, p),
                                 "exp_avg_sq": self.arena.vec_view(self.exp_avg_sq, p)}

    def reload(self):
        self._import_state()
        self._publish_state()

    def _refresh_tables(self):
        hyper = []
        for gi in self.group_of:
            g = self.opt.param_groups[gi]
            b1, b2 = float(g["betas"][0]), float(g["betas"][1])
            hyper.append([float(g["lr"]), b1, b2, float(g["eps"]), float(g["weight_decay"]), 1 - b1, 1 - b2, 0.0])
        if hyper != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: hyper-parameters changed during graph capture")
            self.hyper.copy_(torch.tensor(hyper, dtype=torch.float32))
            self._hyper_host = hyper
        if any(p.data_ptr() != q for p, q in zip(self.params, self._ptrs)):
            raise RuntimeError("FusedAdam: a parameter's storage moved (p.data was reassigned); rebuild the optimizer")

    def step(self):
        self._refresh_tables()
        _native.adam_step(self.arena.grad, self.exp_avg, self.exp_avg_sq, self.win_table, self.hyper, self.step_dev,
                          self.arena.P)
        self.step_dev.add_(1.0)  # a node of the graph when captured

    def dense_state_dict(self):
        """The wrapped optimizer's state_dict with moments as dense clones (views would drag the flat buffers)."""
        sd = self.opt.state_dict()
        for st in sd["state"].values():
            for k in ("exp_avg", "exp_avg_sq", "step"):
                st[k] = st[k].detach().clone().contiguous()
        return sd
