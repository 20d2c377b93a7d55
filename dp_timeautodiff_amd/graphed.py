"""GraphedDPStep: one whole DP-SGD training step captured into a hipGraph.

The reference's step (Model_Code/timeautoencoder.py:732-751: zero_grad, forward, loss, backward, step) is
launch-bound on any GPU: the Linear-cell GRU's T-step Python loop (:84-94) alone issues ~1000 small kernels per
step.  On MI355X the idiomatic answer is not a tracing compiler but a HIP graph: the eager step — autograd
kernels, the hooks' stacking copies, the dpk per-sample-gradient launches (plain hipLaunchKernel calls on the
capturing stream), K6/K7 and a capturable Adam — is captured once and replayed with one host call.

Constraints (checked or documented): fixed physical batch size and static input tensors (copy each batch into
them); the arena must have reached its capacity before capture (the warm-up steps do that); the Philox noise
offset lives in device memory and is advanced by a node of the graph, so every replay draws fresh noise; the
accountant's step hook is host code and is called once per replay.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import gemm_tuning
from .grad_sample import GradSampleModule
from .optimizer import DistributedDPOptimizer, DPOptimizer


def _make_capturable(optimizer: torch.optim.Optimizer):
    """Adam/AdamW-family optimizers need capturable=True (device-side step counters) under graph capture."""
    for group in optimizer.param_groups:
        if "capturable" in group and not group["capturable"]:
            group["capturable"] = True
            if group.get("fused"):
                continue
    for p, st in optimizer.state.items():
        if isinstance(st, dict) and "step" in st and torch.is_tensor(st["step"]) and not st["step"].is_cuda:
            st["step"] = st["step"].to(p.device, dtype=torch.float32)


class GraphedDPStep:
    def __init__(self, module: GradSampleModule, optimizer: DPOptimizer, loss_fn: Callable[..., torch.Tensor],
                 warmup: int = 3, loader=None):
        """`loss_fn()` runs forward + loss on STATIC device tensors and returns the scalar loss.
        With `loader` (a DevicePoissonLoader) every step first draws a Poisson batch on the device into the loader's
        static `[capacity, ...]` tensors and calls `loss_fn(batch_tensors)`; the padding rows are masked by K6."""
        if not isinstance(module, GradSampleModule) or not isinstance(optimizer, DPOptimizer):
            raise TypeError("GraphedDPStep needs the GradSampleModule and DPOptimizer returned by PrivacyEngine")
        self.module, self.optimizer, self.loss_fn = module, optimizer, loss_fn
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_loss: Optional[torch.Tensor] = None
        self.warmup = warmup
        self.loader = loader
        self.replays = 0
        # data-parallel: with an RCCL ("nccl") process group the flat all-reduce, the shared-noise kernel and the wrapped
        # optimizer are nodes of the same graph (RCCL collectives are stream-ordered kernels: capturable).  Any other
        # backend (gloo in CPU tests / one-GPU rehearsals) runs on the host: then only the communication-free part
        # is captured and the exchange + noise + optimizer run eagerly after each replay (4-15 launches).
        self.distributed = isinstance(optimizer, DistributedDPOptimizer)
        self.exchange_captured = False

    def _loss(self):
        if self.loader is None:
            return self.loss_fn()
        batch, n_valid = self.loader.sample_static()
        self.optimizer.set_valid_rows(n_valid)
        return self.loss_fn(batch)

    def _eager_step(self):
        opt = self.optimizer
        opt.zero_grad(set_to_none=True)
        loss = self._loss()
        loss.backward()
        opt.step()
        return loss.detach()

    def _local_part(self):
        opt = self.optimizer
        opt.zero_grad(set_to_none=True)
        loss = self._loss()
        loss.backward()
        opt.local_step()
        return loss.detach()

    def capture(self):
        opt = self.optimizer
        if opt._step_skip_queue:
            raise RuntimeError("capture a step outside BatchMemoryManager (pending skip-step signals)")
        if opt.generator is not None:
            raise RuntimeError("a torch noise generator cannot be captured; use the Philox stream (default)")
        full = not self.distributed or self._collective_capturable()
        if full:
            _make_capturable(opt.original_optimizer)
            opt.enable_device_noise_offset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):  # real training steps: allocate arena / optimizer state, count in accountant
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gemm_tuning.freeze()  # the warm-up steps have tuned every GEMM shape of the step; no timing runs under capture
        capacity = self.module.arena.capacity
        hook, opt.step_hook = opt.step_hook, None  # host-side accounting happens per replay, not at capture
        def capture_once(whole: bool):
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: a collective library's watchdog thread (RCCL) may query events while this thread captures;
            # in the default "global" mode that would invalidate the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_loss = self._eager_step() if whole else self._local_part()

        try:
            try:
                capture_once(full)
            except Exception as e:  # noqa: BLE001
                if not (self.distributed and full):
                    raise
                # the collective library refused capture (whole-step capture with RCCL inside has only run on a one-rank
                # group so far): keep the communication-free part as the graph and run the exchange eagerly after each
                # replay.  The optimizer's device-side noise offset and capturable Adam state work eagerly too.
                import warnings
                warnings.warn(f"capture of the data-parallel step with the all-reduce inside failed ({type(e).__name__}: "
                              f"{e}); capturing the local part only, the exchange runs eagerly")
                torch.cuda.synchronize()
                opt.zero_grad(set_to_none=True)
                full = False
                capture_once(False)
        finally:
            opt.step_hook = hook
        self.exchange_captured = self.distributed and full
        opt._graph_captured = True  # set_noise_state() refuses a different seed from here on (the key is baked in)
        if self.loader is not None:
            self.loader._graph_captured = True
        if self.module.arena.capacity != capacity:
            raise RuntimeError("the per-sample-gradient arena grew during capture; warm up with the capture batch size")
        return self

    def _collective_capturable(self) -> bool:
        import os
        import torch.distributed as dist
        if os.environ.get("DPK_GRAPH_EXCHANGE", "1") == "0":  # operator's switch: keep the exchange out of the graph
            return False
        try:
            return dist.get_backend(self.optimizer.process_group) == "nccl"
        except Exception:
            return False

    def __call__(self) -> torch.Tensor:
        if self.graph is None:
            self.capture()
        fa = self.optimizer._fused_adam
        if fa is not None and (not self.distributed or self.exchange_captured):
            fa._refresh_tables()  # lr / betas / weight_decay changed since capture (schedulers): the graph reads this table
        self.graph.replay()
        self.replays += 1
        if self.distributed and not self.exchange_captured:
            self.optimizer.finish_step()  # calls the accountant hook itself
        elif self.optimizer.step_hook:
            self.optimizer.step_hook(self.optimizer)
        return self.static_loss
