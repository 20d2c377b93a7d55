"""DPOptimizer / DistributedDPOptimizer on the fused HIP tail (K6 + K7).

Drop-in for the optimizer `PrivacyEngine.make_private_with_epsilon` returns at
Model_Code/timeautoencoder.py:674 and the reference drives with `optimizer_ae.zero_grad()` (:732) and
`optimizer_ae.step()` (:751).  Behaviour follows Opacus' `DPOptimizer`:

    step() -> pre_step():
        clip_and_accumulate   per-sample norm over ALL parameter groups, clip factor
                              min(1, C/(norm+1e-6)), clipped weighted sum into p.summed_grad
        [skip flag set -> return]      (BatchMemoryManager: non-final physical chunk)
        add_noise             p.grad = p.summed_grad + N(0, (sigma*C)^2)
        scale_grad            p.grad /= expected_batch_size * accumulated_iterations
        step_hook             accountant.step(...)
    -> original_optimizer.step()

MI355X-first: the per-sample norms were already emitted by the grad-sample kernels, so
clip_and_accumulate is one tiny launch (K6) plus ONE streaming launch over the whole flat arena that
reads every per-sample gradient exactly once (K7); on a real (non-skipped) step noise and scale are fused
into that same launch.  Noise comes from a counter-based Philox4x32-10 stream keyed by (seed, step), so
every rank of a data-parallel job draws the identical tensor with no communication.
"""
from __future__ import annotations

import os
from collections import deque
from typing import Callable, List, Optional

import torch
from torch.optim import Optimizer

from . import _native, fused_adam
from .rng import stream_key


def _generate_seed() -> int:
    return int.from_bytes(os.urandom(8), "little")


class DPOptimizer(Optimizer):
    def __init__(self, optimizer: Optimizer, *, noise_multiplier: float, max_grad_norm: float,
                 expected_batch_size: Optional[int], loss_reduction: str = "mean", generator=None,
                 secure_mode: bool = False, noise_seed: Optional[int] = None, fused_inner_step: bool = True,
                 **kwargs):
        if loss_reduction not in ("mean", "sum"):
            raise ValueError(f"Unexpected value for loss_reduction: {loss_reduction}")
        self.fused_inner_step = fused_inner_step  # plain torch.optim.Adam -> one K9 launch on the arena gradient
        self._fused_adam = None
        if loss_reduction == "mean" and expected_batch_size is None:
            raise ValueError("You must provide expected batch size of the loss reduction is mean")
        self.original_optimizer = optimizer
        self.noise_multiplier = noise_multiplier
        self.max_grad_norm = max_grad_norm
        self.loss_reduction = loss_reduction
        self.expected_batch_size = expected_batch_size
        self.step_hook: Optional[Callable[["DPOptimizer"], None]] = None
        self.generator = generator
        self.secure_mode = secure_mode
        self.param_groups = self.original_optimizer.param_groups
        self.defaults = self.original_optimizer.defaults
        self.state = self.original_optimizer.state
        self._step_skip_queue = deque()
        self._is_last_step_skipped = False
        self.noise_seed = _generate_seed() if noise_seed is None else int(noise_seed)
        self._noise_offset = 0  # Philox blocks consumed so far (advances by P/4 per real step)
        self._noise_offset_dev: Optional[torch.Tensor] = None  # device-side offset (hipGraph-captured steps)
        self._injected_noise: Optional[torch.Tensor] = None
        self._arena = None
        self._gsm = None
        self._n_valid_dev: Optional[torch.Tensor] = None  # device int32: real rows of a fixed-capacity Poisson batch
        for p in self.params:
            p.summed_grad = None

    # ---- wiring -------------------------------------------------------------------------------------
    @property
    def params(self) -> List[torch.nn.Parameter]:
        return [p for group in self.param_groups for p in group["params"] if p.requires_grad]

    def attach_grad_sample_module(self, gsm):
        """Bind to the GradSampleModule whose arena holds this optimizer's per-sample gradients."""
        self._gsm = gsm
        self._arena = gsm.arena
        missing = [i for i, p in enumerate(self.params) if id(p) not in self._arena.windows]
        if missing:
            raise ValueError("optimizer holds trainable parameters that are not in the GradSampleModule")

    @property
    def arena(self):
        if self._arena is None:
            ps = self.params
            a = getattr(ps[0], "_dpk_arena", None) if ps else None
            if a is None:
                raise ValueError("DPOptimizer is not attached to a GradSampleModule (parameters carry no "
                                 "per-sample-gradient arena); wrap the module with GradSampleModule first")
            if any(getattr(p, "_dpk_arena", None) is not a for p in ps):
                raise ValueError("all parameters of a DPOptimizer must belong to one GradSampleModule")
            self._arena = a
        return self._arena

    # ---- Opacus-visible properties ------------------------------------------------------------------
    @property
    def grad_samples(self) -> List[torch.Tensor]:
        ret = []
        for p in self.params:
            gs = getattr(p, "grad_sample", None)
            if gs is None:
                raise ValueError("Per sample gradient not found. Are you using GradSampleModule?")
            ret.append(torch.cat(gs, dim=0) if isinstance(gs, list) else gs)
        return ret

    @property
    def accumulated_iterations(self) -> int:
        n = len(self.arena.pass_B)
        return max(n, 1)

    def signal_skip_step(self, do_skip=True):
        self._step_skip_queue.append(do_skip)

    def _check_skip_next_step(self, pop_next=True):
        if self._step_skip_queue:
            if pop_next:
                return self._step_skip_queue.popleft()
            return self._step_skip_queue[0]
        return False

    def attach_step_hook(self, fn: Callable[["DPOptimizer"], None]):
        self.step_hook = fn

    def enable_device_noise_offset(self):
        """Keep the Philox offset in a device word that the step itself advances (K7 reads it through
        `offset_dev`), so a step captured into a hipGraph draws fresh noise on every replay."""
        if self._noise_offset_dev is None:
            self._noise_offset_dev = torch.tensor([self._noise_offset], dtype=torch.int64, device=self.arena.device)
            self._noise_offset = 0

    @property
    def noise_key(self) -> int:
        """Philox key of the DP-noise stream: derived from `noise_seed`, never the seed itself (rng.py), so a sampler
        seeded with the same number draws an independent stream."""
        return stream_key(self.noise_seed, "noise")

    def set_noise_state(self, seed: int, offset: int):
        """Restore the Philox stream position (checkpoints).  After enable_device_noise_offset() the live offset is
        the device word; the host-side offset stays 0 then."""
        if getattr(self, "_graph_captured", False) and int(seed) != self.noise_seed:
            # the Philox KEY is a by-value kernel argument baked into the captured graph (only the offset lives in a
            # device word): a different seed would be reported by noise_state() and ignored by every replay
            raise RuntimeError("the DP-noise seed cannot change once a step has been captured into a hipGraph: restore "
                               "the checkpoint before GraphedDPStep.capture(), or capture again")
        self.noise_seed = int(seed)
        if self._noise_offset_dev is not None:
            self._noise_offset_dev.fill_(int(offset))
            self._noise_offset = 0
        else:
            self._noise_offset = int(offset)

    def noise_state(self):
        off = int(self._noise_offset)
        if self._noise_offset_dev is not None:
            off += int(self._noise_offset_dev.item())
        return {"seed": self.noise_seed, "offset": off}

    def inject_noise(self, flat_noise: torch.Tensor):
        """Parity/testing: use this [P_arena] tensor (already scaled by sigma*C) for the next real step."""
        self._injected_noise = flat_noise

    # ---- the DP tail --------------------------------------------------------------------------------
    def _prepare(self):
        arena = self.arena
        for p in self.params:
            if getattr(p, "grad_sample", None) is None:
                raise ValueError("Per sample gradient is not initialized. Not updated in backward pass?")
        if self._gsm is not None:
            self._gsm.check_complete()
        if arena.consumed:
            raise ValueError("Per-sample gradients were already processed by a previous step(); call "
                             "zero_grad() before the next backward pass")
        return arena, arena.rows

    def set_valid_rows(self, n_valid_dev: Optional[torch.Tensor]):
        """Rows >= n_valid_dev[0] (a device int32) of the next batches are padding (DevicePoissonLoader's
        fixed-capacity batches): K6 gives them clip factor 0.  None = every row is real."""
        self._n_valid_dev = n_valid_dev

    def _clip_factors(self, arena, B):
        if B > 0:
            _native.clip_factors(arena.part, arena.n_slots, arena.clip, arena.norms, B, self.max_grad_norm,
                                 n_valid_dev=self._n_valid_dev)

    def clip_and_accumulate(self):
        """K6 + K7a: p.summed_grad (+)= sum_b clip_b * grad_sample_b  — one pass over the arena."""
        arena, B = self._prepare()
        arena.ensure(max(B, 1))
        self._clip_factors(arena, B)
        ws = arena.ensure_workspace(B)
        _native.clip_reduce(_native.FlatView(arena.buf, 0, arena.P), arena.clip,
                            arena.summed if arena.summed_valid else None, arena.summed, B, ws)
        arena.summed_valid = True
        arena.consumed = True
        self._publish_summed(arena)

    def _publish_summed(self, arena):
        for p in self.params:
            p.summed_grad = arena.vec_view(arena.summed, p)

    def _publish_grad(self, arena):
        for p in self.params:
            p.grad = arena.vec_view(arena.grad, p)

    def _noise_args(self, arena):
        std = float(self.noise_multiplier) * float(self.max_grad_norm)
        if self._injected_noise is not None:
            z = self._injected_noise
            self._injected_noise = None
            if z.numel() != arena.P or z.device != arena.summed.device:
                raise ValueError(f"injected noise must be a device tensor of {arena.P} floats (arena layout)")
            return z, 0, 0, 0.0
        if std == 0:
            return None, 0, 0, 0.0
        if self.generator is not None:  # Opacus' `noise_generator=` argument
            z = torch.normal(mean=0.0, std=std, size=(arena.P,), device=arena.device, generator=self.generator)
            return z, 0, 0, 0.0
        off = self._noise_offset
        if self._noise_offset_dev is None:
            self._noise_offset += arena.P // 4
        return None, self.noise_key, off, std

    def _advance_device_offset(self, arena, used_philox: bool):
        if used_philox and self._noise_offset_dev is not None:
            self._noise_offset_dev.add_(arena.P // 4)  # a graph node when captured

    def add_noise(self):
        """K7b with unit scale: p.grad = p.summed_grad + N(0, (sigma*C)^2)."""
        arena = self.arena
        noise, seed, off, std = self._noise_args(arena)
        _native.noise_scale(arena.summed, noise, seed, off, std, 1.0, arena.grad, arena.P,
                            offset_dev=self._noise_offset_dev)
        self._advance_device_offset(arena, noise is None and std != 0)
        self._publish_grad(arena)

    def scale_grad(self):
        if self.loss_reduction == "mean":
            self.arena.grad.div_(self.expected_batch_size * self.accumulated_iterations)

    def _inv_scale(self) -> float:
        if self.loss_reduction == "mean":
            return 1.0 / (self.expected_batch_size * self.accumulated_iterations)
        return 1.0

    def _reduce_noise_scale(self, arena, B):
        """Real step: K7 fused (clipped sum + noise + scale in the same pass over the arena)."""
        noise, seed, off, std = self._noise_args(arena)
        ws = arena.ensure_workspace(B)
        _native.clip_reduce_noise(_native.FlatView(arena.buf, 0, arena.P), arena.clip,
                                  arena.summed if arena.summed_valid else None, noise, seed, off, std,
                                  self._inv_scale(), arena.summed, arena.grad, B, ws,
                                  offset_dev=self._noise_offset_dev)
        self._advance_device_offset(arena, noise is None and std != 0)
        arena.summed_valid = True

    def _step_without_backward(self):
        """No backward pass since the last zero_grad: nothing to clip (upstream returns True here)."""
        return True

    def pre_step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:
        ps = self.params
        if not ps or all(getattr(p, "grad_sample", None) is None for p in ps) and not self.arena.pass_B:
            return self._step_without_backward()
        arena, B = self._prepare()
        arena.ensure(max(B, 1))
        if self._check_skip_next_step():
            self._clip_factors(arena, B)
            ws = arena.ensure_workspace(B)
            _native.clip_reduce(_native.FlatView(arena.buf, 0, arena.P), arena.clip,
                                arena.summed if arena.summed_valid else None, arena.summed, B, ws)
            arena.summed_valid = True
            arena.consumed = True
            self._publish_summed(arena)
            self._is_last_step_skipped = True
            return False
        self._clip_factors(arena, B)
        self._reduce_noise_scale(arena, B)
        arena.consumed = True
        self._publish_summed(arena)
        self._publish_grad(arena)
        if self.step_hook:
            self.step_hook(self)
        self._is_last_step_skipped = False
        return True

    def _inner_step(self):
        """The wrapped optimizer's update on p.grad (= views of arena.grad)."""
        if self.fused_inner_step and self._fused_adam is None and self.arena.device.type == "cuda" \
                and fused_adam.supported(self.original_optimizer):
            self._fused_adam = fused_adam.FusedAdam(self.original_optimizer, self.arena)
        if self._fused_adam is not None and self.fused_inner_step:
            return self._fused_adam.step()
        return self.original_optimizer.step()

    def step(self, closure: Optional[Callable[[], float]] = None) -> Optional[float]:
        if closure is not None:
            with torch.enable_grad():
                closure()
        if self.pre_step():
            return self._inner_step()
        return None

    def zero_grad(self, set_to_none: bool = False):
        for p in self.params:
            p.grad_sample = None
            if not self._is_last_step_skipped:
                p.summed_grad = None
        arena = self._arena if self._arena is not None else getattr(self.params[0], "_dpk_arena", None)
        if arena is not None:
            if not self._is_last_step_skipped:
                arena.summed_valid = False
            arena.reset_pass()
        if self._gsm is not None:
            self._gsm._reset_step_state()  # also drops what a backward pass that raised left queued
        if not set_to_none and arena is not None and arena.grad is not None and self._grads_live_in(arena):
            arena.grad.zero_()  # every p.grad is a view of this one vector: one fill instead of one per parameter
        else:
            self.original_optimizer.zero_grad(set_to_none)

    def _grads_live_in(self, arena) -> bool:
        base = arena.grad.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    # ---- last-step diagnostics (device tensors, valid after step()) ------------------------------------
    @property
    def per_sample_norms(self) -> torch.Tensor:
        return self.arena.norms[: self.arena.rows]

    @property
    def per_sample_clip_factor(self) -> torch.Tensor:
        return self.arena.clip[: self.arena.rows]

    # ---- torch.optim.Optimizer plumbing ---------------------------------------------------------------
    def __repr__(self):
        return self.original_optimizer.__repr__()

    def state_dict(self):
        if self._fused_adam is not None:
            return self._fused_adam.dense_state_dict()
        return self.original_optimizer.state_dict()

    def load_state_dict(self, state_dict) -> None:
        self.original_optimizer.load_state_dict(state_dict)
        if self._fused_adam is not None:
            self._fused_adam.reload()


class DistributedDPOptimizer(DPOptimizer):
    """Data-parallel DP-SGD: each rank clips ITS examples locally; the flat clipped sums are all-reduced
    once (RCCL over xGMI when the process group is nccl); identical Philox noise is then added on every
    rank and the result divided by the *global* expected batch size.

    Equivalent to Opacus' DistributedDPOptimizer (rank-0 noise, per-parameter all_reduce, / world_size —
    SURVEY.md Appendix A.7) but with ONE collective per step on one flat [P] buffer instead of one per
    parameter, and no dependence on rank 0 for the noise.  `expected_batch_size` is per rank, as upstream.
    """

    def __init__(self, optimizer: Optimizer, *, noise_multiplier: float, max_grad_norm: float,
                 expected_batch_size: Optional[int], loss_reduction: str = "mean", generator=None,
                 secure_mode: bool = False, noise_seed: Optional[int] = None, process_group=None, **kwargs):
        super().__init__(optimizer, noise_multiplier=noise_multiplier, max_grad_norm=max_grad_norm,
                         expected_batch_size=expected_batch_size, loss_reduction=loss_reduction,
                         generator=generator, secure_mode=secure_mode, noise_seed=noise_seed)
        import torch.distributed as dist
        self._dist = dist
        self.process_group = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        # one shared noise seed: rank 0's
        dev = self.params[0].device
        seed_t = torch.tensor([self.noise_seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev)
        dist.broadcast(seed_t, src=0, group=process_group)
        self.noise_seed = int(seed_t.item())
        if generator is not None:
            raise ValueError("DistributedDPOptimizer draws noise from the shared Philox stream; "
                             "a torch generator cannot be shared across ranks")

    def _inv_scale(self) -> float:
        s = super()._inv_scale()
        return s / self.world_size if self.loss_reduction == "mean" else s

    def _step_without_backward(self):
        """A rank whose Poisson draw came out empty, and whose loop therefore ran no forward/backward, is still a
        participant of the step's ONE collective: the other ranks are already waiting in the all-reduce.  It contributes
        what it has accumulated (earlier physical chunks of this logical batch, else zeros) and takes the same noise and
        scale as everyone.  Returning early here, as the single-process optimizer does, would hang the job."""
        arena = self.arena
        if self._check_skip_next_step():
            self._is_last_step_skipped = True   # a non-final chunk with nothing in it: nothing to add, no exchange yet
            return False
        if not arena.summed_valid:
            arena.summed.zero_()
            arena.summed_valid = True
        self._exchange_noise_scale(arena)
        self._publish_summed(arena)
        self._publish_grad(arena)
        if self.step_hook:
            self.step_hook(self)
        self._is_last_step_skipped = False
        return True

    def _reduce_noise_scale(self, arena, B):
        self._local_reduce(arena, B)
        self._exchange_noise_scale(arena)

    def _local_reduce(self, arena, B):
        ws = arena.ensure_workspace(B)
        _native.clip_reduce(_native.FlatView(arena.buf, 0, arena.P), arena.clip,
                            arena.summed if arena.summed_valid else None, arena.summed, B, ws)
        arena.summed_valid = True

    def _exchange_noise_scale(self, arena):
        self._dist.all_reduce(arena.summed, op=self._dist.ReduceOp.SUM, group=self.process_group)
        noise, seed, off, std = self._noise_args(arena)
        _native.noise_scale(arena.summed, noise, seed, off, std, self._inv_scale(), arena.grad, arena.P,
                            offset_dev=self._noise_offset_dev)
        self._advance_device_offset(arena, noise is None and std != 0)

    # -- two-phase form used by GraphedDPStep: the local phase is captured, the exchange stays eager ------------
    def local_step(self):
        """zero-communication part of a real step: K6 + K7a on this rank's examples."""
        arena, B = self._prepare()
        arena.ensure(max(B, 1))
        if self._check_skip_next_step():
            raise RuntimeError("local_step() is for real steps only (skip-step signal pending)")
        self._clip_factors(arena, B)
        self._local_reduce(arena, B)
        arena.consumed = True

    def finish_step(self):
        """all-reduce of the flat clipped sum, shared-Philox noise + scale, accountant hook, wrapped optimizer."""
        arena = self.arena
        self._exchange_noise_scale(arena)
        self._publish_summed(arena)
        self._publish_grad(arena)
        if self.step_hook:
            self.step_hook(self)
        self._is_last_step_skipped = False
        return self._inner_step()

    def add_noise(self):
        raise NotImplementedError("use step(): the distributed path all-reduces between sum and noise")
