"""PrivacyEngine — the entry point the reference uses (Model_Code/timeautoencoder.py:672-682, 759):

    privacy_engine = PrivacyEngine()
    ae, optimizer_ae, train_loader = privacy_engine.make_private_with_epsilon(
        module=ae, optimizer=optimizer_ae, data_loader=train_loader, epochs=n_epochs,
        target_epsilon=epsilon, target_delta=delta, max_grad_norm=max_grad_norm)
    ...
    epsilon = privacy_engine.get_epsilon(delta)

Same names, argument meaning and error behaviour as Opacus' PrivacyEngine; the
objects it returns run the DP-SGD arithmetic in hand-written gfx950 kernels.
"""
from __future__ import annotations

import os
import warnings
from typing import IO, Any, BinaryIO, Dict, List, Optional, Tuple, Union

import torch
from torch import nn, optim
from torch.utils.data import DataLoader

from .accountants import create_accountant
from .accountants.utils import get_noise_multiplier
from .data_loader import DevicePoissonLoader, DPDataLoader, switch_generator
from .distributed import DifferentiallyPrivateDistributedDataParallel as DPDDP
from .grad_sample import GradSampleModule
from .layers.fused_gru import fuse_recurrences
from .optimizer import DistributedDPOptimizer, DPOptimizer
from .validators import ModuleValidator


class PrivacyEngine:
    def __init__(self, *, accountant: str = "prv", secure_mode: bool = False):
        self.accountant = create_accountant(mechanism=accountant)
        self.secure_mode = secure_mode
        self.secure_rng = None
        self.dataset = None  # only used to detect switching to a different dataset
        self.fuse_recurrence = True  # make_private(fuse_recurrence=False) keeps the hooked T-call protocol
        if self.secure_mode:
            # Opacus' secure mode needs a cryptographically safe generator (torchcsprng) for sampling and noise, and
            # draws the noise as a sum of four samples; this engine's noise is Philox4x32-10 (not a CSPRNG) and there is
            # no CSPRNG in this image.  Refuse rather than hand back the ordinary path under that name.
            raise NotImplementedError("secure_mode=True is not supported: DP noise and Poisson sampling come from "
                                      "Philox4x32-10 counter streams, which are not cryptographically secure")

    # ---- internals ----------------------------------------------------------------------------------
    def _prepare_optimizer(self, *, optimizer: optim.Optimizer, noise_multiplier: float, max_grad_norm,
                           expected_batch_size: int, loss_reduction: str = "mean", distributed: bool = False,
                           clipping: str = "flat", noise_generator=None, grad_sample_mode="hooks",
                           noise_seed: Optional[int] = None, **kwargs) -> DPOptimizer:
        if isinstance(optimizer, DPOptimizer):
            optimizer = optimizer.original_optimizer
        if clipping != "flat":
            raise NotImplementedError(f"clipping='{clipping}': only flat clipping (per-sample norm over all "
                                      "parameter groups) is implemented")
        cls = DistributedDPOptimizer if distributed else DPOptimizer
        return cls(optimizer=optimizer, noise_multiplier=noise_multiplier, max_grad_norm=max_grad_norm,
                   expected_batch_size=expected_batch_size, loss_reduction=loss_reduction,
                   generator=noise_generator, secure_mode=self.secure_mode, noise_seed=noise_seed, **kwargs)

    def _prepare_data_loader(self, data_loader: DataLoader, *, poisson_sampling: bool,
                             distributed: bool, device_sampler_on=None) -> DataLoader:
        if self.dataset is None:
            self.dataset = data_loader.dataset
        elif self.dataset != data_loader.dataset:
            warnings.warn(f"PrivacyEngine detected new dataset object. Was: {self.dataset}, got: "
                          f"{data_loader.dataset}. Privacy accounting works per dataset, please initialize new "
                          "PrivacyEngine if you're using different dataset. You can ignore this warning if two "
                          "datasets above represent the same logical dataset")
        if poisson_sampling and device_sampler_on is not None:
            # K11: Bernoulli draws, compaction and the row gather all happen in HBM (no per-sample collation)
            rank = torch.distributed.get_rank() if distributed else 0
            world = torch.distributed.get_world_size() if distributed else 1
            seed = None
            if distributed:  # one base seed for the job (rank 0's); each rank derives its own stream from it
                t = torch.tensor([int.from_bytes(os.urandom(7), "little")], dtype=torch.int64, device=device_sampler_on)
                torch.distributed.broadcast(t, src=0)
                seed = int(t.item())
            return DevicePoissonLoader.from_data_loader(data_loader, device=device_sampler_on, seed=seed, rank=rank,
                                                        world_size=world)
        if poisson_sampling:
            return DPDataLoader.from_data_loader(data_loader, generator=self.secure_rng, distributed=distributed)
        elif self.secure_mode:
            return switch_generator(data_loader=data_loader, generator=self.secure_rng)
        return data_loader

    def _prepare_model(self, module: nn.Module, *, batch_first: bool = True, loss_reduction: str = "mean",
                       grad_sample_mode: str = "hooks") -> GradSampleModule:
        self.validate(module=module, optimizer=None, data_loader=None)
        if grad_sample_mode not in ("hooks",):
            raise NotImplementedError(f"grad_sample_mode='{grad_sample_mode}': only 'hooks' is implemented")
        if isinstance(module, GradSampleModule):
            if module.batch_first != batch_first or module.loss_reduction != loss_reduction:
                raise ValueError("Pre-existing GradSampleModule doesn't match new arguments."
                                 f"Got: module.batch_first: {module.batch_first}, module.loss_reduction: "
                                 f"{module.loss_reduction}Requested: batch_first:{batch_first}, "
                                 f"loss_reduction: {loss_reduction}. Please pass vanilla nn.Module instead")
            return module
        inner = module.module if isinstance(module, DPDDP) else module
        if self.fuse_recurrence and next(inner.parameters()).is_cuda:
            # CustomGRU-shaped loops -> the K8 recurrence kernels (same Linear submodules and parameters, so the
            # optimizer built on them stays valid)
            inner = fuse_recurrences(inner)
        return GradSampleModule(inner, batch_first=batch_first, loss_reduction=loss_reduction)

    def is_compatible(self, *, module: nn.Module, optimizer: Optional[optim.Optimizer],
                      data_loader: Optional[DataLoader]) -> bool:
        return ModuleValidator.is_valid(module)

    def validate(self, *, module: nn.Module, optimizer: Optional[optim.Optimizer],
                 data_loader: Optional[DataLoader]):
        m = module._module if isinstance(module, GradSampleModule) else module
        ModuleValidator.validate(m, strict=True)

    @classmethod
    def get_compatible_module(cls, module: nn.Module) -> nn.Module:
        module = ModuleValidator.fix(module)
        ModuleValidator.validate(module, strict=True)
        return module

    # ---- public API -----------------------------------------------------------------------------------
    def make_private(self, *, module: nn.Module, optimizer: optim.Optimizer, data_loader: DataLoader,
                     noise_multiplier: float, max_grad_norm: Union[float, List[float]],
                     batch_first: bool = True, loss_reduction: str = "mean", poisson_sampling: bool = True,
                     clipping: str = "flat", noise_generator=None, grad_sample_mode: str = "hooks",
                     noise_seed: Optional[int] = None, **kwargs
                     ) -> Tuple[GradSampleModule, DPOptimizer, DataLoader]:
        if noise_generator and self.secure_mode:
            raise ValueError("Passing seed is prohibited in secure mode")
        self.fuse_recurrence = bool(kwargs.pop("fuse_recurrence", getattr(self, "fuse_recurrence", True)))
        device_sampler = bool(kwargs.pop("device_sampler", False))

        # compare module parameter with optimizer parameters
        model_parameters = set(module.parameters())
        for p in [p for g in optimizer.param_groups for p in g["params"]]:
            if p not in model_parameters:
                raise ValueError("Module parameters are different than optimizer Parameters")

        distributed = isinstance(module, (DPDDP, nn.parallel.DistributedDataParallel))
        if isinstance(module, nn.parallel.DistributedDataParallel):
            raise NotImplementedError("wrap the module in DifferentiallyPrivateDistributedDataParallel (DPDDP), "
                                      "not torch DDP: gradients are all-reduced by DistributedDPOptimizer")

        module = self._prepare_model(module, batch_first=batch_first, loss_reduction=loss_reduction,
                                     grad_sample_mode=grad_sample_mode)
        if poisson_sampling:
            module.forbid_grad_accumulation = True

        dev_on = next(module.parameters()).device if device_sampler else None
        data_loader = self._prepare_data_loader(data_loader, distributed=distributed,
                                                poisson_sampling=poisson_sampling, device_sampler_on=dev_on)

        sample_rate = 1 / len(data_loader)
        expected_batch_size = int(len(data_loader.dataset) * sample_rate)

        # expected_batch_size is the *per worker* batch size
        if distributed:
            world_size = torch.distributed.get_world_size()
            expected_batch_size /= world_size

        optimizer = self._prepare_optimizer(optimizer=optimizer, noise_multiplier=noise_multiplier,
                                            max_grad_norm=max_grad_norm, expected_batch_size=expected_batch_size,
                                            loss_reduction=loss_reduction, noise_generator=noise_generator,
                                            distributed=distributed, clipping=clipping,
                                            grad_sample_mode=grad_sample_mode, noise_seed=noise_seed, **kwargs)
        optimizer.attach_grad_sample_module(module)
        optimizer.attach_step_hook(self.accountant.get_optimizer_hook_fn(sample_rate=sample_rate))
        return module, optimizer, data_loader

    def make_private_with_epsilon(self, *, module: nn.Module, optimizer: optim.Optimizer,
                                  data_loader: DataLoader, target_epsilon: float, target_delta: float,
                                  epochs: int, max_grad_norm: Union[float, List[float]],
                                  batch_first: bool = True, loss_reduction: str = "mean",
                                  poisson_sampling: bool = True, clipping: str = "flat", noise_generator=None,
                                  grad_sample_mode: str = "hooks", **kwargs):
        sample_rate = 1 / len(data_loader)
        if len(self.accountant) > 0:
            warnings.warn("You're calling make_private_with_epsilon with non-zero privacy budget already spent. "
                          "Returned noise_multiplier assumes zero starting point, so your overall privacy "
                          "budget will be higher.")
        # options of this engine (not of the sigma search) travel on to make_private
        engine_kw = {k: kwargs.pop(k) for k in ("device_sampler", "fuse_recurrence", "noise_seed") if k in kwargs}
        return self.make_private(
            module=module, optimizer=optimizer, data_loader=data_loader,
            noise_multiplier=get_noise_multiplier(target_epsilon=target_epsilon, target_delta=target_delta,
                                                  sample_rate=sample_rate, epochs=epochs,
                                                  accountant=self.accountant.mechanism(), **kwargs),
            max_grad_norm=max_grad_norm, batch_first=batch_first, loss_reduction=loss_reduction,
            noise_generator=noise_generator, grad_sample_mode=grad_sample_mode,
            poisson_sampling=poisson_sampling, clipping=clipping, **engine_kw)

    def get_epsilon(self, delta):
        # the reference asks every epoch (timeautoencoder.py:759); PRV's FFT composition is the expensive part of
        # that call, so the answer is cached per (ledger, delta)
        key = (repr(getattr(self.accountant, "history", None)), float(delta))
        if getattr(self, "_eps_cache", (None, None))[0] != key:
            self._eps_cache = (key, self.accountant.get_epsilon(delta))
        return self._eps_cache[1]

    def epsilon_report(self, delta) -> Dict[str, Any]:
        """The epsilon the engine's accountant reports AND the RDP bound of the same ledger.  PrivacyEngine() defaults to
        the PRV accountant, as Opacus does (the reference's call, timeautoencoder.py:672): its epsilon is checked here as
        a valid, tight bound by an independent composition (tests/test_prv_independent.py) but its equality with Opacus'
        own PRV figure cannot be pinned offline; the RDP figure is the one pinned to Opacus' unit-test value."""
        from .accountants import RDPAccountant
        hist = list(getattr(self.accountant, "history", []))
        key = (repr(hist), float(delta))
        if getattr(self, "_rdp_cache", (None, None))[0] != key:
            r = RDPAccountant()
            r.history = hist
            self._rdp_cache = (key, float(r.get_epsilon(delta)) if hist else 0.0)
        return {"accountant": self.accountant.mechanism(), "epsilon": float(self.get_epsilon(delta)),
                "rdp_epsilon": self._rdp_cache[1], "delta": float(delta)}

    # ---- checkpoints (unused by the reference; saves what its own save_autoencoder forgets) -----------
    def save_checkpoint(self, *, path: Union[str, os.PathLike, BinaryIO, IO[bytes]],
                        module: GradSampleModule, optimizer: Optional[DPOptimizer] = None,
                        noise_scheduler=None, grad_clip_scheduler=None,
                        checkpoint_dict: Optional[Dict[str, Any]] = None,
                        module_state_dict_kwargs: Optional[Dict[str, Any]] = None,
                        torch_save_kwargs: Optional[Dict[str, Any]] = None):
        checkpoint_dict = checkpoint_dict or {}
        checkpoint_dict["module_state_dict"] = module.state_dict(**(module_state_dict_kwargs or {}))
        checkpoint_dict["privacy_accountant_state_dict"] = self.accountant.state_dict()
        if optimizer is not None:
            checkpoint_dict["optimizer_state_dict"] = optimizer.state_dict()
            checkpoint_dict["dp_noise_state"] = optimizer.noise_state()  # host offset + the graph's device word
        torch.save(checkpoint_dict, path, **(torch_save_kwargs or {}))

    def load_checkpoint(self, *, path: Union[str, os.PathLike, BinaryIO, IO[bytes]],
                        module: GradSampleModule, optimizer: Optional[DPOptimizer] = None,
                        noise_scheduler=None, grad_clip_scheduler=None,
                        module_load_dict_kwargs: Optional[Dict[str, Any]] = None,
                        torch_load_kwargs: Optional[Dict[str, Any]] = None) -> Dict:
        checkpoint = torch.load(path, **(torch_load_kwargs or {}))
        module.load_state_dict(checkpoint["module_state_dict"], **(module_load_dict_kwargs or {}))
        self.accountant.load_state_dict(checkpoint["privacy_accountant_state_dict"])
        optimizer_state_dict = checkpoint.pop("optimizer_state_dict", {})
        if optimizer is not None and len(optimizer_state_dict) > 0:
            optimizer.load_state_dict(optimizer_state_dict)
        elif (optimizer is not None) ^ (len(optimizer_state_dict) > 0):
            warnings.warn(f"optimizer_state_dict has {len(optimizer_state_dict)} items"
                          f" but optimizer is {'' if optimizer else 'not'} provided.")
        ns = checkpoint.get("dp_noise_state")
        if optimizer is not None and ns:
            optimizer.set_noise_state(ns["seed"], ns["offset"])  # lands in the device word once a graph owns the offset
        return checkpoint
