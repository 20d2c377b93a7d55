"""IAccountant: history bookkeeping shared by the RDP / GDP / PRV accountants.

Behavioural restatement of Opacus' `opacus.accountants.accountant.IAccountant`
(SURVEY.md Appendix A.8), reached from the reference through
`PrivacyEngine.make_private_with_epsilon` / `get_epsilon`
(Model_Code/timeautoencoder.py:672-682, 759).  Pure host code, float64.
"""
from __future__ import annotations

import abc
from collections import OrderedDict
from copy import deepcopy
from typing import Any, Callable, List, Mapping, Optional, Tuple, TypeVar

T_state_dict = TypeVar("T_state_dict", bound=Mapping[str, Any])


class IAccountant(abc.ABC):
    def __init__(self):
        self.history: List[Tuple[float, float, int]] = []  # (noise_multiplier, sample_rate, num_steps)

    def step(self, *, noise_multiplier: float, sample_rate: float):
        if len(self.history) >= 1:
            last_noise_multiplier, last_sample_rate, num_steps = self.history.pop()
            if last_noise_multiplier == noise_multiplier and last_sample_rate == sample_rate:
                self.history.append((last_noise_multiplier, last_sample_rate, num_steps + 1))
            else:
                self.history.append((last_noise_multiplier, last_sample_rate, num_steps))
                self.history.append((noise_multiplier, sample_rate, 1))
        else:
            self.history.append((noise_multiplier, sample_rate, 1))

    @abc.abstractmethod
    def get_epsilon(self, delta: float, *args, **kwargs) -> float:
        ...

    @abc.abstractmethod
    def __len__(self) -> int:
        ...

    @classmethod
    @abc.abstractmethod
    def mechanism(cls) -> str:
        ...

    def get_optimizer_hook_fn(self, sample_rate: float) -> Callable:
        """Hook attached to DPOptimizer.step: one accountant step per *real* optimizer step."""

        def hook_fn(optim):
            self.step(noise_multiplier=optim.noise_multiplier,
                      sample_rate=sample_rate * optim.accumulated_iterations)

        return hook_fn

    def state_dict(self, destination: Optional[T_state_dict] = None) -> T_state_dict:
        if destination is None:
            destination = OrderedDict()
        destination["history"] = deepcopy(self.history)
        destination["mechanism"] = self.__class__.mechanism()
        return destination

    def load_state_dict(self, state_dict: T_state_dict):
        if state_dict is None or len(state_dict) == 0:
            raise ValueError("state dict is either None or empty and hence cannot be loaded into Privacy Accountant.")
        if "history" not in state_dict.keys():
            raise ValueError("state_dict does not have the key `history`. Cannot be loaded into Privacy Accountant.")
        if "mechanism" not in state_dict.keys():
            raise ValueError("state_dict does not have the key `mechanism`. Cannot be loaded into Privacy Accountant.")
        if self.__class__.mechanism() != state_dict["mechanism"]:
            raise ValueError(
                f"state_dict of {state_dict['mechanism']} cannot be loaded into "
                f" Privacy Accountant with mechanism {self.__class__.mechanism()}")
        self.history = state_dict["history"]
