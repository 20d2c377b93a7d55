"""get_noise_multiplier: sigma calibration by doubling + bisection on an accountant.

Imported (but never called directly) by the reference at Model_Code/timeautoencoder.py:20; called for
it by `PrivacyEngine.make_private_with_epsilon` (:674-682).  Restated from SURVEY.md Appendix A.8.
"""
from __future__ import annotations

from typing import Optional

from . import create_accountant

MAX_SIGMA = 1e6


def get_noise_multiplier(*, target_epsilon: float, target_delta: float, sample_rate: float,
                         epochs: Optional[int] = None, steps: Optional[int] = None, accountant: str = "rdp",
                         epsilon_tolerance: float = 0.01, **kwargs) -> float:
    if (steps is None) == (epochs is None):
        raise ValueError("get_noise_multiplier takes as input EITHER a number of steps or a number of epochs")
    if steps is None:
        steps = int(epochs / sample_rate)

    eps_high = float("inf")
    acc = create_accountant(mechanism=accountant)

    sigma_low, sigma_high = 0, 10
    while eps_high > target_epsilon:
        sigma_high = 2 * sigma_high
        acc.history = [(sigma_high, sample_rate, steps)]
        eps_high = acc.get_epsilon(delta=target_delta, **kwargs)
        if sigma_high > MAX_SIGMA:
            raise ValueError("The privacy budget is too low.")

    while target_epsilon - eps_high > epsilon_tolerance:
        sigma = (sigma_low + sigma_high) / 2
        acc.history = [(sigma, sample_rate, steps)]
        eps = acc.get_epsilon(delta=target_delta, **kwargs)
        if eps < target_epsilon:
            sigma_high = sigma
            eps_high = eps
        else:
            sigma_low = sigma
    return sigma_high
