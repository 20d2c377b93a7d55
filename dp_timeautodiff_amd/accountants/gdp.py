"""Gaussian-DP accountant (CLT approximation; Bu, Dong, Long, Su 2019), as Opacus' GaussianAccountant
(SURVEY.md Appendix A.8 soft known answer: sigma=1.5, q=0.04, 2250 steps, delta=1e-5 -> eps ~ 6.59803).
Only a single (sigma, q) regime is supported, as upstream."""
from __future__ import annotations

import warnings

import numpy as np
from scipy import optimize
from scipy.stats import norm

from .accountant import IAccountant


def compute_mu_uniform(*, steps: int, noise_multiplier: float, sample_rate: float) -> float:
    c = sample_rate * np.sqrt(steps)
    return (np.sqrt(2) * c * np.sqrt(np.exp(noise_multiplier ** (-2)) * norm.cdf(1.5 / noise_multiplier)
                                     + 3 * norm.cdf(-0.5 / noise_multiplier) - 2))


def compute_mu_poisson(*, steps: int, noise_multiplier: float, sample_rate: float) -> float:
    return np.sqrt(np.exp(noise_multiplier ** (-2)) - 1) * np.sqrt(steps) * sample_rate


def delta_eps_mu(*, eps: float, mu: float) -> float:
    return norm.cdf(-eps / mu + mu / 2) - np.exp(eps) * norm.cdf(-eps / mu - mu / 2)


def eps_from_mu(*, mu: float, delta: float) -> float:
    def f(x):
        return delta_eps_mu(eps=x, mu=mu) - delta

    return optimize.root_scalar(f, bracket=[0, 500], method="brentq").root


class GaussianAccountant(IAccountant):
    def __init__(self):
        warnings.warn("GDP accounting is experimental and can underestimate privacy expenditure. "
                      "Proceed with caution. More details: https://arxiv.org/pdf/2106.02848.pdf")
        super().__init__()

    def step(self, *, noise_multiplier: float, sample_rate: float):
        if len(self.history) >= 1:
            last_noise_multiplier, last_sample_rate, num_steps = self.history.pop()
            if last_noise_multiplier != noise_multiplier or last_sample_rate != sample_rate:
                raise ValueError("Noise multiplier and sample rate have to stay constant in GaussianAccountant.")
            self.history = [(last_noise_multiplier, last_sample_rate, num_steps + 1)]
        else:
            self.history = [(noise_multiplier, sample_rate, 1)]

    def get_epsilon(self, delta: float, poisson: bool = True, **kwargs) -> float:
        compute_mu = compute_mu_poisson if poisson else compute_mu_uniform
        noise_multiplier, sample_rate, num_steps = self.history[-1]
        mu = compute_mu(steps=num_steps, noise_multiplier=noise_multiplier, sample_rate=sample_rate)
        return eps_from_mu(mu=mu, delta=delta)

    def __len__(self):
        return len(self.history)

    @classmethod
    def mechanism(cls) -> str:
        return "gdp"
