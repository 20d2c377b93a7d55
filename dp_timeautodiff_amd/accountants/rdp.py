"""Renyi-DP accountant for the Poisson-subsampled Gaussian mechanism (closed form, float64).

Exact restatement of the algorithm of Opacus' `accountants/analysis/rdp.py` + `RDPAccountant`
as recorded in SURVEY.md Appendix A.8 (Mironov, Talwar, Zhang 2019, "Renyi Differential Privacy
of the Sampled Gaussian Mechanism"; epsilon conversion of Balle et al. 2020).  Reached from the
reference via `PrivacyEngine.get_epsilon` (Model_Code/timeautoencoder.py:759) when the engine is
built with accountant="rdp", and always used by the PRV accountant to bound its domain.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
from scipy import special

from .accountant import IAccountant

DEFAULT_ALPHAS = [1 + x / 10.0 for x in range(1, 100)] + list(range(12, 64))


def _log_add(logx: float, logy: float) -> float:
    a, b = min(logx, logy), max(logx, logy)
    if a == -np.inf:
        return b
    return math.log1p(math.exp(a - b)) + b


def _log_sub(logx: float, logy: float) -> float:
    if logx < logy:
        raise ValueError("The result of subtraction must be non-negative.")
    if logy == -np.inf:
        return logx
    if logx == logy:
        return -np.inf
    try:
        return math.log(math.expm1(logx - logy)) + logy
    except OverflowError:
        return logx


def _log_erfc(x: float) -> float:
    return math.log(2) + special.log_ndtr(-x * 2 ** 0.5)


def _compute_log_a_for_int_alpha(q: float, sigma: float, alpha: int) -> float:
    log_a = -np.inf
    for i in range(alpha + 1):
        log_coef_i = math.log(special.binom(alpha, i)) + i * math.log(q) + (alpha - i) * math.log(1 - q)
        s = log_coef_i + (i * i - i) / (2 * (sigma ** 2))
        log_a = _log_add(log_a, s)
    return float(log_a)


def _compute_log_a_for_frac_alpha(q: float, sigma: float, alpha: float) -> float:
    log_a0, log_a1 = -np.inf, -np.inf
    i = 0
    z0 = sigma ** 2 * math.log(1 / q - 1) + 0.5
    while True:
        coef = special.binom(alpha, i)
        log_coef = math.log(abs(coef))
        j = alpha - i
        log_t0 = log_coef + i * math.log(q) + j * math.log(1 - q)
        log_t1 = log_coef + j * math.log(q) + i * math.log(1 - q)
        log_e0 = math.log(0.5) + _log_erfc((i - z0) / (math.sqrt(2) * sigma))
        log_e1 = math.log(0.5) + _log_erfc((z0 - j) / (math.sqrt(2) * sigma))
        log_s0 = log_t0 + (i * i - i) / (2 * (sigma ** 2)) + log_e0
        log_s1 = log_t1 + (j * j - j) / (2 * (sigma ** 2)) + log_e1
        if coef > 0:
            log_a0 = _log_add(log_a0, log_s0)
            log_a1 = _log_add(log_a1, log_s1)
        else:
            log_a0 = _log_sub(log_a0, log_s0)
            log_a1 = _log_sub(log_a1, log_s1)
        i += 1
        if max(log_s0, log_s1) < -30:
            break
    return _log_add(log_a0, log_a1)


def _compute_log_a(q: float, sigma: float, alpha: float) -> float:
    if float(alpha).is_integer():
        return _compute_log_a_for_int_alpha(q, sigma, int(alpha))
    return _compute_log_a_for_frac_alpha(q, sigma, alpha)


def _compute_rdp(q: float, sigma: float, alpha: float) -> float:
    if q == 0:
        return 0
    if sigma == 0:
        return np.inf
    if q == 1.0:
        return alpha / (2 * sigma ** 2)
    if np.isinf(alpha):
        return np.inf
    return _compute_log_a(q, sigma, alpha) / (alpha - 1)


def compute_rdp(*, q: float, noise_multiplier: float, steps: int,
                orders: Union[Sequence[float], float]) -> Union[np.ndarray, float]:
    if isinstance(orders, float):
        rdp = _compute_rdp(q, noise_multiplier, orders)
    else:
        rdp = np.array([_compute_rdp(q, noise_multiplier, order) for order in orders])
    return rdp * steps


def get_privacy_spent(*, orders: Union[Sequence[float], float], rdp: Union[Sequence[float], float],
                      delta: float) -> Tuple[float, float]:
    orders_vec = np.atleast_1d(orders)
    rdp_vec = np.atleast_1d(rdp)
    if len(orders_vec) != len(rdp_vec):
        raise ValueError(f"Input lists must have the same length.\n\torders_vec = {orders_vec}\n\trdp_vec = {rdp_vec}\n")
    eps = (rdp_vec - (np.log(delta) + np.log(orders_vec)) / (orders_vec - 1)
           + np.log((orders_vec - 1) / orders_vec))
    if np.isnan(eps).all():
        return np.inf, np.nan
    idx_opt = np.nanargmin(eps)
    return float(eps[idx_opt]), float(orders_vec[idx_opt])


class RDPAccountant(IAccountant):
    DEFAULT_ALPHAS = DEFAULT_ALPHAS

    def __init__(self):
        super().__init__()

    def get_privacy_spent(self, *, delta: float, alphas: Optional[List[Union[float, int]]] = None):
        if not self.history:
            return 0, 0
        if alphas is None:
            alphas = self.DEFAULT_ALPHAS
        rdp = sum(compute_rdp(q=sample_rate, noise_multiplier=noise_multiplier, steps=num_steps, orders=alphas)
                  for (noise_multiplier, sample_rate, num_steps) in self.history)
        return get_privacy_spent(orders=alphas, rdp=rdp, delta=delta)

    def get_epsilon(self, delta: float, alphas: Optional[List[Union[float, int]]] = None, **kwargs) -> float:
        eps, _ = self.get_privacy_spent(delta=delta, alphas=alphas)
        return eps

    def __len__(self):
        return len(self.history)

    @classmethod
    def mechanism(cls) -> str:
        return "rdp"
