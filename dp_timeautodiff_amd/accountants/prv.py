"""PRV accountant (Gopi, Lee, Wutschitz 2021, "Numerical Composition of Differential Privacy"):
the default of `PrivacyEngine()` in the Opacus versions the reference was written against, hence the
accountant behind `privacy_engine.get_epsilon(delta)` at Model_Code/timeautoencoder.py:759 and behind
the sigma calibration of `make_private_with_epsilon` (:674-682).

Restated from the published algorithm / Opacus' `accountants/analysis/prv` as recorded in SURVEY.md
Appendix A.8: Poisson-subsampled-Gaussian privacy random variable -> truncate to [-L, L] (L from an
RDP bound) -> mean-matching discretisation on a mesh of
2*eps_error / sqrt(2 n ln(12/delta_error)) -> n-fold self composition by real FFT -> upper epsilon
estimate.  Pure numpy/scipy float64.  Opacus itself is not available offline, so equality with
*Opacus'* epsilon is unpinned; the build's own epsilon is reproducible bit for bit.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np
from scipy import integrate
from scipy.fft import irfft, rfft
from scipy.signal import convolve
from scipy.special import erfc

from .accountant import IAccountant
from .rdp import RDPAccountant

SQRT2 = np.sqrt(2)


def _norm_cdf(x):
    # scipy.stats.norm.cdf without the distribution-object overhead (same formula: ndtr)
    return 0.5 * erfc(-np.asarray(x, dtype=np.float64) / SQRT2)


@dataclass
class Domain:
    """Uniform grid on [t_min, t_max] with `size` points; `shifts` tracks mean-matching offsets."""
    t_min: float
    t_max: float
    size: int
    shifts: float = 0.0

    def __post_init__(self):
        if not isinstance(self.size, int):
            raise TypeError("`size` must be an integer")
        if self.size % 2 != 0:
            raise ValueError("`size` must be even")

    @classmethod
    def create_aligned(cls, t_min: float, t_max: float, dt: float) -> "Domain":
        t_min = np.floor(t_min / dt) * dt
        t_max = np.ceil(t_max / dt) * dt
        size = int(np.round((t_max - t_min) / dt)) + 1
        if size % 2 == 1:
            size += 1
            t_max += dt
        domain = cls(t_min, t_max, size)
        if np.abs(domain.dt - dt) / dt >= 1e-8:
            raise RuntimeError
        return domain

    def shift_right(self, dt: float) -> "Domain":
        return Domain(t_min=self.t_min + dt, t_max=self.t_max + dt, size=self.size, shifts=self.shifts + dt)

    @property
    def dt(self):
        return (self.t_max - self.t_min) / (self.size - 1)

    @property
    def ts(self):
        return np.linspace(self.t_min, self.t_max, self.size)

    def __getitem__(self, i: int) -> float:
        return self.t_min + i * self.dt

    def __len__(self):
        return self.size


class PoissonSubsampledGaussianPRV:
    r"""Privacy loss random variable of the Poisson-subsampled Gaussian mechanism (remove adjacency)."""

    def __init__(self, sample_rate: float, noise_multiplier: float):
        self.sample_rate = sample_rate
        self.noise_multiplier = noise_multiplier

    def pdf(self, t):
        q = self.sample_rate
        sigma = self.noise_multiplier
        z = np.log((np.exp(t) + q - 1) / q)
        return np.where(
            t > np.log(1 - q),
            sigma * np.exp(-(sigma ** 2) * z ** 2 / 2 - 1 / (8 * sigma ** 2) + 2 * t)
            / (SQRT2 * np.sqrt(np.pi) * (np.exp(t) + q - 1) * ((np.exp(t) + q - 1) / q) ** 0.5),
            0.0,
        )

    def cdf(self, t):
        q = self.sample_rate
        sigma = self.noise_multiplier
        with np.errstate(invalid="ignore", divide="ignore"):
            z = np.log((np.exp(t) + q - 1) / q)
            return np.where(
                t > np.log(1 - q),
                -q * _norm_cdf(-z * sigma + 1 / (2 * sigma)) - (1 - q) * _norm_cdf(-z * sigma - 1 / (2 * sigma)) + 1.0,
                0.0,
            )

    def rdp(self, alpha: float) -> float:
        from .rdp import _compute_rdp
        return _compute_rdp(self.sample_rate, self.noise_multiplier, alpha)


class TruncatedPrivacyRandomVariable:
    def __init__(self, prv, t_min: float, t_max: float):
        self._prv = prv
        self.t_min = t_min
        self.t_max = t_max
        self._remaining_mass = self._prv.cdf(t_max) - self._prv.cdf(t_min)

    def pdf(self, t):
        return np.where(t < self.t_min, 0.0,
                        np.where(t < self.t_max, self._prv.pdf(t) / self._remaining_mass, 0.0))

    def cdf(self, t):
        return np.where(t < self.t_min, 0.0,
                        np.where(t < self.t_max,
                                 (self._prv.cdf(t) - self._prv.cdf(self.t_min)) / self._remaining_mass, 1.0))

    def mean(self) -> float:
        """E[t] = int t pdf(t) dt by parts over log-spaced panels around 0."""
        points = np.concatenate([[self.t_min], -np.logspace(-5, -1, 5)[::-1], np.logspace(-5, -1, 5), [self.t_max]])
        mean = 0.0
        for left, right in zip(points[:-1], points[1:]):
            integral, _ = integrate.quad(self.cdf, left, right, limit=500)
            mean += right * self.cdf(right) - left * self.cdf(left) - integral
        return float(mean)


@dataclass
class DiscretePRV:
    pmf: np.ndarray
    domain: Domain

    def __len__(self) -> int:
        if len(self.pmf) != self.domain.size:
            raise ValueError("pmf and domain must have the same length")
        return len(self.pmf)

    def compute_epsilon(self, delta: float, delta_error: float, eps_error: float) -> Tuple[float, float, float]:
        if delta <= 0:
            return (float("inf"),) * 3
        if np.finfo(np.longdouble).eps * self.domain.size > delta - delta_error:
            raise ValueError("Floating point errors will dominate for such small values of delta. "
                             "Increase delta or reduce domain size.")
        t = self.domain.ts
        th = np.flip(np.flip(self.pmf).cumsum())
        tl = np.flip(np.flip(self.pmf * np.exp(-t)).cumsum())
        ndelta = np.exp(t) * tl - th

        def find_epsilon(delta_target):
            i = np.searchsorted(ndelta, -delta_target, side="left")
            if i <= 0:
                raise RuntimeError("Cannot compute epsilon")
            return np.log((th[i] - delta_target) / tl[i])

        eps_upper = find_epsilon(delta - delta_error) + eps_error
        eps_lower = find_epsilon(delta + delta_error) - eps_error
        eps_estimate = find_epsilon(delta)
        return float(eps_lower), float(eps_estimate), float(eps_upper)

    def compute_delta_estimate(self, eps: float) -> float:
        return float(np.where(self.domain.ts > eps, self.pmf * (1.0 - np.exp(eps) * np.exp(-self.domain.ts)), 0.0).sum())


def discretize(prv, domain: Domain) -> DiscretePRV:
    tC = domain.ts
    tL = tC - domain.dt / 2
    tR = tC + domain.dt / 2
    discrete_pmf = prv.cdf(tR) - prv.cdf(tL)
    mean_d = np.dot(domain.ts, discrete_pmf)
    mean_c = prv.mean()
    mean_shift = mean_c - mean_d
    if np.abs(mean_shift) >= domain.dt / 2:
        raise RuntimeError("Discrete mean differs significantly from continuous mean.")
    domain_shifted = domain.shift_right(mean_shift)
    return DiscretePRV(pmf=discrete_pmf, domain=domain_shifted)


def _compose_fourier(dprv: DiscretePRV, num_self_composition: int) -> DiscretePRV:
    if len(dprv) % 2 != 0:
        raise ValueError("Can only compose evenly sized discrete PRVs")
    composed_pmf = irfft(rfft(dprv.pmf) ** num_self_composition)
    m = num_self_composition - 1
    if num_self_composition % 2 == 0:
        m += len(composed_pmf) // 2
    composed_pmf = np.roll(composed_pmf, m)
    domain = dprv.domain.shift_right(dprv.domain.shifts * (num_self_composition - 1))
    return DiscretePRV(pmf=composed_pmf, domain=domain)


def _compose_two(dprv_left: DiscretePRV, dprv_right: DiscretePRV) -> DiscretePRV:
    pmf = convolve(dprv_left.pmf, dprv_right.pmf, mode="same")
    domain = dprv_left.domain.shift_right(dprv_right.domain.shifts)
    return DiscretePRV(pmf=pmf, domain=domain)


def _compose_convolution_tree(dprvs: Sequence[DiscretePRV]) -> DiscretePRV:
    dprvs = list(dprvs)
    while len(dprvs) > 1:
        dprvs_conv = []
        if len(dprvs) % 2 == 1:
            dprvs_conv.append(dprvs.pop())
        for dprv_left, dprv_right in zip(dprvs[:-1:2], dprvs[1::2]):
            dprvs_conv.append(_compose_two(dprv_left, dprv_right))
        dprvs = dprvs_conv
    return dprvs[0]


def compose_heterogeneous(dprvs: Sequence[DiscretePRV], num_self_compositions: Sequence[int]) -> DiscretePRV:
    if len(dprvs) != len(num_self_compositions):
        raise ValueError("dprvs and num_self_compositions must have the same length")
    dprvs = [_compose_fourier(dprv, n) for dprv, n in zip(dprvs, num_self_compositions)]
    return _compose_convolution_tree(dprvs)


def compute_safe_domain_size(prvs, max_self_compositions: Sequence[int], eps_error: float,
                             delta_error: float) -> float:
    """L such that the truncated mass is negligible: RDP epsilon bounds at delta_error-scaled deltas."""
    total_compositions = sum(max_self_compositions)
    rdp_accountant = RDPAccountant()
    for prv, max_self_composition in zip(prvs, max_self_compositions):
        rdp_accountant.history.append((prv.noise_multiplier, prv.sample_rate, max_self_composition))
    L_max = rdp_accountant.get_epsilon(delta=delta_error / 4)
    for prv, _ in zip(prvs, max_self_compositions):
        rdp_accountant = RDPAccountant()
        rdp_accountant.history = [(prv.noise_multiplier, prv.sample_rate, 1)]
        L_max = max(L_max, rdp_accountant.get_epsilon(delta=delta_error / (8 * total_compositions)))
    L_max = max(L_max, eps_error) + 3
    return L_max


class PRVAccountant(IAccountant):
    def __init__(self):
        super().__init__()

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

    def get_epsilon(self, delta: float, *, eps_error: float = 0.01, delta_error: float = None) -> float:
        if delta_error is None:
            delta_error = delta / 1000
        dprv = self._get_dprv(eps_error=eps_error, delta_error=delta_error)
        _, _, eps_upper = dprv.compute_epsilon(delta, delta_error, eps_error)
        return eps_upper

    def _get_dprv(self, eps_error, delta_error) -> DiscretePRV:
        prvs = [PoissonSubsampledGaussianPRV(sample_rate, noise_multiplier)
                for noise_multiplier, sample_rate, _ in self.history]
        num_self_compositions = [steps for _, _, steps in self.history]
        domain = self._get_domain(prvs=prvs, num_self_compositions=num_self_compositions, eps_error=eps_error,
                                  delta_error=delta_error)
        tprvs = [TruncatedPrivacyRandomVariable(prv, domain.t_min, domain.t_max) for prv in prvs]
        dprvs = [discretize(tprv, domain) for tprv in tprvs]
        return compose_heterogeneous(dprvs=dprvs, num_self_compositions=num_self_compositions)

    def _get_domain(self, prvs, num_self_compositions: List[int], eps_error: float, delta_error: float) -> Domain:
        total_self_compositions = sum(num_self_compositions)
        L = compute_safe_domain_size(prvs=prvs, max_self_compositions=num_self_compositions, eps_error=eps_error,
                                     delta_error=delta_error)
        mesh_size = 2 * eps_error / np.sqrt(2 * total_self_compositions * np.log(12 / delta_error))
        return Domain.create_aligned(-L, L, mesh_size)

    def __len__(self):
        return len(self.history)

    @classmethod
    def mechanism(cls) -> str:
        return "prv"
