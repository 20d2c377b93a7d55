"""PRV accountant (the default of PrivacyEngine(), reference call Model_Code/timeautoencoder.py:672, read at :759)
against computations that share no code with it (tests/prv_checker.py).

Opacus itself is absent (requirements.txt pins nothing, no wheel offline), so EQUALITY with Opacus' PRV epsilon stays
unpinned; what these tests establish is that the epsilon this package reports is a VALID and TIGHT bound:
eps_lower <= eps_true <= eps_upper with eps_upper - eps_true within ~2 eps_error, for small exact compositions and
for the two tuples the north star names (the dp_job.sh run: q = 1/187, 3.74 M steps; Appendix A.8's: sigma = 1.5,
q = 0.04, 2250 steps).  The calibration outputs recalled from Opacus' own unit tests (accountants_test.py: target
eps 8, delta 1e-5, q 0.04, 90 epochs / 2000 steps -> 1.416 / 1.3562 RDP, 1.3477 / 1.2915 PRV) are matched digit for
digit by the bisection, which pins eps(sigma) around the target to the bisection's tolerance."""
import warnings

import pytest

import prv_checker as C
from dp_timeautodiff_amd.accountants import PRVAccountant, RDPAccountant
from dp_timeautodiff_amd.accountants.utils import get_noise_multiplier


def _prv(sigma, q, n, delta, eps_error=0.01):
    a = PRVAccountant()
    a.history = [(sigma, q, n)]
    d = a._get_dprv(eps_error=eps_error, delta_error=delta / 1000)
    return d.compute_epsilon(delta, delta / 1000, eps_error), d.domain


@pytest.mark.parametrize("sigma,q,n,delta", [(1.0, 0.3, 1, 1e-3), (0.8, 0.1, 2, 1e-4), (1.2, 0.5, 2, 1e-5),
                                             (2.0, 1.0, 2, 1e-5), (1.0, 0.2, 3, 1e-3)])
def test_prv_brackets_the_exact_epsilon_of_small_compositions(sigma, q, n, delta):
    warnings.simplefilter("ignore")
    exact = C.exact_epsilon(sigma, q, n, delta)
    (lo, est, up), _ = _prv(sigma, q, n, delta)
    assert lo <= exact <= up, (lo, exact, up)
    assert abs(est - exact) <= 0.01 and up - exact <= 0.025
    if q == 1.0 and n == 2:  # plain Gaussian, two steps = one step at sigma / sqrt(2): the analytic Gaussian delta
        from scipy.stats import norm
        mu = 2 ** 0.5 / sigma
        d = norm.cdf(-exact / mu + mu / 2) - 2.718281828459045 ** exact * norm.cdf(-exact / mu - mu / 2)
        assert abs(d - delta) < 1e-9


def test_one_step_closed_form_is_a_probability_statement():
    # h1 against brute-force quadrature of E_P[(1 - e^{eps - L})_+]
    from scipy import integrate
    import numpy as np
    sigma, q, eps = 0.9, 0.25, 0.4
    f = lambda x: C._p_density(x, sigma, q) * max(0.0, 1.0 - np.exp(eps - C._loss(x, sigma, q)))
    xs = float(C._xstar(eps, sigma, q))
    val = integrate.quad(f, xs, 12.0, epsabs=1e-13)[0]
    assert abs(val - C.h1(eps, sigma, q)) < 1e-10


@pytest.mark.parametrize("sigma,q,n,delta,name", [
    (1.5, 0.04, 2250, 1e-5, "A.8"),
    (9.27734375, 1 / 187, 3_740_000, 1e-5, "dp_job"),   # AirQuality dp_job.sh: 20000 epochs x 187 steps
])
def test_prv_upper_bounds_a_four_times_finer_independent_composition(sigma, q, n, delta, name):
    warnings.simplefilter("ignore")
    (lo, est, up), dom = _prv(sigma, q, n, delta)
    indep, mass_err = C.fine_grid_epsilon(sigma, q, n, delta, mesh=dom.dt / 4, L=dom.t_max)
    assert mass_err < 1e-9
    assert lo <= indep <= up, (name, lo, indep, up)
    assert abs(est - indep) <= 0.005 and 0 <= up - indep <= 0.025, (name, est, indep, up)
    r = RDPAccountant()
    r.history = [(sigma, q, n)]
    assert indep < r.get_epsilon(delta)          # and the RDP bound (the pinned accountant) is looser, as it must be


def test_calibration_matches_the_values_recalled_from_opacus_unit_tests():
    """opacus/tests/accountants_test.py (recalled; the parameter set was mis-remembered in SURVEY A.8, the outputs were
    not): get_noise_multiplier(target_epsilon=8, target_delta=1e-5, sample_rate=0.04, ...)."""
    warnings.simplefilter("ignore")
    kw = dict(target_epsilon=8.0, target_delta=1e-5, sample_rate=0.04)
    assert round(get_noise_multiplier(epochs=90, accountant="rdp", **kw), 4) == 1.4160
    assert round(get_noise_multiplier(steps=2000, accountant="rdp", **kw), 4) == 1.3562
    assert round(get_noise_multiplier(epochs=90, accountant="prv", **kw), 4) == 1.3477
    assert round(get_noise_multiplier(steps=2000, accountant="prv", **kw), 4) == 1.2915
