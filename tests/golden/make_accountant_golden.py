#!/usr/bin/env python3
"""Accountant known answers computed by this repository's own accountants in the build container.

Opacus is not installable offline, so these pin REPRODUCIBILITY ("same inputs -> bit-identical epsilon on
every box"), not equality with Opacus' own numbers; the one value recalled from Opacus' unit tests
(RDP eps 7.32911117143 for sigma=1.5, q=0.04, 2250 steps, delta=1e-5) is asserted separately in
tests/test_accountants.py.   Usage: python tests/golden/make_accountant_golden.py
"""
import json, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from dp_timeautodiff_amd.accountants import create_accountant
from dp_timeautodiff_amd.accountants.utils import get_noise_multiplier

CASES = [
    # (sigma, q, steps, delta)
    (1.5, 0.04, 2250, 1e-5), (1.0, 0.01, 9000, 1e-5), (0.8, 1 / 187, 187 * 20, 1e-5), (3.0, 0.1, 500, 1e-6),
    (9.27734375, 1 / 187, 3740000, 1e-5),
]
out = {"epsilon": [], "history": [], "noise_multiplier": []}
for mech in ("rdp", "gdp", "prv"):
    for sigma, q, steps, delta in CASES:
        if mech == "prv" and steps > 100000:
            continue  # 25M-point FFT: covered by the sigma-calibration case below instead
        a = create_accountant(mech)
        a.history = [(sigma, q, steps)]
        out["epsilon"].append(dict(mech=mech, sigma=sigma, q=q, steps=steps, delta=delta, eps=a.get_epsilon(delta)))
for mech in ("rdp", "prv"):
    a = create_accountant(mech)
    a.history = [(1.2, 0.02, 300), (0.9, 0.05, 100), (1.2, 0.02, 50)]
    out["history"].append(dict(mech=mech, history=a.history, delta=1e-5, eps=a.get_epsilon(1e-5)))
for mech, eps, q, epochs in (("rdp", 2.0, 0.01, 90), ("prv", 2.0, 0.01, 90), ("rdp", 1.0, 1 / 187, 20000),
                             ("rdp", 5.0, 1 / 187, 20000), ("prv", 5.0, 1 / 12, 30)):
    out["noise_multiplier"].append(dict(mech=mech, target_epsilon=eps, delta=1e-5, q=q, epochs=epochs,
                                        sigma=get_noise_multiplier(target_epsilon=eps, target_delta=1e-5,
                                                                   sample_rate=q, epochs=epochs, accountant=mech)))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "accountant.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
