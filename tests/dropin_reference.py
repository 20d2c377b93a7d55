"""Helper of test_reference_dropin.py — runs in its own process, in the BUILD container only.

Imports the reference's OWN training module (Model_Code/timeautoencoder.py) with the repository root first on
sys.path, so that its four `from opacus ...` lines (:18-21) resolve to this repository's `opacus/` alias, patches the
ctypes layer with the CPU oracle stand-ins (tests/cpu_backend.py: there is no GPU here) and calls the reference's
`train_dp_autoencoder` (:603) unchanged on a slice of its own AirQuality.csv.  Prints one JSON line.
Nothing of the reference is copied or shipped: this file only names paths under /root/reference.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/TimeAutoDiff-main"
sys.path[:0] = [ROOT, HERE, os.path.join(REF, "Model_Code")]

import pandas as pd  # noqa: E402
import torch  # noqa: E402

import cpu_backend  # noqa: E402
cpu_backend.install()

import opacus  # noqa: E402
assert os.path.dirname(os.path.abspath(opacus.__file__)) == os.path.join(ROOT, "opacus"), opacus.__file__

import DP as dp  # noqa: E402  (the reference's data windowing, not differential privacy)
import timeautoencoder as tae  # noqa: E402
assert tae.__file__.startswith(REF)
assert tae.PrivacyEngine is opacus.PrivacyEngine

from dp_timeautodiff_amd.grad_sample import GradSampleModule  # noqa: E402
from dp_timeautodiff_amd.layers.fused_gru import FusedLinearCellGRU, fuse_recurrences, looks_like_linear_cell_gru  # noqa: E402

torch.manual_seed(0)
real_df = pd.read_csv(os.path.join(REF, "Dataset", "Single-Sequence", "AirQuality.csv")).iloc[:224]
real_df1 = real_df.drop("date", axis=1)
processed = dp.splitData(real_df1, 24, 1)[:200]          # the reference's load_dataset (timeautodiff_run.py:44)
time_info = dp.splitTimeData(real_df, processed.shape[1])  # noqa: F841  (as the reference does; unused by the DP loop)

out = tae.train_dp_autoencoder(real_df1, processed, channels=64, hidden_size=200, num_layers=1, lr=2e-4,
                               weight_decay=1e-6, n_epochs=1, batch_size=50, threshold=1, min_beta=1e-5, max_beta=0.1,
                               emb_dim=128, time_dim=8, lat_dim=7, device="cpu", epsilon=5.0)
ae, latent, output, losses, recons_loss, mu_z, logvar_z, eps = out

# the column-typing rule restated in dp_timeautodiff_amd/data_types.py against the reference's parser, on its own CSVs
import process_edited as pce  # noqa: E402
from dp_timeautodiff_amd import data_types  # noqa: E402
typing_ok = {}
for name in ("AirQuality", "Hurricane", "Metro_Traffic"):
    df = pd.read_csv(os.path.join(REF, "Dataset", "Single-Sequence", name + ".csv")).drop("date", axis=1)
    for thr in (1, 0.05):
        want = pce.DataFrameParser().fit(df, thr).datatype_info()
        got = data_types.datatype_info(df, thr)
        typing_ok[f"{name}@{thr}"] = (got == {k: (list(v) if k == "cards" else v) for k, v in want.items()}, got)

# the model class the reference builds, untouched, for the key comparison and the fuse predicate
ref_model = tae.DeapStack_DP(64, 50, 24, 0, 0, 13, [], 13, 200, 1, 128, 8, 7)
fused = fuse_recurrences(tae.DeapStack_DP(64, 50, 24, 0, 0, 13, [], 13, 200, 1, 128, 8, 7))
res = {
    "len": len(out), "eps": float(eps), "eps_type": type(eps).__name__,
    "is_gsm": isinstance(ae, GradSampleModule),
    "keys_equal": list(ae.state_dict().keys()) == list(ref_model.state_dict().keys()),
    "n_keys": len(ae.state_dict()),
    "custom_gru_recognised": bool(looks_like_linear_cell_gru(ref_model.encoder_mu)
                                  and looks_like_linear_cell_gru(ref_model.encoder_logvar)),
    "custom_gru_class": type(ref_model.encoder_mu).__name__,
    "fused_types": [type(fused.encoder_mu).__name__, type(fused.encoder_logvar).__name__],
    "fused_keys_equal": list(fused.state_dict().keys()) == list(ref_model.state_dict().keys()),
    "latent_shape": list(latent.shape), "output_is_tuple_or_tensor": type(output).__name__,
    "mu_shape": list(mu_z.shape), "losses": [len(losses), len(recons_loss)],
    "n_windows": int(processed.shape[0]),
    "decoder_forwarded": callable(getattr(ae, "decoder", None)),
    "finite": bool(torch.isfinite(latent).all()),
    "typing": typing_ok,
}
print(json.dumps(res), flush=True)
