"""Pin the CPU oracle (oracle/opacus_ref.py) against the golden vectors produced from the
reference's own model classes (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import opacus_ref as O

TOL = 1e-5  # north_star: 1e-5 relative in fp32
FULL = ["vae_small_mixed.npz", "vae_small_nums.npz", "denoiser_small.npz"]


def _layer_kind(g, lname):
    if g.has(f"gs/{lname}.weight") and g[f"gs/{lname}.weight"].ndim == 3 and g[f"A/{lname}/0"].dtype.kind == "i":
        return "embedding"
    if "layer_norm" in lname:
        return "layernorm"
    return "linear"


@pytest.mark.parametrize("case", FULL)
def test_samplers_match_reference_autograd(golden, case):
    g = golden(case)
    B = g.meta["B"]
    checked = 0
    for lname, ncalls in g.layers.items():
        As = [torch.from_numpy(g[f"A/{lname}/{k}"]) for k in range(ncalls)]
        Gs = [torch.from_numpy(g[f"G/{lname}/{k}"]) for k in range(ncalls)]
        kind = _layer_kind(g, lname)
        if kind == "linear":
            gw, gb = O.accumulate_linear_calls(As, Gs, float(B), has_bias=g.has(f"gs/{lname}.bias"))
            assert rel_err(gw.numpy(), g[f"gs/{lname}.weight"]) < TOL, lname
            if gb is not None:
                assert rel_err(gb.numpy(), g[f"gs/{lname}.bias"]) < TOL, lname
        elif kind == "embedding":
            V = g[f"gs/{lname}.weight"].shape[1]
            acc = sum(O.embedding_grad_sample(a, gg * float(B), V) for a, gg in zip(As, Gs))
            assert rel_err(acc.numpy(), g[f"gs/{lname}.weight"]) < TOL, lname
        else:
            C = As[0].shape[-1]
            gw, gb = O.layernorm_grad_sample(As[0], Gs[0] * float(B), (C,), 1e-5)
            assert rel_err(gw.numpy(), g[f"gs/{lname}.weight"]) < TOL, lname
            assert rel_err(gb.numpy(), g[f"gs/{lname}.bias"]) < TOL, lname
        checked += 1
    assert checked >= 8


def test_repeated_layer_is_called_T_times(golden):
    g = golden("vae_small_nums.npz")
    T = g.meta["T"]
    assert g.layers["encoder_mu.update_gate"] == T
    assert g.layers["encoder_logvar.new_gate"] == T
    assert g.layers["Emb.mlp_nums.0"] == 1


@pytest.mark.parametrize("case", FULL)
def test_tail_matches_float64_formulas(golden, case):
    g = golden(case)
    names = g.names
    gs = [torch.from_numpy(g["gs/" + n]) for n in names]
    for ci, C in enumerate(g.meta["C_list"]):
        norms, clip, summed = O.clip_and_accumulate(gs, C)
        assert rel_err(norms.numpy(), g[f"tail{ci}/norms_f64"]) < TOL
        assert rel_err(clip.numpy(), g[f"tail{ci}/clip_f64"]) < TOL
        flat = np.concatenate([s.reshape(-1).numpy() for s in summed])
        assert rel_err(flat, g[f"tail{ci}/summed_f64"]) < TOL
        noise = torch.from_numpy(g[f"tail{ci}/noise"])
        offs = np.cumsum([0] + [s.numel() for s in summed])
        noises = [noise[o:e] for o, e in zip(offs[:-1], offs[1:])]
        grads = O.add_noise_and_scale(summed, noises, g.meta["sigma"], C, g.meta["expected_batch_size"])
        gflat = np.concatenate([x.reshape(-1).numpy() for x in grads])
        assert rel_err(gflat, g[f"tail{ci}/grad_f64"]) < TOL
    # both clamp branches are exercised by the second C (median norm)
    c1 = g["tail1/clip_f64"]
    assert (c1 < 1.0).any() and (c1 >= 1.0).any()


def test_flat_tail_equals_per_param_tail(golden):
    g = golden("vae_small_mixed.npz")
    flat = torch.from_numpy(g.gs_flat())
    C = g.meta["C_list"][1]
    noise = torch.from_numpy(g["tail1/noise"])
    norms, clip, summed, grad = O.dp_step_flat(flat, C, noise, g.meta["expected_batch_size"])
    assert rel_err(norms.numpy(), g["tail1/norms_f64"]) < TOL
    assert rel_err(summed.numpy(), g["tail1/summed_f64"]) < TOL
    assert rel_err(grad.numpy(), g["tail1/grad_f64"]) < TOL


def test_airquality_shape_fixture_is_consistent(golden):
    g = golden("vae_airquality_b64.npz")
    assert g.meta["P"] == 649400 and g.meta["B"] == 64 and g.meta["expected_batch_size"] == 49
    assert len(g.names) == 34
    # at the reference's C=1.0 nothing is clipped at init-scale weights
    assert g["tail0/norms_f64"].shape == (64,)
    c1 = g["tail1/clip_f64"]
    assert (c1 < 1.0).any() and (c1 >= 1.0).any()


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    z = O.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    z = O.philox4x32_10(f, np.full(2, 0xFFFFFFFF, np.uint32))[0]
    assert [hex(int(v)) for v in z] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    c = np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32)
    z = O.philox4x32_10(c, np.array([0xA4093822, 0x299F31D0], np.uint32))[0]
    assert [hex(int(v)) for v in z] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_normal_moments_and_offsets():
    z = O.philox_normal(1 << 18, seed=1234, offset=0, std=2.0)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 2.0) < 0.02
    # stream property: offset shifts by whole 4-blocks
    z2 = O.philox_normal(1024, seed=1234, offset=16, std=2.0)
    assert np.array_equal(z2, z[64:64 + 1024])
    assert not np.array_equal(O.philox_normal(64, 1235, 0, 2.0), z[:64])
