"""dp_job harness: the reference's train_dp_autoencoder protocol end to end on the GPU —
budget respected, graph and eager modes, checkpoint resume continues the same ledger and random streams."""
import pytest
import torch

pytestmark = pytest.mark.gpu

INFO = dict(n_bins=0, n_cats=0, n_nums=5, cards=[])
ARGS = dict(channels=64, hidden_size=24, num_layers=1, lr=1e-3, weight_decay=1e-6, batch_size=40, threshold=1,
            min_beta=1e-5, max_beta=0.1, emb_dim=16, time_dim=8, lat_dim=3, device="cuda:0", epsilon=3.0)
KW = dict(delta=1e-5, max_grad_norm=1.0, accountant="rdp", log=lambda *a: None)


def _data(n=800):
    return torch.rand(n, 8, 5, generator=torch.Generator().manual_seed(3))


def _train(data, n_epochs, info=INFO, **kw):
    """The reference's positional call (timeautodiff_run.py:315-333), engine options as keywords."""
    from dp_timeautodiff_amd.train import train_dp_autoencoder
    a = ARGS
    return train_dp_autoencoder(info, data, a["channels"], a["hidden_size"], a["num_layers"], a["lr"], a["weight_decay"],
                                n_epochs, a["batch_size"], a["threshold"], a["min_beta"], a["max_beta"], a["emb_dim"],
                                a["time_dim"], a["lat_dim"], a["device"], a["epsilon"], **{**KW, **kw})


@pytest.mark.parametrize("use_graph", [True, False])
def test_training_respects_the_budget(use_graph):
    torch.manual_seed(0)
    out = _train(_data(), 3, use_graph=use_graph, max_physical_batch_size=128 if use_graph else 16, noise_seed=1,
                 sampler_seed=2)
    assert len(out) == 8  # the reference's tuple (timeautoencoder.py:786)
    ae, latent, output, losses, recons_loss, mu, logvar, eps = out
    assert 0 < eps <= 3.0 * 1.02  # sigma is calibrated so that n_epochs spend (at most) the target
    assert latent.shape == (800, 8, 3) and torch.isfinite(latent).all()
    assert mu.shape == logvar.shape == (800, 8, 3) and "nums" in output
    assert len(losses) == 3 and len(recons_loss) == 1 and all(l == l for l in losses)
    assert all(torch.isfinite(p).all() for p in ae.parameters())


@pytest.mark.parametrize("use_graph", [True, False])
def test_three_steps_match_the_oracle_step_by_step(use_graph, monkeypatch):
    """f4 parity: the harness — device Poisson sampler, chunking / graph replay, K1..K7, Philox noise,
    K9 Adam — against the CPU oracle (reference op order) driven with the SAME Poisson indices (oracle.poisson_select)
    and the SAME noise (oracle.philox_normal): parameters after 3 logical steps within 1e-5."""
    import numpy as np
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.accountants.utils import get_noise_multiplier
    from oracle import opacus_ref as O
    # the VAE draws its reparametrisation noise from the device RNG: make it a fixed function of (mu, logvar) on both sides
    monkeypatch.setattr(models.DeapStackDP, "reparametrize", lambda self, mu, logvar: mu + 0.5 * torch.exp(0.5 * logvar))
    N, S_SEED, Z_SEED, C = 120, 21, 22, 0.05
    data = _data(N)
    a = ARGS
    torch.manual_seed(0)
    ae, *_ = _train(data, 1, use_graph=use_graph, max_physical_batch_size=128 if use_graph else 16, noise_seed=Z_SEED,
                    sampler_seed=S_SEED, max_grad_norm=C)
    arena = ae.arena

    torch.manual_seed(0)  # the harness builds its model on the CPU under this seed, then moves it
    ref = models.DeapStackDP(a["channels"], a["batch_size"], 8, 0, 0, 5, [], 5, a["hidden_size"], a["num_layers"],
                             a["emb_dim"], a["time_dim"], a["lat_dim"]).train()
    O.OracleGradSampleModule(ref)
    params = list(ref.parameters())
    adam = torch.optim.Adam(params, lr=a["lr"], weight_decay=a["weight_decay"])
    q = 1.0 / 3.0  # len(DataLoader(120 windows, batch 40)) = 3 -> sample rate 1/3, 3 steps per epoch
    sigma = get_noise_multiplier(target_epsilon=a["epsilon"], target_delta=1e-5, sample_rate=q, epochs=1,
                                 accountant="rdp")
    expected = int(N * q)
    blocks = (N + 3) // 4
    gpu_params = list(ae.parameters())
    for k in range(3):
        idx = O.poisson_select(N, q, O.stream_key(S_SEED, "sampler"), k * blocks)
        for p in params:
            p.grad_sample = None
        adam.zero_grad()
        models.vae_loss(ref, data[idx], beta=a["max_beta"]).backward()
        _, _, summed = O.clip_and_accumulate([p.grad_sample for p in params], C)
        z = torch.from_numpy(O.philox_normal(arena.P, O.stream_key(Z_SEED, "noise"), k * (arena.P // 4), sigma * C))
        for p, pg, s in zip(params, gpu_params, summed):
            p.grad = ((s + arena.vec_view(z, pg).reshape(s.shape)) / expected).to(torch.float32)
        adam.step()
    worst = 0.0
    for p, pg in zip(params, gpu_params):
        d = (pg.detach().cpu().double() - p.detach().double()).abs().max()
        worst = max(worst, float(d / p.detach().double().abs().max().clamp_min(1e-30)))
    assert worst < 1e-5, worst


def test_resume_continues_the_same_run(tmp_path):
    """A 4-epoch job interrupted after its first checkpoint and resumed == the same job run in one go (eager mode:
    identical launch sequence; ledger, DP-noise stream, sampler stream, device RNG and Adam state all restored)."""
    path = str(tmp_path / "ck.pt")

    torch.manual_seed(0)
    ae_full, *_, eps_full = _train(_data(), 4, use_graph=False, noise_seed=1, sampler_seed=2)

    class Stop(Exception):
        pass

    def log(msg):
        if str(msg).startswith("Epoch 2/4"):  # logged before epoch 2's checkpoint is written
            raise Stop

    torch.manual_seed(0)
    with pytest.raises(Stop):
        _train(_data(), 4, use_graph=False, noise_seed=1, sampler_seed=2, save_path=path, log=log)
    ck = torch.load(path, weights_only=True)
    assert ck["train_state"]["epoch"] == 1 and len(ck["privacy_accountant_state_dict"]["history"]) >= 1
    torch.manual_seed(123)  # a resumed process starts from different seeds everywhere: the checkpoint must win
    ae_res, *_, eps_res = _train(_data(), 4, use_graph=False, noise_seed=99, sampler_seed=98, load_path=path)
    assert abs(eps_res - eps_full) < 1e-9
    for a, b in zip(ae_res.parameters(), ae_full.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
