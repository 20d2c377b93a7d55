"""dp_job harness: the reference's train_dp_autoencoder protocol end to end on the GPU —
budget respected, graph and eager modes, checkpoint resume continues the same ledger and random streams."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(n_bins=0, n_cats=0, n_nums=5, cards=[], channels=64, hidden_size=24, num_layers=1, lr=1e-3,
          weight_decay=1e-6, batch_size=40, min_beta=1e-5, max_beta=0.1, emb_dim=16, time_dim=8, lat_dim=3,
          device="cuda:0", epsilon=3.0, delta=1e-5, max_grad_norm=1.0, accountant="rdp", log=lambda *a: None)


def _data():
    return torch.rand(800, 8, 5, generator=torch.Generator().manual_seed(3))


@pytest.mark.parametrize("use_graph", [True, False])
def test_training_respects_the_budget(use_graph):
    from dp_timeautodiff_amd.train import train_dp_autoencoder
    torch.manual_seed(0)
    ae, latent, output, mu, logvar, eps = train_dp_autoencoder(
        _data(), n_epochs=3, use_graph=use_graph, max_physical_batch_size=128 if use_graph else 16,
        noise_seed=1, sampler_seed=2, **KW)
    assert 0 < eps <= 3.0 * 1.02  # sigma is calibrated so that n_epochs spend (at most) the target
    assert latent.shape == (800, 8, 3) and torch.isfinite(latent).all()
    assert all(torch.isfinite(p).all() for p in ae.parameters())


def test_resume_continues_the_same_run(tmp_path):
    """A 4-epoch job interrupted after its first checkpoint and resumed == the same job run in one go (eager mode:
    identical launch sequence; ledger, DP-noise stream, sampler stream, device RNG and Adam state all restored)."""
    from dp_timeautodiff_amd.train import train_dp_autoencoder
    path = str(tmp_path / "ck.pt")

    torch.manual_seed(0)
    ae_full, *_, eps_full = train_dp_autoencoder(_data(), n_epochs=4, use_graph=False, noise_seed=1, sampler_seed=2,
                                                 **KW)

    class Stop(Exception):
        pass

    def log(msg):
        if str(msg).startswith("Epoch 2/4"):  # logged before epoch 2's checkpoint is written
            raise Stop

    torch.manual_seed(0)
    with pytest.raises(Stop):
        train_dp_autoencoder(_data(), n_epochs=4, use_graph=False, noise_seed=1, sampler_seed=2, save_path=path,
                             **{**KW, "log": log})
    ck = torch.load(path, weights_only=True)
    assert ck["train_state"]["epoch"] == 1 and len(ck["privacy_accountant_state_dict"]["history"]) >= 1
    torch.manual_seed(123)  # a resumed process starts from different seeds everywhere: the checkpoint must win
    ae_res, *_, eps_res = train_dp_autoencoder(_data(), n_epochs=4, use_graph=False, noise_seed=99, sampler_seed=98,
                                               load_path=path, **KW)
    assert abs(eps_res - eps_full) < 1e-9
    for a, b in zip(ae_res.parameters(), ae_full.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
