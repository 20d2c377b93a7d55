"""Host-side pieces that need no kernels: Poisson loader, BatchMemoryManager, ModuleValidator, DP layers."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

from dp_timeautodiff_amd.batch_memory_manager import BatchMemoryManager
from dp_timeautodiff_amd.data_loader import DPDataLoader
from dp_timeautodiff_amd.layers import DPGRU, DPLSTM, DPMultiheadAttention
from dp_timeautodiff_amd.validators import ModuleValidator


def test_dp_data_loader_matches_reference_settings():
    # AirQuality dp_job.sh: N=9333, batch 50 -> len 187, q=1/187, expected batch int(9333/187)=49
    data = torch.zeros(9333, 2)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=50, shuffle=True)
    assert len(dl) == 187
    g = torch.Generator().manual_seed(0)
    dp = DPDataLoader.from_data_loader(dl, generator=g)
    assert dp.sample_rate == 1 / 187 and len(dp) == 187
    assert int(len(dp.dataset) * dp.sample_rate) == 49
    sizes = [b[0].shape[0] for b in dp]
    assert len(sizes) == 187 and 35 < np.mean(sizes) < 65 and len(set(sizes)) > 5


def test_empty_poisson_batch_collates_to_zero_length():
    data = torch.zeros(3, 4, 5)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=1)
    dp = DPDataLoader(dl.dataset, sample_rate=1e-9)
    b = next(iter(dp))
    assert b[0].shape == (0, 4, 5) and b[0].dtype == torch.float32


class _FakeOpt:
    def __init__(self):
        self.q = []

    def signal_skip_step(self, do_skip=True):
        self.q.append(do_skip)


def test_batch_memory_manager_chunks_and_flags():
    data = torch.arange(1000).float().unsqueeze(1)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=250)
    dp = DPDataLoader.from_data_loader(dl, generator=torch.Generator().manual_seed(1))
    opt = _FakeOpt()
    with BatchMemoryManager(data_loader=dp, max_physical_batch_size=128, optimizer=opt) as safe:
        sizes = [b[0].shape[0] for b in safe]
    assert all(s <= 128 for s in sizes)
    assert len(opt.q) == len(sizes) and opt.q.count(False) == len(dp)  # one real step per logical batch
    # chunks of one logical batch: all but the last are skip=True
    i = 0
    for _ in range(len(dp)):
        n = 0
        while opt.q[i]:
            i += 1; n += 1
        i += 1
        assert n == math.ceil((sum(sizes[i - n - 1:i])) / 128) - 1


def test_validator_fix_replaces_rnn_and_keeps_outputs():
    torch.manual_seed(0)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = nn.LSTM(5, 6, 2, batch_first=True, bidirectional=True)
            self.gru = nn.GRU(12, 4, batch_first=True)
            self.mha = nn.MultiheadAttention(4, 2, batch_first=True)
            self.fc = nn.Linear(4, 3)

        def forward(self, x):
            h = self.gru(self.lstm(x)[0])[0]
            h = self.mha(h, h, h)[0]
            return self.fc(h)

    net = Net()
    errs = ModuleValidator.validate(net)
    assert len(errs) == 3 and not ModuleValidator.is_valid(net)
    with pytest.raises(ValueError):
        ModuleValidator.validate(net, strict=True)
    fixed = ModuleValidator.fix(net)
    assert ModuleValidator.is_valid(fixed)
    assert isinstance(fixed.lstm, DPLSTM) and isinstance(fixed.gru, DPGRU)
    assert isinstance(fixed.mha, DPMultiheadAttention)
    bn = ModuleValidator.fix(nn.Sequential(nn.BatchNorm1d(4)))
    assert isinstance(bn[0], nn.GroupNorm) and not ModuleValidator.is_valid(bn)  # no GroupNorm kernel in this build
    assert net.lstm is not fixed.lstm and isinstance(net.lstm, nn.LSTM)  # deep clone, original untouched
    x = torch.randn(3, 7, 5)
    net.eval(); fixed.eval()
    assert (net(x) - fixed(x)).abs().max() < 1e-5
    # state_dict interchange under the nn.LSTM names
    assert [k for k in fixed.state_dict() if k.startswith("lstm.")] == [k for k in net.state_dict() if k.startswith("lstm.")]


def test_deapstack_needs_no_fix_and_param_inventory():
    from dp_timeautodiff_amd import models
    m = models.DeapStackDP(64, 50, 24, 0, 0, 13, [], 13, 200, 1, 128, 8, 7)
    assert ModuleValidator.is_valid(m)
    assert sum(p.numel() for p in m.parameters()) == 649400 and len(list(m.parameters())) == 34
    d = models.BiRNNScoreDP(7, 250, 2, 100)
    assert sum(p.numel() for p in d.parameters()) == 2830257 and len(list(d.parameters())) == 32
    assert ModuleValidator.is_valid(d)
    m2 = models.DeapStackDP(64, 1024, 24, 0, 0, 32, [], 32, 128, 1, 128, 8, 7)
    assert sum(p.numel() for p in m2.parameters()) == 824398


def test_oracle_poisson_select_is_a_bernoulli_draw_per_sample():
    from oracle import opacus_ref as O
    N, q = 200000, 0.01
    a = O.poisson_select(N, q, seed=5, offset=0)
    assert np.array_equal(a, O.poisson_select(N, q, seed=5, offset=0))          # counter-based: reproducible
    assert not np.array_equal(a, O.poisson_select(N, q, seed=5, offset=N // 4))  # next draw of the stream
    assert (np.diff(a) > 0).all() and a.min() >= 0 and a.max() < N
    assert abs(len(a) - N * q) < 6 * (N * q * (1 - q)) ** 0.5
    # consecutive draws tile the stream: draw k uses blocks [k*ceil(N/4), (k+1)*ceil(N/4))
    b = O.poisson_select(8, 0.5, seed=1, offset=2)
    c = O.poisson_select(16, 0.5, seed=1, offset=0)
    assert np.array_equal(b, c[c >= 8] - 8)
    assert len(O.poisson_select(64, 0.0, 1, 0)) == 0 and len(O.poisson_select(64, 1.0, 1, 0)) == 64


def test_gemm_tuning_is_a_no_op_without_a_gpu():
    import torch
    from dp_timeautodiff_amd import gemm_tuning
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() is False
    gemm_tuning.freeze()  # never raises, tuned or not


def test_noise_and_sampler_streams_are_domain_separated():
    """one --seed used to key both the Poisson sampler and the DP noise with the same Philox word,
    so the published noise was a function of the inclusion uniforms.  Equal user seeds must give unrelated streams."""
    from dp_timeautodiff_amd import rng
    from oracle import opacus_ref as O
    for seed in (0, 1, 5, 2**63 + 5):
        kn, ks = rng.stream_key(seed, "noise"), rng.stream_key(seed, "sampler")
        assert kn != ks and kn != seed and ks != seed
        assert kn == O.stream_key(seed, "noise") and ks == O.stream_key(seed, "sampler")  # checker restates it
        ctr = np.zeros((4, 4), np.uint32)
        ctr[:, 0] = np.arange(4)
        key = lambda k: np.array([k & 0xFFFFFFFF, k >> 32], np.uint32)
        bn, bs = O.philox4x32_10(ctr, key(kn)), O.philox4x32_10(ctr, key(ks))
        assert not (bn == bs).any()  # block 0..3: not one 32-bit word in common
    # per-rank sampler keys differ, and none of them is the noise key
    ks = {rng.stream_key((9 + rng.RANK_STRIDE * r) & (2**64 - 1), "sampler") for r in range(8)}
    assert len(ks) == 8 and rng.stream_key(9, "noise") not in ks


def test_secure_mode_is_refused_not_degraded():
    """Opacus fails hard when it cannot provide a cryptographically secure generator; so does this build
    (the Philox noise kernels and the device sampler are not CSPRNG-backed)."""
    from dp_timeautodiff_amd.privacy_engine import PrivacyEngine
    with pytest.raises(NotImplementedError):
        PrivacyEngine(secure_mode=True)
    PrivacyEngine(secure_mode=False)
