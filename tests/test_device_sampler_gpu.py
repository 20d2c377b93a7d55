"""K11 — on-device Poisson sampler and fixed-capacity masked batches.

Checker: oracle.poisson_select (numpy restatement of Opacus' `(rand(N) < q).nonzero()` on the kernel's Philox
stream) — index lists must match exactly (integer work: bit-exact).  The masked-batch property (padding rows get
clip factor 0) is checked against the same step on the exact n-row batch."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,q,seed,offset,capacity", [
    (9333, 50 / 9333, 7, 0, 128), (9333, 50 / 9333, 7, 12345678901, 128), (1, 0.5, 3, 0, 4), (4097, 0.3, 11, 5, 4097),
    (100000, 0.01, 2**63 + 5, 2**40, 1200), (5000, 0.2, 1, 0, 300), (64, 1.0, 9, 0, 64), (64, 0.0, 9, 0, 8),
])
def test_poisson_sample_matches_oracle(N, q, seed, offset, capacity):
    from dp_timeautodiff_amd import _native
    from oracle import opacus_ref as O
    idx = torch.full((capacity,), -1, dtype=torch.int64, device="cuda")
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    _native.poisson_sample(seed, offset, q, N, idx, counts)
    want = O.poisson_select(N, q, seed, offset)
    n_kept, n_sel = counts.cpu().tolist()
    assert n_sel == len(want) and n_kept == min(len(want), capacity)
    got = idx.cpu().numpy()
    assert np.array_equal(got[:n_kept], want[:n_kept])
    assert (got[n_kept:] == 0).all()
    if N == 5000:
        assert n_sel > capacity  # this case exercises truncation


def test_loader_iterates_like_a_poisson_dataloader():
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    from oracle import opacus_ref as O
    data = torch.arange(2000 * 3 * 2, dtype=torch.float32).reshape(2000, 3, 2)
    labels = torch.arange(2000)
    dl = DevicePoissonLoader([data, labels], sample_rate=0.02, device="cuda", seed=21)
    assert len(dl) == 50 and len(dl.dataset) == 2000
    blocks = (2000 + 3) // 4
    seen = []
    for k, (xb, yb) in enumerate(dl):
        want = O.poisson_select(2000, 0.02, O.stream_key(21, "sampler"), k * blocks)
        assert np.array_equal(yb.cpu().numpy(), want)
        assert torch.equal(xb.cpu(), data[want])
        seen.append(len(want))
        if k == 4:
            break
    assert len(set(seen)) > 1 and dl.overflow_steps() == 0


def test_a_draw_beyond_the_capacity_is_never_silently_truncated():
    """the iterator grows the capacity and repeats the same draw (every selected sample is yielded, the
    Philox position of the next step is the one an untruncated run has); the shape-static form cannot grow (a hipGraph
    holds its tensors), counts the overflow on the device and check_overflow() raises."""
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader, PoissonCapacityError
    from oracle import opacus_ref as O
    N, q, seed = 5000, 0.2, 1                     # ~1000 selected per draw
    labels = torch.arange(N)
    blocks = (N + 3) // 4
    dl = DevicePoissonLoader([labels], sample_rate=q, device="cuda", seed=seed, capacity=300, steps=3)
    for k, (yb,) in enumerate(dl):
        want = O.poisson_select(N, q, O.stream_key(seed, "sampler"), k * blocks)
        assert len(want) > 300 and np.array_equal(yb.cpu().numpy(), want)
    assert dl.capacity >= 1000 and dl.overflow_steps() == 0
    dl.check_overflow()
    st = DevicePoissonLoader([labels], sample_rate=q, device="cuda", seed=seed, capacity=300)
    st.sample_static()
    with pytest.raises(PoissonCapacityError, match="truncated"):
        st.check_overflow()
    ok = DevicePoissonLoader([labels], sample_rate=q, device="cuda", seed=seed, capacity=1400)
    ok.sample_static()
    ok.check_overflow()


def test_rank_shards_are_disjoint_exhaustive_and_independently_keyed():
    """Data-parallel sampling: rank r of W owns samples r, r+W, ...; the shards partition the dataset,
    every rank draws from its own Philox key, and the union of one step's draws is what the per-rank oracle streams
    select — a Poisson sample of the whole dataset at the same rate."""
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    from oracle import opacus_ref as O
    N, W, q, seed = 1003, 4, 0.05, 17   # N not a multiple of W
    ids = torch.arange(N)
    loaders = [DevicePoissonLoader([ids], sample_rate=q, device="cuda", seed=seed, rank=r, world_size=W) for r in range(W)]
    owned = [l.tensors[0].cpu().numpy() for l in loaders]
    assert sorted(np.concatenate(owned).tolist()) == list(range(N))          # disjoint and jointly exhaustive
    assert all(np.array_equal(o, np.arange(r, N, W)) for r, o in enumerate(owned))
    assert len({l.key for l in loaders}) == W                                # a stream per rank
    drawn = []
    for r, l in enumerate(loaders):
        (yb,) = next(iter(l))
        want_local = O.poisson_select(len(owned[r]), q, l.key, 0)
        assert np.array_equal(yb.cpu().numpy(), owned[r][want_local])
        drawn.append(yb.cpu().numpy())
    allv = np.concatenate(drawn)
    assert len(set(allv.tolist())) == len(allv)                              # no sample drawn by two ranks


def _small_vae(dev, seed=0):
    from dp_timeautodiff_amd import models
    torch.manual_seed(seed)
    return models.DeapStackDP(64, 16, 8, 0, 0, 5, [], 5, 24, 1, 16, 8, 3).to(dev).train()


def test_masked_padding_rows_leave_the_step_unchanged():
    """A [capacity] batch whose rows >= n are padding gives the same DP gradient as the exact n-row batch."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    dev = torch.device("cuda:0")
    cap, n = 40, 23
    x = torch.rand(cap, 8, 5, device=dev)
    x[n:] = x[0]  # what the sampler's padding looks like
    eps = torch.randn(cap, 8, 3, device=dev)
    grads = []
    for masked in (False, True):
        m = _small_vae(dev)
        gsm = GradSampleModule(m)
        opt = DPOptimizer(torch.optim.SGD(m.parameters(), lr=0.0), noise_multiplier=1.1, max_grad_norm=0.05,
                          expected_batch_size=25, noise_seed=5)
        opt.attach_grad_sample_module(gsm)
        rows = cap if masked else n
        if masked:
            opt.set_valid_rows(torch.tensor([n], dtype=torch.int32, device=dev))
        opt.zero_grad()
        m.injected_eps = eps[:rows]
        models.vae_loss(gsm, x[:rows]).backward()
        opt.step()
        if masked:
            assert float(opt.per_sample_clip_factor[n:].abs().max()) == 0.0
            assert float(opt.per_sample_clip_factor[:n].min()) > 0.0
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy())
    assert rel_err(grads[1], grads[0]) < 1e-5


def test_graph_replay_draws_a_fresh_poisson_batch_every_step():
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.graphed import GraphedDPStep
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    from oracle import opacus_ref as O
    dev = torch.device("cuda:0")
    N, q = 3000, 0.01
    data = torch.rand(N, 8, 5)
    dl = DevicePoissonLoader([data], sample_rate=q, device=dev, seed=33)
    m = _small_vae(dev)
    gsm = GradSampleModule(m)
    opt = DPOptimizer(torch.optim.Adam(m.parameters(), lr=1e-3), noise_multiplier=1.0, max_grad_norm=0.1,
                      expected_batch_size=int(N * q), noise_seed=5)
    opt.attach_grad_sample_module(gsm)
    eps = torch.randn(dl.capacity, 8, 3, device=dev)

    def loss_fn(batch):
        m.injected_eps = eps
        return models.vae_loss(gsm, batch[0])

    steps = []
    opt.attach_step_hook(lambda o: steps.append(1))
    g = GraphedDPStep(gsm, opt, loss_fn, warmup=2, loader=dl).capture()
    blocks = (N + 3) // 4
    w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    for k in range(4):
        loss = g()
        draw = 2 + 1 + k  # 2 warm-up draws, 1 draw recorded at capture time (not executed), then the replays
        want = O.poisson_select(N, q, O.stream_key(33, "sampler"), (2 + k) * blocks)
        n_kept = int(dl.counts[0].item())
        assert n_kept == len(want), (k, draw)
        assert np.array_equal(dl.idx[:n_kept].cpu().numpy(), want)
        assert torch.isfinite(loss)
    assert len(steps) == 2 + 4 and dl.overflow_steps() == 0
    w1 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert not torch.equal(w0, w1)


def test_privacy_engine_device_sampler_and_batch_memory_manager():
    """The reference's call sequence (timeautoencoder.py:664-751) with device_sampler=True: same protocol, batches
    arrive as device tensors, physical chunks <= max_physical_batch_size, one real step per logical batch."""
    from opacus import PrivacyEngine
    from opacus.utils.batch_memory_manager import BatchMemoryManager
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    dev = torch.device("cuda:0")
    data = torch.rand(600, 8, 5)
    m = _small_vae(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=60, shuffle=True)
    pe = PrivacyEngine(accountant="rdp")
    gsm, dpo, loader = pe.make_private(module=m, optimizer=opt, data_loader=dl, noise_multiplier=1.0,
                                       max_grad_norm=0.1, device_sampler=True)
    assert isinstance(loader, DevicePoissonLoader) and len(loader) == 10 and dpo.expected_batch_size == 60
    sizes, real_steps = [], []
    dpo.attach_step_hook(lambda o: real_steps.append(1))
    with BatchMemoryManager(data_loader=loader, max_physical_batch_size=32, optimizer=dpo) as safe:
        for (xb,) in safe:
            assert xb.is_cuda and xb.shape[0] <= 32
            sizes.append(xb.shape[0])
            dpo.zero_grad()
            m.injected_eps = torch.randn(xb.shape[0], 8, 3, device=dev)
            models.vae_loss(gsm, xb).backward()
            dpo.step()
    assert len(real_steps) == 10 and sum(sizes) > 300 and max(sizes) == 32


def test_rank_shards_partition_the_dataset_and_draw_independent_streams():
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    labels = torch.arange(1001)
    loaders = [DevicePoissonLoader([labels], sample_rate=0.05, device="cuda", seed=4, rank=r, world_size=3)
               for r in range(3)]
    owned = [dl.tensors[0].cpu() for dl in loaders]
    assert sorted(torch.cat(owned).tolist()) == list(range(1001))  # a partition: no sample on two ranks
    assert [len(o) for o in owned] == [334, 334, 333]
    draws = [next(iter(dl))[0].cpu() for dl in loaders]
    for r, d in enumerate(draws):
        assert set(d.tolist()) <= set(owned[r].tolist())
    assert len({dl.key for dl in loaders}) == 3
    # pooled over ranks the draw is a Poisson sample of the whole dataset at the same rate
    n = sum(len(d) for d in draws)
    assert abs(n - 1001 * 0.05) < 6 * (1001 * 0.05 * 0.95) ** 0.5
