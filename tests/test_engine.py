"""End-to-end parity of the Opacus-shaped engine against golden vectors from the reference model classes.

Every test runs twice:
  * backend "hip"   (marked gpu): the product path — HIP kernels through the C-ABI on cuda:0;
  * backend "oracle" (CPU): the same host code with `_native` patched by tests/cpu_backend.py, which
    checks the hook / stacking / arena / optimizer logic in the GPU-less container.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err, row_err

TOL = 1e-5


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), "oracle"])
def dev(request, monkeypatch):
    if request.param == "hip":
        assert torch.cuda.is_available()
        from dp_timeautodiff_amd import _native
        _native.load()
        return torch.device("cuda:0")
    import cpu_backend
    cpu_backend.install(monkeypatch)
    return torch.device("cpu")


def _build_vae(g, dev):
    from dp_timeautodiff_amd import models
    m = models.DeapStackDP(**g.meta["ctor"])
    if g.has("w/" + g.names[0]):
        m.load_state_dict({n: torch.from_numpy(g["w/" + n]) for n in g.names})
    else:
        _fill(m, g.meta["seed_w"])
    return m.to(dev).train()


def _fill(model, seed):
    # same deterministic fill as tests/golden/make_golden.py::fill_params
    rng = np.random.Generator(np.random.PCG64(seed))
    with torch.no_grad():
        for name, p in model.named_parameters():
            fan_in = p.shape[-1] if p.dim() > 1 else max(p.shape[0], 1)
            v = rng.standard_normal(tuple(p.shape)).astype(np.float32) / np.sqrt(np.float32(fan_in))
            p.copy_(torch.from_numpy(v))


def _arena_noise(opt, names, params, flat_noise, dev):
    """Lay a reference-ordered flat noise vector out like an arena row (windows may be row-padded)."""
    arena = opt.arena
    z = torch.zeros(arena.P)
    off = 0
    for p in params:
        n = p.numel()
        arena.vec_view(z, p).copy_(torch.from_numpy(flat_noise[off:off + n]).view(p.shape))
        off += n
    return z.to(dev)


def _flat(params, attr):
    return np.concatenate([getattr(p, attr).detach().cpu().reshape(-1).numpy() for p in params])


def _run_vae(g, dev, ci, chunks=None):
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    m = _build_vae(g, dev)
    params = [p for _, p in m.named_parameters()]
    gsm = GradSampleModule(m)
    C = g.meta["C_list"][ci]
    opt = DPOptimizer(torch.optim.SGD(m.parameters(), lr=0.0), noise_multiplier=g.meta["sigma"], max_grad_norm=C,
                      expected_batch_size=g.meta["expected_batch_size"])
    opt.attach_grad_sample_module(gsm)
    x = torch.from_numpy(g["x"]).to(dev)
    eps = torch.from_numpy(g["eps"]).to(dev)
    B = x.shape[0]
    chunks = chunks or [B]
    norms, clips, gs_seen = [], [], []
    b0 = 0
    for k, n in enumerate(chunks):
        last = k == len(chunks) - 1
        opt.signal_skip_step(do_skip=not last)
        opt.zero_grad()
        m.injected_eps = eps[b0:b0 + n]
        # loss is a mean over the chunk: per-sample grads are chunk-size independent
        loss = models.vae_loss(gsm, x[b0:b0 + n], beta=g.meta["beta"])
        loss.backward()
        if last:
            if g.has(f"tail{ci}/noise"):
                noise = g[f"tail{ci}/noise"]
            else:
                seed = int(g[f"tail{ci}/noise_seed"][0])
                noise = (np.random.Generator(np.random.PCG64(seed)).standard_normal(g.meta["P"])
                         * g.meta["sigma"] * C).astype(np.float32)
            opt.inject_noise(_arena_noise(opt, g.names, params, noise, dev))
        gs_seen.append([p.grad_sample.detach().cpu().numpy().copy() for p in params])
        opt.step()
        norms.append(opt.per_sample_norms.cpu().numpy().copy())
        clips.append(opt.per_sample_clip_factor.cpu().numpy().copy())
        b0 += n
    return dict(model=m, params=params, opt=opt, loss=float(loss), norms=np.concatenate(norms),
                clips=np.concatenate(clips), gs=gs_seen)


@pytest.mark.parametrize("case", ["vae_small_mixed.npz", "vae_small_nums.npz"])
def test_vae_grad_samples_match_reference_autograd(golden, dev, case):
    g = golden(case)
    r = _run_vae(g, dev, 0)
    assert abs(r["loss"] - g.meta["loss"]) <= 2e-6 * max(1.0, abs(g.meta["loss"]))
    for n, got in zip(g.names, r["gs"][0]):
        ref = g["gs/" + n]
        assert got.shape == ref.shape, n
        assert rel_err(got, ref) < TOL, n
        assert row_err(got, ref) < TOL, n  # sample by sample, each on its own scale


@pytest.mark.parametrize("case", ["vae_small_mixed.npz", "vae_small_nums.npz"])
@pytest.mark.parametrize("ci", [0, 1])
def test_vae_dp_step_matches_float64_formulas(golden, dev, case, ci):
    g = golden(case)
    r = _run_vae(g, dev, ci)
    assert rel_err(r["norms"], g[f"tail{ci}/norms_f64"]) < TOL
    assert rel_err(r["clips"], g[f"tail{ci}/clip_f64"]) < TOL
    assert rel_err(_flat(r["params"], "summed_grad"), g[f"tail{ci}/summed_f64"]) < TOL
    assert rel_err(_flat(r["params"], "grad"), g[f"tail{ci}/grad_f64"]) < TOL


def test_physical_chunks_equal_one_logical_batch(golden, dev):
    """BatchMemoryManager protocol: skip-step accumulation over chunks == one step over the whole batch."""
    g = golden("vae_small_nums.npz")
    r = _run_vae(g, dev, 1, chunks=[5, 4, 3])
    assert rel_err(r["norms"], g["tail1/norms_f64"]) < TOL
    assert rel_err(_flat(r["params"], "summed_grad"), g["tail1/summed_f64"]) < TOL
    assert rel_err(_flat(r["params"], "grad"), g["tail1/grad_f64"]) < TOL


def test_airquality_dp_job_shape(golden, dev):
    """BASELINE.json configs[0]: real AirQuality windows, B=64, T=24, H=200, P=649400, expected_bs=49."""
    g = golden("vae_airquality_b64.npz")
    if dev.type == "cpu":
        torch.set_num_threads(8)
    for ci in (0, 1):
        r = _run_vae(g, dev, ci)
        assert abs(r["loss"] - g.meta["loss"]) <= 2e-6 * max(1.0, abs(g.meta["loss"]))
        assert rel_err(r["norms"], g[f"tail{ci}/norms_f64"]) < TOL
        assert rel_err(r["clips"], g[f"tail{ci}/clip_f64"]) < TOL
        idx = g[f"tail{ci}/idx"]
        assert rel_err(_flat(r["params"], "summed_grad")[idx], g[f"tail{ci}/summed_f64"]) < TOL
        assert rel_err(_flat(r["params"], "grad")[idx], g[f"tail{ci}/grad_f64"]) < TOL
    # per-parameter norms in the op order Opacus uses (float32 golden)
    ppn = np.stack([np.sqrt((x.reshape(64, -1).astype(np.float64) ** 2).sum(1)) for x in r["gs"][0]], 1)
    assert rel_err(ppn, g["per_param_norms_f32"]) < TOL
    assert row_err(ppn, g["per_param_norms_f32"]) < TOL
    assert np.abs(r["norms"] / g["tail0/norms_f64"] - 1).max() < TOL  # every sample's norm, relative to itself


def test_denoiser_dplstm_matches_reference_lstm_autograd(golden, dev):
    """DPLSTM (Linear cells, K=T stacked) reproduces the per-sample grads of the reference's nn.LSTM."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    g = golden("denoiser_small.npz")
    m = models.BiRNNScoreDP(**g.meta["ctor"])
    m.load_state_dict({n: torch.from_numpy(g["w/" + n]) for n in g.names})
    m = m.to(dev).train()
    ref_names = models.reference_param_names(m)
    params = dict(zip(ref_names, [p for _, p in m.named_parameters()]))
    assert set(ref_names) == set(g.names)
    gsm = GradSampleModule(m)
    ci = 1
    C = g.meta["C_list"][ci]
    opt = DPOptimizer(torch.optim.SGD(m.parameters(), lr=0.0), noise_multiplier=g.meta["sigma"], max_grad_norm=C,
                      expected_batch_size=g.meta["expected_batch_size"])
    opt.attach_grad_sample_module(gsm)
    opt.zero_grad()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    loss = models.denoiser_loss(gsm, t("x_noisy"), t("t"), t("i"), t("time_info"), t("noise_target"))
    loss.backward()
    assert abs(float(loss) - g.meta["loss"]) <= 2e-6 * max(1.0, abs(g.meta["loss"]))
    for n in g.names:
        assert rel_err(params[n].grad_sample.cpu().numpy(), g["gs/" + n]) < TOL, n
        assert row_err(params[n].grad_sample.cpu().numpy(), g["gs/" + n]) < TOL, n
    ordered = [params[n] for n in g.names]
    opt.inject_noise(_arena_noise(opt, g.names, ordered, g[f"tail{ci}/noise"], dev))
    opt.step()
    assert rel_err(opt.per_sample_norms.cpu().numpy(), g[f"tail{ci}/norms_f64"]) < TOL
    assert rel_err(_flat(ordered, "grad"), g[f"tail{ci}/grad_f64"]) < TOL


def test_reference_call_sequence(dev):
    """The literal call sequence of train_dp_autoencoder (timeautoencoder.py:653-759) on synthetic data."""
    from opacus import PrivacyEngine
    from opacus.utils.batch_memory_manager import BatchMemoryManager
    from opacus.validators import ModuleValidator
    from dp_timeautodiff_amd import models
    torch.manual_seed(0)
    N, T, K = 96, 6, 5
    data = torch.rand(N, T, K)
    ae = models.DeapStackDP(64, 16, T, 0, 0, K, [], K, 12, 1, 16, 8, 3).to(dev)
    ae = ModuleValidator.fix(ae)
    assert ModuleValidator.is_valid(ae) and ModuleValidator.validate(ae, strict=False) == []
    opt = torch.optim.Adam(ae.parameters(), lr=2e-4, weight_decay=1e-6)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data), batch_size=16, shuffle=True,
                                         drop_last=False)
    pe = PrivacyEngine(accountant="rdp")
    ae, opt, loader = pe.make_private_with_epsilon(module=ae, optimizer=opt, data_loader=loader, epochs=3,
                                                   target_epsilon=5.0, target_delta=1e-5, max_grad_norm=1.0)
    assert opt.expected_batch_size == 16 and opt.noise_multiplier > 0
    w0 = ae.fc_mu.weight.detach().clone()
    eps_hist, steps = [], 0
    for epoch in range(3):
        ae.train()
        with BatchMemoryManager(data_loader=loader, max_physical_batch_size=8, optimizer=opt) as safe:
            for batch in safe:
                opt.zero_grad()
                x = batch[0].to(dev)
                loss = models.vae_loss(ae, x, beta=0.1)
                loss.backward()
                opt.step()
                steps += 1
        eps_hist.append(pe.get_epsilon(1e-5))
    assert len(pe.accountant.history) == 1 and pe.accountant.history[0][2] == 3 * 6  # 6 logical steps/epoch
    assert steps >= 18
    assert eps_hist[0] < eps_hist[1] < eps_hist[2] <= 5.0 + 0.01
    assert not torch.equal(w0, ae.fc_mu.weight.detach())
    assert all(not k.startswith("_module.") for k in ae.state_dict().keys())
    assert "state" in opt.state_dict() and callable(ae.decoder)
    # full-dataset forward through the wrapper in eval mode leaves no stashed activations (reference :779)
    ae.eval()
    with torch.no_grad():
        out, latent, _, _ = ae(data.to(dev))
    assert latent.shape == (N, T, 3)
    assert all(len(l.acts) == 0 for l in ae._layers)


def test_poisson_forbids_grad_accumulation_and_missing_layer_raises(dev):
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2)).to(dev)
    gsm = GradSampleModule(net)
    gsm.forbid_grad_accumulation = True
    x = torch.randn(3, 4, device=dev)
    gsm(x).sum().backward()
    with pytest.raises(ValueError, match="Poisson sampling is not compatible"):
        gsm(x).sum().backward()
    # virtual-batch accumulation (poisson off): two passes are concatenated along the batch dim
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2)).to(dev)
    gsm2 = GradSampleModule(net2, loss_reduction="sum")
    opt = DPOptimizer(torch.optim.SGD(net2.parameters(), lr=0.1), noise_multiplier=0.0, max_grad_norm=1e9,
                      expected_batch_size=None, loss_reduction="sum")
    opt.attach_grad_sample_module(gsm2)
    x1, x2 = torch.randn(3, 4, device=dev), torch.randn(5, 4, device=dev)
    gsm2(x1).sum().backward()
    gsm2(x2).sum().backward()
    assert opt.accumulated_iterations == 2 and opt.grad_samples[0].shape[0] == 8
    gsm2.disable_hooks()
    ref = torch.autograd.grad(net2(torch.cat([x1, x2])).sum(), net2[0].weight)[0]
    gsm2.enable_hooks()
    opt.step()
    assert rel_err(net2[0].weight.grad.cpu().numpy(), ref.cpu().numpy()) < TOL
    with pytest.raises(ValueError, match="already processed"):
        opt.step()


def test_backward_that_raises_leaves_nothing_behind(dev):
    """at B <= 256 the Linear contractions are queued for ONE launch at the end of backward.  A
    backward pass that raises never runs its end-of-pass callback; after zero_grad the next pass must be a clean one
    (no stale job sharing its rows), and must queue its own flush."""
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer

    class Boom(RuntimeError):
        pass

    def build():
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(8, 12), torch.nn.Tanh(), torch.nn.Linear(12, 4)).to(dev)
        gsm = GradSampleModule(net)
        opt = DPOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), noise_multiplier=0.0, max_grad_norm=0.3,
                          expected_batch_size=6)
        opt.attach_grad_sample_module(gsm)
        return net, gsm, opt

    g = torch.Generator().manual_seed(5)
    x_bad = torch.randn(6, 5, 8, generator=g).to(dev)
    x = torch.randn(6, 5, 8, generator=g).to(dev)

    net1, gsm1, opt1 = build()
    opt1.zero_grad()
    gsm1(x).square().mean().backward()
    opt1.step()

    net2, gsm2, opt2 = build()
    opt2.zero_grad()
    h = net2[0](x_bad)

    def explode(_):
        raise Boom()

    h.register_hook(explode)  # fires after the last layer's backward hook has queued its contraction
    with pytest.raises(Boom):
        gsm2._module[2](gsm2._module[1](h)).square().mean().backward()
    opt2.zero_grad()
    assert not gsm2._multi_pending and all(not l.pairs for l in gsm2._layers)
    # the first layer's stashed input never met its backprop; like Opacus' activation stacks it is not zero_grad's to
    # drop (a forward may legally precede zero_grad), and leaving it makes the next step fail loudly in check_complete
    for l in gsm2._layers:
        l.acts.clear()
        l.counter = 0
    for p in net2.parameters():
        p._forward_counter = 0
    gsm2(x).square().mean().backward()
    assert not gsm2._multi_pending  # this pass queued (and ran) its own flush
    opt2.step()
    for p1, p2 in zip(net1.parameters(), net2.parameters()):
        assert torch.equal(p1.grad_sample, p2.grad_sample)
        assert torch.equal(p1.grad, p2.grad)


def test_empty_poisson_batch_gives_pure_noise(dev):
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    net = torch.nn.Sequential(torch.nn.Linear(4, 3)).to(dev)
    gsm = GradSampleModule(net)
    opt = DPOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), noise_multiplier=1.0, max_grad_norm=2.0,
                      expected_batch_size=4, noise_seed=11)
    opt.attach_grad_sample_module(gsm)
    opt.zero_grad()
    gsm(torch.zeros(0, 4, device=dev)).sum().backward()
    opt.step()
    from oracle import opacus_ref as O
    z = O.philox_normal(opt.arena.P, O.stream_key(11, "noise"), 0, 2.0) / 4.0
    w = opt.arena.window(net[0].weight)
    assert np.abs(net[0].weight.grad.cpu().reshape(-1).numpy() - z[w.offset:w.offset + 12]).max() < 1e-5


def test_empty_poisson_batch_flows_through_the_vae(dev):
    """A Poisson draw of zero samples reaches the model as [0, T, K] tensors (the reference's loop does not guard it,
    timeautoencoder.py:731-751): forward, loss, backward and step must run — fused recurrence included on the HIP
    backend — and the step is pure noise."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    torch.manual_seed(0)
    m = models.DeapStackDP(64, 8, 5, 0, 0, 4, [], 4, 16, 1, 8, 8, 3).to(dev).train()
    if dev.type == "cuda":
        from dp_timeautodiff_amd.layers.fused_gru import fuse_recurrences
        m = fuse_recurrences(m)
    gsm = GradSampleModule(m)
    opt = DPOptimizer(torch.optim.SGD(m.parameters(), lr=0.1), noise_multiplier=1.0, max_grad_norm=2.0,
                      expected_batch_size=4, noise_seed=11)
    opt.attach_grad_sample_module(gsm)
    opt.zero_grad()
    m.injected_eps = torch.zeros(0, 5, 3, device=dev)
    models.vae_loss(gsm, torch.zeros(0, 5, 4, device=dev)).backward()
    opt.step()
    from oracle import opacus_ref as O
    z = O.philox_normal(opt.arena.P, O.stream_key(11, "noise"), 0, 2.0) / 4.0
    for p in m.parameters():
        w = opt.arena.window(p)
        got = opt.arena.grad[w.offset:w.offset + w.numel].cpu().numpy()
        assert np.abs(got - z[w.offset:w.offset + w.numel]).max() < 1e-5


def test_optimizer_keys_its_kernel_with_the_derived_noise_key(dev):
    from dp_timeautodiff_amd import rng
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    net = torch.nn.Linear(3, 2).to(dev)
    gsm = GradSampleModule(net)
    opt = DPOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), noise_multiplier=1.0, max_grad_norm=1.0,
                      expected_batch_size=2, noise_seed=5)
    opt.attach_grad_sample_module(gsm)
    assert opt.noise_key == rng.stream_key(5, "noise") != 5
    _, key, off, std = opt._noise_args(opt.arena)
    assert key == opt.noise_key and off == 0 and std == 1.0
    # checkpoint round trip of the stream position, before and after a graph owns the offset
    st = opt.noise_state()
    assert st == {"seed": 5, "offset": opt.arena.P // 4}
    opt.set_noise_state(5, 1000)
    assert opt.noise_state()["offset"] == 1000
    if dev.type == "cuda":
        opt.enable_device_noise_offset()
        assert opt.noise_state()["offset"] == 1000
        opt.set_noise_state(5, 64)  # must land in the device word, not be added to it
        assert opt.noise_state()["offset"] == 64 and opt._noise_offset == 0


@pytest.mark.gpu
def test_graph_captured_step_matches_eager_steps():
    """A step replayed from a hipGraph (fresh Philox noise per replay via the device-side offset) trains exactly
    like eager steps with the same seed."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.accountants import RDPAccountant
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.graphed import GraphedDPStep
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    dev = torch.device("cuda:0")
    B, T, K = 32, 6, 5
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, T, K, generator=g).to(dev)
    eps = torch.randn(B, T, 3, generator=g).to(dev)

    def build():
        torch.manual_seed(1)
        m = models.DeapStackDP(64, B, T, 0, 0, K, [], K, 12, 1, 16, 8, 3).to(dev).train()
        m.injected_eps = eps
        gsm = GradSampleModule(m)
        opt = DPOptimizer(torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-6), noise_multiplier=0.7,
                          max_grad_norm=0.05, expected_batch_size=B, noise_seed=99)
        opt.attach_grad_sample_module(gsm)
        acc = RDPAccountant()
        opt.attach_step_hook(acc.get_optimizer_hook_fn(sample_rate=0.1))
        return m, gsm, opt, acc

    n_steps = 7
    m1, gsm1, opt1, acc1 = build()
    for _ in range(n_steps):
        opt1.zero_grad()
        models.vae_loss(gsm1, x).backward()
        opt1.step()
    m2, gsm2, opt2, acc2 = build()
    step = GraphedDPStep(gsm2, opt2, lambda: models.vae_loss(gsm2, x), warmup=2).capture()
    losses = [float(step()) for _ in range(n_steps - 2)]
    torch.cuda.synchronize()
    assert acc1.history == acc2.history == [(0.7, 0.1, n_steps)]
    assert int(opt2._noise_offset_dev.item()) == n_steps * (gsm2.arena.P // 4)
    assert losses[0] != losses[-1]
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert rel_err(p2.detach().cpu().numpy(), p1.detach().cpu().numpy()) < 1e-5, n


@pytest.mark.gpu
def test_learning_rate_change_between_graph_replays_takes_effect():
    """The captured Adam launch reads lr / weight decay / betas from a device table refreshed before every replay:
    a scheduler (or the user) changing `param_groups[0]["lr"]` between replays changes the update, exactly as it
    does for eager steps — and lr = 0 freezes the parameters while the noise stream still advances."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.graphed import GraphedDPStep
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    dev = torch.device("cuda:0")
    B, T, K = 24, 5, 4
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, T, K, generator=g).to(dev)
    eps = torch.randn(B, T, 3, generator=g).to(dev)
    lrs = [1e-2, 1e-2, 1e-2, 3e-3, 0.0, 2e-2]  # the first two are GraphedDPStep's eager warm-up steps

    def build():
        torch.manual_seed(2)
        m = models.DeapStackDP(64, B, T, 0, 0, K, [], K, 12, 1, 16, 8, 3).to(dev).train()
        m.injected_eps = eps
        gsm = GradSampleModule(m)
        opt = DPOptimizer(torch.optim.Adam(m.parameters(), lr=lrs[0]), noise_multiplier=0.5, max_grad_norm=0.05,
                          expected_batch_size=B, noise_seed=3)
        opt.attach_grad_sample_module(gsm)
        return m, gsm, opt

    def snap(m):
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()

    m1, gsm1, opt1 = build()
    eager = []
    for lr in lrs:
        opt1.param_groups[0]["lr"] = lr
        opt1.zero_grad()
        models.vae_loss(gsm1, x).backward()
        opt1.step()
        eager.append(snap(m1))
    m2, gsm2, opt2 = build()
    step = GraphedDPStep(gsm2, opt2, lambda: models.vae_loss(gsm2, x), warmup=2).capture()
    replayed = [None, snap(m2)]
    for lr in lrs[2:]:
        opt2.param_groups[0]["lr"] = lr
        step()
        replayed.append(snap(m2))
    torch.cuda.synchronize()
    for k in range(1, len(lrs)):
        assert rel_err(replayed[k].cpu().numpy(), eager[k].cpu().numpy()) < 1e-5, k
    assert torch.equal(replayed[4], replayed[3])          # lr = 0: nothing moved
    assert not torch.equal(replayed[5], replayed[4])      # and moves again at the next lr
    d3 = (replayed[3] - replayed[2]).abs().max()
    d2 = (replayed[2] - replayed[1]).abs().max()
    assert float(d3) < 0.6 * float(d2)                    # Adam's step size follows lr (3e-3 after 1e-2)


@pytest.mark.gpu
def test_seeds_cannot_change_under_a_captured_graph():
    """the Philox keys (DP noise, device sampler) are by-value arguments of the captured launches.  Restoring
    a DIFFERENT seed after capture would be reported by noise_state() and ignored by every replay: it raises; restoring
    the same seed with another offset (what a checkpoint of the same job does) lands in the device word."""
    from dp_timeautodiff_amd import models
    from dp_timeautodiff_amd.data_loader import DevicePoissonLoader
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.graphed import GraphedDPStep
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    dev = torch.device("cuda:0")
    T, K = 6, 5
    torch.manual_seed(1)
    m = models.DeapStackDP(64, 16, T, 0, 0, K, [], K, 12, 1, 16, 8, 3).to(dev).train()
    data = torch.rand(400, T, K)
    loader = DevicePoissonLoader([data], sample_rate=0.05, device=dev, seed=5)
    gsm = GradSampleModule(m)
    opt = DPOptimizer(torch.optim.Adam(m.parameters(), lr=1e-3), noise_multiplier=0.7, max_grad_norm=0.05,
                      expected_batch_size=20, noise_seed=99)
    opt.attach_grad_sample_module(gsm)
    opt.set_noise_state(98, 0)   # before capture: anything goes
    opt.set_noise_state(99, 0)
    loader.seed = 6
    loader.seed = 5

    def loss_fn(batch):
        inner = gsm._module
        inner.injected_eps = torch.zeros(batch[0].shape[0], T, 3, device=dev)
        return models.vae_loss(gsm, batch[0])

    step = GraphedDPStep(gsm, opt, loss_fn, warmup=2, loader=loader).capture()
    step()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="captured"):
        opt.set_noise_state(7, 0)
    with pytest.raises(RuntimeError, match="captured"):
        loader.seed = 8
    opt.set_noise_state(99, 4096)
    loader.seed = 5
    assert opt.noise_state() == {"seed": 99, "offset": 4096}
    step()
    torch.cuda.synchronize()
    assert opt.noise_state()["offset"] == 4096 + gsm.arena.P // 4
