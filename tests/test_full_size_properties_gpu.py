"""Full-size parity through size-independent properties (BASELINE.json configs[1] and configs[2] at their real
batch sizes, where the CPU oracle would take minutes):

  * sum:        with clipping off (C huge) and no noise, p.grad = (1/B) sum_b grad_sample_b must equal the gradient of
                the mean loss from plain autograd with the hooks disabled (an independent code path: hipBLASLt GEMMs);
  * norms:      per-sample norms from the kernels' fused partials == K5's norm of the materialised arena rows;
  * clip+sum:   K7's clipped sum == clip @ arena computed by torch on the same arena; every clipped row has norm <= C;
  * noise:      grad(z) - grad(0) == z / B  (linearity of the tail in the injected noise).
Tolerance 2e-5 relative: fp32 sums over up to 4096 samples in a different order."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_size_properties(cfg):
    from dp_timeautodiff_amd import _native, workloads
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    dev = torch.device("cuda:0")
    wl = workloads.build(cfg, dev)
    B = wl.B
    model = wl.model
    gsm = GradSampleModule(model)
    opt = DPOptimizer(torch.optim.SGD(model.parameters(), lr=0.0), noise_multiplier=0.0, max_grad_norm=1e9,
                      expected_batch_size=B)
    opt.attach_grad_sample_module(gsm)
    params = list(model.parameters())
    arena = gsm.arena

    # reference: plain autograd on the same module, hooks off
    gsm.disable_hooks()
    model.zero_grad(set_to_none=True)
    wl.loss_fn(gsm).backward()
    plain = [p.grad.detach().clone() for p in params]
    gsm.enable_hooks()

    # ---- sum property ---------------------------------------------------------------------------------------------
    opt.zero_grad()
    wl.loss_fn(gsm).backward()
    opt.step()
    assert float(opt.per_sample_clip_factor.min()) == 1.0
    for p, g in zip(params, plain):
        assert rel_err(p.grad.cpu().numpy(), g.cpu().numpy()) < TOL
    grad0 = arena.grad.clone()

    # ---- norms: fused partials vs a separate pass over the materialised rows -----------------------------------------
    norms = opt.per_sample_norms.clone()
    part = torch.zeros(B, _native.normsq_slots(arena.P), device=dev)
    _native.normsq_partials(_native.FlatView(arena.buf, 0, arena.P), B, part, 0)
    ref_norms = part.double().sum(1).sqrt()
    assert rel_err(norms.cpu().numpy(), ref_norms.cpu().numpy()) < TOL

    # ---- clip + sum at a clipping threshold that bites ---------------------------------------------------------------
    C = float(norms.median())
    opt.max_grad_norm = C
    opt.zero_grad()
    wl.loss_fn(gsm).backward()
    opt.step()
    clip = opt.per_sample_clip_factor
    assert 0.3 < float((clip < 1).float().mean()) < 0.7
    assert float((clip * norms).max()) <= C * (1 + 1e-5)
    ref_sum = torch.zeros(arena.P, device=dev, dtype=torch.float64)
    for lo in range(0, B, 256):  # float64 accumulation over row blocks of the same arena
        ref_sum += clip[lo:lo + 256].double() @ arena.buf[lo:lo + 256].double()
    assert rel_err(arena.summed.cpu().numpy(), ref_sum.cpu().numpy()) < TOL
    grad_c = arena.grad.clone()

    # ---- linearity in the noise -------------------------------------------------------------------------------------
    z = torch.randn(arena.P, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * C
    opt.zero_grad()
    wl.loss_fn(gsm).backward()
    opt.inject_noise(z)
    opt.step()
    assert rel_err((arena.grad - grad_c).cpu().numpy(), (z / B).cpu().numpy()) < TOL
    assert not torch.equal(grad0, grad_c)
