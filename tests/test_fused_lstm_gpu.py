"""K10 — fused DPLSTM recurrence (denoiser) against torch's nn.LSTM.

Checker: nn.LSTM in float64 on the CPU (the module the reference's BiRNN_score holds, Model_Code/timediffusion.py:106,
and which ModuleValidator.fix turns into DPLSTM with the same state_dict keys).  End to end the per-sample gradients
of the fused path are compared with the hooked T-call path, which tests/test_engine.py pins to the reference's autograd
(tests/golden/denoiser_small.npz).  Tolerance 1e-5 relative (fp32)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("B,T,I,H,L,bi,with_state,batch_first", [
    (5, 3, 4, 8, 1, False, False, True), (19, 7, 6, 10, 2, True, True, True), (16, 5, 9, 33, 1, True, False, False),
    (21, 6, 16, 250, 2, True, False, True), (40, 12, 8, 128, 1, False, True, True),
])
def test_fused_dplstm_matches_nn_lstm(B, T, I, H, L, bi, with_state, batch_first):
    from dp_timeautodiff_amd.layers import DPLSTM
    torch.manual_seed(B + H)
    ref = torch.nn.LSTM(I, H, L, batch_first=batch_first, bidirectional=bi).double()
    dp = DPLSTM(I, H, L, batch_first=batch_first, bidirectional=bi)
    dp.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    dp = dp.cuda()
    D = 2 if bi else 1
    shape = (B, T, I) if batch_first else (T, B, I)
    x = torch.randn(*shape, dtype=torch.float64)
    st = (torch.randn(L * D, B, H, dtype=torch.float64) * 0.5, torch.randn(L * D, B, H, dtype=torch.float64) * 0.5) \
        if with_state else None
    oshape = (B, T, D * H) if batch_first else (T, B, D * H)
    w = torch.randn(*oshape, dtype=torch.float64)
    wh = torch.randn(L * D, B, H, dtype=torch.float64)
    wc = torch.randn(L * D, B, H, dtype=torch.float64)  # the loss also reads c_n (its gradient was dropped)

    xr = x.clone().requires_grad_(True)
    str_ = tuple(s.clone().requires_grad_(True) for s in st) if st else None
    o_r, (h_r, c_r) = ref(xr, str_)
    ((o_r * w).sum() + (h_r * wh).sum() + (c_r * wc).sum()).backward()

    xg = x.float().cuda().requires_grad_(True)
    stg = tuple(s.float().cuda().requires_grad_(True) for s in st) if st else None
    o_g, (h_g, c_g) = dp(xg, stg)
    ((o_g * w.float().cuda()).sum() + (h_g * wh.float().cuda()).sum() + (c_g * wc.float().cuda()).sum()).backward()
    torch.cuda.synchronize()

    for got, want in ((o_g, o_r), (h_g, h_r), (c_g, c_r), (xg.grad, xr.grad)):
        assert got.shape == want.shape
        assert rel_err(got.detach().cpu().numpy(), want.detach().numpy()) < TOL
    if st:
        for got, want in zip(stg, str_):
            assert rel_err(got.grad.cpu().numpy(), want.grad.numpy()) < TOL
    ref_grads = dict(ref.named_parameters())
    for n, v in dp.state_dict().items():
        pass  # names only
    for name, (mod, attr) in dp._rename.items():
        assert rel_err(getattr(mod, attr).grad.cpu().numpy(), ref_grads[name].grad.numpy()) < TOL, name


def test_fused_and_hooked_dplstm_agree_on_cfg3_shapes():
    """B=24 slice of BASELINE configs[2] (denoiser, T=48, H=250, 2 layers, bidirectional): per-sample gradients of
    every parameter, fused recurrence vs the hooked T-call protocol."""
    from dp_timeautodiff_amd import workloads
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.layers import DPLSTM
    dev = torch.device("cuda:0")
    res = []
    for fuse in (False, True):
        wl = workloads.build("cfg3", dev, batch=24)
        for m in wl.model.modules():
            if isinstance(m, DPLSTM):
                m.fuse = fuse
        gsm = GradSampleModule(wl.model)
        wl.loss_fn(gsm).backward()
        gsm.check_complete()
        res.append({n: p.grad_sample.detach().cpu().numpy().copy() for n, p in wl.model.named_parameters()})
    for n in res[0]:
        assert rel_err(res[1][n], res[0][n]) < TOL, n
