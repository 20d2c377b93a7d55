"""Attention-style layers (north_star): `nn.MultiheadAttention` -> DPMultiheadAttention, whose
four projections are nn.Linear (K1) and whose add_bias_kv tokens are SequenceBias (K1 with a unit activation).

Checker: the untouched `nn.MultiheadAttention` in float64, one sample at a time (B=1 autograd loop): sample b's row of
`p.grad_sample` must be d loss_b / d p.  The reference repo has no live attention layer (Model_Code/
timeautoencoder.py:179-212 is commented out), so there is no reference fixture for this: torch's own module is the
statement of the semantics Opacus' DPMultiheadAttention follows.  Tolerance 1e-5 relative (fp32 path).

Runs twice like test_engine.py: "hip" (gpu: the HIP kernels through the C-ABI) and "oracle" (CPU: same host code).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import rel_err
from test_engine import dev  # noqa: F401  (fixture: hip | oracle)

TOL = 1e-5

CASES = {
    # name: (mha kwargs, Tq, Tk, use key_padding_mask, use attn_mask)
    "packed": (dict(embed_dim=32, num_heads=4, batch_first=True), 6, 6, False, False),
    "packed_seq_first_masks": (dict(embed_dim=24, num_heads=3, batch_first=False), 5, 7, True, True),
    "separate_qkv": (dict(embed_dim=32, num_heads=2, kdim=20, vdim=12, batch_first=True), 4, 9, True, False),
    "no_bias": (dict(embed_dim=16, num_heads=4, bias=False, batch_first=True), 3, 3, False, True),
    "bias_kv": (dict(embed_dim=32, num_heads=4, add_bias_kv=True, batch_first=True), 5, 6, True, True),
    "bias_kv_zero_attn_separate": (dict(embed_dim=16, num_heads=2, add_bias_kv=True, add_zero_attn=True, kdim=10,
                                        vdim=6, batch_first=False), 4, 5, False, False),
    "wide": (dict(embed_dim=256, num_heads=8, batch_first=True), 24, 24, False, False),
}


class Net(nn.Module):
    """attention + a Linear head; query/key/value come in as data (first layers of the model)."""

    def __init__(self, **kw):
        super().__init__()
        self.mha = nn.MultiheadAttention(**kw)
        self.head = nn.Linear(kw["embed_dim"], 5)

    def forward(self, q, k, v, kpm=None, am=None):
        out, _ = self.mha(q, k, v, key_padding_mask=kpm, attn_mask=am, need_weights=False)
        return self.head(out)


def _inputs(kw, B, Tq, Tk, kpm, am, seed):
    g = torch.Generator().manual_seed(seed)
    E, kd, vd = kw["embed_dim"], kw.get("kdim") or kw["embed_dim"], kw.get("vdim") or kw["embed_dim"]
    q, k, v = (torch.randn(B, T, F, generator=g, dtype=torch.float64) for T, F in ((Tq, E), (Tk, kd), (Tk, vd)))
    w = torch.randn(B, Tq, 5, generator=g, dtype=torch.float64)
    pad = None
    if kpm:
        pad = torch.zeros(B, Tk, dtype=torch.bool)
        for b in range(B):
            pad[b, Tk - (b % 3):] = b % 3 > 0  # 0, 1 or 2 padded keys at the end
    mask = torch.randn(Tq, Tk, generator=g, dtype=torch.float64) if am else None
    return q, k, v, w, pad, mask


def _reference_grad_samples(net64, kw, q, k, v, w, pad, mask):
    """float64, one sample at a time, on the original nn.MultiheadAttention."""
    bf = kw["batch_first"]
    B = q.shape[0]
    out = {n: [] for n, _ in net64.named_parameters()}
    for b in range(B):
        net64.zero_grad(set_to_none=True)
        args = [t[b:b + 1] if bf else t[b:b + 1].transpose(0, 1) for t in (q, k, v)]
        y = net64(*args, None if pad is None else pad[b:b + 1], mask)
        y = y if bf else y.transpose(0, 1)
        (y * w[b:b + 1]).sum().backward()
        for n, p in net64.named_parameters():
            out[n].append(p.grad.detach().clone())
    return {n: torch.stack(v) for n, v in out.items()}


def _split(ref, E, packed):
    """nn.MultiheadAttention's gradients under DPMultiheadAttention's parameter names."""
    o = {}
    if packed:
        wq, wk, wv = ref["mha.in_proj_weight"].split(E, 1)
    else:
        wq, wk, wv = ref["mha.q_proj_weight"], ref["mha.k_proj_weight"], ref["mha.v_proj_weight"]
    o["mha.qlinear.weight"], o["mha.klinear.weight"], o["mha.vlinear.weight"] = wq, wk, wv
    if "mha.in_proj_bias" in ref:
        o["mha.qlinear.bias"], o["mha.klinear.bias"], o["mha.vlinear.bias"] = ref["mha.in_proj_bias"].split(E, 1)
        o["mha.out_proj.bias"] = ref["mha.out_proj.bias"]
    o["mha.out_proj.weight"] = ref["mha.out_proj.weight"]
    if "mha.bias_k" in ref:
        B = wq.shape[0]
        o["mha.seq_bias_k.bias"] = ref["mha.bias_k"].reshape(B, E)
        o["mha.seq_bias_v.bias"] = ref["mha.bias_v"].reshape(B, E)
    o["head.weight"], o["head.bias"] = ref["head.weight"], ref["head.bias"]
    return o


@pytest.mark.parametrize("case", list(CASES))
def test_attention_per_sample_gradients_match_a_float64_loop_over_nn_mha(dev, case):  # noqa: F811
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.layers import DPMultiheadAttention
    from dp_timeautodiff_amd.optimizer import DPOptimizer
    from dp_timeautodiff_amd.validators import ModuleValidator
    kw, Tq, Tk, kpm, am = CASES[case]
    B = 7
    torch.manual_seed(list(CASES).index(case))
    net = Net(**kw)
    with torch.no_grad():  # default in_proj_bias / bias_k inits are 0 / tiny: move them off zero
        for p in net.parameters():
            if p.dim() < 2 or p.shape[0] == 1:
                p.normal_(0, 0.3)
    assert not ModuleValidator.is_valid(net)
    fixed = ModuleValidator.fix(net)
    assert isinstance(fixed.mha, DPMultiheadAttention) and ModuleValidator.is_valid(fixed)
    q, k, v, w, pad, mask = _inputs(kw, B, Tq, Tk, kpm, am, seed=len(case))
    ref = _split(_reference_grad_samples(net.double(), kw, q, k, v, w, pad, mask), kw["embed_dim"],
                 packed=kw.get("kdim") is None)

    fixed = fixed.to(dev).train()
    assert {n for n, _ in fixed.named_parameters()} == set(ref)
    gsm = GradSampleModule(fixed, batch_first=kw["batch_first"])  # loss_reduction="mean": rows are B * d(mean)/dp
    f32 = lambda t: None if t is None else t.float().to(dev)
    args = [f32(t) if kw["batch_first"] else f32(t).transpose(0, 1).contiguous() for t in (q, k, v)]
    y = gsm(*args, None if pad is None else pad.to(dev), f32(mask))
    y = y if kw["batch_first"] else y.transpose(0, 1)
    ((y * f32(w)).sum() / B).backward()
    gsm.check_complete()
    for n, p in fixed.named_parameters():
        got = p.grad_sample.detach().cpu().numpy()
        assert got.shape == tuple(ref[n].shape), n
        if float(ref[n].abs().max()) < 1e-12:
            # klinear.bias without add_bias_kv: a shift common to all keys cancels in the softmax, the gradient is 0
            assert n == "mha.klinear.bias" and np.abs(got).max() < 1e-6
            continue
        assert rel_err(got, ref[n].numpy()) < TOL, n
        # row by row: no sample may hide behind a larger one
        for b in range(B):
            assert rel_err(got[b], ref[n][b].numpy()) < 5 * TOL, (n, b)

    # and through the optimizer tail: norms and the clipped sum against float64
    C = 0.5
    opt = DPOptimizer(torch.optim.SGD(fixed.parameters(), lr=0.0), noise_multiplier=0.0, max_grad_norm=C,
                      expected_batch_size=B)
    opt.attach_grad_sample_module(gsm)
    opt.step()
    flat = torch.cat([ref[n].reshape(B, -1) for n, _ in fixed.named_parameters()], 1)
    norms = flat.norm(dim=1)
    clip = (C / (norms + 1e-6)).clamp(max=1.0)
    assert rel_err(opt.per_sample_norms.cpu().numpy(), norms.numpy()) < TOL
    got = torch.cat([p.grad.reshape(-1).cpu() for p in fixed.parameters()]).numpy()
    assert rel_err(got, ((flat * clip[:, None]).sum(0) / B).numpy()) < TOL


def test_sequence_bias_called_twice_and_sequence_first(dev):  # noqa: F811
    """SequenceBias on its own: appended token, [T,B,E] layout, two calls per forward (gradients add up)."""
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.layers import SequenceBias

    class Two(nn.Module):
        def __init__(self):
            super().__init__()
            self.sb = SequenceBias(12)
            self.fc = nn.Linear(12, 3)

        def forward(self, x):
            return self.fc(self.sb(self.sb(x)))

    torch.manual_seed(3)
    m = Two()
    x = torch.randn(4, 6, 12, dtype=torch.float64)  # [T, B, E]
    w = torch.randn(6, 6, 3, dtype=torch.float64)
    ref = []
    m64 = Two().double()
    m64.load_state_dict(m.state_dict())
    for b in range(6):
        m64.zero_grad(set_to_none=True)
        (m64(x[:, b:b + 1]) * w[:, b:b + 1]).sum().backward()
        ref.append(m64.sb.bias.grad.clone())
    m = m.to(dev).train()
    gsm = GradSampleModule(m, batch_first=False, loss_reduction="sum")
    (gsm(x.float().to(dev)) * w.float().to(dev)).sum().backward()
    gsm.check_complete()
    assert rel_err(m.sb.bias.grad_sample.cpu().numpy(), torch.stack(ref).numpy()) < TOL
