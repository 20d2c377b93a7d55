"""Synthetic workloads of BASELINE.json `configs`: model + resident inputs + loss closure.

Seeds follow SURVEY.md §8(d): weights torch.manual_seed(0) (PyTorch default init), data seed 1,
reparametrisation-epsilon seed 2, DP-noise seed 3.  Numeric-only feature layout for the VAE configs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict

import torch

from . import models
from .layers.fused_gru import fuse_recurrences

CONFIGS: Dict[str, dict] = {
    # name: kind, per-GPU batch, dims
    "cfg1": dict(kind="vae", B=64, T=24, K=13, H=200, desc="VAE dp_job.sh shape (AirQuality-like), B=64 T=24 K=13 H=200"),
    "cfg1_b50": dict(kind="vae", B=50, T=24, K=13, H=200, desc="VAE dp_job.sh literal batch 50"),
    "cfg2": dict(kind="vae", B=1024, T=24, K=32, H=128, cpu_sample=256,
                 desc="VAE encoder/decoder DP-SGD, B=1024 T=24 K=32 H=128"),
    "cfg3": dict(kind="denoiser", B=4096, T=48, D=7, H=250, L=2, cpu_sample=64, desc="latent diffusion denoiser DP-SGD, B=4096 T=48 H=250"),
    # configs[3]: both models under DP-SGD, global batch 16384 over 8 GPUs = 2048 per GPU through the VAE and the denoiser
    "cfg4": dict(kind="pair", parts=("cfg4_vae", "cfg4_den"), B=2048, cpu_sample=64,
                 desc="full VAE + latent diffusion denoiser DP-SGD, 2048 samples per GPU through both models "
                      "(global batch 16384 on 8 GPUs), one flat all-reduce of each model's clipped sum"),
    "cfg4_vae": dict(kind="vae", B=2048, T=24, K=32, H=128, desc="cfg4's VAE (cfg2 model), 2048 per GPU"),
    "cfg4_den": dict(kind="denoiser", B=2048, T=48, D=7, H=250, L=2, desc="cfg4's denoiser (cfg3 model), 2048 per GPU"),
    "cfg5": dict(kind="vae", B=4096, T=256, K=128, H=512, phys=2048, cpu_sample=8, no_gemm_tuning=True,
                 desc="long-horizon VAE, T=256 K=128 H=512, 4096/GPU in physical chunks"),
}


@dataclass
class Workload:
    name: str
    model: torch.nn.Module
    B: int
    n_params: int
    loss_fn: Callable  # loss_fn(wrapped_model, lo, hi) -> scalar loss over samples [lo, hi)
    flops_per_sample: float  # 2*T*sum(in*out) over Linear layers (per-sample-gradient contraction)
    io_bytes_per_sample: float  # 4*T*sum(in+out): activation + backprop reads


def _linear_dims(model):
    from .layers import RNNLinear
    dims = []
    for m in model.modules():
        if type(m) in (torch.nn.Linear, RNNLinear):
            dims.append((m.in_features, m.out_features))
    return dims


def build(name: str, device, batch: int = None, seed_offset: int = 0, fuse_recurrence: bool = True) -> Workload:
    c = dict(CONFIGS[name])
    if c["kind"] == "pair":
        raise ValueError(f"{name} is a pair of models: build its parts {c['parts']} one by one")
    B = batch or c["B"]
    torch.manual_seed(0)
    gd = torch.Generator().manual_seed(1 + seed_offset)
    ge = torch.Generator().manual_seed(2 + seed_offset)
    T = c["T"]
    if c["kind"] == "vae":
        K, H = c["K"], c["H"]
        model = models.DeapStackDP(64, B, T, 0, 0, K, [], K, H, 1, 128, 8, 7).to(device).train()
        if fuse_recurrence and torch.device(device).type == "cuda":
            model = fuse_recurrences(model)  # what PrivacyEngine.make_private does to CustomGRU-shaped modules
        x = torch.rand(B, T, K, generator=gd).to(device)
        eps = torch.randn(B, T, 7, generator=ge).to(device)

        def loss_fn(wrapped, lo=0, hi=B):
            inner = getattr(wrapped, "_module", wrapped)
            inner.injected_eps = eps[lo:hi]
            return models.vae_loss(wrapped, x[lo:hi], beta=0.1)
    else:
        D, H, L = c["D"], c["H"], c["L"]
        model = models.BiRNNScoreDP(D, H, L, 100).to(device).train()
        x0 = torch.randn(B, T, D, generator=gd)
        t = torch.rand(B, T, 1, generator=gd).sort(1)[0]
        i = torch.randint(0, 100, (B, 1, 1), generator=gd).float().expand(B, T, 1).contiguous()
        time_info = torch.rand(B, T, 8, generator=gd) * 2 - 1
        noise = torch.randn(B, T, D, generator=ge)
        alpha = models.diffusion_alphas(100)[i.long()]
        x_noisy = torch.sqrt(alpha) * x0 + torch.sqrt(1 - alpha) * noise
        x_noisy, t, i, time_info, noise = (v.to(device) for v in (x_noisy, t, i, time_info, noise))

        def loss_fn(wrapped, lo=0, hi=B):
            return models.denoiser_loss(wrapped, x_noisy[lo:hi], t[lo:hi], i[lo:hi], time_info[lo:hi], noise[lo:hi])
    dims = _linear_dims(model)
    n_params = sum(p.numel() for p in model.parameters())
    return Workload(name=name, model=model, B=B, n_params=n_params, loss_fn=loss_fn,
                    flops_per_sample=2.0 * T * sum(i_ * o for i_, o in dims),
                    io_bytes_per_sample=4.0 * T * sum(i_ + o for i_, o in dims))
