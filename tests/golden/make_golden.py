#!/usr/bin/env python3
"""Golden-vector generator for the DP-SGD hot path (run in the BUILD container only).

This script imports the *reference* model/loss code from /root/reference (read-only) to
produce input/output vectors.  Nothing from the reference is copied: only tensors
(inputs, weights drawn from a numpy Generator, captured activations/backprops and the
expected per-sample gradients) are written to ``tests/golden/*.npz``.

The reference keeps all DP-SGD arithmetic inside Opacus, which is not installed here
(SURVEY.md §8c).  Its top-of-file ``from opacus import ...`` lines
(timeautoencoder.py:18-21) are satisfied with empty placeholder modules so that the
*model and loss* classes can be imported; no Opacus behaviour is emulated by them.

Ground truth for per-sample gradients = a micro-batch (B=1) autograd loop over the
reference model, valid because every loss term is a mean over the batch
(auto_loss timeautoencoder.py:439-453, KL :745-746, get_loss timediffusion.py:169-170),
so Opacus' "backprop x B" equals d(loss_b)/d(theta).

Ground truth for the clip / sum / noise / scale tail = the DPOptimizer formulas
 evaluated in float64 on those per-sample gradients.

Usage:  python tests/golden/make_golden.py   (writes next to this file)
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference/TimeAutoDiff-main"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    for name, attrs in [
        ("opacus", ["PrivacyEngine"]),
        ("opacus.validators", ["ModuleValidator"]),
        ("opacus.accountants", []),
        ("opacus.accountants.utils", ["get_noise_multiplier"]),
        ("opacus.utils", []),
        ("opacus.utils.batch_memory_manager", ["BatchMemoryManager"]),
    ]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    sys.path.insert(0, os.path.join(REF, "Model_Code"))
    import timeautoencoder as tae  # noqa
    import timediffusion as tdf  # noqa
    import DP as dpmod  # noqa

    return tae, tdf, dpmod


def fill_params(model, seed):
    """Deterministic, torch-init-independent weights: N(0,1)/sqrt(fan_in) from PCG64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    with torch.no_grad():
        for name, p in model.named_parameters():
            fan_in = p.shape[-1] if p.dim() > 1 else max(p.shape[0], 1)
            v = rng.standard_normal(tuple(p.shape)).astype(np.float32) / np.sqrt(np.float32(fan_in))
            if name.endswith("layer_norm.weight"):
                v = v + np.float32(1.0)
            p.copy_(torch.from_numpy(v))


HOOKED = (torch.nn.Linear, torch.nn.Embedding, torch.nn.LayerNorm)


class Capture:
    """Record (input, grad_output) of every call of every Linear/Embedding/LayerNorm."""

    def __init__(self, model):
        self.A = {}
        self.G = {}
        self.handles = []
        for name, mod in model.named_modules():
            if type(mod) in HOOKED:
                self.A[name] = []
                self.G[name] = []
                self.handles.append(mod.register_forward_hook(self._fwd(name)))
                self.handles.append(mod.register_full_backward_hook(self._bwd(name)))

    def _fwd(self, name):
        def hook(mod, inp, out):
            self.A[name].append(inp[0].detach().clone())

        return hook

    def _bwd(self, name):
        def hook(mod, gin, gout):
            # backward visits calls in reverse order -> prepend
            self.G[name].insert(0, gout[0].detach().clone())

        return hook

    def remove(self):
        for h in self.handles:
            h.remove()


def dp_tail(gs_flat32, C, sigma, expected_bs, noise32):
    """Appendix A.3 in float64 and in float32 (torch op order)."""
    out = {}
    g64 = gs_flat32.astype(np.float64)
    n64 = np.sqrt((g64 * g64).sum(1))
    c64 = np.minimum(1.0, C / (n64 + 1e-6))
    s64 = (c64[:, None] * g64).sum(0)
    grad64 = (s64 + noise32.astype(np.float64)) / expected_bs
    out["norms_f64"] = n64
    out["clip_f64"] = c64
    out["summed_f64"] = s64
    out["grad_f64"] = grad64
    return out


def opacus_order_f32(per_param_gs, C):
    """float32, in the op order DPOptimizer.clip_and_accumulate uses (A.3)."""
    per_param_norms = [g.reshape(len(g), -1).norm(2, dim=-1) for g in per_param_gs]
    per_sample_norms = torch.stack(per_param_norms, dim=1).norm(2, dim=1)
    clip = (C / (per_sample_norms + 1e-6)).clamp(max=1.0)
    summed = [torch.einsum("i,i...", clip, g) for g in per_param_gs]
    return per_sample_norms, clip, summed, torch.stack(per_param_norms, dim=1)


def pack_case(out, model, names, gs, cap, meta, C_list, sigma, seed_noise, store_full=True):
    B = meta["B"]
    P = sum(p.numel() for p in model.parameters())
    gs_flat = np.concatenate([gs[n].reshape(B, -1) for n in names], axis=1).astype(np.float32)
    assert gs_flat.shape == (B, P)
    rng = np.random.Generator(np.random.PCG64(seed_noise))
    meta["names"] = names
    meta["shapes"] = [list(gs[n].shape[1:]) for n in names]
    meta["P"] = int(P)
    # C=None -> use the median per-sample norm so that both branches of the clamp are exercised
    med = float(np.float32(np.median(np.sqrt((gs_flat.astype(np.float64) ** 2).sum(1)))))
    C_list = [med if C is None else C for C in C_list]
    meta["C_list"] = C_list
    meta["sigma"] = sigma
    for ci, C in enumerate(C_list):
        noise = (rng.standard_normal(P) * sigma * C).astype(np.float32)
        tail = dp_tail(gs_flat, C, sigma, meta["expected_batch_size"], noise)
        pn, clip32, summed32, ppn = opacus_order_f32([torch.from_numpy(gs[n]) for n in names], C)
        if store_full:
            out[f"tail{ci}/noise"] = noise
            out[f"tail{ci}/summed_f64"] = tail["summed_f64"]
            out[f"tail{ci}/grad_f64"] = tail["grad_f64"]
            out[f"tail{ci}/summed_f32"] = np.concatenate([s.reshape(-1).numpy() for s in summed32])
        else:
            # large case: keep the first 64 elements of every parameter only
            offs = np.cumsum([0] + [int(np.prod(s)) for s in meta["shapes"]])
            idx = np.concatenate([np.arange(o, min(o + 64, e)) for o, e in zip(offs[:-1], offs[1:])])
            out[f"tail{ci}/idx"] = idx
            out[f"tail{ci}/noise_seed"] = np.array([seed_noise + 1000 * ci], dtype=np.int64)
            noise = (np.random.Generator(np.random.PCG64(seed_noise + 1000 * ci)).standard_normal(P)
                     * sigma * C).astype(np.float32)
            tail = dp_tail(gs_flat, C, sigma, meta["expected_batch_size"], noise)
            out[f"tail{ci}/summed_f64"] = tail["summed_f64"][idx]
            out[f"tail{ci}/grad_f64"] = tail["grad_f64"][idx]
        out[f"tail{ci}/norms_f64"] = tail["norms_f64"]
        out[f"tail{ci}/clip_f64"] = tail["clip_f64"]
        out[f"tail{ci}/norms_f32"] = pn.numpy()
        out[f"tail{ci}/clip_f32"] = clip32.numpy()
        if ci == 0:
            out["per_param_norms_f32"] = ppn.numpy()
    if store_full:
        for n in names:
            out["gs/" + n] = gs[n].astype(np.float32)
        for lname in cap.A:
            for k, a in enumerate(cap.A[lname]):
                out[f"A/{lname}/{k}"] = a.numpy()
            for k, g in enumerate(cap.G[lname]):
                out[f"G/{lname}/{k}"] = g.numpy()
        out["layers"] = np.array(json.dumps({l: len(v) for l, v in cap.A.items()}))
    out["meta"] = np.array(json.dumps(meta))


def vae_loss(tae, model, x, eps, n_bins, n_nums, n_cats, cards, beta):
    # forward with injected reparametrisation noise (randn_like at timeautoencoder.py:377)
    model.reparametrize = lambda mu, logvar: mu + eps * torch.exp(0.5 * logvar)
    outputs, _, mu_z, logvar_z = model(x)
    disc_loss, num_loss = tae.auto_loss(x, outputs, n_bins, n_nums, n_cats, beta, cards)
    temp = 1 + logvar_z - mu_z.pow(2) - logvar_z.exp()
    loss_kld = -0.5 * torch.mean(temp.mean(-1).mean())
    return num_loss + disc_loss + beta * loss_kld, mu_z, logvar_z


def make_vae_case(tae, fname, ctor, B, x, beta, seed_w, seed_eps, C_list, sigma, store_full=True,
                  expected_bs=None, extra_meta=None):
    torch.manual_seed(0)
    model = tae.DeapStack_DP(**ctor)
    fill_params(model, seed_w)
    model.train()
    T = x.shape[1]
    lat = ctor["lat_dim"]
    eps = torch.from_numpy(
        np.random.Generator(np.random.PCG64(seed_eps)).standard_normal((B, T, lat)).astype(np.float32))
    n_bins, n_cats, n_nums, cards = ctor["n_bins"], ctor["n_cats"], ctor["n_nums"], ctor["cards"]
    names = [n for n, _ in model.named_parameters()]

    # full-batch pass with capture
    cap = Capture(model)
    loss, mu, logvar = vae_loss(tae, model, x, eps, n_bins, n_nums, n_cats, cards, beta)
    model.zero_grad()
    loss.backward()
    cap.remove()
    full_grad = {n: p.grad.detach().clone().numpy() for n, p in model.named_parameters()}

    # micro-batch loop
    gs = {n: np.zeros((B,) + tuple(p.shape), np.float32) for n, p in model.named_parameters()}
    for b in range(B):
        model.zero_grad()
        lb, _, _ = vae_loss(tae, model, x[b:b + 1], eps[b:b + 1], n_bins, n_nums, n_cats, cards, beta)
        lb.backward()
        for n, p in model.named_parameters():
            gs[n][b] = p.grad.detach().numpy()
    err = max(np.abs(gs[n].mean(0) - full_grad[n]).max() for n in names)
    print(f"{fname}: mean_b(gs) vs full-batch grad max abs err = {err:.3e}")
    assert err < 1e-6

    out = {}
    meta = dict(kind="vae", ctor=ctor, B=B, T=T, beta=beta, seed_w=seed_w, seed_eps=seed_eps,
                expected_batch_size=expected_bs or B, g_scale=B, loss=float(loss))
    if extra_meta:
        meta.update(extra_meta)
    out["x"] = x.numpy()
    out["eps"] = eps.numpy()
    out["mu"] = mu.detach().numpy() if store_full else mu.detach().numpy()[:2]
    out["logvar"] = logvar.detach().numpy() if store_full else logvar.detach().numpy()[:2]
    if store_full:
        for n, p in model.named_parameters():
            out["w/" + n] = p.detach().numpy()
    pack_case(out, model, names, gs, cap, meta, C_list, sigma, seed_noise=seed_w + 7, store_full=store_full)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    return out


def make_denoiser_case(tdf, fname, D, H, L, steps, B, T, seed_w, seed_d, C_list, sigma):
    torch.manual_seed(0)
    model = tdf.BiRNN_score(D, H, L, steps)
    fill_params(model, seed_w)
    model.train()
    rng = np.random.Generator(np.random.PCG64(seed_d))
    x0 = torch.from_numpy(rng.standard_normal((B, T, D)).astype(np.float32))
    t = torch.from_numpy(np.sort(rng.random((B, T, 1)).astype(np.float32), axis=1))
    i = torch.from_numpy(rng.integers(0, steps, size=(B, 1, 1)).astype(np.float32)).expand(B, T, 1).contiguous()
    time_info = torch.from_numpy((rng.random((B, T, 8)) * 2 - 1).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((B, T, D)).astype(np.float32))
    # add_noise (timediffusion.py:26-42) with identity covariance and injected noise
    alpha = tdf.alphas[i.long()].to(x0)
    x_noisy = torch.sqrt(alpha) * x0 + torch.sqrt(1 - alpha) * noise
    names = [n for n, _ in model.named_parameters()]

    def loss_fn(sl):
        pred = model(x_noisy[sl], t[sl], i[sl], time_info[sl])
        return torch.mean((pred - noise[sl]) ** 2)

    cap = Capture(model)
    model.zero_grad()
    loss = loss_fn(slice(0, B))
    loss.backward()
    cap.remove()
    full_grad = {n: p.grad.detach().clone().numpy() for n, p in model.named_parameters()}
    gs = {n: np.zeros((B,) + tuple(p.shape), np.float32) for n, p in model.named_parameters()}
    for b in range(B):
        model.zero_grad()
        loss_fn(slice(b, b + 1)).backward()
        for n, p in model.named_parameters():
            gs[n][b] = p.grad.detach().numpy()
    err = max(np.abs(gs[n].mean(0) - full_grad[n]).max() for n in names)
    print(f"{fname}: mean_b(gs) vs full-batch grad max abs err = {err:.3e}")
    assert err < 1e-6
    out = {}
    meta = dict(kind="denoiser", ctor=dict(input_size=D, hidden_size=H, num_layers=L, diffusion_steps=steps),
                B=B, T=T, seed_w=seed_w, seed_d=seed_d, expected_batch_size=B, g_scale=B, loss=float(loss))
    out["x_noisy"] = x_noisy.numpy()
    out["t"] = t.numpy()
    out["i"] = i.numpy()
    out["time_info"] = time_info.numpy()
    out["noise_target"] = noise.numpy()
    for n, p in model.named_parameters():
        out["w/" + n] = p.detach().numpy()
    pack_case(out, model, names, gs, cap, meta, C_list, sigma, seed_noise=seed_w + 7)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def main():
    tae, tdf, dpmod = _import_reference()
    torch.set_num_threads(8)

    # --- case 1: small VAE with binary + categorical + numeric columns (Embedding path) -------------
    ctor = dict(channels=64, batch_size=8, seq_len=6, n_bins=1, n_cats=2, n_nums=3, cards=[3, 5],
                input_size=6, hidden_size=16, num_layers=1, cat_emb_dim=8, time_dim=8, lat_dim=4)
    B, T = 8, 6
    rng = np.random.Generator(np.random.PCG64(11))
    x = np.zeros((B, T, 6), np.float32)
    x[:, :, 0] = rng.integers(0, 2, (B, T))
    x[:, :, 1] = rng.integers(0, 3, (B, T))
    x[:, :, 2] = rng.integers(0, 5, (B, T))
    x[:, :, 3:] = rng.random((B, T, 3))
    make_vae_case(tae, "vae_small_mixed.npz", ctor, B, torch.from_numpy(x), beta=0.1, seed_w=101,
                  seed_eps=102, C_list=[1.0, None], sigma=1.0)

    # --- case 2: small numeric-only VAE (the layout of the synthetic bench configs) -----------------
    ctor = dict(channels=64, batch_size=12, seq_len=7, n_bins=0, n_cats=0, n_nums=5, cards=[],
                input_size=5, hidden_size=24, num_layers=1, cat_emb_dim=16, time_dim=8, lat_dim=3)
    B, T = 12, 7
    x = np.random.Generator(np.random.PCG64(21)).random((B, T, 5)).astype(np.float32)
    make_vae_case(tae, "vae_small_nums.npz", ctor, B, torch.from_numpy(x), beta=0.1, seed_w=201,
                  seed_eps=202, C_list=[1.0, None], sigma=0.8)

    # --- case 3: small denoiser (nn.LSTM + LayerNorm; DPLSTM must reproduce the lstm.* grads) -------
    make_denoiser_case(tdf, "denoiser_small.npz", D=4, H=8, L=2, steps=100, B=6, T=5, seed_w=301,
                       seed_d=302, C_list=[1.0, None], sigma=1.1)

    # --- case 4: true dp_job.sh shape on real AirQuality windows (BASELINE.json configs[0]) ---------
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "Dataset/Single-Sequence/AirQuality.csv"))
    df1 = df.drop("date", axis=1) if "date" in df.columns else df
    processed = dpmod.splitData(df1, 24, 1)  # timeautodiff_run.py:44
    print("AirQuality processed:", tuple(processed.shape))
    B = 64
    idx = np.random.Generator(np.random.PCG64(41)).choice(processed.shape[0], B, replace=False)
    x = processed[np.sort(idx)].float()
    ctor = dict(channels=64, batch_size=B, seq_len=24, n_bins=0, n_cats=0, n_nums=13, cards=[],
                input_size=13, hidden_size=200, num_layers=1, cat_emb_dim=128, time_dim=8, lat_dim=7)
    make_vae_case(tae, "vae_airquality_b64.npz", ctor, B, x, beta=0.1, seed_w=401, seed_eps=402,
                  C_list=[1.0, None], sigma=1.0, store_full=False, expected_bs=49,
                  extra_meta=dict(n_windows=int(processed.shape[0]), window_idx=np.sort(idx).tolist()))


if __name__ == "__main__":
    main()
