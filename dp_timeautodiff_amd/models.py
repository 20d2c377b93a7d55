"""The build's own structurally identical copies of the two networks the DP-SGD step serves, used by
bench.py, smoke() and the parity tests (the reference's Python never travels to the GPU box).

Same parameter names and shapes, same forward arithmetic, as
  * `DeapStack_DP` / `CustomGRU` / `Embedding_data` / `compute_sine_cosine`
    (Model_Code/timeautoencoder.py:30-52, 55-96, 114-175, 317-416) and the loss `auto_loss` + KL (:419-457, :745-748);
  * `BiRNN_score` / `PositionalEncoding` / `FeedForward` (Model_Code/timediffusion.py:54-134) with the
    epsilon-prediction loss of `get_loss` (:162-170) and the linear-beta schedule (:9-16).
Differences kept on purpose: no module-level `device` global (tensors stay where the input is), the
reparametrisation noise can be injected (the reference's randn_like at :377 is unseedable across batch
shapes), and the denoiser's LSTM is built as DPLSTM directly (what ModuleValidator.fix would produce).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import DPLSTM, run_together


def sine_cosine_features(v: torch.Tensor, num_terms: int = 8) -> torch.Tensor:
    freqs = (2.0 ** torch.arange(num_terms, device=v.device, dtype=torch.float32)) * math.pi
    ang = freqs * v.unsqueeze(-1)
    s = torch.sin(ang)
    c = torch.cos(ang)
    flat = s.shape[-2] * s.shape[-1]  # explicit: reshape(-1) is ambiguous for an EMPTY Poisson batch (0 samples)
    return torch.cat((s.reshape(*s.shape[:-2], flat), c.reshape(*c.shape[:-2], flat)), dim=-1)


class LinearCellGRU(nn.Module):
    """GRU written with nn.Linear gates only (the reference's CustomGRU): every gate is one Linear
    called T times per forward — the K=T stacking case of the per-sample-gradient kernel."""

    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.batch_first = num_layers, batch_first
        self.input_layer = nn.Linear(input_size, hidden_size)
        self.update_gate = nn.Linear(2 * hidden_size, hidden_size)
        self.reset_gate = nn.Linear(2 * hidden_size, hidden_size)
        self.new_gate = nn.Linear(2 * hidden_size, hidden_size)

    def forward(self, x, h=None):
        if not self.batch_first:
            x = x.transpose(0, 1)
        B, T, _ = x.shape
        if h is None:
            h = x.new_zeros(B, self.hidden_size)
        x = self.input_layer(x)
        outs = []
        for t in range(T):
            x_t = x[:, t]
            comb = torch.cat([x_t, h], dim=1)
            z = torch.sigmoid(self.update_gate(comb))
            r = torch.sigmoid(self.reset_gate(comb))
            n = torch.tanh(self.new_gate(torch.cat([x_t, r * h], dim=1)))
            h = (1 - z) * n + z * h
            outs.append(h)
        return torch.stack(outs, dim=1), h.unsqueeze(0)


class FeatureEmbedding(nn.Module):
    def __init__(self, input_size, emb_dim, n_bins, n_cats, n_nums, cards):
        super().__init__()
        self.n_bins, self.n_cats, self.n_nums, self.cards = n_bins, n_cats, n_nums, cards
        self.n_disc = n_bins + n_cats
        self.num_categorical_list = [2] * n_bins + list(cards)
        if self.n_disc != 0:
            self.embeddings_list = nn.ModuleList([nn.Embedding(c, emb_dim) for c in self.num_categorical_list])
        if n_nums != 0:
            self.mlp_nums = nn.Sequential(nn.Linear(16 * n_nums, 16 * n_nums), nn.SiLU(),
                                          nn.Linear(16 * n_nums, 16 * n_nums))
        self.mlp_output = nn.Sequential(nn.Linear(emb_dim * self.n_disc + 16 * n_nums, emb_dim), nn.ReLU(),
                                        nn.Linear(emb_dim, input_size))

    def forward(self, x):
        parts = []
        if self.n_disc != 0:
            x_disc = x[:, :, : self.n_disc].long()
            parts.append(torch.cat([emb(x_disc[:, :, i]) for i, emb in enumerate(self.embeddings_list)], dim=2))
        if self.n_nums != 0:
            x_nums = x[:, :, self.n_disc: self.n_disc + self.n_nums]
            parts.append(self.mlp_nums(sine_cosine_features(x_nums, 8)))
        return self.mlp_output(torch.cat(parts, dim=2) if len(parts) > 1 else parts[0])


class DeapStackDP(nn.Module):
    """VAE of the reference's DP path (DeapStack_DP).  Argument order matches the reference constructor."""

    def __init__(self, channels, batch_size, seq_len, n_bins, n_cats, n_nums, cards, input_size, hidden_size,
                 num_layers, cat_emb_dim, time_dim, lat_dim):
        super().__init__()
        self.Emb = FeatureEmbedding(input_size, cat_emb_dim, n_bins, n_cats, n_nums, cards)
        self.encoder_mu = LinearCellGRU(input_size, hidden_size, num_layers, batch_first=True)
        self.encoder_logvar = LinearCellGRU(input_size, hidden_size, num_layers, batch_first=True)
        self.fc_mu = nn.Linear(hidden_size, lat_dim)
        self.fc_logvar = nn.Linear(hidden_size, lat_dim)
        self.decoder_mlp = nn.Sequential(nn.Linear(lat_dim, hidden_size), nn.ReLU(),
                                         nn.Linear(hidden_size, hidden_size))
        self.channels, self.n_bins, self.n_cats, self.n_nums = channels, n_bins, n_cats, n_nums
        self.cards = list(cards)
        self.disc = n_bins + n_cats
        self.bins_linear = nn.Linear(hidden_size, n_bins) if n_bins else None
        self.cats_linears = nn.ModuleList([nn.Linear(hidden_size, c) for c in cards]) if n_cats else None
        self.nums_linear = nn.Linear(hidden_size, n_nums) if n_nums else None
        self.injected_eps: Optional[torch.Tensor] = None

    def reparametrize(self, mu, logvar):
        std = torch.exp(0.5 * logvar)
        eps = self.injected_eps if self.injected_eps is not None else torch.randn_like(std)
        return mu + eps * std

    def encoder(self, x):
        x = self.Emb(x)
        # the two encoders read the same x and do not depend on each other: fused ones share one launch
        (mu_z, _), (logvar_z, _) = run_together([self.encoder_mu, self.encoder_logvar], [x, x])
        mu_z = self.fc_mu(mu_z)
        logvar_z = self.fc_logvar(logvar_z)
        return self.reparametrize(mu_z, logvar_z), mu_z, logvar_z

    def decoder(self, latent_feature):
        out = {}
        h = self.decoder_mlp(latent_feature)
        if self.bins_linear is not None:
            out["bins"] = self.bins_linear(h)
        if self.cats_linears is not None:
            out["cats"] = [lin(h) for lin in self.cats_linears]
        if self.nums_linear is not None:
            out["nums"] = torch.sigmoid(self.nums_linear(h))
        return out

    def forward(self, x):
        emb, mu_z, logvar_z = self.encoder(x)
        return self.decoder(emb), emb, mu_z, logvar_z


def auto_loss(inputs, reconstruction, n_bins, n_nums, n_cats, beta, cards):
    """(disc_loss, num_loss), every term mean-reduced (BCE-with-logits, mean of CEs, MSE)."""
    B, L, K = inputs.shape
    bins = inputs[:, :, :n_bins]
    cats = inputs[:, :, n_bins: n_bins + n_cats].long()
    nums = inputs[:, :, n_bins + n_cats: n_bins + n_cats + n_nums]
    disc_loss = 0
    num_loss = 0
    if "bins" in reconstruction:
        disc_loss = disc_loss + F.binary_cross_entropy_with_logits(reconstruction["bins"], bins)
    if "cats" in reconstruction:
        ces = [F.cross_entropy(reconstruction["cats"][i].reshape(B * L, cards[i]), cats[:, :, i].reshape(B * L))
               for i in range(len(reconstruction["cats"]))]
        disc_loss = disc_loss + torch.stack(ces).mean()
    if "nums" in reconstruction:
        num_loss = F.mse_loss(reconstruction["nums"], nums)
    return disc_loss, num_loss


def vae_loss(model: nn.Module, x: torch.Tensor, beta: float = 0.1):
    """loss_Auto of the reference's DP loop: num + disc + beta * KL, all means over the batch."""
    inner = getattr(model, "_module", model)
    outputs, _, mu_z, logvar_z = model(x)
    disc_loss, num_loss = auto_loss(x, outputs, inner.n_bins, inner.n_nums, inner.n_cats, beta, inner.cards)
    temp = 1 + logvar_z - mu_z.pow(2) - logvar_z.exp()
    loss_kld = -0.5 * torch.mean(temp.mean(-1).mean())
    return num_loss + disc_loss + beta * loss_kld


# ---- latent diffusion denoiser -------------------------------------------------------------------------


def diffusion_alphas(steps: int, device=None) -> torch.Tensor:
    ind = torch.linspace(0, 1, steps, device=device)
    betas = 1e-4 * (1 - ind) + 0.2 * ind
    return torch.cumprod(1 - betas, dim=0)


class PositionalEncoding(nn.Module):
    def __init__(self, dim: int, max_value: float):
        super().__init__()
        self.max_value = max_value
        linear_dim = dim // 2
        periodic_dim = dim - linear_dim
        scale = torch.exp(-2 * torch.arange(0, periodic_dim).float() * math.log(self.max_value) / periodic_dim)
        shift = torch.zeros(periodic_dim)
        shift[::2] = 0.5 * math.pi
        # plain tensors in the reference (not parameters/buffers -> absent from its state_dict): keep them
        # out of the state_dict too
        self.register_buffer("scale", scale, persistent=False)
        self.register_buffer("shift", shift, persistent=False)
        self.linear_proj = nn.Linear(1, linear_dim)

    def forward(self, t):
        periodic = torch.sin(t * self.scale + self.shift)
        linear = self.linear_proj(t / self.max_value)
        return torch.cat([linear, periodic], -1)


class FeedForward(nn.Module):
    def __init__(self, in_dim: int, hidden_dims: List[int], out_dim: int, final_relu: bool = False):
        super().__init__()
        dims = list(hidden_dims) + [out_dim]
        layers: List[nn.Module] = [nn.Linear(in_dim, dims[0])]
        for i in range(len(dims) - 1):
            layers += [nn.ReLU(), nn.Linear(dims[i], dims[i + 1])]
        if final_relu:
            layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


class BiRNNScoreDP(nn.Module):
    """BiRNN_score with its bidirectional LSTM in DP (Linear-cell) form."""

    def __init__(self, input_size, hidden_size, num_layers, diffusion_steps):
        super().__init__()
        self.hidden_size, self.num_layers = hidden_size, num_layers
        self.input_proj = FeedForward(input_size, [], hidden_size)
        self.t_enc = PositionalEncoding(hidden_size, max_value=1)
        self.i_enc = PositionalEncoding(hidden_size, max_value=diffusion_steps)
        self.proj = FeedForward(4 * hidden_size, [], hidden_size, final_relu=True)
        self.lstm = DPLSTM(hidden_size, hidden_size, num_layers, batch_first=True, bidirectional=True)
        self.layer_norm = nn.LayerNorm(2 * hidden_size)
        self.fc = nn.Linear(2 * hidden_size, input_size)
        self.time_encode = nn.Sequential(nn.Linear(8, hidden_size), nn.ReLU(), nn.Linear(hidden_size, hidden_size))

    def forward(self, x, t, i, time_info):
        shape = x.shape
        x = x.view(-1, *shape[-2:])
        t = t.view(-1, shape[-2], 1)
        i = i.view(-1, shape[-2], 1)
        x = self.input_proj(x)
        t = self.t_enc(t)
        i = self.i_enc(i)
        time_info = self.time_encode(time_info)
        x = self.proj(torch.cat([x, t, i, time_info], -1))
        out, _ = self.lstm(x)
        return self.fc(self.layer_norm(out))


def denoiser_loss(model: nn.Module, x_noisy, t, i, time_info, noise):
    return torch.mean((model(x_noisy, t, i, time_info) - noise) ** 2)


def reference_param_names(model: nn.Module) -> List[str]:
    """Parameter names as the reference's state_dict spells them (DPLSTM cells -> lstm.weight_ih_l0...)."""
    ids = {}
    for name, sub in model.named_modules():
        if isinstance(sub, DPLSTM):
            for new, (mod, attr) in sub._rename.items():
                ids[id(getattr(mod, attr))] = (name + "." if name else "") + new
    return [ids.get(id(p), n) for n, p in model.named_parameters()]
