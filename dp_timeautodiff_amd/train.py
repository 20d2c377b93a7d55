"""train_dp_autoencoder — the reference's DP training loop (Model_Code/timeautoencoder.py:603-786) on this engine,
as a fixed, resumable harness.

Same protocol and defaults as the reference: ModuleValidator.fix -> Adam(lr, weight_decay) ->
PrivacyEngine().make_private_with_epsilon(epochs, target_epsilon, target_delta, max_grad_norm) -> per epoch a pass over
the Poisson loader under BatchMemoryManager(max_physical_batch_size=128) with zero_grad / forward / auto_loss + beta*KL
/ backward / step, then get_epsilon(delta) and the beta-annealing rule.  What is fixed relative to the reference:
  * `lambd` (undefined at :773) is a parameter (the non-DP loop's 0.7);
  * the loader is not iterated an extra time per epoch just to count batches (:711-717 burns a Poisson pass);
  * checkpoints also carry the accountant history, the DP-noise / sampler stream positions, torch's device RNG state
    and beta, so a resumed
    run continues the same privacy ledger (the reference's save_autoencoder :460-470 drops all of that).
MI355X-first execution: the dataset tensor lives in HBM and is Poisson-sampled on the device (K11); when a logical
batch fits one physical chunk the whole step — sampling included — is one hipGraph replay (fixed-capacity masked
batch); otherwise the BatchMemoryManager chunk loop runs eagerly on device-resident chunks.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import data_types, models
from .batch_memory_manager import BatchMemoryManager
from .data_loader import DevicePoissonLoader
from .graphed import GraphedDPStep
from .privacy_engine import PrivacyEngine
from .validators import ModuleValidator


def train_dp_autoencoder(real_df, processed_data: torch.Tensor, channels: int, hidden_size: int, num_layers: int,
                         lr: float, weight_decay: float, n_epochs: int, batch_size: int, threshold, min_beta: float,
                         max_beta: float, emb_dim: int, time_dim: int, lat_dim: int, device, epsilon: float,
                         delta: float = 1e-5, max_grad_norm: float = 1.0, save_path: Optional[str] = None,
                         load_path: Optional[str] = None, *, lambd: float = 0.7, max_physical_batch_size: int = 128,
                         accountant: str = "prv", use_graph: bool = True, tune_gemms: bool = False,
                         noise_seed: Optional[int] = None, sampler_seed: Optional[int] = None, log=print):
    """Same positional arguments, defaults and 8-tuple as the reference's function (timeautoencoder.py:603-625, :786):

        (ae, latent_features.detach(), output, losses, recons_loss, mu_z, logvar_z, epsilon)

    so `run_dp_experiments` (timeautodiff_run.py:315-336: `[0]`, `[1]`, `[7]`) can call it as is.  `real_df` is the
    reference's DataFrame (typed by data_types.datatype_info, its parser's rule), a fitted parser, or the
    {n_bins, n_cats, n_nums, cards} dict.  `losses` / `recons_loss` hold one value per epoch (the reference creates the
    lists and never appends, :688-690, :775-777); `mu_z` / `logvar_z` are the encoder outputs of the final full-data
    forward (the reference returns whatever the last training batch left in those names).  Everything after `load_path`
    is keyword-only and belongs to this engine."""
    info = data_types.resolve(real_df, threshold)
    n_bins, n_cats, n_nums, cards = info["n_bins"], info["n_cats"], info["n_nums"], info["cards"]
    device = torch.device(device)
    if tune_gemms and device.type == "cuda":
        # per-shape hipBLASLt/rocBLAS solution for the model's own GEMMs (+3 % at cfg2); the choice is timing-based, so
        # two runs may round differently: leave it off where bit-identical resumption matters
        from . import gemm_tuning
        gemm_tuning.enable()
    N, seq_len, input_size = processed_data.shape
    ae = models.DeapStackDP(channels, batch_size, seq_len, n_bins, n_cats, n_nums, list(cards), input_size, hidden_size,
                            num_layers, emb_dim, time_dim, lat_dim).to(device)
    ae = ModuleValidator.fix(ae)
    if not ModuleValidator.is_valid(ae):
        log(ModuleValidator.validate(ae, strict=False))
        raise ValueError("Autoencoder model is not compatible with DP training")
    optimizer_ae = torch.optim.Adam(ae.parameters(), lr=lr, weight_decay=weight_decay)
    train_loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(processed_data), batch_size=batch_size,
                                               shuffle=True, drop_last=False)
    privacy_engine = PrivacyEngine(accountant=accountant)
    ckpt = torch.load(load_path, map_location=device, weights_only=True) if load_path else None
    if ckpt is not None:  # restore the ledger first: make_private_with_epsilon warns if budget was already spent
        ae.load_state_dict(ckpt["module_state_dict"])
    ae, optimizer_ae, train_loader = privacy_engine.make_private_with_epsilon(
        module=ae, optimizer=optimizer_ae, data_loader=train_loader, epochs=n_epochs, target_epsilon=epsilon,
        target_delta=delta, max_grad_norm=max_grad_norm, device_sampler=True, noise_seed=noise_seed)
    assert isinstance(train_loader, DevicePoissonLoader)
    if sampler_seed is not None:
        train_loader.seed = int(sampler_seed)
    beta_t = torch.tensor(float(max_beta), device=device)
    start_epoch, best_train_loss, patience = 0, float("inf"), 0
    if ckpt is not None:
        privacy_engine.accountant.load_state_dict(ckpt["privacy_accountant_state_dict"])
        optimizer_ae.load_state_dict(ckpt["optimizer_state_dict"])
        ns, ts = ckpt["dp_noise_state"], ckpt["train_state"]
        optimizer_ae.set_noise_state(int(ns["seed"]), int(ns["offset"]))
        train_loader.seed = int(ts["sampler_seed"])
        train_loader._offset_dev.fill_(int(ts["sampler_offset"]))
        beta_t.fill_(float(ts["beta"]))
        if device.type == "cuda":
            torch.cuda.set_rng_state(ckpt["cuda_rng_state"].cpu(), device)  # the VAE's reparameterisation draws
        start_epoch, best_train_loss, patience = int(ts["epoch"]), float(ts["best_train_loss"]), int(ts["patience"])
        log(f"Resumed from epoch {start_epoch}: eps spent {privacy_engine.get_epsilon(delta):.3f}")

    def loss_fn(batch):
        return models.vae_loss(ae, batch[0], beta=beta_t)

    graphed, steps_done_by_capture = None, 0
    if use_graph and train_loader.capacity <= max_physical_batch_size and start_epoch < n_epochs:
        # the capture's warm-up steps are real training steps (they size the arena and count in the accountant):
        # they are the first steps of the first epoch
        steps_done_by_capture = min(2, len(train_loader))
        ae.train()
        graphed = GraphedDPStep(ae, optimizer_ae, loss_fn, warmup=steps_done_by_capture, loader=train_loader).capture()
    eps_now, loss_val = 0.0, float("nan")
    losses, recons_loss = [], []
    for epoch in range(start_epoch, n_epochs):
        ae.train()
        epoch_loss = torch.zeros((), device=device)
        num_batches = 0
        if graphed is not None:
            for _ in range(len(train_loader) - steps_done_by_capture):
                epoch_loss += graphed()
                num_batches += 1
            steps_done_by_capture = 0
            loss_t = graphed.static_loss
        else:
            with BatchMemoryManager(data_loader=train_loader, max_physical_batch_size=max_physical_batch_size,
                                    optimizer=optimizer_ae) as memory_safe_data_loader:
                for batch_data in memory_safe_data_loader:
                    optimizer_ae.zero_grad()
                    loss_t = loss_fn(batch_data)  # an empty Poisson draw flows through: the step is pure noise
                    loss_t.backward()
                    optimizer_ae.step()
                    if batch_data[0].shape[0] > 0:
                        epoch_loss += loss_t.detach()
                        num_batches += 1
        train_loader.check_overflow()  # a truncated Poisson draw voids the accounting: raise before epsilon is reported
        rep = privacy_engine.epsilon_report(delta)
        eps_now = rep["epsilon"]
        loss_val = float(loss_t.detach())
        losses.append(float(epoch_loss) / max(num_batches, 1))
        # which accountant produced the figure, and the RDP bound (the accountant pinned to Opacus' value) beside it
        log(f"Epoch {epoch + 1}/{n_epochs} - Avg Loss: {losses[-1]:.4f} - eps: {eps_now:.2f} ({rep['accountant']}; "
            f"rdp bound {rep['rdp_epsilon']:.2f})")
        if loss_val < best_train_loss:
            best_train_loss, patience = loss_val, 0
        else:
            patience += 1
            if patience == 10 and float(beta_t) > min_beta:
                beta_t.mul_(lambd)
        if save_path is not None:
            os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
            privacy_engine.save_checkpoint(
                path=save_path, module=ae, optimizer=optimizer_ae,
                checkpoint_dict={"cuda_rng_state": torch.cuda.get_rng_state(device), "train_state": {
                    "epoch": epoch + 1, "best_train_loss": best_train_loss, "patience": patience,
                    "beta": float(beta_t), "sampler_seed": train_loader.seed,
                    "sampler_offset": int(train_loader._offset_dev.item()), "loss": loss_val}})
    ae.eval()
    with torch.no_grad():
        output, latent_features, mu_z, logvar_z = ae(processed_data.to(device))
        recons_loss.append(float(sum(l for l in models.auto_loss(processed_data.to(device), output, n_bins, n_nums,
                                                                  n_cats, 0.0, cards))))
    train_dp_autoencoder.last_epsilon_report = privacy_engine.epsilon_report(delta)  # the 8-tuple is the reference's
    return ae, latent_features.detach(), output, losses, recons_loss, mu_z, logvar_z, eps_now
