"""Gradient-free loss evaluation for `LatentDiffusion`.

The HIP denoiser runs outside autograd, so of the reference's training side only what needs no gradient is reusable: the loss
VALUE of a batch (validation curves of a checkpoint), the EMA shadow weights, and the with / without-EMA evaluation.  The member
names, call signatures and the keys of the returned dictionaries are the reference module's API (the script's callbacks and
loggers read them; reference diffusion/latent_diffusion.py:280-293 ema_scope, :447-476 forward, :487-495 validation_step,
:502-515 get_loss, :517-551 p_losses); the values are pinned on tests/golden/train_side.npz (tests/test_training_side.py).
"""
from contextlib import contextmanager

import torch

import contextlib


def _device_of(t: torch.Tensor):
    """Make a CUDA(HIP) tensor's device current for the body (kernels go to the current stream of the current device)."""
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


def _per_sample_error(kind: str, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Element-wise |d| ("l1") or d^2 ("l2") of d = target - pred."""
    d = target - pred
    if kind == "l1":
        return d.abs()
    if kind == "l2":
        return d * d
    raise NotImplementedError(f"unknown loss type '{kind}'")


class LossEvaluationMixin:
    """Mixed into LatentDiffusion; uses its schedule buffers, `apply_model`, `q_sample`, `get_input` and the encoders."""

    # ------------------------------------------------------------------ EMA shadow weights
    @contextmanager
    def ema_scope(self, context=None):
        """Evaluate the body with the EMA weights in the denoiser; the live weights come back afterwards (no-op without use_ema)."""
        swapped = bool(self.use_ema)
        live = self.torch_nn_module
        if swapped:
            self.model_ema.store(live.parameters())
            self.model_ema.copy_to(live)
        if swapped and context is not None:
            print(f"{context}: Switched to EMA weights")
        try:
            yield None
        finally:
            if swapped:
                self.model_ema.restore(live.parameters())
                if context is not None:
                    print(f"{context}: Restored training weights")

    def on_train_batch_end(self, *args, **kwargs):
        if self.use_ema:
            self.model_ema(self.torch_nn_module)

    # ------------------------------------------------------------------ loss of a batch (values only)
    def get_loss(self, pred, target, mean=True):
        err = _per_sample_error(self.loss_type, pred, target)
        return err.mean() if mean else err

    @torch.no_grad()
    def p_losses(self, x_start, cond, t, noise=None):
        """(loss, loss_dict) of the denoising objective for latents `x_start` noised to steps `t`."""
        if self.parameterization not in ("eps", "x0"):
            raise NotImplementedError(f"parameterization '{self.parameterization}'")
        if noise is None:
            noise = torch.randn_like(x_start)
        prediction = self.apply_model(self.q_sample(x_start=x_start, t=t, noise=noise), t, cond)
        wanted = noise if self.parameterization == "eps" else x_start
        # one error per sample: mean over every axis but the batch axis
        err = self.get_loss(prediction, wanted, mean=False).mean(dim=self.loss_mean_dim)
        logvar = self.logvar[t]
        weighted = err / logvar.exp() + logvar                       # learned-variance weighting (logvar = 0: identity)
        vlb = (self.lvlb_weights[t] * err).mean()                    # variational-bound term
        total = self.l_simple_weight * weighted.mean() + self.original_elbo_weight * vlb
        tag = "train" if self.training else "val"
        out = {f"{tag}/loss_simple": err.mean()}
        if self.learn_logvar:
            out[f"{tag}/loss_gamma"] = weighted.mean()
            out["logvar"] = self.logvar.data.mean()
        out[f"{tag}/loss_vlb"] = vlb
        out[f"{tag}/loss"] = total
        return total, out

    @torch.no_grad()
    def forward(self, batch, verbose=False):
        """(loss, loss_dict) of one batch: `get_input` (dataset dependent, subclass) -> first-stage latents -> random steps ->
        conditioning latents -> `p_losses`."""
        with _device_of(self.betas):
            target, cond = self.get_input(batch)[:2]
            dev = self.betas.device
            target = target.to(dev)
            n = target.shape[self.batch_axis]
            z = target
            if self.first_stage_model is not None:
                z = self._from_frames(self.encode_first_stage(self._to_frames(target)), n)
            steps = torch.randint(0, self.num_timesteps, (n,), device=dev).long()
            if self.cond_stage_model is None:
                zc = cond if isinstance(cond, torch.Tensor) else cond.get("y", None)
            else:
                assert cond is not None
                zc = self.cond_stage_forward(cond)
                if self.shorten_cond_schedule:          # (latent_diffusion.py:469-471: noise shaped like the RAW condition, as there)
                    raw = cond if isinstance(cond, torch.Tensor) else cond["y"]
                    zc = self.q_sample(zc, self.cond_ids[steps], noise=torch.randn_like(raw.float().to(dev)))
            return self.p_losses(z, zc, steps, noise=None)

    def training_step(self, batch, batch_idx):
        raise NotImplementedError("prediff_amd.LatentDiffusion evaluates losses (forward / validation_step) but cannot train: the "
                                  "denoiser forward runs on HIP kernels outside autograd (no backward pass)")

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """Loss dictionaries of a batch with the live weights and (keys suffixed `_ema`) with the EMA weights; both are logged."""
        live = self(batch)[1]
        with self.ema_scope():
            shadow = {k + "_ema": v for k, v in self(batch)[1].items()}
        for d in (live, shadow):
            self.log_dict(d, prog_bar=False, logger=True, on_step=False, on_epoch=True)
        return {**live, **shadow}
