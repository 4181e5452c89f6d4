"""Oracle: diffusion schedule, DDPM ancestral step, DDIM step and the sampling loop.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations are relative to
/root/reference/src/prediff/diffusion/.

PARITY-UNPINNED: the reference ships no DDIM sampler, only the two
schedule helpers restated in `ddim_timesteps` / `ddim_sampling_parameters`.
`ddim_step` follows the stable-diffusion lineage those helpers come from
(utils.py:1) and is pinned only by self-consistency tests.
"""
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


def beta_schedule(schedule: str, n: int, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3) -> np.ndarray:
    """make_beta_schedule.  utils.py:17-39 (float64)."""
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=np.float64) ** 2
    if schedule == "cosine":
        ts = np.arange(n + 1, dtype=np.float64) / n + cosine_s
        a = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n, dtype=np.float64) ** 0.5
    raise ValueError(schedule)


def schedule_buffers(betas: np.ndarray, v_posterior: float = 0.0) -> Dict[str, np.ndarray]:
    """LatentDiffusion.register_schedule: float64 math, fp32 buffers.  latent_diffusion.py:228-268."""
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    buf = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    return {k: v.astype(np.float32) for k, v in buf.items()}


def ddim_timesteps(num_ddim: int, num_ddpm: int, method: str = "uniform") -> np.ndarray:
    """make_ddim_timesteps.  utils.py:42-56."""
    if method == "uniform":
        c = num_ddpm // num_ddim
        steps = np.asarray(list(range(0, num_ddpm, c)))
    elif method == "quad":
        steps = (np.linspace(0, np.sqrt(num_ddpm * 0.8), num_ddim) ** 2).astype(int)
    else:
        raise NotImplementedError(method)
    return steps + 1


def ddim_sampling_parameters(alphacums: np.ndarray, steps: np.ndarray, eta: float):
    """make_ddim_sampling_parameters.  utils.py:59-70."""
    a = alphacums[steps]
    a_prev = np.asarray([alphacums[0]] + alphacums[steps[:-1]].tolist())
    sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return sig, a, a_prev


def _bcast(v: Tensor, x: Tensor) -> Tensor:
    return v.reshape((-1,) + (1,) * (x.dim() - 1))


def ddpm_step(buf: Dict[str, Tensor], zt: Tensor, eps: Tensor, t: Tensor, noise: Tensor,
              mean_shift: Optional[Tensor] = None, clip_denoised: bool = False,
              temperature: float = 1.0) -> Tensor:
    """p_sample given the denoiser output (eps-parameterisation).

    latent_diffusion.py:553-566 (z0, posterior mean/logvar), :592-596 (aligned mean),
    :620-631 (noise, nonzero mask).  `mean_shift` is alignment_fn's return value.
    """
    z0 = _bcast(buf["sqrt_recip_alphas_cumprod"][t], zt) * zt - _bcast(buf["sqrt_recipm1_alphas_cumprod"][t], zt) * eps
    if clip_denoised:
        z0 = z0.clamp(-1.0, 1.0)
    mean = _bcast(buf["posterior_mean_coef1"][t], zt) * z0 + _bcast(buf["posterior_mean_coef2"][t], zt) * zt
    logvar = _bcast(buf["posterior_log_variance_clipped"][t], zt)
    if mean_shift is not None:
        mean = mean - (0.5 * logvar).exp() * mean_shift
    nonzero = _bcast(1 - (t == 0).float(), zt)
    return mean + nonzero * (0.5 * logvar).exp() * (noise * temperature)


def ddim_step(zt: Tensor, eps: Tensor, a_t: Tensor, a_prev: Tensor, sigma: Tensor, noise: Tensor,
              sqrt_recip: Tensor = None, sqrt_recipm1: Tensor = None) -> Tensor:
    """PARITY-UNPINNED.  z_prev = sqrt(a_prev) z0 + sqrt(1-a_prev-sigma^2) eps + sigma n,
    with z0 from the same formula as predict_start_from_noise (latent_diffusion.py:553-557)."""
    a_t, a_prev, sigma = _bcast(a_t, zt), _bcast(a_prev, zt), _bcast(sigma, zt)
    z0 = (zt - (1 - a_t).sqrt() * eps) / a_t.sqrt()
    return a_prev.sqrt() * z0 + (1 - a_prev - sigma ** 2).clamp_min(0).sqrt() * eps + sigma * noise


def cond_schedule(num_timesteps: int, num_timesteps_cond: int) -> np.ndarray:
    """cond_ids of make_cond_schedule (latent_diffusion.py:295-299): the first num_timesteps_cond entries are that many levels spread
    evenly over [0, T-1] (rounded half to even, as torch.round), every later entry is T-1."""
    ids = np.full(num_timesteps, num_timesteps - 1, dtype=np.int64)
    ids[:num_timesteps_cond] = np.rint(np.linspace(0.0, num_timesteps - 1, num_timesteps_cond, dtype=np.float32)).astype(np.int64)
    return ids


def ddpm_sample_loop(buf, denoiser: Callable, zc: Tensor, noise_tape: Sequence[Tensor], timesteps: int,
                     align_fn: Optional[Callable] = None, clip_denoised=False, cond_ids: Optional[np.ndarray] = None,
                     cond_tape: Optional[Sequence[Tensor]] = None) -> List[Tensor]:
    """p_sample_loop with an explicit noise tape [x_T, n_{T-1}, ..., n_0].  latent_diffusion.py:633-684.
    Returns [z_T, z_{T-1}, ..., z_0].  cond_ids / cond_tape: shorten_cond_schedule (:665-667) -- in front of step i the condition is
    replaced by q_sample(condition, cond_ids[i], cond_tape[k]) (cumulatively: the re-noised condition is what the next step re-noises)."""
    z = noise_tape[0]
    traj = [z]
    B = z.shape[0]
    for k, i in enumerate(reversed(range(timesteps))):
        t = torch.full((B,), i, dtype=torch.long)
        if cond_ids is not None:
            lvl = int(cond_ids[i])
            zc = float(buf["sqrt_alphas_cumprod"][lvl]) * zc + float(buf["sqrt_one_minus_alphas_cumprod"][lvl]) * cond_tape[k]
        eps = denoiser(z, t, zc)
        shift = align_fn(z, t) if align_fn is not None else None
        z = ddpm_step(buf, z, eps, t, noise_tape[1 + k], mean_shift=shift, clip_denoised=clip_denoised)
        traj.append(z)
    return traj


def ddim_sample_loop(alphas_cumprod: np.ndarray, denoiser: Callable, zc: Tensor, noise_tape: Sequence[Tensor],
                     num_steps: int, eta: float = 0.0) -> List[Tensor]:
    """PARITY-UNPINNED DDIM loop over the reference's uniform timestep subset (utils.py:42-70).
    The denoiser is queried at t = steps[i] (the helper's "+1" indices, clipped to T-1)."""
    T = alphas_cumprod.shape[0]
    steps = ddim_timesteps(num_steps, T)
    steps = np.minimum(steps, T - 1)
    sig, a, a_prev = ddim_sampling_parameters(alphas_cumprod.astype(np.float64), steps, eta)
    z = noise_tape[0]
    traj = [z]
    B = z.shape[0]
    for k, idx in enumerate(reversed(range(len(steps)))):
        t = torch.full((B,), int(steps[idx]), dtype=torch.long)
        eps = denoiser(z, t, zc)
        f = lambda v: torch.full((B,), float(v), dtype=torch.float32)
        z = ddim_step(z, eps, f(a[idx]), f(a_prev[idx]), f(sig[idx]), noise_tape[1 + k])
        traj.append(z)
    return traj
