"""Diffusion schedules (host side, float64 -> fp32 tables), same functions as the reference's
``prediff.diffusion.utils`` (make_beta_schedule :17-39, make_ddim_timesteps :42-56,
make_ddim_sampling_parameters :59-70) and ``LatentDiffusion.register_schedule`` (latent_diffusion.py:228-268)."""
from typing import Dict

import numpy as np


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3) -> np.ndarray:
    # config files written for OmegaConf carry "1e-4"-style scalars that plain YAML loaders return as str
    linear_start, linear_end, cosine_s = float(linear_start), float(linear_end), float(cosine_s)
    if schedule == "linear":
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    elif schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        alphas = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        alphas = alphas / alphas[0]
        betas = np.clip(1 - alphas[1:] / alphas[:-1], a_min=0, a_max=0.999)
    elif schedule == "sqrt_linear":
        betas = np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    elif schedule == "sqrt":
        betas = np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas


def schedule_tables(betas: np.ndarray, v_posterior: float = 0.0) -> Dict[str, np.ndarray]:
    """The 12 per-timestep buffers LatentDiffusion registers (float64 math, cast to fp32 last)."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    t = {
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
    return {k: v.astype(np.float32) for k, v in t.items()}


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=False) -> np.ndarray:
    """The DDPM indices a DDIM run visits, each shifted up by one (so that the last one reaches the final alpha).  `verbose` is
    accepted for signature compatibility and ignored."""
    T, S = int(num_ddpm_timesteps), int(num_ddim_timesteps)
    if ddim_discr_method == "uniform":
        picked = np.arange(0, T, T // S)                          # every (T // S)-th training step
    elif ddim_discr_method == "quad":
        picked = np.square(np.linspace(0.0, np.sqrt(0.8 * T), S)).astype(int)   # quadratic spacing over the first 80 %
    else:
        raise NotImplementedError(f"unknown ddim discretization method '{ddim_discr_method}' (uniform | quad)")
    return picked + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=False):
    """(sigma_t, alpha_bar_t, alpha_bar_{t-1}) on the DDIM grid; the step before the first grid point is training step 0."""
    ac = np.asarray(alphacums)
    steps = np.asarray(ddim_timesteps)
    a_t = ac[steps]
    a_before = np.concatenate([ac[:1], ac[steps[:-1]]])
    sigma = eta * np.sqrt((1 - a_before) / (1 - a_t) * (1 - a_t / a_before))
    return sigma, a_t, a_before
