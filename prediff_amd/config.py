"""Config / script front end: reference YAML -> constructor kwargs -> engine objects.

Mirrors what ``PreDiffSEVIRPLModule.__init__`` does with its OmegaConf tree
(reference scripts/prediff/sevirlr/train_sevirlr_prediff.py:72-206): code defaults of ``get_model_config`` (:311-463)
deep-merged with the YAML (``prediff_sevirlr_v1.yaml``), then the explicit keyword mapping into
CuboidTransformerUNet / AutoencoderKL / LatentDiffusion / SEVIRAvgIntensityAlignment.  OmegaConf is not required:
plain PyYAML returns exponent-form scalars without a decimal point ("1e-4") as *strings* (SURVEY.md Q15), so they
are coerced here.  ``evaluate_context`` is the sampling part of ``test_step`` (:905-979): aligned and un-aligned
samples per context, ``.npy`` naming and skill-score update.
"""
import os
import re
from typing import Any, Dict, Optional

import torch

_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")

# the part of get_model_config() the sampling path reads and the YAML may omit (train_sevirlr_prediff.py:316-342)
DIFFUSION_DEFAULTS = dict(timesteps=1000, beta_schedule="linear", use_ema=True, log_every_t=100, clip_denoised=False,
                          linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None, original_elbo_weight=0.,
                          v_posterior=0., l_simple_weight=1., parameterization="eps", learn_logvar=None, logvar_init=0.,
                          cond_stage_model="__is_first_stage__", num_timesteps_cond=None, cond_stage_trainable=False,
                          cond_stage_forward=None, scale_by_std=False, scale_factor=1.0)


def _coerce(v):
    if isinstance(v, str) and _FLOAT_RE.match(v.strip()):
        return float(v)
    if isinstance(v, dict):
        return {k: _coerce(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_coerce(x) for x in v]
    return v


def load_config(path: str) -> Dict[str, Any]:
    import yaml
    with open(path) as f:
        return _coerce(yaml.safe_load(f))


def unet_kwargs(latent_model_cfg: Dict[str, Any]) -> Dict[str, Any]:
    """train_sevirlr_prediff.py:85-137 (note: `self_pattern` -> block_attn_patterns, `down_up_linear_init_mode` feeds both
    down_ and up_linear_init_mode; use_dec_* keys of the YAML are not passed)."""
    c = latent_model_cfg
    n = len(c["depth"])
    pats = [c["self_pattern"]] * n if isinstance(c["self_pattern"], str) else list(c["self_pattern"])
    keys = ["input_shape", "target_shape", "base_units", "scale_alpha", "num_heads", "attn_drop", "proj_drop", "ffn_drop",
            "downsample", "downsample_type", "upsample_type", "upsample_kernel_size", "depth", "num_global_vectors",
            "use_global_vector_ffn", "use_global_self_attn", "separate_global_qkv", "global_dim_ratio", "ffn_activation",
            "gated_ffn", "norm_layer", "padding_type", "checkpoint_level", "pos_embed_type", "use_relative_pos",
            "self_attn_use_final_proj", "attn_linear_init_mode", "ffn_linear_init_mode", "ffn2_linear_init_mode",
            "attn_proj_linear_init_mode", "conv_init_mode", "global_proj_linear_init_mode", "norm_init_mode",
            "time_embed_channels_mult", "time_embed_use_scale_shift_norm", "time_embed_dropout", "unet_res_connect"]
    kw = {k: c[k] for k in keys}
    kw["block_attn_patterns"] = pats
    kw["down_linear_init_mode"] = kw["up_linear_init_mode"] = c["down_up_linear_init_mode"]
    return kw


def vae_kwargs(vae_cfg: Dict[str, Any]) -> Dict[str, Any]:
    """train_sevirlr_prediff.py:139-149"""
    keys = ["down_block_types", "in_channels", "block_out_channels", "act_fn", "latent_channels", "up_block_types",
            "norm_num_groups", "layers_per_block", "out_channels"]
    return {k: vae_cfg[k] for k in keys}


def diffusion_kwargs(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """train_sevirlr_prediff.py:158-188"""
    d = dict(DIFFUSION_DEFAULTS)
    d.update(cfg["model"]["diffusion"])
    d.pop("latent_cond_shape", None)
    d["layout"] = cfg["layout"]["layout"]
    d["loss_type"] = cfg.get("optim", {}).get("loss_type", "l2")
    d["monitor"] = cfg.get("optim", {}).get("monitor", "val/loss")
    d["data_shape"], d["latent_shape"] = tuple(d["data_shape"]), tuple(d["latent_shape"])
    return d


def build_prediff(cfg: Dict[str, Any], precision: str = "bf16", pretrained_dir: Optional[str] = None, device: str = "cuda",
                  unet_ckpt: Optional[str] = None):
    """-> (LatentDiffusion, SEVIRAvgIntensityAlignment or None).  Checkpoints (plain state_dicts, the reference's schema) are
    loaded strictly when `pretrained_dir` holds the files named in the config; otherwise the modules keep their initialisation."""
    from .alignment import SEVIRAvgIntensityAlignment
    from .autoencoder_kl import AutoencoderKL
    from .cuboid_transformer_unet import CuboidTransformerUNet
    from .latent_diffusion import LatentDiffusion
    m = cfg["model"]
    net = CuboidTransformerUNet(**unet_kwargs(m["latent_model"]), precision=precision)
    vae = AutoencoderKL(**vae_kwargs(m["vae"]), precision=precision)

    def ckpt(name):
        """Path of a named checkpoint.  Random initialisation is kept only when no name or no pretrained_dir is given: a name whose
        file is absent raises, as the reference's torch.load does (final_proj is zero-initialised, so a silently skipped denoiser
        checkpoint would give eps = 0 and plausible-looking but meaningless samples)."""
        if pretrained_dir is None or name is None:
            return None
        p = os.path.join(pretrained_dir, name)
        if not os.path.exists(p):
            raise FileNotFoundError(f"checkpoint {name!r} not found in {pretrained_dir!r}")
        return p
    p = ckpt(m["vae"].get("pretrained_ckpt_path"))
    if p:
        vae.load_state_dict(torch.load(p, map_location="cpu"))
    if unet_ckpt and ckpt(unet_ckpt):
        net.load_state_dict(torch.load(ckpt(unet_ckpt), map_location="cpu"))
    ldm = LatentDiffusion(torch_nn_module=net, first_stage_model=vae, **diffusion_kwargs(cfg)).to(device).eval()
    align = None
    a = m.get("align", {})
    if a.get("alignment_type") is not None:
        align = SEVIRAvgIntensityAlignment(alignment_type=a["alignment_type"], guide_scale=a["guide_scale"], model_type=a["model_type"],
                                           model_args=a["model_args"], model_ckpt_path=ckpt(a.get("model_ckpt_path")))
        align.model.to(device)
        ldm.set_alignment(align.get_mean_shift)
    return ldm, align


def split_sequence(seq: torch.Tensor, in_len: int, out_len: int):
    """get_input (train_sevirlr_prediff.py:752-759): (B, T, H, W, C) -> context (first in_len) and target (next out_len) frames."""
    return seq[:, :in_len], seq[:, in_len:in_len + out_len]


@torch.no_grad()
def evaluate_context(ldm, seq: torch.Tensor, cfg: Dict[str, Any], batch_idx: int = 0, rank: int = 0, npy_dir: Optional[str] = None,
                     score=None, aligned_score=None, **sample_kwargs):
    """The sampling part of test_step (train_sevirlr_prediff.py:905-979) for one batch of sequences (B, in_len+out_len, H, W, C)."""
    import numpy as np
    from .alignment import get_alignment_kwargs_avg_x
    lay, ev = cfg["layout"], cfg["eval"]
    ctx, tgt = split_sequence(seq, lay["in_len"], lay["out_len"])
    B = seq.shape[0]
    out = {"pred": [], "aligned_pred": []}
    use_align = ldm.alignment_fn is not None and ev.get("eval_aligned", True)
    for i in range(ev.get("num_samples_per_context", 1)):
        if use_align:
            pred = ldm.sample(cond={"y": ctx}, batch_size=B, use_alignment=True,
                              alignment_kwargs=get_alignment_kwargs_avg_x(context_seq=ctx, target_seq=tgt), **sample_kwargs).contiguous()
            if npy_dir:
                np.save(os.path.join(npy_dir, f"batch{batch_idx}_rank{rank}_sample{i}_aligned.npy"), pred.float().cpu().numpy())
            if aligned_score is not None:
                aligned_score.update(pred.float(), tgt)
            out["aligned_pred"].append(pred)
        if ev.get("eval_unaligned", True):
            pred = ldm.sample(cond={"y": ctx}, batch_size=B, **sample_kwargs).contiguous()
            if npy_dir:
                np.save(os.path.join(npy_dir, f"batch{batch_idx}_rank{rank}_sample{i}.npy"), pred.float().cpu().numpy())
            if score is not None:
                score.update(pred.float(), tgt)
            out["pred"].append(pred)
    return out
