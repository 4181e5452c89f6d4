"""Oracle: frame-wise KL-VAE (diffusers-0.13 lineage), functional restatement (CPU fp32).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Keys follow the reference checkpoint
schema.  Citations are relative to /root/reference/src/prediff/taming/.
"""
import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS = 1e-6   # vae.py:42,54,64,146 ; resnet.py:379 -- GroupNorm eps of every VAE norm


def resnet_block_2d(sd, p: str, x: Tensor, groups: int) -> Tensor:
    """ResnetBlock2D.forward with temb=None, output_scale_factor=1.  resnet.py:454-495."""
    h = F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], EPS)
    h = F.conv2d(F.silu(h), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], EPS)
    h = F.conv2d(F.silu(h), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def attention_block(sd, p: str, x: Tensor, groups: int) -> Tensor:
    """AttentionBlock.forward, single head, rescale 1.  attention.py:136-189."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], EPS)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(h, sd[p + "key.weight"], sd[p + "key.bias"])
    v = F.linear(h, sd[p + "value.weight"], sd[p + "value.bias"])
    score = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(C))        # baddbmm(beta=0, alpha=scale), :163-174
    h = torch.softmax(score.float(), dim=-1) @ v
    h = F.linear(h, sd[p + "proj_attn.weight"], sd[p + "proj_attn.bias"])
    return h.transpose(-1, -2).reshape(B, C, H, W) + x


def mid_block(sd, p: str, x: Tensor, groups: int) -> Tensor:
    """UNetMidBlock2D.forward.  unet_2d_blocks.py:158-165."""
    x = resnet_block_2d(sd, p + "resnets.0.", x, groups)
    x = attention_block(sd, p + "attentions.0.", x, groups)
    return resnet_block_2d(sd, p + "resnets.1.", x, groups)


def encoder(sd, x: Tensor, n_blocks: int, layers_per_block: int, groups: int) -> Tensor:
    """Encoder.forward.  vae.py:70-86; DownEncoderBlock2D unet_2d_blocks.py:217-225; Downsample2D resnet.py:181-190."""
    p = "encoder."
    h = F.conv2d(x, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1)
    for b in range(n_blocks):
        for r in range(layers_per_block):
            h = resnet_block_2d(sd, f"{p}down_blocks.{b}.resnets.{r}.", h, groups)
        if b < n_blocks - 1:
            h = F.pad(h, (0, 1, 0, 1))                      # asymmetric right/bottom zero pad, padding=0
            h = F.conv2d(h, sd[f"{p}down_blocks.{b}.downsamplers.0.conv.weight"],
                         sd[f"{p}down_blocks.{b}.downsamplers.0.conv.bias"], stride=2)
    h = mid_block(sd, p + "mid_block.", h, groups)
    h = F.silu(F.group_norm(h, groups, sd[p + "conv_norm_out.weight"], sd[p + "conv_norm_out.bias"], EPS))
    return F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)


def decoder(sd, z: Tensor, n_blocks: int, layers_per_block: int, groups: int) -> Tensor:
    """Decoder.forward.  vae.py:150-166; UpDecoderBlock2D unet_2d_blocks.py:271-279; Upsample2D resnet.py:108-143."""
    p = "decoder."
    h = F.conv2d(z, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1)
    h = mid_block(sd, p + "mid_block.", h, groups)
    for b in range(n_blocks):
        for r in range(layers_per_block + 1):
            h = resnet_block_2d(sd, f"{p}up_blocks.{b}.resnets.{r}.", h, groups)
        if b < n_blocks - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{p}up_blocks.{b}.upsamplers.0.conv.weight"],
                         sd[f"{p}up_blocks.{b}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + "conv_norm_out.weight"], sd[p + "conv_norm_out.bias"], EPS))
    return F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)


def vae_encode_moments(sd, cfg: dict, x: Tensor) -> Tensor:
    """AutoencoderKL.encode up to the moments tensor.  autoencoder_kl.py:80-84."""
    nb = len(cfg["block_out_channels"])
    h = encoder(sd, x, nb, cfg.get("layers_per_block", 1), cfg.get("norm_num_groups", 32))
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def gaussian_mode_and_logvar(moments: Tensor):
    """DiagonalGaussianDistribution.__init__/.mode.  utils/distributions.py:27-35,70-71."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean, torch.clamp(logvar, -30.0, 20.0)


def vae_encode_mode(sd, cfg: dict, x: Tensor) -> Tensor:
    return gaussian_mode_and_logvar(vae_encode_moments(sd, cfg, x))[0]


def vae_decode(sd, cfg: dict, z: Tensor) -> Tensor:
    """AutoencoderKL.decode.  autoencoder_kl.py:86-113."""
    nb = len(cfg["block_out_channels"])
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder(sd, z, nb, cfg.get("layers_per_block", 1), cfg.get("norm_num_groups", 32))
