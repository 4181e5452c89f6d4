"""Host-side logic of the e4m3 operand path (no GPU): weight packing, scales, saturating cast, constructor flags."""
import math

import pytest
import torch

from prediff_amd.packing import FP8_MAX, fp8_weight_scale, pack_conv_fp8, pack_linear_fp8, pad128, to_fp8


def test_fp8_weight_scale_is_a_power_of_two_below_the_range_top():
    g = torch.Generator().manual_seed(5)
    for amp in (1e-3, 0.07, 1.0, 30.0, 500.0):
        w = torch.randn(64, 96, generator=g) * amp
        s = fp8_weight_scale(w)
        assert s > 0 and math.log2(s) == int(math.log2(s))                      # exact in fp32, folded into alpha
        top = float(w.abs().max()) * s
        assert FP8_MAX / 2 < top <= FP8_MAX                                      # the largest weight lands in the top binade
    assert fp8_weight_scale(torch.zeros(4, 4)) == 1.0


def test_to_fp8_saturates_and_rounds_to_nearest():
    x = torch.tensor([0.0, 1.0, -1.0, 1.0625, 447.0, 449.0, 1e6, -1e6, 2.0 ** -9, 2.0 ** -11])
    q = to_fp8(x, 1.0).float()
    assert q.tolist()[:3] == [0.0, 1.0, -1.0]
    assert q[3] in (1.0, 1.125)                                                 # 3 mantissa bits: spacing 0.125 in [1, 2)
    assert q[4] == 448.0 and q[5] == 448.0 and q[6] == 448.0 and q[7] == -448.0  # saturating, never inf / nan
    assert q[8] == 2.0 ** -9 and q[9] == 0.0                                    # smallest subnormal, then flush to zero
    assert torch.isfinite(q).all()
    # the scale is applied before the cast
    assert float(to_fp8(torch.tensor([3.0]), 16.0).float()) == 48.0


def test_pack_conv_and_linear_fp8_layout():
    g = torch.Generator().manual_seed(9)
    w = torch.randn(40, 200, 3, 3, 3, generator=g) * 0.05
    w8, s = pack_conv_fp8(w)
    assert w8.dtype == torch.float8_e4m3fn and w8.shape == (27, 40, pad128(200)) and w8.is_contiguous()
    # taps enumerated kernel-index-major (kt, kh, kw), K contiguous, zero padded
    ref = w.reshape(40, 200, 27).permute(2, 0, 1)
    got = w8.float()[:, :, :200] / s
    assert float((got - ref).abs().max()) <= float(ref.abs().max()) * 2 ** -4   # e4m3 rounding: half a step of the top binade at most
    assert float(w8.float()[:, :, 200:].abs().max()) == 0.0
    lw = torch.randn(96, 512, generator=g) * 0.03
    l8, ls = pack_linear_fp8(lw)
    assert l8.shape == (96, 512) and float(((l8.float() / ls) - lw).abs().max()) <= float(lw.abs().max()) * 2 ** -4
    assert pad128(1) == 128 and pad128(128) == 128 and pad128(129) == 256


def test_precision_fp8_constructor_flags():
    from _cases import TINY_UNET_CFGS, TINY_VAE_CFG
    from prediff_amd.autoencoder_kl import AutoencoderKL
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    net = CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="fp8")
    assert net.fp8_conv and net.fp8_linear and net.precision == "bf16"           # the bf16 engine with e4m3 convolution + long-K linear operands
    net_c = CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="fp8_conv")
    assert net_c.fp8_conv and not net_c.fp8_linear and net_c.precision == "bf16"   # e4m3 for the convolutions only
    assert not CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="bf16").fp8_conv
    assert AutoencoderKL(**TINY_VAE_CFG, precision="fp8").precision == "bf16"    # the VAE has no fp8 launches
    # precision="fp16": the bf16 engine's launches with IEEE half as the 16-bit operand type (per-module call options, no fp8)
    net_h = CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="fp16")
    assert net_h.precision == "bf16" and net_h.operand == "fp16" and net_h.op_dtype == torch.float16 and net_h.opts.operand == 1
    assert not net_h.fp8_conv and net.opts.operand == 0 and net.op_dtype == torch.bfloat16
    assert AutoencoderKL(**TINY_VAE_CFG, precision="fp16").op_dtype == torch.float16
    with pytest.raises(ValueError):
        CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision="fp64")
