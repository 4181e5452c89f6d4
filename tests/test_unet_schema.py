"""CPU: the product modules keep the reference constructor surface and checkpoint schema."""
import json
import os

import pytest
import torch

from _cases import TINY_UNET_CFGS, V1_UNET_CFG
from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _schema(name, sub=None):
    with open(os.path.join(GOLDEN, name)) as f:
        s = json.load(f)
    return s[sub] if sub else s


@pytest.mark.parametrize("name", list(TINY_UNET_CFGS))
def test_tiny_unet_state_dict_schema(name):
    net = CuboidTransformerUNet(**TINY_UNET_CFGS[name])
    ref = _schema("tiny_unet_schema.json", name)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())          # same names, same registration order
    for k, v in sd.items():
        assert list(v.shape) == ref[k], k


def test_v1_unet_state_dict_schema_and_attributes():
    net = CuboidTransformerUNet(**V1_UNET_CFG)
    ref = _schema("v1_unet_schema.json")
    sd = net.state_dict()
    assert len(sd) == 688 and list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert [list(v.shape), str(v.dtype).replace("torch.", "")] == ref[k], k
    assert sum(p.numel() for p in net.parameters()) == 136_800_000 or abs(sum(p.numel() for p in net.parameters()) / 136.8e6 - 1) < 5e-3
    # public attributes the reference exposes
    assert net.block_units == [256, 512]
    assert net.mem_shapes == [(13, 16, 16, 256), (13, 8, 8, 512)]
    assert net.data_shape == (13, 16, 16, 65) and net.in_len == 7 and net.out_len == 6
    assert net.block_cuboid_size[0] == [(13, 1, 1), (1, 16, 1), (1, 1, 16)]
    assert net.block_cuboid_size[1] == [(13, 1, 1), (1, 8, 1), (1, 1, 8)]
    # default init: output layers are zero like the reference
    assert float(net.final_proj.weight.abs().max()) == 0
    assert float(net.down_self_blocks[0][0].attn_l[0].proj.weight.abs().max()) == 0
    assert float(net.down_time_embed_blocks[0].out_layers[3].weight.abs().max()) == 0
    # strict load of a reference-schema checkpoint
    net.load_state_dict({k: torch.zeros(v[0], dtype=getattr(torch, v[1])) for k, v in ref.items()}, strict=True)


def test_checkpoint_level_and_unsupported_options():
    cfg = dict(TINY_UNET_CFGS["axial"])
    CuboidTransformerUNet(**{**cfg, "checkpoint_level": 2})        # accepted and ignored (reference crashes)
    with pytest.raises(NotImplementedError):
        CuboidTransformerUNet(**{**cfg, "num_global_vectors": 4})
    with pytest.raises(ValueError):
        CuboidTransformerUNet(**cfg, precision="fp64")


def test_cpu_forward_fails_loudly():
    from prediff_amd._lib import PrediffHipError
    cfg = TINY_UNET_CFGS["axial"]
    net = CuboidTransformerUNet(**cfg)
    with pytest.raises(PrediffHipError):
        net(torch.zeros(1, *cfg["target_shape"]), torch.zeros(1, dtype=torch.long), torch.zeros(1, *cfg["input_shape"]))


# ------------------------------------------------------------------------------------------------ VAE
def test_vae_state_dict_schema():
    from _cases import TINY_VAE_CFG, V1_VAE_CFG
    from prediff_amd.autoencoder_kl import AutoencoderKL
    vae = AutoencoderKL(**TINY_VAE_CFG)
    ref = _schema("tiny_vae_schema.json")
    sd = vae.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert list(v.shape) == ref[k], k
    vae = AutoencoderKL(**V1_VAE_CFG)
    ref = _schema("v1_vae_schema.json")
    sd = vae.state_dict()
    assert len(sd) == 248 and list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert [list(v.shape), "float32"] == ref[k], k
    from prediff_amd._lib import PrediffHipError
    with pytest.raises(PrediffHipError):
        vae.encode(torch.zeros(1, 1, 128, 128))
