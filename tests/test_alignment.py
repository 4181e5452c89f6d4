"""Knowledge alignment (PyTorch autograd by design): parity of prediff_amd.alignment with the reference on CPU, and of the
aligned sampling step on the GPU engine."""
import json
import os

import numpy as np
import pytest
import torch

import _templates as TP
from _cases import TINY_ALIGN_ARGS, TINY_UNET_CFGS, V1_ALIGN_ARGS
from _weights import seeded_input, seeded_state_dict
from prediff_amd.alignment import NoisyCuboidTransformerEncoder, SEVIRAvgIntensityAlignment, get_alignment_kwargs_avg_x

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _tiny_alignment():
    al = SEVIRAvgIntensityAlignment(alignment_type="avg_x", guide_scale=50.0, model_type="cuboid", model_args=dict(TINY_ALIGN_ARGS))
    ref = json.load(open(os.path.join(GOLDEN, "tiny_align_schema.json")))
    sd = al.model.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert list(v.shape) == ref[k], k
    al.model.load_state_dict(seeded_state_dict(sd, 700), strict=True)
    return al


def test_alignment_network_and_gradient_cpu(golden):
    g = golden("alignment")
    al = _tiny_alignment()
    B = 2
    zt = seeded_input("azt", (B,) + tuple(TINY_ALIGN_ARGS["input_shape"]), 9)
    t = torch.tensor([999, 3])
    avg = torch.as_tensor(g["avg_x_gt"])
    with torch.no_grad():
        u = al.model(zt, t, zc="ignored", y="ignored")          # extra kwargs are swallowed
    assert rel_l2(u, g["u"]) < 1e-5
    with torch.no_grad():                                        # the hook is called under no_grad and re-enables grad itself
        shift = al.get_mean_shift(zt, t, y=None, zc=None, avg_x_gt=avg)
    assert rel_l2(shift, g["shift"]) < 1e-4
    # the L2 norm couples the batch (sevir.py:81-82): a sample's guidance changes when its batch-mates change
    alone = al.get_mean_shift(zt[:1], t[:1], avg_x_gt=avg[:1])
    assert rel_l2(alone, shift[:1]) > 1e-3


def test_v1_alignment_network_cpu(golden):
    net = NoisyCuboidTransformerEncoder(**V1_ALIGN_ARGS)
    ref = json.load(open(os.path.join(GOLDEN, "v1_align_schema.json")))
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and all(list(v.shape) == ref[k] for k, v in sd.items())
    assert abs(sum(p.numel() for p in net.parameters()) / 8.94e6 - 1) < 1e-2
    net.load_state_dict(seeded_state_dict(sd, 701))
    z = seeded_input("v1az", (2, 6, 16, 16, 64), 11)
    with torch.no_grad():
        assert rel_l2(net(z, torch.tensor([400, 20])), golden("v1_alignment")["u"]) < 1e-5


def test_alignment_kwargs():
    tgt = torch.rand(3, 6, 8, 8, 1)
    kw = get_alignment_kwargs_avg_x(target_seq=tgt)
    assert kw["avg_x_gt"].shape == (3, 1) and torch.allclose(kw["avg_x_gt"][:, 0], 2 * tgt.reshape(3, -1).mean(1))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32"])
def test_aligned_sampling_on_gpu(golden, precision):
    """p_sample / sample with use_alignment=True: HIP denoiser + PyTorch-autograd guidance + fused aligned-mean epilogue,
    against the reference's aligned outputs."""
    from prediff_amd.autoencoder_kl import AutoencoderKL
    from prediff_amd.cuboid_transformer_unet import CuboidTransformerUNet
    from prediff_amd.latent_diffusion import LatentDiffusion
    from _cases import TINY_VAE_CFG
    g = golden("alignment")
    cfg = TINY_UNET_CFGS["axial"]
    net = CuboidTransformerUNet(**cfg, precision=precision)
    net.load_state_dict(seeded_state_dict(TP.unet_template(cfg, "tiny_unet_schema.json", "axial"), 600))
    vae = AutoencoderKL(**TINY_VAE_CFG, precision=precision)
    vae.load_state_dict(seeded_state_dict(TP.from_schema("tiny_vae_schema.json"), 601))
    ldm = LatentDiffusion(torch_nn_module=net, layout="NTHWC", data_shape=(2, 32, 32, 1), timesteps=1000, use_ema=False,
                          latent_shape=tuple(cfg["target_shape"]), first_stage_model=vae, cond_stage_model="__is_first_stage__").cuda()
    al = _tiny_alignment()
    al.model.cuda()
    ldm.set_alignment(al.get_mean_shift)
    B = 2
    zt = seeded_input("azt", (B,) + tuple(cfg["target_shape"]), 9).cuda()
    zc = seeded_input("dzc", (B,) + tuple(cfg["input_shape"]), 5).cuda()
    avg = torch.as_tensor(g["avg_x_gt"]).cuda()
    for tt in (500, 0):
        t = torch.full((B,), tt, dtype=torch.long, device="cuda")
        out = ldm.p_sample(zt=zt, zc=zc, t=t, y=None, use_alignment=True, alignment_kwargs={"avg_x_gt": avg},
                           noise=torch.as_tensor(g[f"psample_noise_{tt}"]).cuda())
        assert rel_l2(out, g[f"psample_aligned_{tt}"]) < 1e-4, tt
    y = seeded_input("dy", (B, cfg["input_shape"][0], 32, 32, 1), 8, kind="uniform").cuda()
    lat = ldm.sample(cond={"y": y}, batch_size=B, timesteps=3, use_alignment=True, alignment_kwargs={"avg_x_gt": avg},
                     return_decoded=False, noise_tape=torch.as_tensor(g["tape"]))
    e = rel_l2(lat, g["sample_aligned_latent"])
    print(f"[aligned sample3] latent rel-L2 vs reference {e:.3e}")
    assert e < 1e-3
    # the loop above ran the denoiser as HIP graphs on lane streams concurrently with the autograd guidance; the plain eager
    # path (no graphs, one stream) must give the same latents bit for bit, whatever the number of lanes
    outs = []
    for streams, graph in ((2, True), (1, True), (2, False)):
        ldm.aligned_lanes, ldm.use_hip_graph = streams, graph
        outs.append(ldm.sample(cond={"y": y}, batch_size=B, timesteps=3, use_alignment=True, alignment_kwargs={"avg_x_gt": avg},
                               return_decoded=False, noise_tape=torch.as_tensor(g["tape"])))
    assert torch.equal(outs[0], lat) and torch.equal(outs[1], lat) and torch.equal(outs[2], lat)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,shape", [(65, 128, (6, 16, 16)), (128, 128, (6, 16, 16)), (256, 256, (6, 8, 8)), (32, 48, (3, 5, 7))])
def test_hip_conv3d_autograd_function(cin, cout, shape):
    """The guidance network's 3x3x3 convolutions run on pd_igemm inside autograd (hi/lo-split: fp32-class accuracy): forward and
    the data gradient (the only gradient the guidance needs) against PyTorch's fp32 Conv3d."""
    from prediff_amd import alignment as AL
    g = torch.Generator().manual_seed(cin + cout)
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (27 * cin) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    conv = conv.cuda().requires_grad_(False)
    x = torch.randn((2, cin) + shape, generator=g).cuda()
    gout = torch.randn((2, cout) + shape, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    ya = AL._conv3d(conv, xa)
    (ga,) = torch.autograd.grad(ya, xa, gout)
    xb = x.clone().requires_grad_(True)
    yb = conv(xb)
    (gb,) = torch.autograd.grad(yb, xb, gout)
    assert ya.shape == yb.shape and rel_l2(ya.detach(), yb.detach()) < 3e-5 and rel_l2(ga, gb) < 3e-5
    # the frozen weights are packed once per weight version
    assert conv._hip_packs[0][2] == conv.weight._version


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dim,heads,cuboid,shift,strategy,padding", [
    ((6, 16, 16), 128, 4, (6, 1, 1), (0, 0, 0), ("l", "l", "l"), "zeros"),
    ((6, 16, 16), 128, 4, (1, 16, 1), (0, 0, 0), ("l", "l", "l"), "zeros"),
    ((6, 8, 8), 256, 4, (1, 1, 8), (0, 0, 0), ("l", "l", "l"), "zeros"),
    ((5, 7, 6), 64, 2, (2, 4, 4), (1, 2, 2), ("l", "l", "l"), "zeros"),       # padded, shifted, masked
    ((4, 8, 8), 64, 2, (2, 4, 4), (0, 0, 0), ("d", "d", "d"), "ignore"),      # dilated
    ((3, 6, 5), 32, 1, (3, 4, 4), (0, 2, 2), ("l", "l", "l"), "ignore"),      # padding masked out ("ignore")
])
def test_hip_cuboid_attention_autograd_function(shape, dim, heads, cuboid, shift, strategy, padding):
    """The guidance network's cuboid attention on pd_cuboid_attention / pd_cuboid_attention_bwd inside autograd, against the
    PyTorch statement of the same layer: output and the data gradient."""
    from prediff_amd import alignment as AL
    from prediff_amd.cuboid_geometry import attention_tables
    from prediff_amd.cuboid_transformer_unet import CuboidSelfAttentionLayer
    torch.manual_seed(dim + heads + sum(cuboid))
    at = CuboidSelfAttentionLayer(dim, heads, cuboid_size=cuboid, shift_size=shift, strategy=strategy, padding_type=padding)
    with torch.no_grad():
        for p in at.parameters():
            p.copy_(torch.randn(p.shape) * (0.5 if p.ndim == 2 and p.shape[-1] == heads else 1.0 / p.shape[-1] ** 0.5 if p.ndim == 2 else 1.0))
        at.norm.weight.add_(1.0)
    at = at.cuda().requires_grad_(False)
    x = torch.randn((2,) + shape + (dim,)).cuda()
    gout = torch.randn((2,) + shape + (dim,)).cuda()
    res = []
    for use in (True, False):
        AL.USE_HIP_ATTN = use
        try:
            tables = attention_tables(shape, cuboid, shift, strategy, padding)
            xa = x.clone().requires_grad_(True)
            y = AL.attention_forward(at, xa, tables)
            (g,) = torch.autograd.grad(y, xa, gout)
            res.append((y.detach(), g))
        finally:
            AL.USE_HIP_ATTN = True
    ey, eg = rel_l2(res[0][0], res[1][0]), rel_l2(res[0][1], res[1][1])
    print(f"[hip cuboid attention autograd {shape} {cuboid}] out {ey:.2e} grad {eg:.2e}")
    assert ey < 2e-5 and eg < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,emb,shape", [(64, 128, False, (6, 16, 16)), (128, 128, True, (6, 16, 16)), (256, 256, True, (6, 8, 8)),
                                                (64, 64, True, (3, 5, 7))])
def test_hip_resblock_autograd_nodes(cin, cout, emb, shape):
    """TimeEmbedResBlock of the guidance network as two row-layout GroupNorm -> SiLU -> Conv3d nodes (pd_groupnorm_silu + pd_igemm,
    backward pd_igemm dgrad + pd_groupnorm_silu_bwd) against the PyTorch statement: output and data gradient."""
    from prediff_amd import alignment as AL
    from prediff_amd.cuboid_transformer_unet import TimeEmbedResBlock
    torch.manual_seed(cin + cout)
    m = TimeEmbedResBlock(channels=cin, emb_channels=96 if emb else None, dropout=0.0, out_channels=cout, use_embed=emb, dims=3)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape) * (1.0 / (27 * p.shape[1]) ** 0.5 if p.ndim == 5 else 0.3))
        for gn in (m.in_layers[0], m.out_layers[0]):
            gn.weight.add_(1.0)
    m = m.cuda().requires_grad_(False)
    x = (torch.randn((2,) + shape + (cin,)) * 1.5 + 0.3).cuda()
    e = torch.randn(2, 96).cuda() if emb else None
    gout = torch.randn((2,) + shape + (cout,)).cuda()
    res = []
    for use in (True, False):
        AL.USE_HIP_RESBLOCK = use
        try:
            xa = x.clone().requires_grad_(True)
            y = AL.resblock_forward(m, xa, e)
            (g,) = torch.autograd.grad(y, xa, gout)
            res.append((y.detach(), g))
        finally:
            AL.USE_HIP_RESBLOCK = True
    ey, eg = rel_l2(res[0][0], res[1][0]), rel_l2(res[0][1], res[1][1])
    print(f"[hip resblock autograd {cin}->{cout} {shape}] out {ey:.2e} grad {eg:.2e}")
    assert ey < 3e-5 and eg < 3e-5
