"""Module-level GPU tests of the denoiser's building blocks against the goldens recorded from the imported reference
(tests/golden/resblock3d.npz, small_layers.npz; generator tests/golden/gen_golden.py): TimeEmbedResBlock,
PositionwiseFFN (a9), PatchMerging3D (a10), Upsample3DLayer (a11), each run through the engine's own method on a stand-alone layer
(the whole-UNet tests reach them only through 100+ other launches).  fp32 engine <= 1e-5 (hi/lo split: 3e-5 where two GEMMs chain),
bf16 engine <= 1e-2."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _templates as TP  # noqa: E402
from _cases import RESBLOCK3D_CASES, TINY_UNET_CFGS  # noqa: E402
from _weights import seeded_input, seeded_state_dict  # noqa: E402
from prediff_amd import _lib as L  # noqa: E402
from prediff_amd.cuboid_transformer_unet import (CuboidTransformerUNet, PatchMerging3D, PositionwiseFFN, TimeEmbedResBlock,  # noqa: E402
                                                 Upsample3DLayer)

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"fp32": 3e-5, "bf16": 1e-2}


def rel_l2(a, b):
    a, b = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / b.norm())


def _host(precision):
    """A denoiser instance only as the owner of the engine's methods, workspace and packing functions."""
    return CuboidTransformerUNet(**TINY_UNET_CFGS["axial"], precision=precision).cuda()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("i", range(len(RESBLOCK3D_CASES)))
def test_time_embed_resblock_vs_reference_golden(golden, i, precision):
    c = RESBLOCK3D_CASES[i]
    m = TimeEmbedResBlock(channels=c["cin"], dropout=0.0, emb_channels=c["emb"], out_channels=c["cout"], use_embed=c["emb"] is not None,
                          use_scale_shift_norm=c["ssn"])
    m.load_state_dict(seeded_state_dict(TP.resblock3d(c["cin"], c["cout"], c["emb"], c["ssn"]), 300 + i), strict=True)
    m = m.cuda()
    net = _host(precision)
    P = {}
    net._packers(P, torch.device(DEV, 0))["resblock"]("rb", m)
    B, (T, H, W) = 2, c["shape"]
    x = seeded_input(f"rb{i}", (B, c["cin"]) + tuple(c["shape"]), 1).permute(0, 2, 3, 4, 1).contiguous()
    xr = x.reshape(B * T * H * W, c["cin"]).cuda().contiguous()
    emb = None
    if c["emb"] is not None:
        e_in = seeded_input(f"rbe{i}", (B, c["emb"]), 1).cuda()
        n = m.emb_layers[1].out_features
        emb = torch.empty(B, n, device=DEV)
        L.linear_small(e_in, P["rb.emb.w"], P["rb.emb.b"], emb, B, c["emb"], n, act_in="silu")      # emb_layers = SiLU -> Linear
    out = torch.empty(B * T * H * W, c["cout"], device=DEV)
    with torch.cuda.device(0):
        net._resblock(P, "rb", m, xr, B, (T, H, W), emb, torch.device(DEV, 0), out=out)
    torch.cuda.synchronize()
    e = rel_l2(out.reshape(B, T, H, W, c["cout"]), golden("resblock3d")[f"y_{i}"])
    print(f"[resblock3d case {i} {precision}] rel-L2 vs reference golden {e:.3e}")
    assert e < TOL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_small_layers_vs_reference_golden(golden, precision):
    g = golden("small_layers")
    dev = torch.device(DEV, 0)
    net = _host(precision)
    # PositionwiseFFN: gelu, leaky, gated gelu (units 32: the fused kernel does not apply; this is the LayerNorm + pd_igemm chain)
    for i, (act, gated) in enumerate([("gelu", False), ("leaky", False), ("gelu", True)]):
        ff = PositionwiseFFN(units=32, hidden_size=128, activation=act, gated_proj=gated)
        ff.load_state_dict(seeded_state_dict(TP.ffn(32, 128, gated), 200 + i), strict=True)
        ff = ff.cuda()
        P = {}
        pk = net._packers(P, dev)
        pk["norm"]("f.ln", ff.layer_norm); pk["lin"]("f.fc1", ff.ffn_1); pk["lin"]("f.fc2", ff.ffn_2)
        if gated:
            pk["lin"]("f.gate", ff.ffn_1_gate)
        x = seeded_input(f"ffn{i}", (2, 3, 4, 4, 32), 1).reshape(96, 32).cuda().contiguous()
        net._ffn(P, "f", ff, x, 2, 48, 32, dev)
        torch.cuda.synchronize()
        e = rel_l2(x.reshape(2, 3, 4, 4, 32), g[f"ffn_{i}"])
        print(f"[ffn {act} gated={gated} {precision}] {e:.3e}")
        assert e < TOL[precision]
    # PatchMerging3D: divisible, zero-padded and nearest-padded (models/utils.py:228-256) shapes
    for i, (shape, ptype) in enumerate([((3, 8, 8), "zeros"), ((3, 7, 6), "zeros"), ((3, 7, 6), "nearest")]):
        pm = PatchMerging3D(dim=16, out_dim=32, downsample=(1, 2, 2), padding_type=ptype)
        pm.load_state_dict(seeded_state_dict(TP.patch_merge(16, 32), 210 + i), strict=True)
        pm = pm.cuda()
        P = {}
        pk = net._packers(P, dev)
        pk["norm"]("d.ln", pm.norm); pk["lin"]("d.red", pm.reduction)
        T, H, W = shape
        x = seeded_input(f"pm{i}", (2,) + shape + (16,), 1).reshape(2 * T * H * W, 16).cuda().contiguous()
        So = T * ((H + 1) // 2) * ((W + 1) // 2)
        out = torch.empty(2 * So, 32, device=DEV)
        net._patch_merge(P, "d", pm, x, 2, shape, 16, 32, (1, 2, 2), out, dev)
        torch.cuda.synchronize()
        e = rel_l2(out.reshape(g[f"pm_{i}"].shape), g[f"pm_{i}"])
        print(f"[patch merging {shape} {ptype} {precision}] {e:.3e}")
        assert e < TOL[precision]
    # Upsample3DLayer: nearest x2 + Conv2d 3x3
    up = Upsample3DLayer(dim=32, out_dim=16, target_size=(3, 8, 8))
    up.load_state_dict(seeded_state_dict(TP.upsample3d(32, 16), 220), strict=True)
    up = up.cuda()
    P = {}
    net._packers(P, dev)["conv"]("u.conv", up.conv)
    x = seeded_input("up0", (2, 3, 4, 4, 32), 1).reshape(96, 32).cuda().contiguous()
    out = torch.empty(2 * 3 * 8 * 8, 16, device=DEV)
    net._upsample(P, "u", x, 2, (3, 4, 4), 32, (8, 8), 16, 3, None, out, dev)
    torch.cuda.synchronize()
    e = rel_l2(out.reshape(2, 3, 8, 8, 16), g["up_0"])
    print(f"[upsample3d {precision}] {e:.3e}")
    assert e < TOL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [3, 6, 7])
def test_attention_layer_padded_vs_reference_golden(golden, case, precision):
    """CuboidSelfAttentionLayer on shapes the cuboid does not divide, against the reference goldens (tests/golden/attn_layer.npz):
    case 3 zero padding, case 6 'ignore' padding with a shifted window, case 7 padding_type="nearest" (models/utils.py:228-270: the padded
    grid is a nearest-neighbour resize of the tokens and the un-padding resizes back, so the token a slot reads and the token that receives
    its result differ -- pd_cuboid_attn_args.tok_out)."""
    from _cases import ATTN_CASES
    from prediff_amd.cuboid_geometry import attention_tables, relative_position_bias
    from prediff_amd.cuboid_transformer_unet import CuboidSelfAttentionLayer
    c = ATTN_CASES[case]
    Cn, heads, shape, cuboid = c["dim"], c["heads"], tuple(c["shape"]), tuple(c["cuboid"])
    at = CuboidSelfAttentionLayer(dim=Cn, num_heads=heads, cuboid_size=cuboid, shift_size=c["shift"], strategy=c["strategy"],
                                  padding_type=c["padding_type"])
    sd = seeded_state_dict(TP.attn_layer(Cn, heads, cuboid), 100 + case)
    at.load_state_dict(sd, strict=True)
    at = at.cuda()
    dev = torch.device(DEV, 0)
    net = _host(precision)
    P = {}
    pk = net._packers(P, dev)
    pk["norm"]("a.ln", at.norm); pk["lin"]("a.qkv", at.qkv); pk["lin"]("a.proj", at.proj)
    geo = attention_tables(shape, cuboid, c["shift"], c["strategy"], c["padding_type"])
    assert (geo["tok_out"] is not None) == (c["padding_type"] == "nearest")
    P["a.bias"] = relative_position_bias(at.relative_position_bias_table, at.relative_position_index.cpu(), geo["vol"]).to(dev)
    tabs = dict(tok=geo["tok_index"].to(dev), mask=geo["mask"].to(dev) if geo["mask"] is not None else None,
                tok_out=geo["tok_out"].to(dev) if geo["tok_out"] is not None else None)
    B, S = c["B"], shape[0] * shape[1] * shape[2]
    x = seeded_input(f"attn{case}", (B,) + shape + (Cn,), 1)
    xd = x.reshape(B * S, Cn).cuda().contiguous()
    x0 = xd.clone()
    net._attention(P, "a", at, xd, B, S, Cn, tabs, geo, dev)
    torch.cuda.synchronize()
    y = (xd - x0).reshape(B, *shape, Cn)
    e = rel_l2(y, golden("attn_layer")[f"y_{case}"])
    print(f"[attention layer case {case} ({c['padding_type']}) {precision}] rel-L2 vs reference golden {e:.3e}")
    assert e < (1e-4 if precision == "fp32" else 1e-2)
