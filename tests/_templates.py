"""Key -> zero tensor templates (checkpoint schema) for single layers and whole nets.

Used with tests/_weights.seeded_state_dict; integer buffers are filled by the oracle's own
`relative_position_index` (pinned against the reference's buffers in test_oracle_golden.py).
"""
import json
import os

import torch

from oracle import unet as OU

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def z(*shape):
    return torch.zeros(*shape)


def attn_layer(dim, heads, cuboid, prefix=""):
    bt, bh, bw = cuboid
    return {
        prefix + "relative_position_bias_table": z((2 * bt - 1) * (2 * bh - 1) * (2 * bw - 1), heads),
        prefix + "relative_position_index": OU.relative_position_index(cuboid),
        prefix + "qkv.weight": z(3 * dim, dim),
        prefix + "proj.weight": z(dim, dim),
        prefix + "proj.bias": z(dim),
        prefix + "norm.weight": z(dim),
        prefix + "norm.bias": z(dim),
    }


def ffn(units, hidden, gated=False):
    t = {"ffn_1.weight": z(hidden, units), "ffn_1.bias": z(hidden)}
    if gated:
        t.update({"ffn_1_gate.weight": z(hidden, units), "ffn_1_gate.bias": z(hidden)})
    t.update({"ffn_2.weight": z(units, hidden), "ffn_2.bias": z(units),
              "layer_norm.weight": z(units), "layer_norm.bias": z(units)})
    return t


def patch_merge(dim, out_dim, ds=(1, 2, 2)):
    k = ds[0] * ds[1] * ds[2] * dim
    return {"reduction.weight": z(out_dim, k), "norm.weight": z(k), "norm.bias": z(k)}


def upsample3d(dim, out_dim, k=3):
    return {"conv.weight": z(out_dim, dim, k, k), "conv.bias": z(out_dim)}


def pos_embed(dim, T, H, W):
    return {"T_embed.weight": z(T, dim), "H_embed.weight": z(H, dim), "W_embed.weight": z(W, dim)}


def time_embed_layer(base, ch):
    return {"layer.0.weight": z(ch, base), "layer.0.bias": z(ch), "layer.2.weight": z(ch, ch), "layer.2.bias": z(ch)}


def resblock3d(cin, cout, emb, ssn):
    t = {"in_layers.0.weight": z(cin), "in_layers.0.bias": z(cin),
         "in_layers.2.weight": z(cout, cin, 3, 3, 3), "in_layers.2.bias": z(cout)}
    if emb is not None:
        n = 2 * cout if ssn else cout
        t.update({"emb_layers.1.weight": z(n, emb), "emb_layers.1.bias": z(n)})
    t.update({"out_layers.0.weight": z(cout), "out_layers.0.bias": z(cout),
              "out_layers.3.weight": z(cout, cout, 3, 3, 3), "out_layers.3.bias": z(cout)})
    if cin != cout:
        t.update({"skip_connection.weight": z(cout, cin, 1, 1, 1), "skip_connection.bias": z(cout)})
    return t


def resnet2d(cin, cout):
    t = {"norm1.weight": z(cin), "norm1.bias": z(cin), "conv1.weight": z(cout, cin, 3, 3), "conv1.bias": z(cout),
         "norm2.weight": z(cout), "norm2.bias": z(cout), "conv2.weight": z(cout, cout, 3, 3), "conv2.bias": z(cout)}
    if cin != cout:
        t.update({"conv_shortcut.weight": z(cout, cin, 1, 1), "conv_shortcut.bias": z(cout)})
    return t


def conv2d(cin, cout, k=3, prefix="conv."):
    return {prefix + "weight": z(cout, cin, k, k), prefix + "bias": z(cout)}


def vae_attention(c):
    t = {"group_norm.weight": z(c), "group_norm.bias": z(c)}
    for n in ("query", "key", "value", "proj_attn"):
        t.update({n + ".weight": z(c, c), n + ".bias": z(c)})
    return t


def from_schema(json_name, sub=None):
    with open(os.path.join(GOLDEN, json_name)) as f:
        schema = json.load(f)
    if sub is not None:
        schema = schema[sub]
    out = {}
    for k, v in schema.items():
        shape, dtype = (v if (len(v) == 2 and isinstance(v[1], str)) else (v, "float32"))
        out[k] = torch.zeros(shape, dtype=torch.int64 if "int" in dtype or k.endswith("relative_position_index")
                             else torch.float32)
    return out


def unet_template(cfg, json_name, sub=None):
    """Schema from the committed JSON; relative_position_index buffers from the oracle geometry."""
    t = from_schema(json_name, sub)
    g = OU.unet_geometry(cfg)
    for k in list(t):
        if k.endswith("relative_position_index"):
            parts = k.split(".")          # {down,up}_self_blocks.{i}.{d}.attn_l.{a}.relative_position_index
            i, a = int(parts[1]), int(parts[4])
            t[k] = OU.relative_position_index(g["sizes"][i][a])
            assert list(t[k].shape) == list(from_schema(json_name, sub)[k].shape), k
    return t
