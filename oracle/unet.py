"""Oracle: Earthformer-UNet denoiser, functional restatement (CPU, fp32, torch ops).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Operates on a plain ``state_dict``
whose keys follow the reference checkpoint schema plus a config
dict holding the reference constructor kwargs.  Every function cites the reference
lines it restates (paths relative to /root/reference/src/prediff/).

The cuboid decomposition is written as explicit index arithmetic (gather / scatter
over flat token ids) rather than the reference's reshape+permute chain, so that it is
an independent statement of the same mapping.
"""
import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# attention patterns: models/cuboid_transformer/cuboid_transformer_patterns.py:11-118
# --------------------------------------------------------------------------------------
def attention_pattern(name: str, shape: Sequence[int]):
    """name -> (cuboid_sizes, strategies, shift_sizes) for a (T, H, W, C) memory shape."""
    T, H, W = shape[0], shape[1], shape[2]
    lll, ddd, z = ("l", "l", "l"), ("d", "d", "d"), (0, 0, 0)
    if name == "full":                                   # patterns.py:11-16
        return [(T, H, W)], [lll], [z]
    if name == "axial":                                  # patterns.py:19-37
        return [(T, 1, 1), (1, H, 1), (1, 1, W)], [lll] * 3, [z] * 3
    if name == "divided_st":                             # patterns.py:53-58
        return [(T, 1, 1), (1, H, W)], [lll] * 2, [z] * 2
    if name == "video_swin" or name.startswith("video_swin_"):   # patterns.py:40-50,66-72
        P, M = 2, 4
        if name != "video_swin":
            p, m = name[len("video_swin_"):].split("x")
            P, M = int(p), int(m)
        P, M = min(P, T), min(M, H, W)
        return [(P, M, M), (P, M, M)], [lll] * 2, [z, (P // 2, M // 2, M // 2)]
    if name == "spatial_lg_v1" or name.startswith("spatial_lg_"):  # patterns.py:76-97
        M = 4 if name == "spatial_lg_v1" else int(name[len("spatial_lg_"):])
        if H <= M and W <= M:
            return [(T, 1, 1), (1, H, W)], [lll] * 2, [z] * 2
        return [(T, 1, 1), (1, M, M), (1, M, M)], [lll, lll, ddd], [z] * 3
    if name.startswith("axial_space_dilate_"):           # patterns.py:100-118
        K = min(int(name[len("axial_space_dilate_"):]), H, W)
        return ([(T, 1, 1), (1, H // K, 1), (1, H // K, 1), (1, 1, W // K), (1, 1, W // K)],
                [lll, ddd, lll, ddd, lll], [z] * 5)
    raise KeyError(name)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def round_to(dat, c):                                    # models/utils.py:143
    return dat + (dat - dat % c) % c


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """models/utils.py:68-88 -- [cos | sin] order, t cast to float."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _activation(name: str):                              # models/utils.py:147-189
    return {
        "leaky": lambda v: F.leaky_relu(v, 0.1),
        "gelu": lambda v: F.gelu(v),
        "relu": F.relu,
        "elu": F.elu,
        "sigmoid": torch.sigmoid,
        "tanh": torch.tanh,
        "identity": lambda v: v,
    }[name]


def _gn_groups(channels: int, norm_groups: int = 32) -> int:   # models/time_embed.py:90-91
    return norm_groups if channels % norm_groups == 0 else channels


def _pad_thw(x: Tensor, pad: Tuple[int, int, int], padding_type: str) -> Tensor:
    """models/utils.py:228-256 (pad at the end of T, H, W)."""
    pt, ph, pw = pad
    if pt == 0 and ph == 0 and pw == 0:
        return x
    B, T, H, W, C = x.shape
    if padding_type == "nearest":
        return F.interpolate(x.permute(0, 4, 1, 2, 3), size=(T + pt, H + ph, W + pw)).permute(0, 2, 3, 4, 1)
    return F.pad(x, (0, 0, 0, pw, 0, ph, 0, pt))


def _unpad_thw(x: Tensor, pad: Tuple[int, int, int], padding_type: str) -> Tensor:
    """models/utils.py:259-270."""
    pt, ph, pw = pad
    if pt == 0 and ph == 0 and pw == 0:
        return x
    B, T, H, W, C = x.shape
    if padding_type == "nearest":
        return F.interpolate(x.permute(0, 4, 1, 2, 3), size=(T - pt, H - ph, W - pw)).permute(0, 2, 3, 4, 1)
    return x[:, :T - pt, :H - ph, :W - pw, :].contiguous()


# --------------------------------------------------------------------------------------
# cuboid decomposition: cuboid_transformer.py:388-467 (reorder / reverse)
# --------------------------------------------------------------------------------------
def cuboid_token_ids(shape: Sequence[int], cuboid: Sequence[int], strategy: Sequence[str]) -> Tensor:
    """(num_cuboids, volume) int64 table: flat index (t*H+h)*W+w of each cuboid member.

    Cuboids are enumerated over (nT, nH, nW), members over (bT, bH, bW).  'l' = the
    members are contiguous (coord = c*b + i); 'd' = dilated with stride n = size/b
    (coord = i*n + c).  cuboid_transformer.py:415-428.
    """
    coords = []
    for size, b, s in zip(shape, cuboid, strategy):
        n = size // b
        c = torch.arange(n).view(n, 1)
        i = torch.arange(b).view(1, b)
        if s == "l":
            coords.append(c * b + i)        # (n, b)
        elif s == "d":
            coords.append(i * n + c)
        else:
            raise NotImplementedError(s)
    (T, H, W) = shape
    ct, ch, cw = coords
    nT, bT = ct.shape
    nH, bH = ch.shape
    nW, bW = cw.shape
    t = ct.view(nT, 1, 1, bT, 1, 1)
    h = ch.view(1, nH, 1, 1, bH, 1)
    w = cw.view(1, 1, nW, 1, 1, bW)
    ids = (t * H + h) * W + w
    return ids.reshape(nT * nH * nW, bT * bH * bW)


def cuboid_reorder(x: Tensor, cuboid, strategy) -> Tensor:
    """(B,T,H,W,C) -> (B, num_cuboids, volume, C).  cuboid_transformer.py:388-429."""
    B, T, H, W, C = x.shape
    ids = cuboid_token_ids((T, H, W), cuboid, strategy)
    return x.reshape(B, T * H * W, C)[:, ids.reshape(-1)].reshape(B, ids.shape[0], ids.shape[1], C)


def cuboid_reorder_reverse(y: Tensor, cuboid, strategy, shape) -> Tensor:
    """inverse of cuboid_reorder.  cuboid_transformer.py:432-467."""
    B, nc, vol, C = y.shape
    T, H, W = shape
    ids = cuboid_token_ids((T, H, W), cuboid, strategy).reshape(-1)
    out = torch.empty(B, T * H * W, C, dtype=y.dtype)
    out[:, ids] = y.reshape(B, nc * vol, C)
    return out.reshape(B, T, H, W, C)


def clamp_cuboid(shape, cuboid, shift, strategy):
    """cuboid_transformer.py:563-592."""
    cuboid, shift = list(cuboid), list(shift)
    for i in range(3):
        if strategy[i] == "d":
            shift[i] = 0
        if shape[i] <= cuboid[i]:
            cuboid[i] = shape[i]
            shift[i] = 0
    return tuple(cuboid), tuple(shift)


def cuboid_attention_mask(shape, cuboid, shift, strategy, padding_type) -> Tensor:
    """(num_cuboids, vol, vol) bool.  cuboid_transformer.py:470-528.

    Region id per padded position and axis: the reference writes three slices in
    order [0,size-b) -> 0, [size-b, size-shift) -> 1, [size-shift, size) -> 2, with
    python slice semantics (for shift == 0 the third slice is the whole axis, so the
    whole axis ends up in region 2).
    """
    T, H, W = shape
    pad = [(b - s % b) % b for s, b in zip(shape, cuboid)]
    full = [s + p for s, p in zip(shape, pad)]
    region = []
    for size, b, sh in zip(full, cuboid, shift):
        p = torch.arange(size)
        r = torch.zeros(size, dtype=torch.long)
        # python: slice(-b) ; slice(-b, -sh) ; slice(-sh, None)
        r[slice(-b)] = 0
        r[slice(-b, -sh)] = 1
        r[slice(-sh, None)] = 2
        region.append(r)
    rid = (region[0].view(-1, 1, 1) * 3 + region[1].view(1, -1, 1)) * 3 + region[2].view(1, 1, -1)
    rid = cuboid_reorder(rid.view(1, *full, 1).float(), cuboid, strategy)[0, :, :, 0]   # (nc, vol)
    mask = rid.unsqueeze(1) == rid.unsqueeze(2)
    if padding_type == "ignore":
        valid = torch.zeros(full, dtype=torch.bool)
        valid[:T, :H, :W] = True
        if any(s > 0 for s in shift):
            valid = torch.roll(valid, shifts=(-shift[0], -shift[1], -shift[2]), dims=(0, 1, 2))
        valid = cuboid_reorder(valid.view(1, *full, 1).float(), cuboid, strategy)[0, :, :, 0] > 0.5
        mask = mask & valid.unsqueeze(1) & valid.unsqueeze(2)
    return mask


def relative_position_index(cuboid: Sequence[int]) -> Tensor:
    """(vol, vol) int64 buffer of CuboidSelfAttentionLayer.  cuboid_transformer.py:719-734."""
    bt, bh, bw = cuboid
    t, h, w = torch.meshgrid(torch.arange(bt), torch.arange(bh), torch.arange(bw), indexing="ij")
    c = torch.stack([t.reshape(-1), h.reshape(-1), w.reshape(-1)])          # (3, vol)
    rel = c[:, :, None] - c[:, None, :]
    return ((rel[0] + bt - 1) * (2 * bh - 1) * (2 * bw - 1) + (rel[1] + bh - 1) * (2 * bw - 1)
            + (rel[2] + bw - 1))


def masked_softmax(score: Tensor, mask: Tensor) -> Tensor:
    """cuboid_transformer.py:531-560 (fp32/bf16 branch: fill -1e18, softmax, * mask)."""
    score = score.masked_fill(~mask, -1e18)
    return torch.softmax(score, dim=-1) * mask


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def cuboid_self_attention(sd: Dict[str, Tensor], p: str, x: Tensor, num_heads: int,
                          cuboid, shift, strategy, padding_type: str,
                          use_relative_pos: bool = True, use_final_proj: bool = True) -> Tensor:
    """CuboidSelfAttentionLayer.forward without global vectors.  cuboid_transformer.py:812-966.

    Returns the layer output (the residual add is done by the caller, :1151).
    """
    B, T, H, W, C = x.shape
    x = F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)        # :813
    cuboid, shift = clamp_cuboid((T, H, W), cuboid, shift, strategy)                   # :821
    pad = tuple((b - s % b) % b for s, b in zip((T, H, W), cuboid))                    # :824-826
    x = _pad_thw(x, pad, padding_type)                                                 # :829
    if any(s > 0 for s in shift):                                                      # :833-836
        x = torch.roll(x, shifts=(-shift[0], -shift[1], -shift[2]), dims=(1, 2, 3))
    Tp, Hp, Wp = T + pad[0], H + pad[1], W + pad[2]
    xr = cuboid_reorder(x, cuboid, strategy)                                           # :839
    _, nc, vol, _ = xr.shape
    mask = cuboid_attention_mask((T, H, W), cuboid, shift, strategy, padding_type)     # :843-847
    hd = C // num_heads
    qkv = F.linear(xr, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"))                   # :849
    qkv = qkv.reshape(B, nc, vol, 3, num_heads, hd).permute(3, 0, 4, 1, 2, 5)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (hd ** -0.5)                                                               # :712,852
    score = q @ k.transpose(-2, -1)                                                    # :853
    if use_relative_pos:                                                               # :855-861
        idx = sd[p + "relative_position_index"][:vol, :vol].reshape(-1)
        bias = sd[p + "relative_position_bias_table"][idx].reshape(vol, vol, num_heads)
        score = score + bias.permute(2, 0, 1).unsqueeze(1)
    att = masked_softmax(score, mask)                                                  # :947
    y = (att @ v).permute(0, 2, 3, 1, 4).reshape(B, nc, vol, C)                        # :949
    if use_final_proj:
        y = F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])                    # :951-952
    y = cuboid_reorder_reverse(y, cuboid, strategy, (Tp, Hp, Wp))                      # :956
    if any(s > 0 for s in shift):                                                      # :958-959
        y = torch.roll(y, shifts=shift, dims=(1, 2, 3))
    return _unpad_thw(y, pad, padding_type)                                            # :962


def positionwise_ffn(sd, p: str, x: Tensor, activation: str = "gelu", gated: bool = False) -> Tensor:
    """PositionwiseFFN.forward, pre-norm.  cuboid_transformer.py:182-208."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    act = _activation(activation)
    if gated:
        h = act(F.linear(h, sd[p + "ffn_1_gate.weight"], sd[p + "ffn_1_gate.bias"])) * \
            F.linear(h, sd[p + "ffn_1.weight"], sd[p + "ffn_1.bias"])
    else:
        h = act(F.linear(h, sd[p + "ffn_1.weight"], sd[p + "ffn_1.bias"]))
    h = F.linear(h, sd[p + "ffn_2.weight"], sd[p + "ffn_2.bias"])
    return h + x


def stack_cuboid_block(sd, p: str, x: Tensor, num_heads, sizes, strategies, shifts, padding_type,
                       activation="gelu", gated=False, use_inter_ffn=True,
                       use_relative_pos=True, use_final_proj=True) -> Tensor:
    """StackCuboidSelfAttentionBlock.forward (eval, no global vectors).  cuboid_transformer.py:1126-1186."""
    n = len(sizes)
    for a in range(n):
        x = x + cuboid_self_attention(sd, f"{p}attn_l.{a}.", x, num_heads, sizes[a], shifts[a], strategies[a],
                                      padding_type, use_relative_pos, use_final_proj)
        if use_inter_ffn:
            x = positionwise_ffn(sd, f"{p}ffn_l.{a}.", x, activation, gated)
    if not use_inter_ffn:
        x = positionwise_ffn(sd, f"{p}ffn_l.0.", x, activation, gated)
    return x


def time_embed_resblock(sd, p: str, x: Tensor, emb, use_scale_shift_norm: bool = False) -> Tensor:
    """TimeEmbedResBlock.forward (dims=3, no up/down).  models/time_embed.py:134-169.

    x: (B, T, H, W, C) channels-last here; the reference works on (B, C, T, H, W).
    """
    xc = x.permute(0, 4, 1, 2, 3)
    Cin = xc.shape[1]
    w1 = sd[p + "in_layers.2.weight"]
    Cout = w1.shape[0]
    h = F.group_norm(xc, _gn_groups(Cin), sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"], 1e-5)
    h = F.conv3d(F.silu(h), w1, sd[p + "in_layers.2.bias"], padding=1)
    gn2 = (_gn_groups(Cout), sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5)
    w2, b2 = sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"]
    if emb is not None and (p + "emb_layers.1.weight") in sd:
        e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
        e = e[:, :, None, None, None]
        if use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = F.group_norm(h, *gn2) * (1 + scale) + shift
            h = F.conv3d(F.silu(h), w2, b2, padding=1)
        else:
            h = F.conv3d(F.silu(F.group_norm(h + e, *gn2)), w2, b2, padding=1)
    else:
        h = F.conv3d(F.silu(F.group_norm(h, *gn2)), w2, b2, padding=1)
    if (p + "skip_connection.weight") in sd:
        sw = sd[p + "skip_connection.weight"]
        xc = F.conv3d(xc, sw, sd[p + "skip_connection.bias"], padding=sw.shape[-1] // 2)
    return (xc + h).permute(0, 2, 3, 4, 1)


def patch_merging_3d(sd, p: str, x: Tensor, downsample=(1, 2, 2), padding_type="zeros") -> Tensor:
    """PatchMerging3D.forward.  cuboid_transformer.py:261-296 (incl. the pad_t quirk at :280)."""
    B, T, H, W, C = x.shape
    d = downsample
    pt, ph, pw = [(di - s % di) % di for s, di in zip((T, H, W), d)]
    if ph or pw:                                           # reference tests `pad_h or pad_h or pad_w`
        x = _pad_thw(x, (pt, ph, pw), padding_type)
        T, H, W = T + pt, H + ph, W + pw
    x = x.reshape(B, T // d[0], d[0], H // d[1], d[1], W // d[2], d[2], C)
    x = x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, T // d[0], H // d[1], W // d[2], d[0] * d[1] * d[2] * C)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    return F.linear(x, sd[p + "reduction.weight"])


def upsample_3d(sd, p: str, x: Tensor, target_thw) -> Tensor:
    """Upsample3DLayer.forward, THWC layout, no temporal upsampling.  cuboid_transformer.py:366-375."""
    B, T, H, W, C = x.shape
    w = sd[p + "conv.weight"]
    y = x.reshape(B * T, H, W, C).permute(0, 3, 1, 2)
    y = F.interpolate(y, size=(target_thw[1], target_thw[2]), mode="nearest")
    y = F.conv2d(y, w, sd[p + "conv.bias"], padding=(w.shape[-2] // 2, w.shape[-1] // 2))
    return y.permute(0, 2, 3, 1).reshape(B, T, target_thw[1], target_thw[2], w.shape[0])


def pos_embed(sd, p: str, x: Tensor, typ: str = "t+h+w") -> Tensor:
    """PosEmbed.forward.  cuboid_transformer.py:65-90."""
    _, T, H, W, C = x.shape
    if typ == "t+h+w":
        return (x + sd[p + "T_embed.weight"][:T].reshape(T, 1, 1, C)
                + sd[p + "H_embed.weight"][:H].reshape(1, H, 1, C)
                + sd[p + "W_embed.weight"][:W].reshape(1, 1, W, C))
    if typ == "t+hw":
        maxW = sd[p + "HW_embed.weight"].shape[0] // H   # only exact when H == maxH
        idx = torch.arange(H).unsqueeze(-1) * maxW + torch.arange(W)
        return x + sd[p + "T_embed.weight"][:T].reshape(T, 1, 1, C) + sd[p + "HW_embed.weight"][idx]
    raise NotImplementedError(typ)


# --------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------
def unet_geometry(cfg: dict):
    """Shapes and cuboid configuration per level.  cuboid_transformer_unet.py:99-106,201-235,377-404."""
    T_in, H, W, C_in = cfg["input_shape"]
    T_out = cfg["target_shape"][0]
    depth = list(cfg.get("depth", [4, 4, 4]))
    nb = len(depth)
    base = cfg.get("base_units", 128)
    ds = cfg.get("downsample", 2)
    if not isinstance(ds, (tuple, list)):
        ds = (1, ds, ds)
    block_units = cfg.get("block_units")
    if block_units is None:
        block_units = [round_to(base * int((max(ds) ** cfg.get("scale_alpha", 1.0)) ** i), 4) for i in range(nb)]
    data_shape = (T_in + T_out, H, W, C_in + 1)
    mem = [(data_shape[0], H, W, base)]
    for i in range(nb - 1):
        t, h, w, _ = mem[-1]
        pt, ph, pw = [(d - s % d) % d for s, d in zip((t, h, w), ds)]
        mem.append(((t + pt) // ds[0], (h + ph) // ds[1], (w + pw) // ds[2], block_units[i + 1]))
    pats = cfg.get("block_attn_patterns")
    sizes, strategies, shifts = [], [], []
    if pats is not None:
        if not isinstance(pats, (list, tuple)):
            pats = [pats] * nb
        for i, name in enumerate(pats):
            s, st, sh = attention_pattern(name, mem[i])
            sizes.append(s), strategies.append(st), shifts.append(sh)
    else:
        def per_level(v):
            return [v] * nb if not isinstance(v[0][0], (list, tuple)) else v
        sizes = per_level(cfg.get("block_cuboid_size", [(4, 4, 4), (4, 4, 4)]))
        strategies = per_level(cfg.get("block_cuboid_strategy", [("l", "l", "l"), ("d", "d", "d")]))
        shifts = per_level(cfg.get("block_cuboid_shift_size", [(0, 0, 0), (0, 0, 0)]))
    return dict(data_shape=data_shape, mem_shapes=mem, block_units=block_units, depth=depth, downsample=tuple(ds),
                sizes=sizes, strategies=strategies, shifts=shifts, in_len=T_in, out_len=T_out)


def unet_forward(sd: Dict[str, Tensor], cfg: dict, x: Tensor, t: Tensor, cond: Tensor) -> Tensor:
    """CuboidTransformerUNet.forward.  cuboid_transformer_unet.py:406-493."""
    if cfg.get("num_global_vectors", 0):
        raise NotImplementedError("global vectors are dead at every shipped config (SURVEY.md §8(a))")
    g = unet_geometry(cfg)
    nh = cfg.get("num_heads", 4)
    pad_t = cfg.get("padding_type", "ignore")
    act = cfg.get("ffn_activation", "leaky")
    kw = dict(activation=act, gated=cfg.get("gated_ffn", False), use_inter_ffn=cfg.get("use_inter_ffn", True),
              use_relative_pos=cfg.get("use_relative_pos", True),
              use_final_proj=cfg.get("self_attn_use_final_proj", True))
    ssn = cfg.get("time_embed_use_scale_shift_norm", False)
    res_connect = cfg.get("unet_res_connect", True)
    hier = cfg.get("hierarchical_pos_embed", False)
    pe_typ = cfg.get("pos_embed_type", "t+h+w")

    x = torch.cat([cond, x], dim=1)                                                    # :425
    ind = torch.ones_like(x[..., :1])
    ind[:, g["in_len"]:] = 0.0                                                         # :426-427
    x = torch.cat([x, ind], dim=-1)                                                    # :428
    x = time_embed_resblock(sd, "first_proj.", x, None)                                # :429-431
    x = pos_embed(sd, "pos_embed.", x, pe_typ)                                         # :435
    temb = timestep_embedding(t, g["block_units"][0])                                  # :437
    temb = F.linear(F.silu(F.linear(temb, sd["time_embed.layer.0.weight"], sd["time_embed.layer.0.bias"])),
                    sd["time_embed.layer.2.weight"], sd["time_embed.layer.2.bias"])
    nb = len(g["depth"])
    skips: List[Tensor] = []
    for i in range(nb):                                                                # :442-465
        if i > 0:
            x = patch_merging_3d(sd, f"downsample_layers.{i - 1}.", x, g["downsample"], pad_t)
            if hier:
                x = pos_embed(sd, f"down_hierarchical_pos_embed_l.{i - 1}.", x, pe_typ)
        for d in range(g["depth"][i]):
            x = time_embed_resblock(sd, f"down_time_embed_blocks.{i}.", x, temb, ssn)
            x = stack_cuboid_block(sd, f"down_self_blocks.{i}.{d}.", x, nh, g["sizes"][i], g["strategies"][i],
                                   g["shifts"][i], pad_t, **kw)
        if res_connect and i < nb - 1:
            skips.append(x)
    for i in range(nb - 1, -1, -1):                                                    # :468-491
        if res_connect and i < nb - 1:
            x = x + skips[i]
        for d in range(g["depth"][i]):
            x = time_embed_resblock(sd, f"up_time_embed_blocks.{i}.", x, temb, ssn)
            x = stack_cuboid_block(sd, f"up_self_blocks.{i}.{d}.", x, nh, g["sizes"][i], g["strategies"][i],
                                   g["shifts"][i], pad_t, **kw)
        if i > 0:
            x = upsample_3d(sd, f"upsample_layers.{i - 1}.", x, g["mem_shapes"][i - 1][:3])
            if hier:
                x = pos_embed(sd, f"up_hierarchical_pos_embed_l.{i - 1}.", x, pe_typ)
    return F.linear(x[:, g["in_len"]:], sd["final_proj.weight"], sd["final_proj.bias"])   # :492
