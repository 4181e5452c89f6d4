"""Host-side geometry of cuboid self-attention: everything the HIP attention kernel needs as flat tables.

The reference reorders activations with pad -> roll -> reshape/permute -> attention -> inverse
(cuboid_transformer.py:821-847, 388-467, 470-528, 956-962).  Here the same mapping is evaluated
once per layer on the host, in coordinate arithmetic, into
    tok_index[cuboid, slot] : flat token id (t*H + h)*W + w feeding that slot, or -1 for a padded slot
    mask[cuboid, slot_q, slot_k] : uint8, only materialised when some entry is 0
    bias[head, slot_q, slot_k] : relative-position bias gathered from the learned table (per weight update)
so that the device never materialises a reordered tensor.
"""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch


def clamp_cuboid(shape, cuboid, shift, strategy):
    """Cuboid / shift actually used for a (T,H,W) input (cuboid_transformer.py:563-592)."""
    cub, sh = list(cuboid), list(shift)
    for ax in range(3):
        if strategy[ax] == "d":
            sh[ax] = 0
        if shape[ax] <= cub[ax]:
            cub[ax] = shape[ax]
            sh[ax] = 0
    return tuple(int(v) for v in cub), tuple(int(v) for v in sh)


def relative_position_index(cuboid: Sequence[int]) -> torch.Tensor:
    """The (vol, vol) int64 buffer registered by the reference layer (cuboid_transformer.py:719-734)."""
    bt, bh, bw = (int(v) for v in cuboid)
    idx = np.arange(bt * bh * bw)
    t, h, w = idx // (bh * bw), (idx // bw) % bh, idx % bw
    dt = t[:, None] - t[None, :] + bt - 1
    dh = h[:, None] - h[None, :] + bh - 1
    dw = w[:, None] - w[None, :] + bw - 1
    return torch.from_numpy((dt * (2 * bh - 1) + dh) * (2 * bw - 1) + dw).long()


def nearest_source_index(n_in: int, n_out: int) -> np.ndarray:
    """src[o] = the input index F.interpolate(mode="nearest") reads for output index o when an axis of n_in is resized to n_out
    (torch: min(floor(o * float32(n_in / n_out)), n_in - 1)), obtained by resizing an index ramp with F.interpolate itself."""
    import torch.nn.functional as F
    ramp = torch.arange(n_in, dtype=torch.float32).reshape(1, 1, n_in)
    return F.interpolate(ramp, size=n_out, mode="nearest").reshape(n_out).long().numpy()


def _slot_coords(padded: Tuple[int, int, int], cuboid, strategy):
    """Per axis: (n_cuboids, block) array of padded-space coordinates. 'l' contiguous, 'd' strided by n."""
    out = []
    for size, b, s in zip(padded, cuboid, strategy):
        n = size // b
        c = np.arange(n)[:, None]
        i = np.arange(b)[None, :]
        if s == "l":
            out.append(c * b + i)
        elif s == "d":
            out.append(i * n + c)
        else:
            raise NotImplementedError(f"unknown cuboid strategy {s!r}")
    return out


def _flatten_slots(ct, ch, cw, fn):
    """Evaluate fn(t, h, w) on the (nT,nH,nW,bT,bH,bW) grid -> (num_cuboids, vol)."""
    nT, bT = ct.shape
    nH, bH = ch.shape
    nW, bW = cw.shape
    t = ct.reshape(nT, 1, 1, bT, 1, 1)
    h = ch.reshape(1, nH, 1, 1, bH, 1)
    w = cw.reshape(1, 1, nW, 1, 1, bW)
    full = np.broadcast_to(fn(t, h, w), (nT, nH, nW, bT, bH, bW))
    return full.reshape(nT * nH * nW, bT * bH * bW)


def attention_tables(shape, cuboid, shift, strategy, padding_type):
    """All index tables of one CuboidSelfAttentionLayer applied to a (T,H,W) token grid."""
    if padding_type not in ("zeros", "ignore", "nearest"):
        raise ValueError(f"padding_type={padding_type!r}")
    T, H, W = (int(v) for v in shape)
    cub, sh = clamp_cuboid((T, H, W), cuboid, shift, strategy)
    pad = tuple((b - s % b) % b for s, b in zip((T, H, W), cub))
    P = (T + pad[0], H + pad[1], W + pad[2])
    ct, ch, cw = _slot_coords(P, cub, strategy)
    # torch.roll(x, -shift): slot at padded coordinate p reads padded position (p + shift) mod P
    st, s_h, sw = (ct + sh[0]) % P[0], (ch + sh[1]) % P[1], (cw + sh[2]) % P[2]
    tok_out = None
    if padding_type == "nearest" and any(pad):
        # models/utils.py:228-270: the padded grid is F.interpolate(x, size=P) (nearest) -- several slots read one token -- and the
        # un-padding is F.interpolate back to (T, H, W).  torch's nearest rule is floor(dst * float32(in / out)) clamped to in - 1, which
        # is NOT floor(dst * in / out) in integers (axis 22 padded to 26: padded position 13 reads token 10, the integer rule says 11),
        # so both maps come from F.interpolate itself, per axis (nearest resizing is separable): exact by construction.
        up = [nearest_source_index(n, Pn) for n, Pn in zip((T, H, W), P)]        # padded position -> source token coordinate
        down = [nearest_source_index(Pn, n) for n, Pn in zip((T, H, W), P)]      # token coordinate -> padded position it receives from
        tok = _flatten_slots(st, s_h, sw, lambda t, h, w: (up[0][t] * H + up[1][h]) * W + up[2][w])
        recv = np.full(P, -1, dtype=np.int64)                 # padded position -> receiving token
        ot, oh, ow = np.meshgrid(np.arange(T), np.arange(H), np.arange(W), indexing="ij")
        recv[down[0][ot], down[1][oh], down[2][ow]] = (ot * H + oh) * W + ow
        tok_out = _flatten_slots(st, s_h, sw, lambda t, h, w: recv[t, h, w])
        assert sorted(tok_out[tok_out >= 0].tolist()) == list(range(T * H * W))
    else:
        tok = _flatten_slots(st, s_h, sw, lambda t, h, w: np.where((t < T) & (h < H) & (w < W), (t * H + h) * W + w, -1))
    nc, vol = tok.shape

    # shifted-window region ids (cuboid_transformer.py:516-525): three python slices per axis, later ones win
    def region(size, b, s):
        r = np.zeros(size, dtype=np.int64)
        r[slice(-b)] = 0
        r[slice(-b, -s)] = 1
        r[slice(-s, None)] = 2
        return r
    rt, rh, rw = region(P[0], cub[0], sh[0]), region(P[1], cub[1], sh[1]), region(P[2], cub[2], sh[2])
    rid = _flatten_slots(ct, ch, cw, lambda t, h, w: (rt[t] * 3 + rh[h]) * 3 + rw[w])
    mask = rid[:, :, None] == rid[:, None, :]
    if padding_type == "ignore":
        valid = tok >= 0
        mask = mask & valid[:, :, None] & valid[:, None, :]
    mask_t = None if mask.all() else torch.from_numpy(mask.astype(np.uint8)).contiguous()
    tok_t = torch.from_numpy(tok.astype(np.int32)).contiguous()
    tok_out_t = torch.from_numpy(tok_out.astype(np.int32)).contiguous() if tok_out is not None else None
    return dict(cuboid=cub, shift=sh, pad=pad, nc=int(nc), vol=int(vol), tok_index=tok_t, mask=mask_t, tok_out=tok_out_t,
                affine=affine_form(tok_t) if tok_out is None else None)


def affine_form(tok_index: torch.Tensor):
    """(n_inner, outer, inner, slot) with tok_index[c, s] == (c // n_inner) * outer + (c % n_inner) * inner + s * slot for every entry,
    or None.  True for the un-shifted, un-padded axial cuboids of the SEVIR grids: pd_attn_block_fused_ex then computes the token ids
    instead of loading the table in front of its row gather."""
    tok = tok_index.cpu().numpy().astype(np.int64)
    nc, vol = tok.shape
    if (tok < 0).any():
        return None
    slot = int(tok[0, 1] - tok[0, 0]) if vol > 1 else 0
    inner = int(tok[1, 0] - tok[0, 0]) if nc > 1 else 0
    c = np.arange(nc)[:, None]
    s = np.arange(vol)[None, :]
    for n_inner in sorted({nc} | {d for d in range(1, nc + 1) if nc % d == 0}):
        outer = int(tok[n_inner, 0] - tok[0, 0]) if n_inner < nc else 0
        if tok[0, 0] == 0 and np.array_equal((c // n_inner) * outer + (c % n_inner) * inner + s * slot, tok):
            if max(abs(outer), abs(inner), abs(slot)) < 2 ** 24:
                return (int(n_inner), outer, inner, slot)
    return None


def relative_position_bias(table: torch.Tensor, rel_index: torch.Tensor, vol: int) -> torch.Tensor:
    """(heads, vol, vol) fp32 bias: table[index[:vol, :vol]] (the slice reproduces quirk Q2 of SURVEY.md:
    when the cuboid was clamped the index is sliced from the *unclamped* table).  cuboid_transformer.py:855-860."""
    idx = rel_index[:vol, :vol].reshape(-1)
    return table.detach().float()[idx].reshape(vol, vol, -1).permute(2, 0, 1).contiguous()
