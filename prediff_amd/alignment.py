"""Knowledge alignment (guided sampling) -- stays in PyTorch autograd by design (BASELINE.json north_star).

Reference: diffusion/knowledge_alignment/sevir.py:7-104 (SEVIRAvgIntensityAlignment), models.py:19-528
(NoisyCuboidTransformerEncoder, AttentionPool3d, QKVAttention), alignment_pl.py:423-446 (get_sample_align_fn).

The guidance term of one denoising step is  guide_scale * grad_{z_t} || mean_tau U_phi(z_t, t)[b, tau] - avg_x_gt[b] ||_2
(the L2 norm spans the whole batch, sevir.py:81-82).  U_phi is a small half-U-Net built from the same blocks as the
denoiser; because its *gradient* is needed it runs as ordinary differentiable PyTorch ops (on the GPU through
PyTorch-ROCm, or on CPU), on modules that keep the reference's constructor keywords and ``state_dict`` schema so the
reference checkpoint ``pretrained_sevirlr_alignment_avg_x_cuboid_v1.pt`` loads strictly.  The cuboid decomposition
reuses prediff_amd.cuboid_geometry (gather / scatter by token index instead of the reference's reshape-permute chain).
"""
import math
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .cuboid_geometry import attention_tables
from .cuboid_transformer_unet import (CuboidSelfAttentionLayer, PatchMerging3D, PosEmbed, PositionwiseFFN,
                                      StackCuboidSelfAttentionBlock, TimeEmbedLayer, TimeEmbedResBlock,
                                      apply_initialization, round_to)
from .patterns import CuboidSelfAttentionPatterns

_ACT = {"leaky": lambda v: F.leaky_relu(v, 0.1), "gelu": F.gelu, "relu": F.relu, "silu": F.silu, "identity": lambda v: v}


# ----------------------------------------------------------------------------------------------------------------------
# differentiable forwards of the parameter-holder modules (channels-last, (B, T, H, W, C))
# ----------------------------------------------------------------------------------------------------------------------
_FREQS: Dict = {}


def timestep_embedding(t, dim, max_period=10000):
    """models/utils.py:68-88"""
    half = dim // 2
    key = (half, max_period, str(t.device))
    freqs = _FREQS.get(key)
    if freqs is None:       # computed on the host exactly as the reference does, then kept on the device (no per-call upload: the
        freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(t.device)
        _FREQS[key] = freqs     # guidance gradient can then be captured in a HIP graph)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class _HipConv3d(torch.autograd.Function):
    """3x3x3, stride-1, pad-1 Conv3d of the guidance network on the engine's implicit-GEMM kernel, inside PyTorch autograd.

    The guidance gradient is only ever taken w.r.t. the noisy latent (the network is frozen), so the backward is data-gradient
    only: a convolution of grad_out with the spatially flipped, channel-transposed filter -- the same `pd_igemm` launch with a
    second weight pack.  Both directions run the hi/lo-split bf16 MFMA form (fp32-class accuracy, tests/test_hip_kernels.py:
    2e-5 per GEMM), so the guidance keeps the reference's fp32 numerics to ~1e-5; PyTorch's own fp32 Conv3d (MIOpen) spent
    ~70 % of the 25 ms guidance gradient at 32 trajectories in these six convolutions."""

    @staticmethod
    def _packs(conv: nn.Conv3d, device):
        from . import _lib as L
        from .packing import pack_conv
        split = not getattr(conv, "_hip_bf16", False)       # False: single-pass bf16 operands (SEVIRAvgIntensityAlignment(hip_precision="bf16"))
        key = (str(device), conv.weight.data_ptr(), conv.weight._version, split)
        cached = getattr(conv, "_hip_packs", None)
        if cached is None or cached[0] != key:
            w = conv.weight.detach().to(device)
            fwd = pack_conv(w, split)                                             # (27, Cout, Cin_p) hi, lo (lo = None without the split)
            bwd = pack_conv(w.flip(2, 3, 4).transpose(0, 1).contiguous(), split)  # (27, Cin, Cout_p): dgrad filter
            bias = conv.bias.detach().float().contiguous().to(device) if conv.bias is not None else None
            cached = (key, fwd, bwd, bias)
            conv._hip_packs = cached
        return cached[1], cached[2], cached[3]

    @staticmethod
    def _run(x_ncthw, w_hi, w_lo, bias, n_out):
        from . import _lib as L
        from .packing import pad64
        B, Cn, T, H, W = x_ncthw.shape
        M = B * T * H * W
        Cp = pad64(Cn)
        rows = x_ncthw.permute(0, 2, 3, 4, 1).reshape(M, Cn).float().contiguous()
        # (the activation is split exactly when the filter is; both halves in one allocation: the 256 x 256 hi/lo kernel reads them
        #  through one buffer descriptor)
        a_both = torch.empty((2 if w_lo is not None else 1, M, Cp), dtype=torch.bfloat16, device=rows.device)
        a_hi, a_lo = a_both[0], (a_both[1] if w_lo is not None else None)
        out = torch.empty((M, n_out), dtype=torch.float32, device=rows.device)
        with L.on_device(rows):
            L.cast_rows(rows, a_hi, a_lo, 1, M, 0, M, Cn, Cn, Cp)          # fp32 -> bf16 hi [+ lo], zero-padded columns, one pass
            L.igemm(a_hi, w_hi, A_lo=a_lo, W_lo=w_lo, M=M, N=n_out, Cin=Cp, taps=27, w_tap_stride=n_out * Cp,
                    geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias, out_f32=out)
        return out.reshape(B, T, H, W, n_out).permute(0, 4, 1, 2, 3)

    @staticmethod
    def forward(ctx, x, conv):
        fwd, bwd, bias = _HipConv3d._packs(conv, x.device)
        ctx.bwd, ctx.n_in = bwd, x.shape[1]
        return _HipConv3d._run(x, fwd[0], fwd[1], bias, conv.out_channels)

    @staticmethod
    def backward(ctx, grad_out):
        return _HipConv3d._run(grad_out, ctx.bwd[0], ctx.bwd[1], None, ctx.n_in), None


USE_HIP_CONV = True      # engine switch: False = nn.Conv3d (MIOpen) for the guidance network's convolutions
# (the token-level nn.Linear layers were tried on pd_igemm the same way: slower -- 16.6 ms against 15.1 ms per gradient at 32
#  trajectories; hipBLASLt's fp32 GEMM is fast enough that the fp32 -> bf16 hi/lo operand passes cost more than they save)


def _conv3d(conv: nn.Conv3d, x):
    """conv(x) on NCTHW x; on a HIP device the 3x3x3 convolutions go through _HipConv3d."""
    if (USE_HIP_CONV and x.is_cuda and conv.kernel_size == (3, 3, 3) and conv.stride == (1, 1, 1) and conv.padding == (1, 1, 1)
            and conv.dilation == (1, 1, 1) and conv.groups == 1 and x.dtype == torch.float32):
        return _HipConv3d.apply(x, conv)
    return conv(x)


class _HipGnSiluConv3d(torch.autograd.Function):
    """conv3x3x3(SiLU(GroupNorm(x))) [+ per-sample vector] [+ residual] on channels-last rows (B, T, H, W, C), forward and data
    gradient, without leaving the row layout: pd_groupnorm_silu writes the convolution's bf16 hi/lo operand directly, pd_igemm
    adds bias / embedding vector / residual in its epilogue; backward = dgrad pd_igemm (flipped, transposed filter) then
    pd_groupnorm_silu_bwd.  Replaces, per convolution and direction, PyTorch's GroupNorm + SiLU + two layout copies + the operand
    split (models/time_embed.py:89-120,134-169).  Frozen network: only x (and the residual) get a gradient."""

    @staticmethod
    def forward(ctx, x, gn, conv, rowvec, residual):
        from . import _lib as L
        B, T, H, W, Cn = x.shape
        S, M, N, G = T * H * W, B * T * H * W, conv.out_channels, gn.num_groups
        fwd, bwd, bias = _HipConv3d._packs(conv, x.device)
        x = x.contiguous()
        dev = x.device
        part = torch.empty(B * L.groupnorm_nchunk(S, Cn) * G * 2, dtype=torch.float64, device=dev)
        a_both = torch.empty((2 if fwd[1] is not None else 1, M, Cn), dtype=torch.bfloat16, device=dev)
        a_hi, a_lo = a_both[0], (a_both[1] if fwd[1] is not None else None)
        out = torch.empty((B, T, H, W, N), dtype=torch.float32, device=dev)
        with L.on_device(x):
            L.groupnorm_silu(x, gn.weight, gn.bias, part, a_hi, a_lo, B, S, Cn, G, Cn, gn.eps, silu=True)
            L.igemm(a_hi, fwd[0], A_lo=a_lo, W_lo=fwd[1], M=M, N=N, Cin=Cn, taps=27, w_tap_stride=N * Cn,
                    geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias, rowvec=rowvec, rows_per_sample=S if rowvec is not None else 0,
                    residual=residual, out_f32=out)
        ctx.save_for_backward(x, part, gn.weight, gn.bias)
        ctx.bwd, ctx.geom, ctx.has_res = bwd, (B, T, H, W, Cn, N, G, gn.eps), residual is not None
        return out

    @staticmethod
    def backward(ctx, d_out):
        from . import _lib as L
        x, part, gamma, beta = ctx.saved_tensors
        B, T, H, W, Cn, N, G, eps = ctx.geom
        S, M = T * H * W, B * T * H * W
        d_out = d_out.contiguous()
        dev = x.device
        g_both = torch.empty((2 if ctx.bwd[1] is not None else 1, M, N), dtype=torch.bfloat16, device=dev)
        g_hi, g_lo = g_both[0], (g_both[1] if ctx.bwd[1] is not None else None)
        da = torch.empty((M, Cn), dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        part_b = torch.empty_like(part)
        with L.on_device(x):
            L.cast_rows(d_out, g_hi, g_lo, 1, M, 0, M, N, N, N)
            L.igemm(g_hi, ctx.bwd[0], A_lo=g_lo, W_lo=ctx.bwd[1], M=M, N=Cn, Cin=N, taps=27, w_tap_stride=Cn * N,
                    geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), out_f32=da)
            L.groupnorm_silu_bwd(x, da, gamma, beta, part, part_b, dx, B, S, Cn, G, eps, silu=True)
        return dx, None, None, None, (d_out if ctx.has_res else None)


USE_HIP_RESBLOCK = True   # the guidance network's GroupNorm -> SiLU -> Conv3d pairs as one row-layout autograd node each


def _hip_resblock_ok(m, x):
    if not (USE_HIP_RESBLOCK and USE_HIP_CONV and x.is_cuda and x.dtype == torch.float32 and not m.use_scale_shift_norm):
        return False
    if m.training:           # the fused node has no Dropout (out_layers[2]): a network left in train mode keeps the torch path
        return False
    for gn, conv in ((m.in_layers[0], m.in_layers[2]), (m.out_layers[0], m.out_layers[3])):
        Cn = conv.in_channels
        if not (conv.kernel_size == (3, 3, 3) and conv.stride == (1, 1, 1) and conv.padding == (1, 1, 1) and conv.dilation == (1, 1, 1)
                and conv.groups == 1 and Cn % 64 == 0 and Cn <= 256 and 256 % Cn == 0 and conv.out_channels % 64 == 0
                and gn.num_groups <= 256 and gn.affine):
            return False
    sk = m.skip_connection
    return isinstance(sk, nn.Identity) or (isinstance(sk, nn.Conv3d) and sk.kernel_size == (1, 1, 1))


def resblock_forward(m: TimeEmbedResBlock, x, emb=None):
    """models/time_embed.py:134-169 on channels-last input."""
    if _hip_resblock_ok(m, x):
        e = None
        if m.use_embed:
            e = m.emb_layers[1](F.silu(emb)).float().contiguous()                   # (B, C_out): depends on t only, no gradient path
            if e.requires_grad:
                e = None
        if e is not None or not m.use_embed:
            sk = m.skip_connection
            res = x if isinstance(sk, nn.Identity) else F.linear(x, sk.weight.reshape(sk.out_channels, sk.in_channels), sk.bias)
            h = _HipGnSiluConv3d.apply(x, m.in_layers[0], m.in_layers[2], e, None)
            return _HipGnSiluConv3d.apply(h, m.out_layers[0], m.out_layers[3], None, res.contiguous())
    xc = x.permute(0, 4, 1, 2, 3)
    h = _conv3d(m.in_layers[2], F.silu(m.in_layers[0](xc)))
    if m.use_embed:
        e = m.emb_layers[1](F.silu(emb)).type(h.dtype)[:, :, None, None, None]
        if m.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = _conv3d(m.out_layers[3], F.silu(m.out_layers[0](h) * (1 + scale) + shift))
        else:
            h = _conv3d(m.out_layers[3], F.silu(m.out_layers[0](h + e)))
    else:
        h = _conv3d(m.out_layers[3], F.silu(m.out_layers[0](h)))
    return (m.skip_connection(xc) + h).permute(0, 2, 3, 4, 1)


class _HipCuboidAttention(torch.autograd.Function):
    """softmax(scale q k^T + bias [masked]) v over the cuboids of one layer, forward (pd_cuboid_attention, fp32 path) and data
    gradient (pd_cuboid_attention_bwd), both on q/k/v in NATURAL token order: the cuboid gather / scatter (cuboid_transformer.py:
    388-467) happens inside the kernels, so the ~30 reorder / bmm / softmax / index_copy launches per layer and direction that
    torch.autograd issues for cuboid_transformer.py:839-861,947-962 become one kernel each.  Only qkv is kept for the backward
    (the probabilities are recomputed).  The relative-position table gets no gradient (sampling-time guidance only)."""

    @staticmethod
    def forward(ctx, qkv, bias, tok32, mask_u8, heads, scale):
        from . import _lib as L
        B, S, C3 = qkv.shape
        Cn = C3 // 3
        nc, vol = tok32.shape
        qkv = qkv.contiguous()
        out = torch.empty(B, S, Cn, dtype=torch.float32, device=qkv.device)
        with L.on_device(qkv):
            L.cuboid_attention(qkv_f32=qkv, tok_index=tok32, bias=bias, mask=mask_u8, out_f32=out, B=B, ntok=S, Cn=Cn, heads=heads,
                               nc=nc, vol=vol, ld_qkv=C3, ld_out=Cn, scale=scale, force_generic=True)
        ctx.save_for_backward(qkv, bias, tok32, mask_u8 if mask_u8 is not None else tok32.new_empty(0))
        ctx.geom = (heads, scale, mask_u8 is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        from . import _lib as L
        qkv, bias, tok32, mask_u8 = ctx.saved_tensors
        heads, scale, has_mask = ctx.geom
        B, S, C3 = qkv.shape
        nc, vol = tok32.shape
        d_out = d_out.contiguous()
        d_qkv = torch.empty_like(qkv)
        with L.on_device(qkv):
            L.cuboid_attention_bwd(qkv=qkv, d_out=d_out, tok_index=tok32, bias=bias, mask=mask_u8 if has_mask else None, d_qkv=d_qkv,
                                   B=B, ntok=S, Cn=C3 // 3, heads=heads, nc=nc, vol=vol, ld_qkv=C3, ld_dout=C3 // 3, ld_dqkv=C3,
                                   scale=scale)
        return d_qkv, None, None, None, None, None


def _attn_bias(at, vol, device):
    """(heads, vol, vol) fp32 bias of one layer (cuboid_transformer.py:856-859), zeros without relative positions; cached per
    table version."""
    tab = at.relative_position_bias_table if at.use_relative_pos else None
    key = (str(device), vol) + ((tab.data_ptr(), tab._version) if tab is not None else ())
    hit = getattr(at, "_hip_bias", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            if tab is not None:
                idx = at.relative_position_index[:vol, :vol].reshape(-1)
                b = tab[idx].reshape(vol, vol, -1).permute(2, 0, 1).float().contiguous()
            else:
                b = torch.zeros(at.num_heads, vol, vol, dtype=torch.float32, device=device)
        at._hip_bias = hit = (key, b)
    return hit[1]


USE_HIP_ATTN = True    # the guidance network's cuboid attention (forward + data gradient) on the HIP kernels when x is on a GPU


def _hip_attention_ok(at, x, vol):
    hd = x.shape[-1] // at.num_heads
    # pd_cuboid_attention_bwd keeps q, k, dO, v and two vol x vol tiles in dynamic LDS (csrc/attention.hip): mirror its bound so that a
    # geometry it refuses (volume 64 with head_dim 128) takes the torch path instead of raising inside the guidance gradient
    lds_bwd = 4 * (2 * vol * hd + 2 * vol * (hd + 1) + 2 * vol * (vol + 1))
    return (USE_HIP_ATTN and x.is_cuda and x.dtype == torch.float32 and at.qkv.bias is None and vol <= 64 and hd <= 128
            and lds_bwd <= 160 * 1024 - 512 and not at.training)


def attention_forward(at: CuboidSelfAttentionLayer, x, tables):
    """cuboid_transformer.py:812-966 (no global vectors): returns the layer output (no residual)."""
    B, T, H, W, C = x.shape
    S = T * H * W
    dk = ("dev", str(x.device))
    if dk not in tables:    # device copies of the static index / mask tables, made once per device
        tok = tables["tok_index"].to(x.device).long()
        tables[dk] = dict(tok=tok, gather=torch.where(tok >= 0, tok, torch.full_like(tok, S)).reshape(-1),
                          mask=tables["mask"].to(x.device).bool() if tables["mask"] is not None else None,
                          tok32=tok.to(torch.int32).contiguous(),
                          mask_u8=tables["mask"].to(x.device).to(torch.uint8).contiguous() if tables["mask"] is not None else None)
    dev_t = tables[dk]
    tok, gather = dev_t["tok"], dev_t["gather"]
    nc, vol = tok.shape
    if _hip_attention_ok(at, x, vol):
        # natural token order end to end: the qkv Linear commutes with the cuboid gather, and a padded slot's q/k/v are zero rows
        # either way (no qkv bias)
        qkv = at.qkv(at.norm(x).reshape(B, S, C))
        y = _HipCuboidAttention.apply(qkv, _attn_bias(at, vol, x.device), dev_t["tok32"], dev_t["mask_u8"], at.num_heads, float(at.scale))
        if at.use_final_proj:
            y = at.proj(y)
        return y.reshape(B, T, H, W, C)
    h = at.norm(x).reshape(B, S, C)
    h = torch.cat([h, h.new_zeros(B, 1, C)], dim=1)                  # row S = the zero padding token
    xr = h[:, gather].reshape(B, nc, vol, C)
    hd = C // at.num_heads
    qkv = at.qkv(xr).reshape(B, nc, vol, 3, at.num_heads, hd).permute(3, 0, 4, 1, 2, 5)
    q, k, v = qkv[0] * at.scale, qkv[1], qkv[2]
    score = q @ k.transpose(-2, -1)
    if at.use_relative_pos:
        idx = at.relative_position_index[:vol, :vol].reshape(-1)
        bias = at.relative_position_bias_table[idx].reshape(vol, vol, -1).permute(2, 0, 1)
        score = score + bias.unsqueeze(1)
    if tables["mask"] is not None:
        mask = dev_t["mask"]
        score = score.masked_fill(~mask, -1e4 if score.dtype == torch.float16 else -1e18)
        att = torch.softmax(score, dim=-1) * mask
    else:
        att = torch.softmax(score, dim=-1)
    y = (att @ v).permute(0, 2, 3, 1, 4).reshape(B, nc * vol, C)
    if at.use_final_proj:
        y = at.proj(y)
    out = y.new_zeros(B, S + 1, C).index_copy(1, gather, y)          # padded slots land in the dummy row
    return out[:, :S].reshape(B, T, H, W, C)


def ffn_forward(ff: PositionwiseFFN, x):
    """cuboid_transformer.py:182-208 (pre-norm)."""
    h = ff.layer_norm(x)
    act = _ACT[ff.activation_name]
    h = act(ff.ffn_1_gate(h)) * ff.ffn_1(h) if ff.gated else act(ff.ffn_1(h))
    return ff.ffn_2(h) + x


def stack_forward(blk: StackCuboidSelfAttentionBlock, x, tables):
    for a, at in enumerate(blk.attn_l):
        x = x + attention_forward(at, x, tables[a])
        if blk.use_inter_ffn:
            x = ffn_forward(blk.ffn_l[a], x)
    if not blk.use_inter_ffn:
        x = ffn_forward(blk.ffn_l[0], x)
    return x


def patch_merge_forward(pm: PatchMerging3D, x):
    """cuboid_transformer.py:261-296"""
    B, T, H, W, C = x.shape
    d = pm.downsample
    pt, ph, pw = [(d[i] - s % d[i]) % d[i] for i, s in enumerate((T, H, W))]
    if ph or pw:
        if pm.padding_type == "nearest":
            x = F.interpolate(x.permute(0, 4, 1, 2, 3), size=(T + pt, H + ph, W + pw)).permute(0, 2, 3, 4, 1)
        else:
            x = F.pad(x, (0, 0, 0, pw, 0, ph, 0, pt))
        T, H, W = T + pt, H + ph, W + pw
    x = x.reshape(B, T // d[0], d[0], H // d[1], d[1], W // d[2], d[2], C).permute(0, 1, 3, 5, 2, 4, 6, 7)
    x = x.reshape(B, T // d[0], H // d[1], W // d[2], d[0] * d[1] * d[2] * C)
    return pm.reduction(pm.norm(x))


def pos_embed_forward(pe: PosEmbed, x):
    _, T, H, W, C = x.shape
    return x + pe.table(T, H, W).reshape(T, H, W, C).to(x.dtype)


# ----------------------------------------------------------------------------------------------------------------------
class AttentionPool3d(nn.Module):
    """Attention-pool read-out over the H*W tokens of one frame plus their mean token (models.py:49-104; QKV attention with the
    ch^-1/4 scaling on q and k, fp32 softmax, models.py:19-46).  Only the mean token's output is read (models.py:100), so only
    its query is formed: one (1 x L) score row per head instead of the (L x L) matrix -- same result, 1/L of the work and of the
    autograd graph.  Parameters (positional_embedding, qkv_proj, c_proj) keep the checkpoint's names and shapes."""

    def __init__(self, data_dim: int, embed_dim: int, num_heads: int, output_dim: int = None, init_mode: str = "0"):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(embed_dim, data_dim + 1) / embed_dim ** 0.5)
        self.qkv_proj = nn.Conv1d(embed_dim, 3 * embed_dim, 1)
        self.c_proj = nn.Conv1d(embed_dim, output_dim or embed_dim, 1)
        self.num_heads = num_heads
        self.init_mode = init_mode

    def forward(self, x):
        b, c = x.shape[:2]
        tok = x.reshape(b, c, -1)
        tok = torch.cat([tok.mean(dim=-1, keepdim=True), tok], dim=-1) + self.positional_embedding.to(x.dtype)   # (b, c, L), token 0 = mean
        heads, ch = self.num_heads, c // self.num_heads
        assert heads * ch == c
        q, k, v = self.qkv_proj(tok).view(b, 3, heads, ch, -1).unbind(dim=1)          # each (b, heads, ch, L)
        s = ch ** -0.25
        row = ((q[..., :1] * s).transpose(-1, -2) @ (k * s)).float()                  # (b, heads, 1, L): the read-out token's scores
        w = torch.softmax(row, dim=-1).to(v.dtype)
        pooled = (v @ w.transpose(-1, -2)).reshape(b, c, 1)
        return self.c_proj(pooled)[:, :, 0]

    def reset_parameters(self):
        self.qkv_proj.reset_parameters()
        self.c_proj.reset_parameters()


class NoisyCuboidTransformerEncoder(nn.Module):
    """Half U-Net U_phi(z_t, t) with a pooled read-out (models.py:107-528); pool="attention" as in every shipped config."""

    def __init__(self, input_shape, out_channels=1, base_units=128, block_units=None, scale_alpha=1.0, depth=[4, 4, 4],
                 downsample=2, downsample_type="patch_merge", block_attn_patterns=None,
                 block_cuboid_size=[(4, 4, 4), (4, 4, 4)], block_cuboid_strategy=[("l", "l", "l"), ("d", "d", "d")],
                 block_cuboid_shift_size=[(0, 0, 0), (0, 0, 0)], num_heads=4, attn_drop=0.0, proj_drop=0.0, ffn_drop=0.0,
                 ffn_activation="gelu", gated_ffn=False, norm_layer="layer_norm", use_inter_ffn=True,
                 hierarchical_pos_embed=False, pos_embed_type="t+h+w", padding_type="zeros", checkpoint_level=True,
                 use_relative_pos=True, self_attn_use_final_proj=True, num_global_vectors=0, use_global_vector_ffn=True,
                 use_global_self_attn=False, separate_global_qkv=False, global_dim_ratio=1, attn_linear_init_mode="0",
                 ffn_linear_init_mode="0", ffn2_linear_init_mode="2", attn_proj_linear_init_mode="2", conv_init_mode="0",
                 down_linear_init_mode="0", global_proj_linear_init_mode="2", norm_init_mode="0",
                 time_embed_channels_mult=4, time_embed_use_scale_shift_norm=False, time_embed_dropout=0.0,
                 pool: str = "attention", readout_seq: bool = True, out_len: int = None):
        super().__init__()
        if num_global_vectors:
            raise NotImplementedError("global vectors are dead at every shipped config")
        if pool != "attention" or downsample_type != "patch_merge" or norm_layer != "layer_norm":
            raise NotImplementedError
        self.input_shape, self.out_channels = input_shape, out_channels
        self.num_blocks, self.depth, self.base_units = len(depth), list(depth), base_units
        if not isinstance(downsample, (tuple, list)):
            downsample = (1, downsample, downsample)
        if block_units is None:
            block_units = [round_to(base_units * int((max(downsample) ** scale_alpha) ** i), 4) for i in range(self.num_blocks)]
        self.block_units = block_units
        self.hierarchical_pos_embed = hierarchical_pos_embed
        self.time_embed_channels = block_units[0] * time_embed_channels_mult
        self.pool, self.readout_seq, self.out_len = pool, readout_seq, out_len
        self.norm_init_mode = norm_init_mode
        T_in, H_in, W_in, C_in = input_shape
        self.first_proj = TimeEmbedResBlock(channels=C_in, emb_channels=None, dropout=proj_drop, out_channels=base_units,
                                            use_embed=False, dims=3)
        self.pos_embed = PosEmbed(embed_dim=base_units, typ=pos_embed_type, maxT=T_in, maxH=H_in, maxW=W_in)
        self.time_embed = TimeEmbedLayer(base_channels=block_units[0], time_embed_channels=self.time_embed_channels)
        if self.num_blocks > 1:
            self.downsample_layers = nn.ModuleList([
                PatchMerging3D(dim=block_units[i], downsample=downsample, padding_type=padding_type, out_dim=block_units[i + 1],
                               linear_init_mode=down_linear_init_mode, norm_init_mode=norm_init_mode)
                for i in range(self.num_blocks - 1)])
            if hierarchical_pos_embed:
                self.down_hierarchical_pos_embed_l = nn.ModuleList([
                    PosEmbed(embed_dim=block_units[i], typ=pos_embed_type, maxT=self.mem_shapes[i][0],
                             maxH=self.mem_shapes[i][1], maxW=self.mem_shapes[i][2]) for i in range(self.num_blocks - 1)])
        if block_attn_patterns is not None:
            if not isinstance(block_attn_patterns, (tuple, list)):
                block_attn_patterns = [block_attn_patterns] * self.num_blocks
            block_cuboid_size, block_cuboid_strategy, block_cuboid_shift_size = [], [], []
            for idx, key in enumerate(block_attn_patterns):
                cs, st, sh = CuboidSelfAttentionPatterns.get(key)(self.mem_shapes[idx])
                block_cuboid_size.append(cs), block_cuboid_strategy.append(st), block_cuboid_shift_size.append(sh)
        else:
            rep = lambda v: [v] * self.num_blocks if not isinstance(v[0][0], (list, tuple)) else v
            block_cuboid_size, block_cuboid_strategy, block_cuboid_shift_size = rep(block_cuboid_size), rep(block_cuboid_strategy), rep(block_cuboid_shift_size)
        self.block_cuboid_size, self.block_cuboid_strategy, self.block_cuboid_shift_size = block_cuboid_size, block_cuboid_strategy, block_cuboid_shift_size
        down_self, down_te = [], []
        for i in range(self.num_blocks):
            down_te.append(TimeEmbedResBlock(channels=self.mem_shapes[i][-1], emb_channels=self.time_embed_channels,
                                             dropout=time_embed_dropout, out_channels=self.mem_shapes[i][-1], use_embed=True,
                                             use_scale_shift_norm=time_embed_use_scale_shift_norm, dims=3))
            down_self.append(nn.ModuleList([
                StackCuboidSelfAttentionBlock(
                    dim=self.mem_shapes[i][-1], num_heads=num_heads, block_cuboid_size=block_cuboid_size[i],
                    block_strategy=block_cuboid_strategy[i], block_shift_size=block_cuboid_shift_size[i],
                    activation=ffn_activation, gated_ffn=gated_ffn, use_inter_ffn=use_inter_ffn, padding_type=padding_type,
                    use_relative_pos=use_relative_pos, use_final_proj=self_attn_use_final_proj,
                    attn_linear_init_mode=attn_linear_init_mode, ffn_linear_init_mode=ffn_linear_init_mode,
                    ffn2_linear_init_mode=ffn2_linear_init_mode, attn_proj_linear_init_mode=attn_proj_linear_init_mode,
                    norm_init_mode=norm_init_mode) for _ in range(depth[i])]))
        self.down_self_blocks = nn.ModuleList(down_self)
        self.down_time_embed_blocks = nn.ModuleList(down_te)
        out_shape = self.mem_shapes[-1]
        cc = out_shape[-1]
        data_dim = int(np.prod(out_shape[1:-1])) if readout_seq else int(np.prod(out_shape[:-1]))
        self.out = nn.Sequential(nn.GroupNorm(min(cc, 32), cc), nn.SiLU(), AttentionPool3d(data_dim, cc, num_heads, out_channels, init_mode="0"))
        apply_initialization(self.out[0], norm_mode=norm_init_mode)
        self.out[2].reset_parameters()
        self._tables = [[attention_tables(self.mem_shapes[i][:3], cs, sh, st, padding_type)
                         for cs, sh, st in zip(block_cuboid_size[i], block_cuboid_shift_size[i], block_cuboid_strategy[i])]
                        for i in range(self.num_blocks)]

    @property
    def mem_shapes(self):
        inner = tuple(self.input_shape)[:3] + (self.base_units,)
        if self.num_blocks == 1:
            return [inner]
        shapes, cur = [inner], inner
        for layer in self.downsample_layers:
            cur = layer.get_out_shape(cur)
            shapes.append(cur)
        return shapes

    def forward(self, x, t, verbose=False, **kwargs):
        """x (B,T,H,W,C), t (B,) -> (B, out_len, out_channels); extra kwargs (zc, y, avg_x_gt, ...) are ignored exactly as
        the reference does (models.py:459)."""
        B, seq_len = x.shape[0], x.shape[1]
        x = resblock_forward(self.first_proj, x)
        x = pos_embed_forward(self.pos_embed, x)
        t_emb = self.time_embed.layer(timestep_embedding(t, self.block_units[0]))
        for i in range(self.num_blocks):
            if i > 0:
                x = patch_merge_forward(self.downsample_layers[i - 1], x)
                if self.hierarchical_pos_embed:
                    x = pos_embed_forward(self.down_hierarchical_pos_embed_l[i - 1], x)
            for d in range(self.depth[i]):
                x = resblock_forward(self.down_time_embed_blocks[i], x, t_emb)
                x = stack_forward(self.down_self_blocks[i][d], x, self._tables[i])
        if self.readout_seq:
            if self.out_len is not None:
                seq_len = self.out_len
                x = x[:, -self.out_len:, ...]
            Bx, T, H, W, C = x.shape
            out = x.permute(0, 1, 4, 2, 3).reshape(Bx * T, C, H * W)          # "(b t) c (h w)"
            return self.out(out).reshape(B, seq_len, -1)
        Bx, T, H, W, C = x.shape
        return self.out(x.permute(0, 4, 1, 2, 3).reshape(Bx, C, T * H * W))


def get_sample_align_fn(sample_align_model):
    """grad of the (scalar-summed) alignment objective w.r.t. the noisy latent (alignment_pl.py:423-446)."""
    def sample_align_fn(x, *args, **kwargs):
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            logits = sample_align_model(x_in, *args, **kwargs)
            return torch.autograd.grad(logits.sum(), x_in, allow_unused=True)[0]
    return sample_align_fn


class SEVIRAvgIntensityAlignment:
    """sevir.py:7-104"""

    def __init__(self, alignment_type: str = "avg_x", guide_scale: float = 1.0, model_type: str = "cuboid",
                 model_args: Dict[str, Any] = None, model_ckpt_path: str = None, hip_precision: str = "fp32"):
        """`hip_precision` (engine option, not in the reference): operand form of the guidance network's 3x3x3 convolutions on a HIP
        device -- "fp32": bf16 hi/lo split, fp32-class accuracy (the form the parity of the guided step is pinned with);
        "bf16": single-pass bf16 operands, the accuracy class of a precision="bf16" / "fp8" denoiser next to it."""
        assert alignment_type in ["avg_x"], f"alignment_type {alignment_type} is not supported"
        if hip_precision not in ("fp32", "bf16"):
            raise ValueError("hip_precision must be 'fp32' or 'bf16'")
        self.alignment_type, self.guide_scale = alignment_type, guide_scale
        if model_type != "cuboid":
            raise NotImplementedError(f"model_type={model_type} is not implemented")
        self.model = NoisyCuboidTransformerEncoder(**(model_args or {}))
        if model_ckpt_path is not None:
            self.model.load_state_dict(torch.load(model_ckpt_path, map_location="cpu"))
        self.model.eval()
        self.model.requires_grad_(False)
        self.hip_precision = hip_precision
        for mod in self.model.modules():
            if isinstance(mod, nn.Conv3d):
                mod._hip_bf16 = hip_precision == "bf16"

    @classmethod
    def model_objective(cls, x, y=None, **kwargs):
        return torch.mean(x, dim=[2, 3, 4], keepdim=False).unsqueeze(-1)

    def alignment_fn(self, zt, t, y=None, zc=None, **kwargs):
        pred = self.model(zt, t, zc=zc, y=y, **kwargs).mean(dim=1)           # b t 1 -> b 1
        return torch.linalg.vector_norm(pred - kwargs.get("avg_x_gt"), ord=2)

    def get_mean_shift(self, zt, t, y=None, zc=None, **kwargs):
        return self.guide_scale * get_sample_align_fn(self.alignment_fn)(zt, t, y=y, zc=zc, **kwargs)


def get_alignment_kwargs_avg_x(context_seq=None, target_seq=None):
    """scripts/prediff/sevirlr/train_sevirlr_prediff.py:48-67: avg_x_gt = 2 * mean(target) per sample."""
    B = target_seq.shape[0]
    return {"avg_x_gt": 2.0 * target_seq.reshape(B, -1).mean(dim=1, keepdim=True)}
