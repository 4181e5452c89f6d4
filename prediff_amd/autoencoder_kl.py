"""Frame-wise KL-VAE on MI355X: reference constructor / encode / decode / state_dict schema, HIP forward.

Drop-in for ``prediff.taming.autoencoder_kl.AutoencoderKL`` (reference taming/autoencoder_kl.py:9-140, Encoder/Decoder
taming/vae.py:9-166, blocks taming/unet_2d_blocks.py:89-279, ResnetBlock2D / Upsample2D / Downsample2D
taming/resnet.py:77-190,367-495, AttentionBlock taming/attention.py:48-189): same keyword set, parameter names and
shapes (diffusers-0.13 schema), ``encode(x NCHW) -> DiagonalGaussianDistribution``,
``decode(z NCHW) -> Tensor NCHW``.

Engine: activations are channels-last (N, H*W, C) fp32; every Conv2d 3x3 / 1x1 / Linear is one pd_igemm launch (the
stride-2 asymmetric-pad down-sampling and the nearest x2 up-sampling are folded into the implicit-GEMM gather);
GroupNorm(eps 1e-6)+SiLU emits the bf16 operand; the single-head mid-block attention is three batched pd_igemm launches
(q k^T, V^T, P V) around a fp32 row softmax.  ``precision`` as in CuboidTransformerUNet.
"""
import math
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from .distributions import DiagonalGaussianDistribution
from .packing import pack_conv, pack_linear, pad64

VAE_EPS = 1e-6


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters; AutoencoderKL.encode/decode run the HIP kernels")


class ResnetBlock2D(_Holder):
    """taming/resnet.py:367-495 with temb_channels=None, output_scale_factor=1."""

    def __init__(self, in_channels, out_channels, groups, eps=VAE_EPS):
        super().__init__()
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None


class AttentionBlock(_Holder):
    """taming/attention.py:48-189, single head (num_head_channels=None)."""

    def __init__(self, channels, norm_num_groups, eps=VAE_EPS):
        super().__init__()
        self.channels = channels
        self.group_norm = nn.GroupNorm(norm_num_groups, channels, eps=eps, affine=True)
        self.query = nn.Linear(channels, channels)
        self.key = nn.Linear(channels, channels)
        self.value = nn.Linear(channels, channels)
        self.proj_attn = nn.Linear(channels, channels, True)     # third positional arg of the reference is `bias`


class _Sampler2D(_Holder):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)   # Downsample2D overrides stride/padding at run time


class DownEncoderBlock2D(_Holder):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, groups)
                                      for i in range(num_layers)])
        if add_downsample:
            ds = _Sampler2D(out_channels)
            ds.conv = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=0)
            self.downsamplers = nn.ModuleList([ds])
        else:
            self.downsamplers = None


class UpDecoderBlock2D(_Holder):
    def __init__(self, in_channels, out_channels, num_layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, groups)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([_Sampler2D(out_channels)]) if add_upsample else None


class UNetMidBlock2D(_Holder):
    def __init__(self, channels, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(channels, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, groups), ResnetBlock2D(channels, channels, groups)])


class Encoder(_Holder):
    def __init__(self, in_channels, out_channels, down_block_types, block_out_channels, layers_per_block, norm_num_groups, double_z=True):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList([])
        oc = block_out_channels[0]
        for i, typ in enumerate(down_block_types):
            if typ not in ("DownEncoderBlock2D", "UNetResDownEncoderBlock2D"):
                raise ValueError(f"{typ} does not exist.")
            ic, oc = oc, block_out_channels[i]
            self.down_blocks.append(DownEncoderBlock2D(ic, oc, layers_per_block, norm_num_groups, i != len(block_out_channels) - 1))
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[-1], eps=VAE_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)


class Decoder(_Holder):
    def __init__(self, in_channels, out_channels, up_block_types, block_out_channels, layers_per_block, norm_num_groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], 3, stride=1, padding=1)
        self.up_blocks = nn.ModuleList([])
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], norm_num_groups)
        rev = list(reversed(block_out_channels))
        oc = rev[0]
        for i, typ in enumerate(up_block_types):
            if typ not in ("UpDecoderBlock2D", "UNetResUpDecoderBlock2D"):
                raise ValueError(f"{typ} does not exist.")
            pc, oc = oc, rev[i]
            self.up_blocks.append(UpDecoderBlock2D(pc, oc, layers_per_block + 1, norm_num_groups, i != len(block_out_channels) - 1))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[0], eps=VAE_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types: Tuple[str] = ("DownEncoderBlock2D",),
                 up_block_types: Tuple[str] = ("UpDecoderBlock2D",), block_out_channels: Tuple[int] = (64,),
                 layers_per_block: int = 1, act_fn: str = "silu", latent_channels: int = 4, norm_num_groups: int = 32,
                 sample_size: int = 32, scaling_factor: float = 0.18215, precision: str = "bf16"):
        super().__init__()
        if act_fn not in ("silu", "swish"):
            raise NotImplementedError(f"act_fn={act_fn!r}: only SiLU is fused in the GroupNorm kernel")
        if precision not in ("bf16", "fp16", "fp16x2", "fp16x2_lin", "fp32", "fp8", "fp8_conv"):
            raise ValueError("precision must be 'bf16', 'fp16', 'fp16x2', 'fp16x2_lin', 'fp32', 'fp8' or 'fp8_conv'")
        if precision in ("fp16x2", "fp16x2_lin"):
            # the denoiser's folded-weight engine (exact weights against the error that ACCUMULATES over the sampling steps); the VAE runs once
            # per sample, its weight rounding does not accumulate: it keeps the one-product fp16 engine
            precision = "fp16"
        # "fp8" is a denoiser option (e4m3 operands for its 3x3x3 convolutions); the VAE has no such launches and runs its bf16 engine.
        # "fp16": the bf16 engine on IEEE-half operands (see CuboidTransformerUNet)
        self.operand = "fp16" if precision == "fp16" else "bf16"
        self.opts = L.CallOpts(self.operand)          # per-call options of every launch of this module (nothing is process-global)
        self.op_dtype = self.opts.dtype
        self.precision = "bf16" if (precision.startswith("fp8") or precision == "fp16") else precision
        self.precision_name = precision
        self.fuse_resblock = True     # bf16 engine: GroupNorm -> SiLU -> Conv2d 3x3 of the ResBlocks as one launch (csrc/conv2d_gn.hip)
        self.latent_channels, self.norm_num_groups = latent_channels, norm_num_groups
        self.encoder = Encoder(in_channels, latent_channels, down_block_types, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, up_block_types, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.use_slicing = False
        self.requires_grad_(False)
        self._packed, self._packed_key, self._ws = None, None, {}

    # ------------------------------------------------------------------------------------------------ packing / workspace
    def _pack(self, device):
        split = self.precision == "fp32"
        P = {}
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                P[name + ".w"] = pack_conv(m.weight.to(device), split, dtype=self.op_dtype)
                P[name + ".b"] = m.bias.detach().float().contiguous().to(device)
            elif isinstance(m, nn.Linear):
                P[name + ".w"] = pack_linear(m.weight.to(device), split, dtype=self.op_dtype)
                P[name + ".b"] = m.bias.detach().float().contiguous().to(device)
            elif isinstance(m, nn.GroupNorm):
                P[name + ".g"] = m.weight.detach().float().contiguous().to(device)
                P[name + ".beta"] = m.bias.detach().float().contiguous().to(device)
        return P

    def _ensure_packed(self, device):
        key = (str(device), self.precision, self.operand) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key != self._packed_key:
            L.lib()
            self._packed, self._packed_key = self._pack(device), key
        return self._packed

    def _buf(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype, str(device))
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=device)
            self._ws[key] = t
        return t

    def _bf(self, name, rows, cols, device):
        if self.precision == "fp32":      # both halves in one allocation (the 256 x 256 hi/lo kernel reads them through one buffer descriptor)
            both = self._buf(name + ".hilo", (2, rows, cols), torch.bfloat16, device)
            return both[0], both[1]
        return self._buf(name, (rows, cols), self.op_dtype, device), None

    # ------------------------------------------------------------------------------------------------ primitives
    def _cast(self, x, rows, C, name, dev):
        ld = pad64(C)
        a, alo = self._bf(name, rows, ld, dev)
        L.cast_rows(x, a, alo, 1, rows, 0, rows, C, C, ld, opts=self.opts)
        return a, alo, ld

    def _gn(self, P, name, x, N, S, C, dev, silu=True):
        ld = pad64(C)
        a, alo = self._bf("gn.a", N * S, ld, dev)
        part = self._buf("gn.part", (N * L.groupnorm_nchunk(S, C) * self.norm_num_groups * 2,), torch.float64, dev)
        L.groupnorm_silu(x, P[name + ".g"], P[name + ".beta"], part, a, alo, N, S, C, self.norm_num_groups, ld, VAE_EPS, silu=silu, opts=self.opts)
        return a, alo, ld

    def _conv(self, P, name, a, alo, ld, N, hw, Cout, out, dev, k=3, mode="same", residual=None):
        """Conv2d on channels-last rows.  mode: same | down (pad (0,1,0,1), stride 2) | up (nearest x2 then 3x3 pad 1)."""
        H, W = hw
        w, wlo = P[name + ".w"]
        if k == 1:
            geom, taps, Ho, Wo = None, 1, H, W
        elif mode == "same":
            geom, taps, Ho, Wo = L.conv_geom(N, (1, H, W), (1, 3, 3), pad=(0, 1, 1)), 9, H, W
        elif mode == "down":
            Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
            geom, taps = L.conv_geom(N, (1, H, W), (1, 3, 3), stride=(1, 2, 2), pad=(0, 0, 0), out_thw=(1, Ho, Wo)), 9
        elif mode == "up":
            Ho, Wo = 2 * H, 2 * W
            geom, taps = L.conv_geom(N, (1, H, W), (1, 3, 3), pad=(0, 1, 1), up=(1, 2, 2)), 9
        else:
            raise ValueError(mode)
        L.igemm(a, w, A_lo=alo, W_lo=wlo, M=N * Ho * Wo, N=Cout, Cin=ld, taps=taps, w_tap_stride=Cout * ld, geom=geom,
                bias=P[name + ".b"], residual=residual, out_f32=out, opts=self.opts)
        return Ho, Wo

    def _gn_conv(self, P, gn_name, conv_name, x, N, hw, Cin, Cout, out, dev, residual=None):
        """GroupNorm -> SiLU -> Conv2d 3x3 (+ residual): ONE fused launch behind the statistics pass where the geometry allows
        (bf16 engine; csrc/conv2d_gn.hip: the halo of a pixel tile is normalised into LDS once and read by all nine taps), else the
        apply pass + implicit GEMM."""
        H, W = hw
        G = self.norm_num_groups
        if self.precision == "bf16" and self.fuse_resblock and L.conv2d_gn_silu_supported(H, W, Cin, Cout, G) and pad64(Cin) == Cin:
            S = H * W
            part = self._buf("gn.part", (N * L.groupnorm_nchunk(S, Cin) * G * 2,), torch.float64, dev)
            stats = self._buf("gn.stats", (N, G, 2), torch.float32, dev)
            L.groupnorm_stats(x, part, stats, N, S, Cin, G, VAE_EPS)
            L.conv2d_gn_silu(x, stats, P[gn_name + ".g"], P[gn_name + ".beta"], P[conv_name + ".w"][0], P[conv_name + ".b"], residual, out,
                             N, H, W, Cin, Cout, G, opts=self.opts)
            return
        a, alo, ld = self._gn(P, gn_name, x, N, H * W, Cin, dev)
        self._conv(P, conv_name, a, alo, ld, N, hw, Cout, out, dev, residual=residual)

    def _resnet(self, P, name, m: ResnetBlock2D, x, N, hw, dev):
        """ResnetBlock2D.forward, temb=None (taming/resnet.py:454-495).  x: fp32 (N*S, Cin) -> fp32 (N*S, Cout)."""
        S = hw[0] * hw[1]
        Cin, Cout = m.in_channels, m.out_channels
        h = self._buf("res.h", (N * S, Cout), torch.float32, dev)
        self._gn_conv(P, name + ".norm1", name + ".conv1", x, N, hw, Cin, Cout, h, dev)
        if m.conv_shortcut is None:
            self._gn_conv(P, name + ".norm2", name + ".conv2", h, N, hw, Cout, Cout, x, dev, residual=x)
            return x
        out = self._buf(f"res.out{Cout}", (N * S, Cout), torch.float32, dev)
        xa, xalo, ldx = self._cast(x, N * S, Cin, "sc.a", dev)
        self._conv(P, name + ".conv_shortcut", xa, xalo, ldx, N, hw, Cout, out, dev, k=1)
        self._gn_conv(P, name + ".norm2", name + ".conv2", h, N, hw, Cout, Cout, out, dev, residual=out)
        return out

    def _attention(self, P, name, x, N, S, C, dev):
        """AttentionBlock.forward (taming/attention.py:136-189): x += proj(softmax(q k^T / sqrt(C)) v)."""
        if S % 64 or C % 64:
            raise NotImplementedError("VAE mid attention needs H*W and C to be multiples of 64")
        h, hlo, ld = self._gn(P, name + ".group_norm", x, N, S, C, dev, silu=False)
        split = self.precision == "fp32"
        q, qlo = self._bf("at.q", N * S, C, dev)
        k, klo = self._bf("at.k", N * S, C, dev)
        for nm, (o, olo) in (("query", (q, qlo)), ("key", (k, klo))):
            w, wlo = P[f"{name}.{nm}.w"]
            L.igemm(h, w, A_lo=hlo, W_lo=wlo, M=N * S, N=C, Cin=ld, bias=P[f"{name}.{nm}.b"], out_bf16=o, out_bf16_lo=olo, opts=self.opts)
        # V^T per frame: vt[c, s] = sum_k Wv[c, k] h[s, k]   (the value bias is added after P V: softmax rows sum to 1)
        vt, vtlo = self._bf("at.vt", N * C, S, dev)
        wv, wvlo = P[name + ".value.w"]
        L.igemm(wv, h, A_lo=wvlo, W_lo=hlo, M=C, N=S, Cin=ld, lda=ld, ldw=ld, nbatch=N, a_batch_stride=0, w_batch_stride=S * ld,
                out_bf16=vt, out_bf16_lo=vtlo, ld_outb=S, outb_batch_stride=C * S, opts=self.opts)
        sc = self._buf("at.sc", (N * S, S), torch.float32, dev)
        L.igemm(q, k, A_lo=qlo, W_lo=klo, M=S, N=S, Cin=C, nbatch=N, a_batch_stride=S * C, w_batch_stride=S * C,
                alpha=1.0 / math.sqrt(C), out_f32=sc, out_batch_stride=S * S, opts=self.opts)
        p, plo = self._bf("at.p", N * S, S, dev)
        L.softmax_rows(sc, p, plo, N * S, S, S, S, opts=self.opts)
        o, olo = self._bf("at.o", N * S, C, dev)
        L.igemm(p, vt, A_lo=plo, W_lo=vtlo, M=S, N=C, Cin=S, nbatch=N, a_batch_stride=S * S, w_batch_stride=C * S,
                bias=P[name + ".value.b"], out_bf16=o, out_bf16_lo=olo, outb_batch_stride=S * C, opts=self.opts)
        wp, wplo = P[name + ".proj_attn.w"]
        L.igemm(o, wp, A_lo=olo, W_lo=wplo, M=N * S, N=C, Cin=C, bias=P[name + ".proj_attn.b"], residual=x, out_f32=x, opts=self.opts)
        return x

    def _mid(self, P, name, mid: UNetMidBlock2D, x, N, hw, C, dev):
        x = self._resnet(P, name + ".resnets.0", mid.resnets[0], x, N, hw, dev)
        x = self._attention(P, name + ".attentions.0", x, N, hw[0] * hw[1], C, dev)
        return self._resnet(P, name + ".resnets.1", mid.resnets[1], x, N, hw, dev)

    def _input(self, x, dev):
        if not x.is_cuda:
            raise L.PrediffHipError("prediff_amd.AutoencoderKL runs only on an MI355X (HIP) device; there is no CPU path")
        N, C, H, W = x.shape
        xl = self._buf("in.nhwc", (N * H * W, C), torch.float32, dev)
        L.nchw_to_nhwc(x.contiguous().float(), xl, N, C, H * W, C)
        return xl, N, C, (H, W)

    # ------------------------------------------------------------------------------------------------ public API
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> DiagonalGaussianDistribution:
        """taming/autoencoder_kl.py:80-84 (Encoder.forward taming/vae.py:70-86)."""
        with L.on_device(x):
            return self._encode(x)

    def _encode(self, x: torch.Tensor) -> DiagonalGaussianDistribution:
        dev = x.device
        P = self._ensure_packed(dev)
        xl, N, C, hw = self._input(x, dev)
        enc = self.encoder
        a, alo, ld = self._cast(xl, N * hw[0] * hw[1], C, "cast.a", dev)
        C = enc.conv_in.out_channels
        cur = self._buf("enc.x0", (N * hw[0] * hw[1], C), torch.float32, dev)
        self._conv(P, "encoder.conv_in", a, alo, ld, N, hw, C, cur, dev)
        for b, blk in enumerate(enc.down_blocks):
            for r, rn in enumerate(blk.resnets):
                cur = self._resnet(P, f"encoder.down_blocks.{b}.resnets.{r}", rn, cur, N, hw, dev)
                C = rn.out_channels
            if blk.downsamplers is not None:
                a, alo, ld = self._cast(cur, N * hw[0] * hw[1], C, "cast.a", dev)
                Ho, Wo = (hw[0] - 2) // 2 + 1, (hw[1] - 2) // 2 + 1
                nxt = self._buf(f"enc.ds{b}", (N * Ho * Wo, C), torch.float32, dev)
                hw = self._conv(P, f"encoder.down_blocks.{b}.downsamplers.0.conv", a, alo, ld, N, hw, C, nxt, dev, mode="down")
                cur = nxt
        cur = self._mid(P, "encoder.mid_block", enc.mid_block, cur, N, hw, C, dev)
        S = hw[0] * hw[1]
        a, alo, ld = self._gn(P, "encoder.conv_norm_out", cur, N, S, C, dev)
        Cz = enc.conv_out.out_channels
        hz = self._buf("enc.z", (N * S, Cz), torch.float32, dev)
        self._conv(P, "encoder.conv_out", a, alo, ld, N, hw, Cz, hz, dev)
        a, alo, ld = self._cast(hz, N * S, Cz, "cast.a", dev)
        mom = self._buf("enc.mom", (N * S, Cz), torch.float32, dev)
        self._conv(P, "quant_conv", a, alo, ld, N, hw, Cz, mom, dev, k=1)
        out = torch.empty((N, Cz, hw[0], hw[1]), dtype=torch.float32, device=dev)
        L.nhwc_to_nchw(mom, out, N, Cz, S, Cz)
        return DiagonalGaussianDistribution(out)

    @torch.no_grad()
    def _decode(self, z: torch.Tensor) -> torch.Tensor:
        """taming/autoencoder_kl.py:86-89 (Decoder.forward taming/vae.py:150-166)."""
        with L.on_device(z):
            return self._decode_impl(z)

    def _decode_impl(self, z: torch.Tensor) -> torch.Tensor:
        dev = z.device
        P = self._ensure_packed(dev)
        zl, N, Cz, hw = self._input(z, dev)
        dec = self.decoder
        S = hw[0] * hw[1]
        a, alo, ld = self._cast(zl, N * S, Cz, "cast.a", dev)
        zq = self._buf("dec.zq", (N * S, Cz), torch.float32, dev)
        self._conv(P, "post_quant_conv", a, alo, ld, N, hw, Cz, zq, dev, k=1)
        a, alo, ld = self._cast(zq, N * S, Cz, "cast.a", dev)
        C = dec.conv_in.out_channels
        cur = self._buf("dec.x0", (N * S, C), torch.float32, dev)
        self._conv(P, "decoder.conv_in", a, alo, ld, N, hw, C, cur, dev)
        cur = self._mid(P, "decoder.mid_block", dec.mid_block, cur, N, hw, C, dev)
        for b, blk in enumerate(dec.up_blocks):
            for r, rn in enumerate(blk.resnets):
                cur = self._resnet(P, f"decoder.up_blocks.{b}.resnets.{r}", rn, cur, N, hw, dev)
                C = rn.out_channels
            if blk.upsamplers is not None:
                nxt = self._buf(f"dec.us{b}", (N * 4 * hw[0] * hw[1], C), torch.float32, dev)
                uname = f"decoder.up_blocks.{b}.upsamplers.0.conv"
                if (self.precision == "bf16" and self.fuse_resblock and pad64(C) == C
                        and L.conv2d_gn_silu_supported(2 * hw[0], 2 * hw[1], C, C, 1)):
                    # Upsample2D in one launch of the fused tile kernel: the halo comes straight from the half-resolution fp32 rows
                    L.conv2d_up2(cur, P[uname + ".w"][0], P[uname + ".b"], nxt, N, 2 * hw[0], 2 * hw[1], C, C, opts=self.opts)
                    hw = (2 * hw[0], 2 * hw[1])
                else:
                    a, alo, ld = self._cast(cur, N * hw[0] * hw[1], C, "cast.a", dev)
                    hw = self._conv(P, uname, a, alo, ld, N, hw, C, nxt, dev, mode="up")
                cur = nxt
        S = hw[0] * hw[1]
        a, alo, ld = self._gn(P, "decoder.conv_norm_out", cur, N, S, C, dev)
        Co = dec.conv_out.out_channels
        y = self._buf("dec.y", (N * S, Co), torch.float32, dev)
        self._conv(P, "decoder.conv_out", a, alo, ld, N, hw, Co, y, dev)
        out = torch.empty((N, Co, hw[0], hw[1]), dtype=torch.float32, device=dev)
        L.nhwc_to_nchw(y, out, N, Co, S, Co)
        return out

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        if self.use_slicing and z.shape[0] > 1:
            return torch.cat([self._decode(s) for s in z.split(1)])
        return self._decode(z)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_posterior: bool = False,
                generator: Optional[torch.Generator] = None):
        posterior = self.encode(sample)
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        dec = self.decode(z)
        return (dec, posterior) if return_posterior else dec
