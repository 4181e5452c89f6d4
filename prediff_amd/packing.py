"""Weight packing for libprediff_hip: fp32 checkpoint tensors -> K-contiguous bf16 (hi[/lo]) GEMM operands.

Done once at load time (and again whenever the parameters change).  Layout consumed by pd_igemm:
    W_packed[tap][n][c]   with c zero padded to a multiple of 64
so that for one filter tap every output channel's C_in weights are contiguous (the MFMA K dimension).
"""
from typing import Optional, Tuple

import torch


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def split_bf16(x: torch.Tensor, want_lo: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x (fp32) -> (hi, lo) bf16 with x ~= hi + lo (about 16 mantissa bits); lo is None unless requested."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16) if want_lo else None
    return hi.contiguous(), (lo.contiguous() if lo is not None else None)


def pack_linear(weight: torch.Tensor, split: bool, k_pad: Optional[int] = None):
    """nn.Linear weight (N, K) -> bf16 (N, Kp)."""
    N, K = weight.shape
    Kp = k_pad if k_pad is not None else pad64(K)
    w = torch.zeros(N, Kp, dtype=torch.float32, device=weight.device)
    w[:, :K] = weight.detach().float()
    return split_bf16(w, split)


def pack_conv(weight: torch.Tensor, split: bool, c_pad: Optional[int] = None):
    """Conv weight (N, C, *kernel) -> bf16 (taps, N, Cp) with taps enumerated kernel-index-major (kt, kh, kw)."""
    N, Cn = weight.shape[:2]
    taps = 1
    for k in weight.shape[2:]:
        taps *= k
    Cp = c_pad if c_pad is not None else pad64(Cn)
    w = torch.zeros(taps, N, Cp, dtype=torch.float32, device=weight.device)
    w[:, :, :Cn] = weight.detach().float().reshape(N, Cn, taps).permute(2, 0, 1)
    return split_bf16(w, split)


# ---------------------------------------------------------------------------------------------------- fp8 (OCP e4m3) operands
FP8_MAX = 448.0          # largest finite e4m3fn value


def pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def fp8_weight_scale(w: torch.Tensor) -> float:
    """Per-tensor power-of-two scale that puts max|w| just below the top of the e4m3 range (powers of two keep the scaling exact)."""
    import math
    amax = float(w.detach().abs().max())
    if amax == 0.0 or not math.isfinite(amax):
        return 1.0
    return 2.0 ** math.floor(math.log2(FP8_MAX / amax))


def to_fp8(x: torch.Tensor, scale: float) -> torch.Tensor:
    """round-to-nearest-even, saturating cast of x * scale to e4m3fn"""
    return (x.float() * scale).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).contiguous()


def pack_conv_fp8(weight: torch.Tensor, c_pad: Optional[int] = None):
    """Conv weight (N, C, *kernel) -> (e4m3 (taps, N, Cp) with Cp % 128 == 0, scale): stored value = weight * scale."""
    N, Cn = weight.shape[:2]
    taps = 1
    for k in weight.shape[2:]:
        taps *= k
    Cp = c_pad if c_pad is not None else pad128(Cn)
    w = torch.zeros(taps, N, Cp, dtype=torch.float32, device=weight.device)
    w[:, :, :Cn] = weight.detach().float().reshape(N, Cn, taps).permute(2, 0, 1)
    s = fp8_weight_scale(w)
    return to_fp8(w, s), s


def pack_linear_fp8(weight: torch.Tensor, k_pad: Optional[int] = None):
    N, K = weight.shape
    Kp = k_pad if k_pad is not None else pad128(K)
    w = torch.zeros(N, Kp, dtype=torch.float32, device=weight.device)
    w[:, :K] = weight.detach().float()
    s = fp8_weight_scale(w)
    return to_fp8(w, s), s
