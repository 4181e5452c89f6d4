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
