"""Weight packing for libprediff_hip: fp32 checkpoint tensors -> K-contiguous bf16 (hi[/lo]) GEMM operands.

Done once at load time (and again whenever the parameters change).  Layout consumed by pd_igemm:
    W_packed[tap][n][c]   with c zero padded to a multiple of 64
so that for one filter tap every output channel's C_in weights are contiguous (the MFMA K dimension).
"""
from typing import Optional, Tuple

import torch


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


F16_MAX = 65504.0


def to_operand(x: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """fp32 -> the engine's 16-bit operand type, round to nearest even; IEEE half saturates at +-65504 like the kernels' packers."""
    if dtype == torch.float16:
        return x.float().clamp(-F16_MAX, F16_MAX).to(torch.float16)
    return x.to(dtype)


def split_bf16(x: torch.Tensor, want_lo: bool, dtype=torch.bfloat16) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x (fp32) -> (hi, lo) with x ~= hi + lo (about 16 mantissa bits; bf16 only); lo is None unless requested.  dtype: the operand
    type of a single-pass engine (torch.bfloat16, or torch.float16 for precision="fp16")."""
    if want_lo and dtype != torch.bfloat16:
        raise ValueError("the hi/lo split exists for bfloat16 operands only")
    hi = to_operand(x, dtype)
    if not want_lo:
        return hi.contiguous(), None
    # both halves in ONE allocation (hi = both[0], lo = both[1]): the 256 x 256 hi/lo kernel reads them through one buffer descriptor
    both = torch.empty((2,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    both[0].copy_(hi)
    both[1].copy_((x - hi.float()).to(torch.bfloat16))
    return both[0], both[1]


def fold_weights(w: torch.Tensor, dtype=torch.float16) -> torch.Tensor:
    """fp32 packed weights (taps, N, Kp) -> (2 taps, N, Kp) 16-bit operands: the slabs of W_hi = round(w) followed by the slabs of
    W_lo = round(w - W_hi) -- pd_igemm_args.w_fold (precision="fp16x2": D = A W_hi^T + A W_lo^T, the weights exact to ~2^-22, the
    activations rounded once).  The returned tensor is tagged `_pd_fold` for prediff_amd._lib.igemm."""
    hi = to_operand(w, dtype)
    lo = to_operand(w - hi.float(), dtype)
    out = torch.cat([hi, lo], dim=0).contiguous()
    out._pd_fold = True
    return out


def pack_linear(weight: torch.Tensor, split: bool, k_pad: Optional[int] = None, dtype=torch.bfloat16, fold: bool = False):
    """nn.Linear weight (N, K) -> 16-bit operands (N, Kp)."""
    N, K = weight.shape
    Kp = k_pad if k_pad is not None else pad64(K)
    w = torch.zeros(N, Kp, dtype=torch.float32, device=weight.device)
    w[:, :K] = weight.detach().float()
    if fold:
        return fold_weights(w[None], dtype), None
    return split_bf16(w, split, dtype)


def pack_conv(weight: torch.Tensor, split: bool, c_pad: Optional[int] = None, dtype=torch.bfloat16, fold: bool = False):
    """Conv weight (N, C, *kernel) -> 16-bit operands (taps, N, Cp) with taps enumerated kernel-index-major (kt, kh, kw)."""
    N, Cn = weight.shape[:2]
    taps = 1
    for k in weight.shape[2:]:
        taps *= k
    Cp = c_pad if c_pad is not None else pad64(Cn)
    w = torch.zeros(taps, N, Cp, dtype=torch.float32, device=weight.device)
    w[:, :, :Cn] = weight.detach().float().reshape(N, Cn, taps).permute(2, 0, 1)
    if fold:
        return fold_weights(w, dtype), None
    return split_bf16(w, split, dtype)


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


# ---------------------------------------------------------------------------------------------------- (attention, FFN) pair kernel
PAIR_CHUNK_BYTES = 32768          # csrc/pair_block.hip: 32 fragments of 1 KB
PAIR_VEC_FLOATS = 3584          # units 256; 6144 at units 512


def _mfma_frags(w: torch.Tensor) -> torch.Tensor:
    """16-bit operands (N, K) -> (N/16, K/32, 64 lanes, 8) fragments of v_mfma_f32_16x16x32_{bf16,f16} with the k order of pd_attn_ffn_pair:
    lane = 16 * kg + f holds W[16 F + f][32 KB + 16 (j >> 2) + 4 kg + (j & 3)], j = 0..7 -- the order in which the PREVIOUS stage's
    accumulator registers (lane = (token, 4 consecutive features per 16-feature tile)) line up as the other MFMA operand."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    v = w.reshape(N // 16, 16, K // 32, 2, 4, 4)          # F, f, KB, jh, kg, jl   (k = 32 KB + 16 jh + 4 kg + jl)
    v = v.permute(0, 2, 4, 1, 3, 5)                        # F, KB, kg, f, jh, jl
    return v.reshape(N // 16, K // 32, 64, 8).contiguous()


def pair_cuboids_per_group(vol: int) -> int:
    """cuboids that share one 16-slot group of pd_attn_ffn_pair (== pd_attn_ffn_pair_cuboids_per_group)"""
    return 2 if vol >= 1 and 2 * vol <= 16 else 1


def _hi_lo(t: torch.Tensor, dtype):
    """fp32 -> (round(t), round(t - round(t))) in the 16-bit operand type: the two weight images of a folded stream"""
    hi = to_operand(t.detach().float(), dtype)
    return hi, to_operand(t.detach().float() - hi.float(), dtype)


def pack_pair_block(wqkv: torch.Tensor, wproj: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, dtype=torch.bfloat16, fold: bool = False) -> torch.Tensor:
    """The weight stream of pd_attn_ffn_pair for units C = 256 (4 heads of 64, hidden 1024) or 512 (4 heads of 128, hidden 2048):
    chunks of 32 KB = 32 fragments in consumption order.  With CW = C / 256, HD = C / 4, DT = HD / 16, CT = C / 16:
      per head h: Wq_h, Wk_h, Wv_h ([HD x C], CW^2 chunks each: fragment i of chunk s = feature tile i % DT, k-step s * 32 / DT + i / DT),
                  Wproj[:, HD h : HD h + HD] ([C x HD], CW^2 chunks: column tile i % CT, k-step s * 32 / CT + i / CT);
      then the 64-wide hidden slices W1_0, W1_1, (W2_j, W1_{j+2}) for j = 0 .. hidden/64 - 3, W2_{n-2}, W2_{n-1}
      (W1_j = rows 64 j .. of (hidden, C), CW chunks: hidden tile i & 3, k-step 8 s + i / 4;  W2_j = columns 64 j .. of (C, hidden), CW
      chunks: column tile i % CT, k-step s * 32 / CT + i / CT): gelu(h_j) runs beside the chunks between W1_j and W2_j.
    fold (precision="fp16x2", pd_call_opts.w_fold): every such group of chunks is followed by the same group of the LOW parts
    (W = W_hi + W_lo in the operand type): twice the chunks, the kernel multiplies both into the same accumulators."""
    Cn = wproj.shape[0]
    assert Cn in (256, 512)
    CW, hid = Cn // 256, 4 * Cn
    assert tuple(wqkv.shape) == (3 * Cn, Cn) and tuple(wproj.shape) == (Cn, Cn) and tuple(w1.shape) == (hid, Cn) and tuple(w2.shape) == (Cn, hid)
    HD, CT = Cn // 4, Cn // 16
    DT, HS = HD // 16, HD // 32
    parts = [_hi_lo(t, dtype) if fold else (to_operand(t.detach(), dtype),) for t in (wqkv, wproj, w1, w2)]
    fq, fp, f1, f2 = ([_mfma_frags(x) for x in pr] for pr in parts)          # per matrix: [hi] or [hi, lo] fragment tensors

    def tile_chunks(frs, F0, nF, K0, nK):
        """fragments [F0, F0 + nF) x [K0, K0 + nK) -> chunks of 32 with i = nF * k_local + f (the hi image, then the lo image)"""
        out = []
        for fr in frs:
            v = fr[F0:F0 + nF, K0:K0 + nK].permute(1, 0, 2, 3).reshape(nF * nK, 64, 8)
            out += list(v.reshape(nF * nK // 32, 32, 64, 8))
        return out

    chunks = []
    for h in range(4):
        for kind in range(3):
            chunks += tile_chunks(fq, (kind * Cn + HD * h) // 16, DT, 0, Cn // 32)
        chunks += tile_chunks(fp, 0, CT, HS * h, HS)
    nj = hid // 64
    w1s = lambda j: tile_chunks(f1, 4 * j, 4, 0, Cn // 32)
    w2s = lambda j: tile_chunks(f2, 0, CT, 2 * j, 2)
    chunks += w1s(0) + w1s(1)
    for j in range(nj):
        chunks += w2s(j)
        if j + 2 < nj:
            chunks += w1s(j + 2)
    out = torch.stack(chunks).contiguous()
    assert out.shape[0] == (16 * CW * CW + 2 * nj * CW) * (2 if fold else 1) and out.numel() * 2 == out.shape[0] * PAIR_CHUNK_BYTES
    return out


def pack_pair_ffn_split(w1: torch.Tensor, w2: torch.Tensor, dtype=torch.bfloat16, nsplit: int = 4, fold: bool = False) -> torch.Tensor:
    """The FFN part of pack_pair_block's stream re-ordered for pd_attn_ffn_pair_split (units 512): `nsplit` independent sub-streams, one
    per quarter q of the hidden units (64-wide slices j = q n .. q n + n - 1, n = hidden / 64 / nsplit), each in the software-pipelined
    order of the kernel's FFN loop: W1_0', W1_1', (W2_j', W1_{j+2}') for j' = 0 .. n - 3, W2_{n-2}', W2_{n-1}' (primes = local slices)."""
    Cn, hid = w2.shape
    assert Cn in (256, 512) and tuple(w1.shape) == (hid, Cn) and hid == 4 * Cn
    CT = Cn // 16
    f1, f2 = ([_mfma_frags(x) for x in (_hi_lo(t, dtype) if fold else (to_operand(t.detach(), dtype),))] for t in (w1, w2))

    def tile_chunks(frs, F0, nF, K0, nK):
        out = []
        for fr in frs:
            v = fr[F0:F0 + nF, K0:K0 + nK].permute(1, 0, 2, 3).reshape(nF * nK, 64, 8)
            out += list(v.reshape(nF * nK // 32, 32, 64, 8))
        return out

    nj = hid // 64
    n = nj // nsplit
    assert nj % nsplit == 0 and n >= 3
    w1s = lambda j: tile_chunks(f1, 4 * j, 4, 0, Cn // 32)
    w2s = lambda j: tile_chunks(f2, 0, CT, 2 * j, 2)
    chunks = []
    for q in range(nsplit):
        j0 = q * n
        chunks += w1s(j0) + w1s(j0 + 1)
        for j in range(n):
            chunks += w2s(j0 + j)
            if j + 2 < n:
                chunks += w1s(j0 + j + 2)
    out = torch.stack(chunks).contiguous()
    assert out.shape[0] == 2 * nj * (Cn // 256) * (2 if fold else 1) and out.numel() * 2 == out.shape[0] * PAIR_CHUNK_BYTES
    return out


def pack_pair_vecs(ln1_g, ln1_b, bproj, ln2_g, ln2_b, b2, b1, rel_bias) -> torch.Tensor:
    """fp32 tables of pd_attn_ffn_pair: LN1 gamma, beta, proj bias, LN2 gamma, beta, FFN-2 bias (units each), FFN-1 bias (hidden), and the
    (4 heads, 16, 16) score table of a 16-slot group: the relative-position bias (4, vol, vol) on the diagonal block of every cuboid the
    group holds (pair_cuboids_per_group(vol)), -inf everywhere else -- a padded slot, or a key of the group's other cuboid."""
    dev = ln1_g.device
    Cn = ln1_g.numel()
    zc = torch.zeros(Cn, device=dev)
    vol = rel_bias.shape[-1]
    rb = torch.full((4, 16, 16), float("-inf"), device=dev)
    for a in range(pair_cuboids_per_group(vol)):
        rb[:, a * vol:(a + 1) * vol, a * vol:(a + 1) * vol] = rel_bias.detach().float()
    parts = [ln1_g, ln1_b, bproj if bproj is not None else zc, ln2_g, ln2_b, b2 if b2 is not None else zc,
             b1 if b1 is not None else torch.zeros(4 * Cn, device=dev), rb.reshape(-1)]
    v = torch.cat([t.detach().float().reshape(-1).to(dev) for t in parts]).contiguous()
    assert v.numel() == 10 * Cn + 1024
    return v
