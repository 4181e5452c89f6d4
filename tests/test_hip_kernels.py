"""GPU: every C-ABI kernel of libprediff_hip.so against a plain PyTorch fp32 statement of the same op
(run on the bf16-rounded operands for the bf16 MFMA paths, on the fp32 operands for the split / fp32 paths)
and against the oracle for the attention core.  Tolerances are written at each check.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from prediff_amd import _lib as L  # noqa: E402
from prediff_amd.packing import pack_conv, pack_linear, split_bf16  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf(x):
    return x.to(torch.bfloat16).float()


def padded_bf16(x2d, split=False):
    """(rows, K) fp32 -> (rows, pad64(K)) bf16 hi[, lo]"""
    rows, K = x2d.shape
    Kp = L.pad64(K)
    xp = torch.zeros(rows, Kp, device=x2d.device)
    xp[:, :K] = x2d
    return split_bf16(xp, split)


# ------------------------------------------------------------------------------------------------ igemm: linear
@pytest.mark.parametrize("M,N,K,tile", [(256, 128, 64, 1), (256, 128, 64, 2), (3328, 768, 256, 0), (1000, 200, 96, 1),
                                        (1000, 200, 96, 2), (37, 5, 32, 2), (832, 2048, 512, 0), (128, 64, 1024, 2),
                                        (3328, 768, 256, 3), (3328, 768, 256, 4), (1000, 200, 96, 3), (1000, 200, 96, 4), (300, 130, 64, 3),
                                        (300, 130, 64, 4), (512, 256, 2048, 3), (512, 256, 2048, 4),
                                        (3328, 768, 256, 5), (1000, 200, 96, 5), (300, 130, 64, 6), (512, 256, 2048, 6), (3328, 768, 256, 6),
                                        (3328, 768, 256, 7), (1000, 200, 96, 7), (300, 130, 64, 7), (512, 256, 2048, 7), (37, 5, 32, 7),
                                        (2048, 512, 192, 7), (106496, 256, 128, 7), (26624, 512, 64, 7), (9000, 300, 64, 7), (209, 256, 64, 7), (415, 256, 64, 7), (1000, 200, 96, 8), (1000, 200, 96, 9), (300, 130, 64, 8), (3328, 768, 256, 9)])
@pytest.mark.parametrize("split", [False, True])
def test_igemm_linear(M, N, K, tile, split):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    # asymmetric data: transposes / row-col swaps cannot cancel
    x = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5 + torch.arange(M)[:, None] * 1e-3).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    a_hi, a_lo = padded_bf16(x, split)
    w_hi, w_lo = pack_linear(w, split)
    out = torch.full((M, N), float("nan"), device=DEV)
    outb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    L.igemm(a_hi, w_hi, A_lo=a_lo, W_lo=w_lo, M=M, N=N, Cin=a_hi.shape[1], bias=bias, residual=res, act="gelu",
            out_f32=out, out_bf16=outb, tile=tile)
    torch.cuda.synchronize()
    if split:
        ref = F.gelu(x @ w.t() + bias) + res
        tol = 2e-5          # hi/lo split: fp32-class accuracy
    else:
        ref = F.gelu(bf(x) @ bf(w).t() + bias) + res
        tol = 2e-6          # same bf16 operands, fp32 accumulate: only summation order differs
    assert rel_l2(out, ref) < tol
    assert rel_l2(outb.float(), ref) < 4e-3      # bf16 output rounding


@pytest.mark.parametrize("heavy", [False, True])
@pytest.mark.parametrize("M,N,K,tile", [(3328, 768, 256, 0), (3328, 768, 256, 3), (26624, 512, 2048, 0), (1000, 200, 96, 4), (300, 130, 64, 7),
                                        (512, 2048, 512, 7), (209, 256, 64, 0)])
def test_igemm_linear_folded_weights(M, N, K, tile, heavy):
    """pd_igemm_args.w_fold (precision="fp16x2"): D = A W_hi^T + A W_lo^T with IEEE-half operands -- the activations rounded once, the
    weights exact to ~2^-22 -- against the fp32 product ON THE SAME rounded activations (what is left: fp32 summation order and the
    2^-22 tail of the weights) and against the one-product fp16 launch (which carries the 2^-12 weight rounding the fold removes)."""
    from prediff_amd.packing import pack_linear as PL
    g = torch.Generator(device="cpu").manual_seed(M + 5 * N + K)
    x = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    if heavy:        # Student-t(3) entries spanning six decades (W_lo down in the fp16 subnormals), 30x outlier rows and activation channels
        t3 = torch.randn(N, K, generator=g) / torch.sqrt((torch.randn(3, N, K, generator=g) ** 2).mean(0))
        w = (t3 / math.sqrt(3 * K)).to(DEV)
        w[::37] *= 30.0
        x[:, ::29] *= 30.0
    bias = torch.randn(N, generator=g).to(DEV)
    Kp = (K + 63) // 64 * 64
    a16 = torch.zeros(M, Kp, dtype=torch.float16, device=DEV)
    a16[:, :K] = x.half()
    opts = L.CallOpts("fp16")
    wf, _ = PL(w, False, dtype=torch.float16, fold=True)
    w1, _ = PL(w, False, dtype=torch.float16)
    assert tuple(wf.shape) == (2, N, Kp) and wf._pd_fold
    out, out1 = (torch.full((M, N), float("nan"), device=DEV) for _ in range(2))
    L.igemm(a16, wf, M=M, N=N, Cin=Kp, bias=bias, out_f32=out, tile=tile, opts=opts)
    L.igemm(a16, w1, M=M, N=N, Cin=Kp, bias=bias, out_f32=out1, tile=tile, opts=opts)
    ref = a16[:, :K].float().double() @ w.double().t() + bias.double()
    e2, e1 = rel_l2(out, ref), rel_l2(out1, ref)
    print(f"[igemm folded weights {M}x{N}x{K} tile {tile}] vs the fp64 product on the same fp16 activations: two products {e2:.2e}, one product {e1:.2e}")
    assert e2 < 2e-6 and e1 > 20 * e2
    with pytest.raises(L.PrediffHipError):                 # folded weights with an e4m3 launch: refused by the binding
        L.igemm(a16, wf, M=M, N=N, Cin=Kp, out_f32=out, fp8=True, opts=opts)


@pytest.mark.parametrize("heavy", [False, True])
@pytest.mark.parametrize("B,T,H,W,Cin,Cout,splitk", [(2, 13, 16, 16, 256, 256, False), (1, 13, 16, 16, 256, 256, True), (2, 13, 8, 8, 512, 512, True),
                                                     (3, 5, 6, 7, 64, 128, False)])
def test_igemm_conv3d_folded_weights(B, T, H, W, Cin, Cout, splitk, heavy):
    """The same for the 3x3x3 convolution (54 weight slabs over 27 activation gathers; 256 x 256 kernel, its split-K form, 128 x 128)."""
    from prediff_amd.packing import pack_conv as PC
    g = torch.Generator(device="cpu").manual_seed(B + T + Cin)
    x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
    w[:, :, 0, 1, 2] += 0.02                                # asymmetric taps
    if heavy:        # Student-t(3) filters, 30x outlier output channels, 30x activation channels (what a GroupNorm with outlier gains hands over)
        t3 = torch.randn(w.shape, generator=g) / torch.sqrt((torch.randn((3,) + tuple(w.shape), generator=g) ** 2).mean(0))
        w = (t3 / math.sqrt(3 * 27 * Cin)).to(DEV)
        w[::19] *= 30.0
        x[..., ::23] *= 30.0
    bias = torch.randn(Cout, generator=g).to(DEV)
    a16 = x.reshape(-1, Cin).half().contiguous()
    wf, _ = PC(w, False, dtype=torch.float16, fold=True)
    w1, _ = PC(w, False, dtype=torch.float16)
    assert tuple(wf.shape) == (54, Cout, Cin)
    M = B * T * H * W
    opts = L.CallOpts("fp16")
    ws = torch.empty(16 * 1024 * 1024, device=DEV) if splitk else None
    out, out1 = (torch.full((M, Cout), float("nan"), device=DEV) for _ in range(2))
    kw = dict(M=M, N=Cout, Cin=Cin, taps=27, w_tap_stride=Cout * Cin, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias, splitk_ws=ws, opts=opts)
    L.igemm(a16, wf, out_f32=out, **kw)
    L.igemm(a16, w1, out_f32=out1, **kw)
    ref = F.conv3d(a16.float().double().reshape(B, T, H, W, Cin).permute(0, 4, 1, 2, 3), w.double(), bias.double(), padding=1)
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, Cout)
    e2, e1 = rel_l2(out, ref), rel_l2(out1, ref)
    print(f"[igemm conv3d folded weights B={B} {T}x{H}x{W} {Cin}->{Cout} splitk={splitk}] vs fp64 on the same fp16 activations: two products {e2:.2e}, one {e1:.2e}")
    assert e2 < 2e-6 and e1 > 20 * e2


def test_igemm_rowvec_alpha_mul_period():
    M, N, K = 512, 128, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    rowvec = torch.randn(4, N, generator=g).to(DEV)         # 4 samples x 128 rows
    mul = torch.randn(M, N, generator=g).to(DEV)
    table = torch.randn(128, N, generator=g).to(DEV)        # periodic residual (positional table)
    a_hi, _ = padded_bf16(x)
    w_hi, _ = pack_linear(w, False)
    out = torch.empty(M, N, device=DEV)
    L.igemm(a_hi, w_hi, M=M, N=N, Cin=K, rowvec=rowvec, rows_per_sample=128, mul=mul, residual=table, res_period=128,
            alpha=0.25, act="silu", out_f32=out)
    ref = F.silu(0.25 * (bf(x) @ bf(w).t()) + rowvec.repeat_interleave(128, 0)) * mul + table.repeat(4, 1)
    assert rel_l2(out, ref) < 2e-6


def test_igemm_batched():
    nb, M, N, K = 3, 256, 256, 128
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(nb, M, K, generator=g).to(DEV)
    w = torch.randn(nb, N, K, generator=g).to(DEV)
    a_hi, _ = split_bf16(x, False)
    w_hi, _ = split_bf16(w, False)
    out = torch.empty(nb, M, N, device=DEV)
    L.igemm(a_hi, w_hi, M=M, N=N, Cin=K, nbatch=nb, a_batch_stride=M * K, w_batch_stride=N * K, out_batch_stride=M * N,
            alpha=0.5, out_f32=out)
    ref = 0.5 * torch.einsum("bmk,bnk->bmn", bf(x), bf(w))
    assert rel_l2(out, ref) < 2e-6


# ------------------------------------------------------------------------------------------------ igemm: convolutions
@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 5, 8, 8, 64, 64), (1, 13, 16, 16, 256, 256), (2, 3, 6, 6, 5, 32), (1, 13, 8, 8, 512, 512)])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("tile", [0, 3, 4, 5, 6, 7, 8, 9])
def test_igemm_conv3d(B, T, H, W, Cin, Cout, split, tile):
    g = torch.Generator(device="cpu").manual_seed(B + T + Cin)
    x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    emb = torch.randn(B, Cout, generator=g).to(DEV)
    a_hi, a_lo = padded_bf16(x.reshape(-1, Cin), split)
    w_hi, w_lo = pack_conv(w, split)
    Cp = a_hi.shape[1]
    M = B * T * H * W
    out = torch.empty(M, Cout, device=DEV)
    L.igemm(a_hi, w_hi, A_lo=a_lo, W_lo=w_lo, M=M, N=Cout, Cin=Cp, taps=27, w_tap_stride=Cout * Cp,
            geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias, rowvec=emb, rows_per_sample=T * H * W, out_f32=out, tile=tile)
    xs, ws = (x, w) if split else (bf(x), bf(w))
    ref = F.conv3d(xs.permute(0, 4, 1, 2, 3), ws, bias, padding=1) + emb[:, :, None, None, None]
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, Cout)
    assert rel_l2(out, ref) < (3e-5 if split else 3e-6)


@pytest.mark.parametrize("kind,shape", [("conv3d", (2, 13, 16, 16, 256, 256)), ("conv3d", (3, 13, 8, 8, 512, 512)), ("conv3d", (2, 5, 7, 9, 64, 192)),
                                        ("linear", (3328, 256, 1024)), ("linear", (1000, 300, 2048)), ("linear", (257, 512, 64))])
def test_igemm256_hi_lo_bit_equal_to_128(kind, shape):
    """The hi/lo (precision="fp32") form of the 256 x 256 kernel -- a K-tile of 32 high + 32 low elements per row from the two operand arrays
    through one buffer descriptor, lo.hi + hi.lo + hi.hi per accumulator -- against the 128 x 128 SPLIT kernel: the same products in the
    same order over the same 32-deep k-steps, so the SAME BITS (the fp32-class engine stays batch-size independent whichever kernel a
    launch size selects); against torch fp32; with all epilogue operands (bias, per-sample row vector, residual, hi/lo 16-bit output);
    and operands whose halves are NOT one allocation (far apart or merely separate): tile 7 falls back to the 128 x 128 kernel."""
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    if kind == "conv3d":
        B, T, H, W, Cin, Cout = shape
        x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
        M, N = B * T * H * W, Cout
        a_hi, a_lo = padded_bf16(x.reshape(-1, Cin), True)
        w_hi, w_lo = pack_conv(w, True)
        Cp = a_hi.shape[1]
        kw = dict(M=M, N=N, Cin=Cp, taps=27, w_tap_stride=Cout * Cp, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), rows_per_sample=T * H * W)
        emb = torch.randn(B, N, generator=g).to(DEV)
        ref = F.conv3d(x.permute(0, 4, 1, 2, 3), w, None, padding=1).permute(0, 2, 3, 4, 1).reshape(M, N) + emb.repeat_interleave(T * H * W, 0)
    else:
        M, N, K = shape
        x = torch.randn(M, K, generator=g).to(DEV)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
        a_hi, a_lo = padded_bf16(x, True)
        w_hi, w_lo = pack_linear(w, True)
        kw = dict(M=M, N=N, Cin=a_hi.shape[1], rows_per_sample=M)
        emb = torch.randn(1, N, generator=g).to(DEV)
        ref = x @ w.t() + emb
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    ref = ref + bias + res
    outs = {}
    for tile in (1, 7):
        out = torch.full((M, N), float("nan"), device=DEV)
        ob = torch.full((2, M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        L.igemm(a_hi, w_hi, A_lo=a_lo, W_lo=w_lo, bias=bias, rowvec=emb, residual=res, out_f32=out, out_bf16=ob[0], out_bf16_lo=ob[1], tile=tile, **kw)
        torch.cuda.synchronize()
        outs[tile] = (out, ob.clone())
    assert rel_l2(outs[7][0], ref) < 3e-5
    assert torch.equal(outs[1][0], outs[7][0]) and torch.equal(outs[1][1], outs[7][1])
    assert rel_l2(outs[7][1][0].float() + outs[7][1][1].float(), ref) < 3e-5
    # halves in separate allocations: still right (whichever kernel the library picks for them)
    a_lo2, w_lo2 = a_lo.clone(), w_lo.clone()
    out = torch.full((M, N), float("nan"), device=DEV)
    L.igemm(a_hi, w_hi, A_lo=a_lo2, W_lo=w_lo2, bias=bias, rowvec=emb, residual=res, out_f32=out, tile=7, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, outs[1][0])
    # lo BELOW hi in memory (the descriptor starts at the lower of the two)
    both = torch.empty((2,) + tuple(a_hi.shape), dtype=torch.bfloat16, device=DEV)
    both[1].copy_(a_hi); both[0].copy_(a_lo)
    out = torch.full((M, N), float("nan"), device=DEV)
    L.igemm(both[1], w_hi, A_lo=both[0], W_lo=w_lo, bias=bias, rowvec=emb, residual=res, out_f32=out, tile=7, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, outs[1][0])


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,fp8", [(2, 13, 16, 16, 256, 256, False), (3, 4, 16, 16, 128, 256, False), (1, 13, 32, 16, 256, 256, False),
                                                  (2, 13, 16, 16, 256, 256, True)])
def test_igemm_conv3d_tap_skip_equals_dense(B, T, H, W, Cin, Cout, fp8):
    """The 256 x 256 Conv3d kernel leaves out the temporal taps whose input frame is out of range for a WHOLE tile (tiles inside the
    first / last frame of a sample when H * W is a multiple of 256: 9 of 27 taps).  The dense tap loop (debug_flags bit 8) streams
    those taps as zero rows instead; both must give the same numbers (adding exact zeros) and agree with F.conv3d."""
    import ctypes
    g = torch.Generator(device="cpu").manual_seed(B + T + Cin)
    x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    M = B * T * H * W
    geom = L.conv_geom(B, (T, H, W), (3, 3, 3))
    if fp8:
        from prediff_amd.packing import pack_conv_fp8, to_fp8
        a = to_fp8(x.reshape(-1, Cin), 16.0)
        w_p, sw = pack_conv_fp8(w)
        kw = dict(alpha=1.0 / (16.0 * sw), fp8=True)
    else:
        a, _ = padded_bf16(x.reshape(-1, Cin), False)
        w_p, _ = pack_conv(w, False)
        kw = {}
    outs = {}
    for flag in (0, 8):
        out = torch.full((M, Cout), float("nan"), device=DEV)
        L.igemm(a, w_p, M=M, N=Cout, Cin=Cin, taps=27, w_tap_stride=Cout * Cin, geom=geom, bias=bias, out_f32=out, debug_flags=flag, **kw)
        torch.cuda.synchronize()
        outs[flag] = out
    assert torch.equal(outs[0], outs[8]) or float((outs[0] - outs[8]).abs().max()) == 0.0       # (-0.0 vs +0.0 would still be equal)
    if not fp8:
        ref = F.conv3d(bf(x).permute(0, 4, 1, 2, 3), bf(w), bias, padding=1).permute(0, 2, 3, 4, 1).reshape(M, Cout)
        assert rel_l2(outs[0], ref) < 3e-6


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,T,H,W,C", [(2, 13, 16, 16, 256), (3, 13, 8, 8, 512), (1, 5, 7, 9, 128)])
def test_igemm256_two_phase_equals_four_phase(B, T, H, W, C, fp8):
    """The 256 x 256 kernel runs a K-tile in two phases of 32 MFMAs (round 5) instead of four of 16 (debug_flags bit 64 keeps the old form for
    A/B): the same products in the same order per accumulator -- bit-equal outputs, bf16 and e4m3 operands."""
    g = torch.Generator(device="cpu").manual_seed(B + C)
    x = torch.randn(B, T, H, W, C, generator=g).to(DEV)
    w = (torch.randn(C, C, 3, 3, 3, generator=g) / math.sqrt(27 * C)).to(DEV)
    M = B * T * H * W
    if fp8:
        from prediff_amd.packing import pack_conv_fp8, to_fp8
        a = to_fp8(x.reshape(-1, C), 16.0)
        w_p, sw = pack_conv_fp8(w)
        kw = dict(alpha=1.0 / (16.0 * sw), fp8=True)
    else:
        a, _ = padded_bf16(x.reshape(-1, C), False)
        w_p, _ = pack_conv(w, False)
        kw = dict(tile=7)
    outs = []
    for flag in (0, 64):
        out = torch.full((M, C), float("nan"), device=DEV)
        L.igemm(a, w_p, M=M, N=C, Cin=C, taps=27, w_tap_stride=C * C, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), out_f32=out, debug_flags=flag, **kw)
        torch.cuda.synchronize()
        outs.append(out)
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("tile", [0, 7])
@pytest.mark.parametrize("mode", ["same", "up2", "up2_odd", "down2"])
def test_igemm_conv2d(mode, tile):
    N_, H, W, Cin, Cout = 3, 8, 8, 64, 96
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(N_, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    a_hi, _ = padded_bf16(x.reshape(-1, Cin))
    w_hi, _ = pack_conv(w, False)
    xc = bf(x).permute(0, 3, 1, 2)
    if mode == "same":
        geom = L.conv_geom(N_, (1, H, W), (1, 3, 3), pad=(0, 1, 1))
        ref = F.conv2d(xc, bf(w), bias, padding=1)
    elif mode == "up2":      # nearest x2 then conv pad 1 (cuboid_transformer.py:373-375 / taming/resnet.py:128-141)
        geom = L.conv_geom(N_, (1, H, W), (1, 3, 3), pad=(0, 1, 1), up=(1, 2, 2))
        ref = F.conv2d(F.interpolate(xc, scale_factor=2.0, mode="nearest"), bf(w), bias, padding=1)
    elif mode == "up2_odd":  # nearest resize to an odd target (15 x 15 from 8 x 8: Upsample3DLayer with a target size, cuboid_transformer.py:363-372)
        geom = L.conv_geom(N_, (1, H, W), (1, 3, 3), pad=(0, 1, 1), up=(1, 2, 2), out_thw=(1, 15, 15), virt_thw=(1, 15, 15))
        ref = F.conv2d(F.interpolate(xc, size=(15, 15), mode="nearest"), bf(w), bias, padding=1)    # (for out = 2 in - 1 torch's source index is i >> 1)
    else:                    # pad (0,1,0,1) then stride-2 conv, padding 0 (taming/resnet.py:183-188)
        geom = L.conv_geom(N_, (1, H, W), (1, 3, 3), stride=(1, 2, 2), pad=(0, 0, 0), out_thw=(1, H // 2, W // 2))
        ref = F.conv2d(F.pad(xc, (0, 1, 0, 1)), bf(w), bias, stride=2)
    M = N_ * geom["Ho"] * geom["Wo"]
    out = torch.empty(M, Cout, device=DEV)
    L.igemm(a_hi, w_hi, M=M, N=Cout, Cin=64, taps=9, w_tap_stride=Cout * 64, geom=geom, bias=bias, out_f32=out, tile=tile)
    assert rel_l2(out, ref.permute(0, 2, 3, 1).reshape(M, Cout)) < 3e-6


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,Cn", [(1000, 256), (333, 512), (64, 1024), (50, 64), (7, 32), (100, 2048)])
def test_layernorm(rows, Cn):
    g = torch.Generator(device="cpu").manual_seed(rows + Cn)
    x = (torch.randn(rows, Cn, generator=g) * 3 + 1.5).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), torch.randn(Cn, generator=g).to(DEV)
    ld = L.pad64(Cn)
    out = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    lo = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    L.layernorm(x, gamma, beta, out, lo, rows, Cn, ld)
    ref = F.layer_norm(x, (Cn,), gamma, beta, 1e-5)
    assert rel_l2(out[:, :Cn].float() + lo[:, :Cn].float(), ref) < 2e-5       # hi+lo ~ fp32
    assert rel_l2(out[:, :Cn].float(), ref) < 4e-3                            # bf16 rounding
    assert float(out[:, Cn:].float().abs().max() if ld > Cn else 0) == 0


@pytest.mark.parametrize("T,H,W,Cn", [(3, 8, 8, 64), (13, 16, 16, 256), (3, 7, 6, 16)])
def test_patch_merge_layernorm(T, H, W, Cn):
    B = 2
    g = torch.Generator(device="cpu").manual_seed(T * H)
    x = torch.randn(B, T, H, W, Cn, generator=g).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(4 * Cn, generator=g)).to(DEV), torch.randn(4 * Cn, generator=g).to(DEV)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    ld = L.pad64(4 * Cn)
    out = torch.zeros(B * T * Ho * Wo, ld, dtype=torch.bfloat16, device=DEV)
    lo = torch.zeros_like(out)
    L.patch_merge_layernorm(x, gamma, beta, out, lo, B, T, H, W, Cn, (1, 2, 2), ld)
    xp = F.pad(x, (0, 0, 0, Wo * 2 - W, 0, Ho * 2 - H))
    xm = xp.reshape(B, T, 1, Ho, 2, Wo, 2, Cn).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B * T * Ho * Wo, 4 * Cn)
    ref = F.layer_norm(xm, (4 * Cn,), gamma, beta, 1e-5)
    assert rel_l2(out[:, :4 * Cn].float() + lo[:, :4 * Cn].float(), ref) < 2e-5


@pytest.mark.parametrize("B,S,Cn,G", [(2, 3328, 256, 32), (2, 832, 512, 32), (2, 320, 5, 5), (3, 100, 65, 65), (2, 64, 64, 32),
                                      (1, 16384, 128, 32), (2, 200, 32, 8)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_silu(B, S, Cn, G, silu):
    g = torch.Generator(device="cpu").manual_seed(S + Cn)
    x = (torch.randn(B, S, Cn, generator=g) * 2 + 0.7).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), torch.randn(Cn, generator=g).to(DEV)
    ld = L.pad64(Cn)
    part = torch.zeros(B * L.groupnorm_nchunk(S, Cn) * G * 2, dtype=torch.float64, device=DEV)
    out = torch.full((B * S, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    lo = torch.full((B * S, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    L.groupnorm_silu(x, gamma, beta, part, out, lo, B, S, Cn, G, ld, 1e-6, silu=silu)
    ref = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-6)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1).reshape(B * S, Cn)
    assert rel_l2(out[:, :Cn].float() + lo[:, :Cn].float(), ref) < 2e-5
    if ld > Cn:
        assert float(out[:, Cn:].float().abs().max()) == 0


@pytest.mark.parametrize("B,S,Cn,G,ss", [(2, 3328, 256, 32, False), (1, 3328, 256, 32, True), (2, 3000, 256, 32, False),
                                         (3, 832, 512, 32, True), (2, 700, 256, 32, False), (2, 1024, 128, 32, False)])
def test_groupnorm_silu_one_pass(B, S, Cn, G, ss):
    """pd_groupnorm_silu with bf16 rows only (the bf16 engine's call): the one-pass kernel (a workgroup keeps a sample's S x 16 / 32
    values in registers: statistics and normalised rows from one read of x) against torch's fp32 GroupNorm -> SiLU at the rounding
    of the bf16 output, and against the statistics + apply pair of launches it replaces (fp64 partial sums there, a two-pass fp32
    mean / centred variance here): the two agree on all but a few last-place roundings.  v1 level-0 / level-1 shapes, row tails,
    per-sample scale / shift."""
    import ctypes
    g = torch.Generator(device="cpu").manual_seed(S + Cn + B)
    x = (torch.randn(B, S, Cn, generator=g) * 2 + 0.7).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), torch.randn(Cn, generator=g).to(DEV)
    sst = (0.3 * torch.randn(B, 2 * Cn, generator=g)).to(DEV) if ss else None
    kw = dict(ss_scale=sst, ss_shift=sst[:, Cn:], ld_ss=2 * Cn) if ss else {}
    part = torch.zeros(B * L.groupnorm_nchunk(S, Cn) * G * 2, dtype=torch.float64, device=DEV)
    outs, sums = [], []
    for two_launches in (0, 1):
        out = torch.full((B * S, Cn), 7.0, dtype=torch.bfloat16, device=DEV)
        part.fill_(float("nan"))
        L.groupnorm_silu(x, gamma, beta, part, out, None, B, S, Cn, G, Cn, 1e-5, silu=True, opts=L.CallOpts(groupnorm_two_launches=two_launches), **kw)
        torch.cuda.synchronize()
        outs.append(out.float())
        sums.append(part.reshape(B, -1, G, 2).sum(1))          # what pd_groupnorm_silu_bwd reduces `partials` to
    # the partial-sum contract of `partials` holds on both paths: (sum, sum of squares) per (sample, group)
    xg = x.double().reshape(B, S, G, Cn // G)
    want = torch.stack([xg.sum((1, 3)), (xg * xg).sum((1, 3))], -1)
    for sm in sums:
        assert bool(torch.isfinite(sm).all()) and float(((sm - want).abs() / want.abs().clamp_min(1.0)).max()) < 1e-5
    ref = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-5)
    if ss:
        ref = ref * (1 + sst[:, :Cn, None]) + sst[:, Cn:, None]
    ref = F.silu(ref).permute(0, 2, 1).reshape(B * S, Cn)
    e1, e0 = rel_l2(outs[0], ref), rel_l2(outs[1], ref)
    same = float((outs[0] == outs[1]).float().mean())
    print(f"[groupnorm one pass B={B} S={S} C={Cn}] vs torch {e1:.3e} (two launches {e0:.3e}); identical elements {same:.5f}")
    assert e1 < 3e-3 and abs(e1 - e0) < 2e-5
    assert same > 0.995 and rel_l2(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("Cn", [64, 128])          # 64: generic kernels, 128: vectorised fast path
def test_groupnorm_scale_shift(Cn):
    B, S, G = 2, 128, 32
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(B, S, Cn, generator=g).to(DEV)
    gamma, beta = torch.randn(Cn, generator=g).to(DEV), torch.randn(Cn, generator=g).to(DEV)
    ss = torch.randn(B, 2 * Cn, generator=g).to(DEV)
    part = torch.zeros(B * L.groupnorm_nchunk(S, Cn) * G * 2, dtype=torch.float64, device=DEV)
    out = torch.zeros(B * S, Cn, dtype=torch.bfloat16, device=DEV)
    lo = torch.zeros_like(out)
    L.groupnorm_silu(x, gamma, beta, part, out, lo, B, S, Cn, G, Cn, 1e-5, silu=True, ss_scale=ss, ss_shift=ss[:, Cn:], ld_ss=2 * Cn)
    ref = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-5) * (1 + ss[:, :Cn, None]) + ss[:, Cn:, None]
    ref = F.silu(ref).permute(0, 2, 1).reshape(B * S, Cn)
    assert rel_l2(out.float() + lo.float(), ref) < 2e-5


def test_cast_rows_slice():
    x = torch.randn(3, 13, 10, device=DEV)           # 3 samples x 13 rows x 10 ch
    out = torch.full((3 * 6, 64), 5.0, dtype=torch.bfloat16, device=DEV)
    lo = torch.zeros_like(out)
    L.cast_rows(x, out, lo, 3, 13, 7, 6, 10, 10, 64)
    ref = x[:, 7:].reshape(18, 10)
    assert rel_l2(out[:, :10].float() + lo[:, :10].float(), ref) < 1e-5
    assert float(out[:, 10:].float().abs().max()) == 0


# ------------------------------------------------------------------------------------------------ attention core
def _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, B, qdtype, force_generic):
    from oracle import unet as OU
    from prediff_amd.cuboid_geometry import attention_tables
    T, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(T * 100 + Cn)
    ntok = T * H * W
    qkv = torch.randn(B, ntok, 3 * Cn, generator=g)
    table_rows = (2 * cuboid[0] - 1) * (2 * cuboid[1] - 1) * (2 * cuboid[2] - 1)
    bias_table = torch.randn(table_rows, heads, generator=g) * 0.5
    relidx = OU.relative_position_index(cuboid)
    tabs = attention_tables(shape, cuboid, shift, strategy, padding_type)
    vol, nc = tabs["vol"], tabs["nc"]
    if qdtype == "bf16":
        qkv = bf(qkv)
    if qdtype == "fp8":          # the oracle sees exactly the e4m3 values the kernel reads (q, k, v * 16 -> e4m3 -> / 16)
        qkv = (qkv * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 16.0
    # ---- oracle statement of the core on the same q/k/v (cuboid_transformer.py:839-861,947-962) ----
    cub, sh = tabs["cuboid"], tabs["shift"]
    pad = tabs["pad"]
    x = qkv.reshape(B, T, H, W, 3 * Cn)
    x = F.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))         # qkv of padded tokens is exactly zero (no qkv bias)
    if any(s > 0 for s in sh):
        x = torch.roll(x, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
    xr = OU.cuboid_reorder(x, cub, strategy)
    hd = Cn // heads
    q, k, v = xr.reshape(B, nc, vol, 3, heads, hd).permute(3, 0, 4, 1, 2, 5)
    score = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = bias_table[relidx[:vol, :vol].reshape(-1)].reshape(vol, vol, heads).permute(2, 0, 1)
    score = score + bias.unsqueeze(1)
    mask = OU.cuboid_attention_mask(shape, cub, sh, strategy, padding_type)
    y = (OU.masked_softmax(score, mask) @ v).permute(0, 2, 3, 1, 4).reshape(B, nc, vol, Cn)
    y = OU.cuboid_reorder_reverse(y, cub, strategy, (T + pad[0], H + pad[1], W + pad[2]))
    if any(s > 0 for s in sh):
        y = torch.roll(y, shifts=sh, dims=(1, 2, 3))
    ref = y[:, :T, :H, :W].reshape(B, ntok, Cn)
    # ---- HIP ----
    tok = tabs["tok_index"].to(DEV)
    m = tabs["mask"].to(DEV) if tabs["mask"] is not None else None
    assert (m is None) == bool(mask.all())
    bias_d = bias.contiguous().to(DEV)
    kw = dict(tok_index=tok, bias=bias_d, mask=m, B=B, ntok=ntok, Cn=Cn, heads=heads, nc=nc, vol=vol, ld_qkv=3 * Cn,
              ld_out=Cn, scale=hd ** -0.5, force_generic=force_generic)
    if qdtype == "bf16":
        out = torch.zeros(B, ntok, Cn, dtype=torch.bfloat16, device=DEV)
        L.cuboid_attention(qkv_bf16=qkv.to(torch.bfloat16).to(DEV), out_bf16=out, **kw)
        return out.float().cpu(), ref
    if qdtype == "fp8":
        q8 = (qkv * 16.0).to(torch.float8_e4m3fn).to(DEV).contiguous()
        out = torch.zeros(B, ntok, Cn, dtype=torch.bfloat16, device=DEV)
        L.cuboid_attention(qkv_bf16=q8, qkv_fp8_log2=4, out_bf16=out, **kw)
        return out.float().cpu(), ref
    out = torch.zeros(B, ntok, Cn, device=DEV)
    L.cuboid_attention(qkv_f32=qkv.to(DEV), out_f32=out, **kw)
    return out.cpu(), ref


LLL, DDD = ("l", "l", "l"), ("d", "d", "d")
AXIAL = [((13, 16, 16), (13, 1, 1), 256, 4), ((13, 16, 16), (1, 16, 1), 256, 4), ((13, 16, 16), (1, 1, 16), 256, 4),
         ((13, 8, 8), (13, 1, 1), 512, 4), ((13, 8, 8), (1, 8, 1), 512, 4), ((13, 8, 8), (1, 1, 8), 512, 4),
         ((5, 8, 8), (5, 1, 1), 64, 2)]


@pytest.mark.parametrize("shape,cuboid,Cn,heads", AXIAL)
@pytest.mark.parametrize("force_generic", [False, True])
def test_attention_axial_bf16(shape, cuboid, Cn, heads, force_generic):
    out, ref = _attn_case(shape, cuboid, (0, 0, 0), LLL, "zeros", Cn, heads, 2, "bf16", force_generic)
    # bf16 probabilities / bf16 output rounding (MFMA path); the generic path keeps fp32 probabilities
    assert rel_l2(out, ref) < 6e-3


# cuboid volumes 17..64: 2-4 key tiles of the MFMA core (full-resolution axial cuboids 25 / 48 / 24; non-axial patterns: 32 / 64 / 16)
MULTI_TILE = [((25, 12, 12), (25, 1, 1), (0, 0, 0), LLL, "zeros", 256, 4), ((5, 48, 6), (1, 48, 1), (0, 0, 0), LLL, "zeros", 256, 4),
              ((25, 24, 4), (1, 24, 1), (0, 0, 0), LLL, "zeros", 512, 4), ((5, 8, 8), (2, 4, 4), (1, 2, 2), LLL, "ignore", 64, 2),
              ((5, 7, 6), (2, 4, 4), (1, 2, 2), LLL, "ignore", 64, 2), ((5, 8, 8), (4, 4, 4), (0, 0, 0), DDD, "ignore", 64, 2),
              ((3, 8, 8), (4, 16, 2), (2, 1, 1), LLL, "ignore", 64, 2)]


@pytest.mark.parametrize("shape,cuboid,shift,strategy,padding_type,Cn,heads", MULTI_TILE)
def test_attention_mfma_multi_tile_bf16(shape, cuboid, shift, strategy, padding_type, Cn, heads):
    out, ref = _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, 2, "bf16", False)
    assert rel_l2(out, ref) < 6e-3
    out_g, _ = _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, 2, "bf16", True)
    assert rel_l2(out, out_g) < 6e-3          # MFMA core (bf16 probabilities) vs the fp32 VALU kernel on the same bf16 q/k/v


@pytest.mark.parametrize("case", [c + ((0, 0, 0), LLL, "zeros") for c in [(a[0], a[1]) for a in AXIAL]] + [m[:5] for m in MULTI_TILE],
                         ids=lambda c: "x".join(map(str, c[0])) + "_" + "x".join(map(str, c[1])) + "_" + c[4])
def test_attention_mfma_fp8_core(case):
    """q k^T and attn v on the fp8 MFMA (pd_cuboid_attn_args.qkv_fp8_log2 > 0; BASELINE.json configs[4] "fp8 MFMA attention",
    cuboid_transformer.py:849-861,947-952): e4m3 q / k / v and e4m3(P * 256) probabilities, fp32 scores / softmax / accumulation, against
    the oracle's fp32 statement of the core ON THE SAME e4m3 q / k / v -- what is left is the 3-mantissa-bit rounding of the
    probabilities (<= 2^-4 per entry, averaged over the keys) -- and against the bf16 core.  Axial cuboids of the SEVIR-LR grid
    (volumes 13 / 16 / 8, head_dim 64 and 128), the full-resolution volumes 25 / 48 / 24 (two and three key tiles), shifted / masked /
    padded / dilated cuboids of 32 and 64 slots."""
    shape, cuboid, shift, strategy, padding_type = case
    Cn, heads = next(((a[2], a[3]) for a in AXIAL if (a[0], a[1]) == (shape, cuboid)), None) or \
        next((m[5], m[6]) for m in MULTI_TILE if m[:5] == case)
    out, ref = _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, 2, "fp8", False)
    assert bool(torch.isfinite(out).all())
    e = rel_l2(out, ref)
    out_b, ref_b = _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, 2, "bf16", False)
    e_b = rel_l2(out, ref_b)          # vs the oracle on UNquantised q / k / v: the whole fp8 error of the core
    print(f"[attention core fp8 {shape} {cuboid} C={Cn}] rel-L2 vs oracle on the same e4m3 q/k/v {e:.3e}; vs oracle on fp32 q/k/v {e_b:.3e}")
    assert e < 4e-2 and e_b < 0.12
    # rows the cuboids do not cover (none here) / empty slots must not have been written with garbage: every token is finite and the
    # masked / padded cases agree with the reference's masked_softmax semantics to the same tolerance (asserted above)


# cuboids of more than 64 slots ("full" / "divided_st" patterns): online-softmax kernel, incl. a shifted + masked and a padded case
LARGE = [((5, 8, 8), (5, 8, 8), (0, 0, 0), LLL, "zeros", 64, 2), ((13, 16, 16), (1, 16, 16), (0, 0, 0), LLL, "zeros", 256, 4),
         ((3, 16, 8), (1, 16, 8), (0, 0, 0), LLL, "zeros", 512, 4), ((5, 12, 12), (2, 8, 8), (1, 4, 4), LLL, "ignore", 64, 2),
         ((5, 11, 10), (3, 6, 6), (0, 0, 0), LLL, "ignore", 128, 4)]


@pytest.mark.parametrize("shape,cuboid,shift,strategy,padding_type,Cn,heads", LARGE)
def test_attention_large_cuboids_bf16(shape, cuboid, shift, strategy, padding_type, Cn, heads):
    out, ref = _attn_case(shape, cuboid, shift, strategy, padding_type, Cn, heads, 2, "bf16", False)
    assert rel_l2(out, ref) < 6e-3


GENERIC = [((5, 8, 8), (2, 4, 4), (0, 0, 0), LLL, "zeros"), ((5, 8, 8), (2, 4, 4), (1, 2, 2), LLL, "zeros"),
           ((5, 8, 8), (2, 4, 4), (1, 2, 2), LLL, "ignore"), ((5, 7, 6), (2, 4, 4), (1, 2, 2), LLL, "ignore"),
           ((5, 8, 8), (1, 4, 4), (0, 0, 0), DDD, "zeros"), ((3, 8, 8), (4, 16, 2), (2, 1, 1), LLL, "ignore"),
           ((5, 8, 8), (4, 4, 4), (0, 0, 0), DDD, "ignore"), ((13, 16, 16), (13, 1, 1), (0, 0, 0), LLL, "zeros")]


@pytest.mark.parametrize("shape,cuboid,shift,strategy,padding_type", GENERIC)
def test_attention_generic_fp32(shape, cuboid, shift, strategy, padding_type):
    out, ref = _attn_case(shape, cuboid, shift, strategy, padding_type, 64, 2, 2, "fp32", False)
    assert rel_l2(out, ref) < 1e-5


def test_softmax_rows():
    x = torch.randn(300, 256, device=DEV) * 4
    out = torch.zeros(300, 256, dtype=torch.bfloat16, device=DEV)
    lo = torch.zeros_like(out)
    L.softmax_rows(x, out, lo, 300, 256, 256, 256)
    assert rel_l2(out.float() + lo.float(), torch.softmax(x, -1)) < 1e-5


# ------------------------------------------------------------------------------------------------ glue
def test_glue_kernels():
    from oracle import unet as OU
    B, Tin, Tout, HW, Cn = 2, 3, 2, 16, 4
    x, cond = torch.randn(B, Tout, HW, Cn, device=DEV), torch.randn(B, Tin, HW, Cn, device=DEV)
    out = torch.full((B, Tin + Tout, HW, 8), 9.0, device=DEV)
    L.unet_build_input(x, cond, out, B, Tin, Tout, HW, Cn, 8)
    ref = torch.cat([cond, x], 1)
    assert torch.equal(out[..., :Cn], ref)
    assert torch.equal(out[:, :Tin, :, Cn], torch.ones(B, Tin, HW, device=DEV)) and float(out[:, Tin:, :, Cn].abs().max()) == 0
    assert float(out[..., Cn + 1:].abs().max()) == 0
    t = torch.tensor([0, 1, 17, 500, 999], device=DEV)
    emb = torch.empty(5, 256, device=DEV)
    L.timestep_embedding(t, L.timestep_freqs(256, device=DEV), emb, 5, 256)
    assert rel_l2(emb, OU.timestep_embedding(t.cpu(), 256)) < 2e-6      # cosf/sinf ulps only (args up to ~1e3 rad)
    xs, Wm, bm = torch.randn(5, 256, device=DEV), torch.randn(1024, 256, device=DEV) / 16, torch.randn(1024, device=DEV)
    o = torch.empty(5, 1024, device=DEV)
    L.linear_small(xs, Wm, bm, o, 5, 256, 1024, act_in="silu", act_out="silu")
    assert rel_l2(o, F.silu(F.linear(F.silu(xs), Wm, bm))) < 1e-5
    a = torch.randn(2, 6, 5, device=DEV)
    tab = torch.randn(6, 5, device=DEV)
    r = a + tab
    L.add_rowtable(a, tab, 2, 6, 5)
    assert torch.allclose(a, r)
    n = torch.randn(3, 7, 5, 5, device=DEV)
    o1 = torch.empty(3, 25, 8, device=DEV)
    L.nchw_to_nhwc(n, o1, 3, 7, 25, 8)
    assert torch.equal(o1[..., :7], n.reshape(3, 7, 25).permute(0, 2, 1)) and float(o1[..., 7].abs().max()) == 0
    o2 = torch.empty(3, 7, 25, device=DEV)
    L.nhwc_to_nchw(o1, o2, 3, 7, 25, 8)
    assert torch.equal(o2.reshape(3, 7, 5, 5), n)


def test_diffusion_steps():
    from oracle import diffusion as OD
    buf = {k: torch.as_tensor(v) for k, v in OD.schedule_buffers(OD.beta_schedule("linear", 1000)).items()}
    coef = torch.stack([buf[k] for k in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                                         "posterior_mean_coef2", "posterior_log_variance_clipped")]).contiguous().to(DEV)
    B, per = 4, 6 * 16 * 16 * 8
    zt, eps, noise, shift = (torch.randn(B, per) for _ in range(4))
    t = torch.tensor([999, 500, 1, 0])
    out = torch.empty(B, per, device=DEV)
    L.ddpm_step(zt.to(DEV), eps.to(DEV), noise.to(DEV), None, t.to(DEV), coef, 1000, out, B, per)
    assert rel_l2(out, OD.ddpm_step(buf, zt, eps, t, noise)) < 2e-6
    L.ddpm_step(zt.to(DEV), eps.to(DEV), noise.to(DEV), shift.to(DEV), t.to(DEV), coef, 1000, out, B, per)
    assert rel_l2(out, OD.ddpm_step(buf, zt, eps, t, noise, mean_shift=shift)) < 2e-6
    a_t, a_prev, sig = buf["alphas_cumprod"][t.clamp_min(21)], buf["alphas_cumprod"][t.clamp_min(21) - 20], torch.rand(B) * 0.1
    c = torch.stack([a_t, a_prev, sig], 1).contiguous().to(DEV)
    L.ddim_step(zt.to(DEV), eps.to(DEV), noise.to(DEV), c, out, B, per)
    assert rel_l2(out, OD.ddim_step(zt, eps, a_t, a_prev, sig, noise)) < 2e-6


# ------------------------------------------------------------------------------------------------ fused FFN
@pytest.mark.parametrize("M,Cn,act", [(256, 64, "gelu"), (1000, 128, "leaky"), (3328, 256, "gelu"), (130, 256, "gelu"), (53248, 256, "gelu")])
def test_ffn_fused(M, Cn, act):
    g = torch.Generator(device="cpu").manual_seed(M + Cn)
    Hd = 4 * Cn
    x = (torch.randn(M, Cn, generator=g) * 2 + 0.3 + torch.arange(M)[:, None] * 1e-4).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), (0.1 * torch.randn(Cn, generator=g)).to(DEV)
    w1 = (torch.randn(Hd, Cn, generator=g) / math.sqrt(Cn)).to(DEV)
    w2 = (torch.randn(Cn, Hd, generator=g) / math.sqrt(Hd)).to(DEV)
    b1, b2 = torch.randn(Hd, generator=g).to(DEV) * 0.1, torch.randn(Cn, generator=g).to(DEV) * 0.1
    w1p, _ = pack_linear(w1, False)
    w2p, _ = pack_linear(w2, False)
    assert L.ffn_fused_supported(Cn, Hd) and not L.ffn_fused_supported(512, 2048)
    out = torch.empty_like(x)
    L.ffn_fused(x, out, gamma, beta, w1p, b1, w2p, b2, M, Cn, Hd, act=act)
    actf = F.gelu if act == "gelu" else (lambda v: F.leaky_relu(v, 0.1))
    # same roundings as the kernel: bf16 LN output, bf16 weights, bf16 hidden, fp32 accumulation
    h = bf(actf(bf(F.layer_norm(x, (Cn,), gamma, beta, 1e-5)) @ bf(w1).t() + b1))
    ref = x + h @ bf(w2).t() + b2
    assert rel_l2(out, ref) < 2e-4          # differs only by bf16 rounding boundaries of LN / hidden values and summation order
    full = x + actf(F.layer_norm(x, (Cn,), gamma, beta, 1e-5) @ w1.t() + b1) @ w2.t() + b2
    assert rel_l2(out, full) < 6e-3         # vs the un-rounded fp32 statement (bf16 engine tolerance)
    xin = x.clone()
    L.ffn_fused(xin, xin, gamma, beta, w1p, b1, w2p, b2, M, Cn, Hd, act=act)      # in place
    assert torch.equal(xin, out)


# ------------------------------------------------------------------------------------------------ fused attention block
ATTN_BLOCK = [((13, 16, 16), (13, 1, 1), (0, 0, 0), "zeros", 256, 4, 2), ((13, 16, 16), (1, 16, 1), (0, 0, 0), "zeros", 256, 4, 3),
              ((13, 16, 16), (1, 1, 16), (0, 0, 0), "zeros", 256, 4, 1), ((5, 8, 8), (1, 8, 1), (0, 0, 0), "zeros", 128, 2, 3),
              ((3, 5, 6), (3, 1, 1), (0, 0, 0), "zeros", 128, 2, 1), ((5, 7, 6), (1, 4, 4), (0, 2, 2), "ignore", 128, 2, 2),
              ((5, 7, 6), (1, 4, 4), (0, 2, 2), "zeros", 256, 4, 1),
              # cuboid volumes 17 .. 64: 2 / 4 key tiles per cuboid, 2 / 1 cuboids per 64-row workgroup (full-resolution grid: 25, 48)
              ((25, 12, 12), (25, 1, 1), (0, 0, 0), "zeros", 256, 4, 1), ((5, 48, 6), (1, 48, 1), (0, 0, 0), "zeros", 256, 4, 2),
              ((4, 6, 48), (1, 1, 48), (0, 0, 0), "zeros", 128, 2, 1), ((4, 8, 8), (2, 4, 4), (1, 2, 2), "zeros", 256, 4, 2),
              ((6, 9, 10), (2, 4, 8), (0, 2, 4), "ignore", 128, 2, 1), ((5, 7, 6), (2, 3, 3), (1, 1, 1), "ignore", 256, 4, 3),
              ((3, 5, 17), (1, 1, 17), (0, 0, 0), "zeros", 128, 2, 2)]


@pytest.mark.parametrize("shape,cuboid,shift,padding_type,Cn,heads,B", ATTN_BLOCK)
@pytest.mark.parametrize("qkv_bias", [False, True])
def test_attn_block_fused(shape, cuboid, shift, padding_type, Cn, heads, B, qkv_bias):
    """pd_attn_block_fused against the un-fused HIP chain LN -> QKV GEMM -> pd_cuboid_attention -> proj GEMM (+x) it replaces
    (that chain is pinned against the oracle / reference goldens by the tests above and tests/test_hip_unet.py)."""
    from oracle import unet as OU
    from prediff_amd.cuboid_geometry import attention_tables
    T, H, W = shape
    ntok = T * H * W
    g = torch.Generator(device="cpu").manual_seed(ntok + Cn + B)
    x = (torch.randn(B, ntok, Cn, generator=g) * 1.5 + 0.2 + torch.arange(ntok)[None, :, None] * 1e-3).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), (0.1 * torch.randn(Cn, generator=g)).to(DEV)
    wqkv = (torch.randn(3 * Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV)
    wp = (torch.randn(Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV)
    bq = (torch.randn(3 * Cn, generator=g) * 0.2).to(DEV) if qkv_bias else None
    bp = (torch.randn(Cn, generator=g) * 0.1).to(DEV)
    tabs = attention_tables(shape, cuboid, shift, LLL, padding_type)
    vol, nc = tabs["vol"], tabs["nc"]
    table_rows = (2 * cuboid[0] - 1) * (2 * cuboid[1] - 1) * (2 * cuboid[2] - 1)
    bias_table = torch.randn(table_rows, heads, generator=g) * 0.5
    relidx = OU.relative_position_index(cuboid)
    bias = bias_table[relidx[:vol, :vol].reshape(-1)].reshape(vol, vol, heads).permute(2, 0, 1).contiguous().to(DEV)
    tok = tabs["tok_index"].to(DEV)
    m = tabs["mask"].to(DEV) if tabs["mask"] is not None else None
    assert L.attn_block_fused_supported(Cn, heads, vol) and not L.attn_block_fused_supported(512, 4, 13)
    wq_p, _ = pack_linear(wqkv, False)
    wp_p, _ = pack_linear(wp, False)
    scale = (Cn // heads) ** -0.5
    # ---- un-fused chain ----
    a = torch.empty(B * ntok, Cn, dtype=torch.bfloat16, device=DEV)
    L.layernorm(x, gamma, beta, a, None, B * ntok, Cn, Cn)
    qkv = torch.empty(B * ntok, 3 * Cn, dtype=torch.bfloat16, device=DEV)
    L.igemm(a, wq_p, M=B * ntok, N=3 * Cn, Cin=Cn, bias=bq, out_bf16=qkv)
    o = torch.zeros(B * ntok, Cn, dtype=torch.bfloat16, device=DEV)
    L.cuboid_attention(qkv_bf16=qkv, out_bf16=o, tok_index=tok, bias=bias, mask=m, B=B, ntok=ntok, Cn=Cn, heads=heads, nc=nc,
                       vol=vol, ld_qkv=3 * Cn, ld_out=Cn, scale=scale)
    ref = torch.empty_like(x)
    L.igemm(o, wp_p, M=B * ntok, N=Cn, Cin=Cn, bias=bp, residual=x, out_f32=ref)
    # ---- fused ----
    out = torch.full_like(x, float("nan"))
    L.attn_block_fused(x, out, gamma, beta, wq_p, bq, wp_p, bp, tok, bias, m, B, ntok, Cn, heads, nc, vol, scale)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())               # every token row is written exactly once
    # same roundings (bf16 LN output / q / k / v / P / O, fp32 accumulation); only the fp32 summation order of the GEMMs differs,
    # which moves a few bf16 rounding boundaries
    assert rel_l2(out - x, ref - x) < 3e-3
    xin = x.clone()
    L.attn_block_fused(xin, xin, gamma, beta, wq_p, bq, wp_p, bp, tok, bias, m, B, ntok, Cn, heads, nc, vol, scale)   # in place
    assert torch.equal(xin, out)


@pytest.mark.parametrize("case", [10, 11, "w", "t25", "h48", "w48", "swin32"])
def test_attn_block_fused_vs_oracle(golden, case):
    """pd_attn_block_fused directly against the oracle's CuboidSelfAttentionLayer statement (+x) and the reference goldens
    (tests/golden/attn_layer.npz cases 10 / 11 = the v1 level-0 axial-T / axial-H layers; the axial-W layer has no golden and is
    checked against the oracle only).  bf16 operands, fp32 accumulation: <= 6e-3 rel-L2 on the layer output."""
    import _templates as TP
    from _cases import ATTN_CASES
    from _weights import seeded_input, seeded_state_dict
    from oracle import unet as OU
    from prediff_amd.cuboid_geometry import attention_tables, relative_position_bias
    big = {   # cuboid volumes 25 / 48 / 32: the 2-, 3- and 2-key-tile variants (full-resolution axial layers; a shifted 3-D window)
        "t25": dict(ATTN_CASES[11], shape=(25, 6, 6), cuboid=(25, 1, 1), B=1), "h48": dict(ATTN_CASES[11], shape=(3, 48, 5), cuboid=(1, 48, 1), B=2),
        "w48": dict(ATTN_CASES[11], shape=(2, 5, 48), cuboid=(1, 1, 48), B=1),
        "swin32": dict(ATTN_CASES[11], shape=(4, 8, 8), cuboid=(2, 4, 4), shift=(1, 2, 2), B=2)}
    if case == "w":
        c, seed, xname = dict(ATTN_CASES[11], cuboid=(1, 1, 16)), 190, "attnw"
    elif case in big:
        c, seed, xname = big[case], 191 + len(case), "attn" + case
    else:
        c, seed, xname = ATTN_CASES[case], 100 + case, f"attn{case}"
    Cn, heads, shape, cuboid = c["dim"], c["heads"], tuple(c["shape"]), tuple(c["cuboid"])
    sd = seeded_state_dict(TP.attn_layer(Cn, heads, cuboid), seed)
    x = seeded_input(xname, (c["B"],) + shape + (Cn,), 1)
    y_ref = OU.cuboid_self_attention(sd, "", x, heads, cuboid, c["shift"], c["strategy"], c["padding_type"])
    tabs = attention_tables(shape, cuboid, c["shift"], c["strategy"], c["padding_type"])
    vol, nc = tabs["vol"], tabs["nc"]
    assert L.attn_block_fused_supported(Cn, heads, vol)
    bias = relative_position_bias(sd["relative_position_bias_table"], sd["relative_position_index"], vol).to(DEV)
    wq_p, _ = pack_linear(sd["qkv.weight"].to(DEV), False)
    wp_p, _ = pack_linear(sd["proj.weight"].to(DEV), False)
    ntok = shape[0] * shape[1] * shape[2]
    xd = x.reshape(c["B"], ntok, Cn).to(DEV).contiguous()
    out = torch.full_like(xd, float("nan"))
    L.attn_block_fused(xd, out, sd["norm.weight"].to(DEV), sd["norm.bias"].to(DEV), wq_p, None, wp_p, sd["proj.bias"].to(DEV),
                       tabs["tok_index"].to(DEV), bias, tabs["mask"].to(DEV) if tabs["mask"] is not None else None,
                       c["B"], ntok, Cn, heads, nc, vol, (Cn // heads) ** -0.5)
    torch.cuda.synchronize()
    y = (out - xd).reshape(y_ref.shape).cpu()
    e = rel_l2(y, y_ref)
    print(f"[attn_block_fused case {case}] layer output rel-L2 vs oracle {e:.3e}")
    assert e < 6e-3
    assert rel_l2(out.reshape(x.shape).cpu(), x + y_ref) < 3e-3          # the block's result x + attn(x)
    if case in (10, 11):
        g = golden("attn_layer")
        assert rel_l2(y[:, :, ::2, ::2, ::4], g[f"y_{case}_slice"]) < 6e-3
        assert abs(float(y.double().abs().sum()) / float(g[f"y_{case}_abs_sum"][0]) - 1) < 6e-3


@pytest.mark.parametrize("shape,cuboid,Cn,heads,B", [((13, 16, 16), (13, 1, 1), 256, 4, 2), ((13, 16, 16), (1, 16, 1), 256, 4, 6),
                                                      ((13, 16, 16), (1, 1, 16), 256, 4, 1), ((5, 8, 8), (1, 8, 1), 128, 2, 3)])
def test_fused_engine_switches(shape, cuboid, Cn, heads, B):
    """pd_call_opts.attn_block_table_ids (the token table loaded instead of arithmetic token ids) changes the address path, not the
    arithmetic: bit-identical rows for the attention block; the FFN (no switch) is deterministic across repeats."""
    from prediff_amd.cuboid_geometry import attention_tables
    T, H, W = shape
    ntok = T * H * W
    g = torch.Generator(device="cpu").manual_seed(ntok + Cn + B)
    x = (torch.randn(B, ntok, Cn, generator=g) * 1.5 + 0.2).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), (0.1 * torch.randn(Cn, generator=g)).to(DEV)
    wq_p, _ = pack_linear((torch.randn(3 * Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV), False)
    wp_p, _ = pack_linear((torch.randn(Cn, Cn, generator=g) / math.sqrt(Cn)).to(DEV), False)
    bp = (torch.randn(Cn, generator=g) * 0.1).to(DEV)
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    assert tabs["affine"] is not None
    bias = (torch.randn(heads, vol, vol, generator=g) * 0.5).to(DEV)
    tok = tabs["tok_index"].to(DEV)
    Hd = 4 * Cn
    w1p, _ = pack_linear((torch.randn(Hd, Cn, generator=g) / math.sqrt(Cn)).to(DEV), False)
    w2p, _ = pack_linear((torch.randn(Cn, Hd, generator=g) / math.sqrt(Hd)).to(DEV), False)
    b1, b2 = torch.randn(Hd, generator=g).to(DEV) * 0.1, torch.randn(Cn, generator=g).to(DEV) * 0.1
    res = {}
    for table_ids in (1, 0):
        xa = x.clone()
        L.attn_block_fused(xa, xa, gamma, beta, wq_p, None, wp_p, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, (Cn // heads) ** -0.5,
                           tok_affine=tabs["affine"], opts=L.CallOpts(attn_block_table_ids=table_ids))
        xf = x.clone().reshape(B * ntok, Cn)
        L.ffn_fused(xf, xf, gamma, beta, w1p, b1, w2p, b2, B * ntok, Cn, Hd, act="gelu")
        torch.cuda.synchronize()
        res[table_ids] = (xa, xf)
    assert bool(torch.isfinite(res[0][0]).all()) and not torch.equal(res[0][0], x)
    assert torch.equal(res[1][0], res[0][0]), "attention block: token ids from the table / from the affine form differ"
    assert torch.equal(res[1][1], res[0][1]), "FFN: not deterministic across repeats"


# ------------------------------------------------------------------------------------------------ split-K (small grids)
@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(1, 13, 16, 16, 256, 256), (4, 13, 8, 8, 512, 512), (2, 5, 8, 8, 128, 192), (3, 13, 16, 16, 256, 256)])
def test_igemm_conv3d_split_k(B, T, H, W, Cin, Cout):
    """Conv3d with a split-K workspace (few trajectories per launch: K-slices run as extra workgroups, slabs summed in slice order
    by the reduce kernel that also applies bias / timestep embedding / residual) against F.conv3d, and bit-reproducible."""
    g = torch.Generator(device="cpu").manual_seed(B + T + Cin)
    x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    emb = torch.randn(B, Cout, generator=g).to(DEV)
    res = torch.randn(B * T * H * W, Cout, generator=g).to(DEV)
    a_hi, _ = padded_bf16(x.reshape(-1, Cin), False)
    w_hi, _ = pack_conv(w, False)
    Cp = a_hi.shape[1]
    M = B * T * H * W
    ws = torch.full((16 * 1024 * 1024,), float("nan"), device=DEV)
    ref = F.conv3d(bf(x).permute(0, 4, 1, 2, 3), bf(w), bias, padding=1) + emb[:, :, None, None, None]
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, Cout) + res
    outs = []
    for _ in range(2):
        out = torch.full((M, Cout), float("nan"), device=DEV)
        L.igemm(a_hi, w_hi, M=M, N=Cout, Cin=Cp, taps=27, w_tap_stride=Cout * Cp, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)),
                bias=bias, rowvec=emb, rows_per_sample=T * H * W, residual=res, out_f32=out, splitk_ws=ws)
        outs.append(out)
    torch.cuda.synchronize()
    assert rel_l2(outs[0], ref) < 3e-6
    assert torch.equal(outs[0], outs[1])
    tiles = ((M + 255) // 256) * ((Cout + 255) // 256)
    used = bool(torch.isfinite(ws[:M * Cout]).all())          # the first slab was written <=> the launch was split
    assert used == (tiles <= 128 and 27 * Cp // 64 >= 32), (tiles, used)
    # without a workspace the launch is never split and agrees to summation order
    out0 = torch.empty(M, Cout, device=DEV)
    L.igemm(a_hi, w_hi, M=M, N=Cout, Cin=Cp, taps=27, w_tap_stride=Cout * Cp, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)),
            bias=bias, rowvec=emb, rows_per_sample=T * H * W, residual=res, out_f32=out0)
    assert rel_l2(outs[0], out0) < 2e-6


@pytest.mark.parametrize("B,S,Cn,G,ss", [(4, 3328, 256, 32, False), (1, 3328, 256, 32, True), (3, 832, 512, 32, True), (2, 700, 256, 32, False),
                                         (4, 3000, 256, 32, False)])
def test_groupnorm_silu_small_grid(B, S, Cn, G, ss):
    """pd_call_opts.small_grid (the engine's small-batch mode): finer channel chunks -- twice the workgroups -- against the default chunks of
    the one-pass kernel: the statistics are summed in another order, the rows agree on all but a few last-place roundings; neighbouring
    samples untouched (row tails)."""
    g = torch.Generator(device="cpu").manual_seed(S + Cn + B)
    x = (torch.randn(B, S, Cn, generator=g) * 2 + 0.7).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cn, generator=g)).to(DEV), torch.randn(Cn, generator=g).to(DEV)
    sst = (0.3 * torch.randn(B, 2 * Cn, generator=g)).to(DEV) if ss else None
    kw = dict(ss_scale=sst, ss_shift=sst[:, Cn:], ld_ss=2 * Cn) if ss else {}
    part = torch.zeros(B * L.groupnorm_nchunk(S, Cn) * G * 2, dtype=torch.float64, device=DEV)
    outs = []
    for small in (0, 1):
        out = torch.full((B * S + 8, Cn), 7.0, dtype=torch.bfloat16, device=DEV)
        L.groupnorm_silu(x, gamma, beta, part, out, None, B, S, Cn, G, Cn, 1e-5, silu=True, opts=L.CallOpts(small_grid=small), **kw)
        torch.cuda.synchronize()
        assert float((out[B * S:].float() - 7.0).abs().max()) == 0
        outs.append(out[:B * S].float())
    ref = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-5)
    if ss:
        ref = ref * (1 + sst[:, :Cn, None]) + sst[:, Cn:, None]
    ref = F.silu(ref).permute(0, 2, 1).reshape(B * S, Cn)
    e0, e1 = rel_l2(outs[0], ref), rel_l2(outs[1], ref)
    same = float((outs[0] == outs[1]).float().mean())
    print(f"[groupnorm small grid B={B} S={S} C={Cn}] vs torch {e1:.3e} (default chunks {e0:.3e}); identical elements {same:.5f}")
    assert e1 < 3e-3 and abs(e1 - e0) < 2e-5 and same > 0.995


def test_igemm_linear_split_k_all_epilogue_operands():
    M, N, K = 300, 260, 4096
    g = torch.Generator(device="cpu").manual_seed(77)
    x = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    bias, mul, res = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    a_hi, _ = padded_bf16(x)
    w_hi, _ = pack_linear(w, False)
    ws = torch.full((4 * 1024 * 1024,), float("nan"), device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    outb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    L.igemm(a_hi, w_hi, M=M, N=N, Cin=K, bias=bias, mul=mul, residual=res, act="gelu", alpha=0.5, out_f32=out, out_bf16=outb, splitk_ws=ws)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ws[:M * N]).all())             # split
    ref = F.gelu(0.5 * (bf(x) @ bf(w).t()) + bias) * mul + res
    assert rel_l2(out, ref) < 3e-6 and rel_l2(outb.float(), ref) < 4e-3


# ------------------------------------------------------------------------------------------------ fp8 (e4m3) operands, scaled MFMA
@pytest.mark.parametrize("M,N,K", [(1000, 300, 256), (4096, 512, 2048), (77, 64, 128)])
def test_igemm_linear_fp8(M, N, K):
    """pd_igemm with e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales, tensor scales in alpha) against the fp32
    product of the SAME quantised operands: only the fp32 summation order differs."""
    from prediff_amd.packing import pack_linear_fp8, to_fp8
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) + torch.linspace(-1, 1, K)[None, :] * 0.5).to(DEV)     # asymmetric operands
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) + torch.arange(N)[:, None] * 1e-3).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    a8, sa = to_fp8(x, 16.0), 16.0
    w8, sw = pack_linear_fp8(w)
    out = torch.empty(M, N, device=DEV)
    L.igemm(a8, w8, M=M, N=N, Cin=K, bias=bias, alpha=1.0 / (sa * sw), out_f32=out, fp8=True)
    ref = (a8.float() @ w8.float().T) / (sa * sw) + bias
    assert rel_l2(out, ref) < 3e-5
    e = rel_l2(out, x @ w.T + bias)
    print(f"[igemm fp8 linear {M}x{N}x{K}] vs the unquantised fp32 product: rel-L2 {e:.3e}")
    assert e < 6e-2


@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 5, 8, 8, 128, 64), (1, 13, 16, 16, 256, 256), (2, 3, 6, 6, 5, 32), (1, 13, 8, 8, 512, 512)])
def test_igemm_conv3d_fp8(B, T, H, W, Cin, Cout):
    from prediff_amd.packing import pack_conv_fp8, pad128, to_fp8
    g = torch.Generator(device="cpu").manual_seed(B + T + Cin)
    x = torch.randn(B, T, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    emb = torch.randn(B, Cout, generator=g).to(DEV)
    Cp, M = pad128(Cin), B * T * H * W
    xp = torch.zeros(M, Cp, device=DEV)
    xp[:, :Cin] = x.reshape(M, Cin)
    sa = 16.0
    a8 = to_fp8(xp, sa)
    w8, sw = pack_conv_fp8(w)
    out = torch.empty(M, Cout, device=DEV)
    L.igemm(a8, w8, M=M, N=Cout, Cin=Cp, taps=27, w_tap_stride=Cout * Cp, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias,
            rowvec=emb, rows_per_sample=T * H * W, alpha=1.0 / (sa * sw), out_f32=out, fp8=True)
    # small grids: the same launch with a split-K workspace (K-slices as extra workgroups, slabs summed in slice order)
    ws = torch.full((8 * 1024 * 1024,), float("nan"), device=DEV)
    out_sk = torch.full((M, Cout), float("nan"), device=DEV)
    L.igemm(a8, w8, M=M, N=Cout, Cin=Cp, taps=27, w_tap_stride=Cout * Cp, geom=L.conv_geom(B, (T, H, W), (3, 3, 3)), bias=bias,
            rowvec=emb, rows_per_sample=T * H * W, alpha=1.0 / (sa * sw), out_f32=out_sk, fp8=True, splitk_ws=ws)
    assert rel_l2(out_sk, out) < 3e-6
    tiles = ((M + 255) // 256) * ((Cout + 255) // 256)
    assert bool(torch.isfinite(ws[:M * Cout]).all()) == (tiles <= 128 and 27 * Cp // 128 >= 32 and Cout % 4 == 0)
    xq = a8.float()[:, :Cin].reshape(B, T, H, W, Cin) / sa
    wq = w8.float()[:, :, :Cin].permute(1, 2, 0).reshape(Cout, Cin, 3, 3, 3) / sw
    ref = F.conv3d(xq.permute(0, 4, 1, 2, 3), wq, bias, padding=1) + emb[:, :, None, None, None]
    ref = ref.permute(0, 2, 3, 4, 1).reshape(M, Cout)
    assert rel_l2(out, ref) < 3e-5
    full = (F.conv3d(x.permute(0, 4, 1, 2, 3), w, bias, padding=1) + emb[:, :, None, None, None]).permute(0, 2, 3, 4, 1).reshape(M, Cout)
    e = rel_l2(out, full)
    print(f"[igemm fp8 conv3d {Cin}->{Cout}] vs the unquantised fp32 convolution: rel-L2 {e:.3e}")
    assert e < 6e-2


@pytest.mark.parametrize("B,S,C,G,ss", [(2, 3328, 256, 32, False), (3, 832, 512, 32, True), (1, 100, 128, 32, False)])
def test_groupnorm_silu_fp8(B, S, C, G, ss):
    """GroupNorm -> SiLU with an e4m3 output (value * scale, RNE, saturating) against torch's cast of the fp32 statement."""
    g = torch.Generator(device="cpu").manual_seed(B + S + C)
    x = (torch.randn(B, S, C, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    emb = (0.3 * torch.randn(B, 2 * C, generator=g)).to(DEV) if ss else None
    part = torch.empty(B * L.groupnorm_nchunk(S, C) * G * 2, dtype=torch.float64, device=DEV)
    out = torch.empty(B * S, C, dtype=torch.float8_e4m3fn, device=DEV)
    kw = dict(ss_scale=emb, ss_shift=emb[:, C:], ld_ss=2 * C) if ss else {}
    L.groupnorm_silu_fp8(x, gamma, beta, part, out, B, S, C, G, 1e-5, 16.0, silu=True, **kw)
    y = F.group_norm(x.permute(0, 2, 1), G, gamma, beta, 1e-5).permute(0, 2, 1)
    if ss:
        y = y * (1 + emb[:, None, :C]) + emb[:, None, C:]
    y = F.silu(y).reshape(B * S, C)
    ref = (y * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = float((out.view(torch.uint8) == ref.view(torch.uint8)).float().mean())
    err = rel_l2(out.float(), ref.float())
    print(f"[gn silu fp8 C={C}] identical bytes {same:.4f}, rel-L2 of the dequantised values {err:.2e}")
    assert same > 0.99 and err < 1e-2
    assert rel_l2(out.float() / 16.0, y) < 4e-2          # e4m3: 3 mantissa bits


@pytest.mark.parametrize("rows,C", [(1000, 512), (333, 256), (64, 1024)])
def test_layernorm_fp8(rows, C):
    """LayerNorm with an e4m3 output (value * scale, RNE, saturating: the A operand of the fp8 linears) against torch's cast of the fp32
    statement; one row is scaled up to exercise the saturation."""
    g = torch.Generator(device="cpu").manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.3 * torch.randn(C, generator=g)).to(DEV)
    gamma[3] = 40.0                                       # |y| * 16 > 448 in that column: saturates instead of turning into NaN
    out = torch.empty(rows, C, dtype=torch.float8_e4m3fn, device=DEV)
    L.layernorm_fp8(x, gamma, beta, out, rows, C, C, 16.0)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    ref = (y * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = float((out.view(torch.uint8) == ref.view(torch.uint8)).float().mean())
    print(f"[layernorm fp8 C={C}] identical bytes {same:.4f}")
    assert bool(torch.isfinite(out.float()).all()) and same > 0.99
    assert rel_l2(out.float()[:, 4:] / 16.0, y[:, 4:]) < 4e-2


@pytest.mark.parametrize("M,N,K,act", [(832, 2048, 512, "gelu"), (300, 512, 2048, "none"), (256, 1536, 512, "none")])
def test_igemm_fp8_output(M, N, K, act):
    """pd_igemm on e4m3 operands whose epilogue (bias, activation) writes e4m3 again (out_fp8_log2: the FFN-1 -> FFN-2 hand-over of
    the fp8 linears) against the fp32 statement on the same quantised operands, quantised the same way."""
    from prediff_amd.packing import pack_linear_fp8, to_fp8
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias = (0.2 * torch.randn(N, generator=g)).to(DEV)
    a8, sa = to_fp8(x, 16.0), 16.0
    w8, sw = pack_linear_fp8(w)
    out8 = torch.empty(M, N, dtype=torch.float8_e4m3fn, device=DEV)
    L.igemm(a8, w8, M=M, N=N, Cin=K, bias=bias, act=act, alpha=1.0 / (sa * sw), out_bf16=out8, ld_outb=N, fp8=True, out_fp8_log2=4)
    y = (a8.float() @ w8.float().T) / (sa * sw) + bias
    if act == "gelu":
        y = F.gelu(y)
    ref = (y * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = float((out8.view(torch.uint8) == ref.view(torch.uint8)).float().mean())
    err = rel_l2(out8.float(), ref.float())
    print(f"[igemm fp8 -> fp8 {M}x{N}x{K} {act}] identical bytes {same:.4f}, rel-L2 of the dequantised values {err:.2e}")
    assert same > 0.98 and err < 2e-2
    # the bf16 output of the same launch agrees with the e4m3 one to e4m3 precision
    outb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.igemm(a8, w8, M=M, N=N, Cin=K, bias=bias, act=act, alpha=1.0 / (sa * sw), out_bf16=outb, fp8=True)
    assert rel_l2(out8.float() / 16.0, outb.float()) < 4e-2


def test_cuboid_attention_fp8_output():
    """The MFMA attention core writing e4m3 (the A operand of an fp8 proj launch) against its bf16 output, quantised by torch."""
    from prediff_amd.cuboid_geometry import attention_tables
    B, Cn, heads, shape, cuboid = 2, 512, 4, (13, 8, 8), (13, 1, 1)
    ntok = shape[0] * shape[1] * shape[2]
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    g = torch.Generator(device="cpu").manual_seed(7)
    qkv = torch.randn(B * ntok, 3 * Cn, generator=g).to(DEV).to(torch.bfloat16)
    bias = (0.5 * torch.randn(heads, vol, vol, generator=g)).to(DEV)
    tok = tabs["tok_index"].to(DEV)
    kw = dict(qkv_bf16=qkv, tok_index=tok, bias=bias, mask=None, B=B, ntok=ntok, Cn=Cn, heads=heads, nc=nc, vol=vol, ld_qkv=3 * Cn, ld_out=Cn,
              scale=(Cn // heads) ** -0.5)
    ob = torch.zeros(B * ntok, Cn, dtype=torch.bfloat16, device=DEV)
    L.cuboid_attention(out_bf16=ob, **kw)
    o8 = torch.zeros(B * ntok, Cn, dtype=torch.float8_e4m3fn, device=DEV)
    L.cuboid_attention(out_bf16=o8, out_fp8_log2=4, **kw)
    assert rel_l2(o8.float() / 16.0, ob.float()) < 4e-2
    # from the fp32 accumulators, not via bf16: matches the quantisation of the bf16 output except at double-rounding boundaries
    ref = (ob.float() * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = float((o8.view(torch.uint8) == ref.view(torch.uint8)).float().mean())
    print(f"[attention core fp8 output] bytes identical to the quantised bf16 output {same:.4f}")
    assert same > 0.95
    with pytest.raises(L.PrediffHipError):                  # the generic fp32 core has no e4m3 output
        L.cuboid_attention(out_bf16=o8, out_fp8_log2=4, force_generic=True, **kw)


# ------------------------------------------------------------------------------------------------ VAE ResBlock: fused GN -> SiLU -> Conv2d
@pytest.mark.parametrize("N,H,W,Cin,Cout,res", [(3, 16, 16, 128, 128, False), (2, 32, 48, 128, 256, True), (5, 8, 16, 512, 512, True),
                                                 (1, 128, 128, 128, 128, True), (2, 16, 32, 256, 128, False), (7, 16, 16, 64, 128, True)])
def test_conv2d_gn_silu(N, H, W, Cin, Cout, res):
    """pd_groupnorm_stats + pd_conv2d_gn_silu (taming/resnet.py:454-495, one (norm, nonlinearity, conv) of ResnetBlock2D) against the
    fp32 torch statement, against the same statement with the kernel's roundings (bf16 activations after SiLU, bf16 weights), and
    against the un-fused HIP chain pd_groupnorm_silu -> pd_igemm it replaces."""
    G = 32 if Cin >= 128 else 16          # (Cin / G) % 4 == 0
    g = torch.Generator(device="cpu").manual_seed(N + H + Cin + Cout)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.3 + torch.linspace(-1, 1, Cin)[None, None, None, :]).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(DEV)
    w[:, :, 0, 2] += 0.05                                             # asymmetric taps: a transposed / mirrored tap order would show
    bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    resid = torch.randn(N, H, W, Cout, generator=g).to(DEV) if res else None
    assert L.conv2d_gn_silu_supported(H, W, Cin, Cout, G) and not L.conv2d_gn_silu_supported(12, 16, 128, 128, 32)
    S = H * W
    part = torch.empty(N * L.groupnorm_nchunk(S, Cin) * G * 2, dtype=torch.float64, device=DEV)
    stats = torch.empty(N, G, 2, device=DEV)
    L.groupnorm_stats(x, part, stats, N, S, Cin, G, 1e-6)
    xg = x.reshape(N, S, G, Cin // G).double()
    mean, var = xg.mean(dim=(1, 3)), xg.var(dim=(1, 3), unbiased=False)
    assert rel_l2(stats[:, :, 0], mean) < 1e-6 and rel_l2(stats[:, :, 1], 1.0 / torch.sqrt(var + 1e-6)) < 1e-5
    w_p, _ = pack_conv(w, False)
    out = torch.full((N, H, W, Cout), float("nan"), device=DEV)
    L.conv2d_gn_silu(x, stats, gamma, beta, w_p, bias, resid, out, N, H, W, Cin, Cout, G)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    act = F.silu(F.group_norm(x.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
    full = F.conv2d(act, w, bias, padding=1).permute(0, 2, 3, 1) + (resid if res else 0)
    same = F.conv2d(bf(act), bf(w), bias, padding=1).permute(0, 2, 3, 1) + (resid if res else 0)
    e_full, e_same = rel_l2(out, full), rel_l2(out, same)
    # the un-fused chain
    a = torch.empty(N * S, Cin, dtype=torch.bfloat16, device=DEV)
    L.groupnorm_silu(x, gamma, beta, part, a, None, N, S, Cin, G, Cin, 1e-6, silu=True)
    ref = torch.empty(N * S, Cout, device=DEV)
    L.igemm(a, w_p, M=N * S, N=Cout, Cin=Cin, taps=9, w_tap_stride=Cout * Cin, geom=L.conv_geom(N, (1, H, W), (1, 3, 3), pad=(0, 1, 1)),
            bias=bias, residual=(resid.reshape(N * S, Cout) if res else None), out_f32=ref)
    e_chain = rel_l2(out.reshape(N * S, Cout), ref)
    print(f"[conv2d_gn_silu {N}x{H}x{W} {Cin}->{Cout}] rel-L2 vs fp32 {e_full:.2e}, same roundings {e_same:.2e}, un-fused HIP chain {e_chain:.2e}")
    assert e_full < 6e-3 and e_same < 3e-4 and e_chain < 3e-4
    if res:      # in place on the residual (how the ResBlock's conv2 is called)
        r2 = resid.clone()
        L.conv2d_gn_silu(x, stats, gamma, beta, w_p, bias, r2, r2, N, H, W, Cin, Cout, G)
        assert torch.equal(r2, out)


@pytest.mark.parametrize("operand", ["bf16", "fp16"])
@pytest.mark.parametrize("N,Hs,Ws,C", [(3, 16, 16, 512), (2, 64, 64, 256), (5, 8, 24, 128), (1, 4, 8, 128)])
def test_conv2d_up2(N, Hs, Ws, C, operand):
    """pd_conv2d_up2 (Upsample2D: nearest x2 -> Conv2d 3x3 pad 1, taming/resnet.py:128-141, as one launch of the fused tile kernel with the halo
    staged from the half-resolution fp32 rows) against torch on the 16-bit-rounded operands and against the route it replaces (cast pass +
    pd_igemm with the up-sampling gather): the same products, another summation order."""
    opts = L.CallOpts(operand)
    dt = opts.dtype
    g = torch.Generator(device="cpu").manual_seed(N + Hs + C)
    x = torch.randn(N, Hs, Ws, C, generator=g).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    wp, _ = pack_conv(w, False, dtype=dt)
    H, W = 2 * Hs, 2 * Ws
    out = torch.full((N * H * W, C), float("nan"), device=DEV)
    L.conv2d_up2(x, wp, bias, out, N, H, W, C, C, opts=opts)
    torch.cuda.synchronize()
    r = lambda t: t.to(dt).float()
    ref = F.conv2d(F.interpolate(r(x).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"), r(w), bias, padding=1).permute(0, 2, 3, 1).reshape(N * H * W, C)
    assert rel_l2(out, ref) < 3e-6
    a = x.reshape(-1, C).to(dt).contiguous()
    out2 = torch.full_like(out, float("nan"))
    L.igemm(a, wp, M=N * H * W, N=C, Cin=C, taps=9, w_tap_stride=C * C, geom=L.conv_geom(N, (1, Hs, Ws), (1, 3, 3), pad=(0, 1, 1), up=(1, 2, 2)),
            bias=bias, out_f32=out2, opts=opts)
    torch.cuda.synchronize()
    assert rel_l2(out, out2) < 2e-6
    out3 = torch.full_like(out, float("nan"))
    L.conv2d_up2(x, wp, bias, out3, N, H, W, C, C, opts=opts)
    torch.cuda.synchronize()
    assert torch.equal(out, out3)


@pytest.mark.parametrize("N,H,W,Cin,Cout,res", [(7, 128, 128, 128, 128, False), (10, 128, 128, 128, 128, True), (8, 32, 32, 512, 512, True)])
def test_conv2d_gn_silu_two_workgroups_per_cu(N, H, W, Cin, Cout, res):
    """More workgroups than CUs (896 / 1280 / 256 of ~70 KB LDS: two resident per CU): 12 launches, every one against the torch statement
    with the kernel's roundings and bit-equal to the first.  A build of this kernel whose scale computation hipcc had packed into a
    v_pk_mul_f32 behind the loads' s_waitcnt got 4 ... 50 of 896 tiles wrong per launch in exactly this configuration
    (profiles/r03_h_conv2d_gn_hazard.md)."""
    G = 32
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.3).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(DEV)
    bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    resid = torch.randn(N, H, W, Cout, generator=g).to(DEV) if res else None
    S = H * W
    part = torch.empty(N * L.groupnorm_nchunk(S, Cin) * G * 2, dtype=torch.float64, device=DEV)
    stats = torch.empty(N, G, 2, device=DEV)
    L.groupnorm_stats(x, part, stats, N, S, Cin, G, 1e-6)
    w_p, _ = pack_conv(w, False)
    act = F.silu(F.group_norm(x.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
    same = F.conv2d(bf(act), bf(w), bias, padding=1).permute(0, 2, 3, 1) + (resid if res else 0)
    first = None
    for k in range(12):
        out = torch.full((N, H, W, Cout), float("nan"), device=DEV)
        L.conv2d_gn_silu(x, stats, gamma, beta, w_p, bias, resid, out, N, H, W, Cin, Cout, G)
        torch.cuda.synchronize()
        worst = float((out - same).abs().max())
        assert worst < 5e-3, (k, worst)
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), k


# ------------------------------------------------------------------------------------------------ (attention, FFN) pair kernel
PAIR_CASES = {
    # name: (shape, cuboid, B, units)  -- 4 heads, hidden 4 x units: the level-0 (256) and level-1 (512) blocks of the SEVIR-LR denoiser
    "t13": ((13, 16, 16), (13, 1, 1), 2, 256),      # axial T: 13 of 16 slots per cuboid, 64 tiles of 8 cuboids
    "h16": ((13, 16, 16), (1, 16, 1), 1, 256),      # axial H
    "w16": ((13, 16, 16), (1, 1, 16), 1, 256),      # axial W
    "tail": ((3, 6, 7), (1, 6, 1), 3, 256),         # 21 cuboids of 6 slots per sample, two per 16-slot group: 11 groups per sample, the last one half empty
    "one": ((2, 5, 3), (2, 1, 1), 1, 256),          # 15 cuboids of 2 slots: 8 groups
    "L1t13": ((13, 8, 8), (13, 1, 1), 2, 512),      # level 1, axial T: one cuboid of 13 per group, 32 tiles of 4 groups
    "L1h8": ((13, 8, 8), (1, 8, 1), 1, 512),        # level 1, axial H: two cuboids of 8 per group, all 16 slots used
    "L1w8": ((13, 8, 8), (1, 1, 8), 1, 512),        # level 1, axial W
    "L1odd": ((3, 5, 3), (1, 5, 1), 3, 512),        # 9 cuboids of 5 slots per sample: 5 groups (the last with ONE cuboid), 6 dead slots per group
}


def _pair_case(name):
    import _templates as TP
    from _weights import seeded_input, seeded_state_dict
    shape, cuboid, B, Cn = PAIR_CASES[name]
    heads, Hd = 4, 4 * Cn
    seed = 300 + sum(map(ord, name))
    sd_a = seeded_state_dict(TP.attn_layer(Cn, heads, cuboid), seed)
    sd_f = seeded_state_dict(TP.ffn(Cn, Hd), seed + 1)
    x = seeded_input("pair" + name, (B,) + shape + (Cn,), 1)
    return shape, cuboid, B, Cn, heads, Hd, sd_a, sd_f, x


@pytest.fixture
def pair_nc(request):
    """pd_call_opts.pair_form: the form of pd_attn_ffn_pair at units 256 (1 = four waves x one group: 64-row tiles, the small-grid form;
    2 = four waves x two groups, 8 = eight waves x one group: 128-row tiles)."""
    return request.param


def _pair_opts(operand, **kw):
    """CallOpts + (operand dtype, fold) of a pair-kernel test case: "fp16x2" = IEEE-half operands with folded (hi + lo) weights"""
    fold = operand == "fp16x2"
    opts = L.CallOpts("fp16" if fold else operand, w_fold=1 if fold else 0, **kw)
    return opts, opts.dtype, fold


@pytest.mark.parametrize("operand", ["bf16", "fp16", "fp16x2"])
@pytest.mark.parametrize("pair_nc", [1, 2, 8], indirect=True)
@pytest.mark.parametrize("name", list(PAIR_CASES))
def test_attn_ffn_pair_vs_oracle(name, pair_nc, operand):
    """pd_attn_ffn_pair (csrc/pair_block.hip) against the oracle's statement of one (CuboidSelfAttentionLayer, PositionwiseFFN) pair
    of StackCuboidSelfAttentionBlock (reference cuboid_transformer.py:1147-1156: x = x + attn(x); x = ffn(x)), against the two round-3
    kernels it replaces, with the token ids from the table and from its affine form, and twice (bit-equal).  bf16 operands, fp32
    accumulation: <= 6e-3 rel-L2 on the update, as for the attention block alone; IEEE-half operands (the pd_f16_* build of the same
    source, 11-bit significands): <= 1e-3; "fp16x2" (pd_call_opts.w_fold: the stream carries W_hi and W_lo chunks, two products per
    k-step, the WP = 2 instantiations): <= 7e-4 -- the once-rounded activations alone."""
    from oracle import unet as OU
    opts, odt, fold = _pair_opts(operand, pair_form=pair_nc)
    tol = {"bf16": 6e-3, "fp16": 1e-3, "fp16x2": 7e-4}[operand]
    if fold and pair_nc == 2:
        pytest.skip("folded weights: one group per wave (the library refuses pair_form 2)")
    from prediff_amd.cuboid_geometry import attention_tables, relative_position_bias
    from prediff_amd.packing import pack_pair_block, pack_pair_vecs
    shape, cuboid, B, Cn, heads, Hd, sd_a, sd_f, x = _pair_case(name)
    if Cn == 512 and pair_nc != 1:
        pytest.skip("units 512: four waves x one group only")
    y1 = x + OU.cuboid_self_attention(sd_a, "", x, heads, cuboid, (0, 0, 0), LLL, "zeros")
    y_ref = OU.positionwise_ffn(sd_f, "", y1, "gelu")
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    assert tabs["mask"] is None and tabs["affine"] is not None and L.attn_ffn_pair_supported(Cn, heads, Hd, vol)
    d = lambda t: t.to(DEV)
    bias = relative_position_bias(sd_a["relative_position_bias_table"], sd_a["relative_position_index"], vol).to(DEV)
    ws = pack_pair_block(d(sd_a["qkv.weight"]), d(sd_a["proj.weight"]), d(sd_f["ffn_1.weight"]), d(sd_f["ffn_2.weight"]), dtype=odt, fold=fold)
    vecs = pack_pair_vecs(d(sd_a["norm.weight"]), d(sd_a["norm.bias"]), d(sd_a["proj.bias"]), d(sd_f["layer_norm.weight"]),
                          d(sd_f["layer_norm.bias"]), d(sd_f["ffn_2.bias"]), d(sd_f["ffn_1.bias"]), bias)
    ntok = shape[0] * shape[1] * shape[2]
    xd = x.reshape(B, ntok, Cn).to(DEV).contiguous()
    tok = tabs["tok_index"].to(DEV)
    scale = (Cn // heads) ** -0.5
    out = torch.full_like(xd, float("nan"))
    L.attn_ffn_pair(xd, out, ws, vecs, tok, B, ntok, nc, vol, scale, tok_affine=tabs["affine"], units=Cn, opts=opts)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()), "a row was not written (or written with garbage)"
    e = rel_l2((out - xd).reshape(x.shape).cpu(), y_ref - x)
    print(f"[attn_ffn_pair {name}, {pair_nc} cuboid(s) per wave, {operand}] update rel-L2 vs oracle {e:.3e}")
    assert e < tol
    assert rel_l2(out.reshape(x.shape).cpu(), y_ref) < tol          # the pair's result (the FFN update is as large as x itself)
    # in place, token ids from the table instead of the affine form, and a repeat: bit-identical
    for aff in (None, tabs["affine"], tabs["affine"]):
        t = xd.clone()
        L.attn_ffn_pair(t, t, ws, vecs, tok, B, ntok, nc, vol, scale, tok_affine=aff, units=Cn, opts=opts)
        torch.cuda.synchronize()
        assert torch.equal(t, out)
    if fold:
        # the same launch with the one-product stream: the folded form must be the more exact one (it removes the weight rounding)
        o1, odt1, _ = _pair_opts("fp16", pair_form=pair_nc)
        ws1 = pack_pair_block(d(sd_a["qkv.weight"]), d(sd_a["proj.weight"]), d(sd_f["ffn_1.weight"]), d(sd_f["ffn_2.weight"]), dtype=odt1)
        t = torch.full_like(xd, float("nan"))
        L.attn_ffn_pair(xd, t, ws1, vecs, tok, B, ntok, nc, vol, scale, tok_affine=tabs["affine"], units=Cn, opts=o1)
        e1 = rel_l2((t - xd).reshape(x.shape).cpu(), y_ref - x)
        print(f"[attn_ffn_pair {name} fp16x2] update rel-L2 vs oracle {e:.3e}; one-product fp16 stream {e1:.3e}")
        assert e < 0.8 * e1
        return
    if Cn != 256:
        return                                          # (the round-3 fused kernels exist for units 256 only)
    # the two launches it replaces (same 16-bit operands; erf GELU there, the 2.5e-5 sigmoid form here; another summation order)
    wq_p, _ = pack_linear(d(sd_a["qkv.weight"]), False, dtype=odt)
    wp_p, _ = pack_linear(d(sd_a["proj.weight"]), False, dtype=odt)
    w1_p, _ = pack_linear(d(sd_f["ffn_1.weight"]), False, dtype=odt)
    w2_p, _ = pack_linear(d(sd_f["ffn_2.weight"]), False, dtype=odt)
    t = xd.clone()
    L.attn_block_fused(t, t, d(sd_a["norm.weight"]), d(sd_a["norm.bias"]), wq_p, None, wp_p, d(sd_a["proj.bias"]), tok, bias, None,
                       B, ntok, Cn, heads, nc, vol, scale, opts=opts)
    L.ffn_fused(t, t, d(sd_f["layer_norm.weight"]), d(sd_f["layer_norm.bias"]), w1_p, d(sd_f["ffn_1.bias"]), w2_p, d(sd_f["ffn_2.bias"]),
                B * ntok, Cn, Hd, act="gelu", opts=opts)
    torch.cuda.synchronize()
    e3 = rel_l2(out - xd, t - xd)
    e_r3 = rel_l2((t - xd).reshape(x.shape).cpu(), y_ref - x)
    print(f"[attn_ffn_pair {name}, {operand}] update rel-L2 vs pd_attn_block_fused + pd_ffn_fused {e3:.3e} (those vs the oracle {e_r3:.3e})")
    assert e3 < 1.5e-3 and e_r3 < tol


@pytest.mark.parametrize("Cn,shape,cuboid", [(256, (13, 16, 16), (13, 1, 1)), (256, (13, 16, 16), (1, 1, 16)), (256, (3, 5, 3), (1, 5, 1)),
                                              (512, (13, 8, 8), (13, 1, 1)), (512, (13, 8, 8), (1, 8, 1)), (512, (3, 5, 3), (1, 5, 1))])
def test_attn_ffn_pair_batch_independent(Cn, shape, cuboid):
    """A trajectory's rows do not depend on the launch they ride in: 8 trajectories at once (units 256: two groups per wave, 128-row
    tiles) == 4 + 4 == one alone (one group per wave) bit for bit -- the instantiations share every fp32 operation (no implicit
    contraction in pair_block.hip) and 16-slot groups never straddle samples (the (3, 5, 3) case: 9 cuboids of 5 slots per sample, an odd
    number, two per group).  This is what the engine's batch-split-reproducible mode (split_k = False) relies on."""
    from prediff_amd.cuboid_geometry import attention_tables
    from prediff_amd.packing import pack_pair_block, pack_pair_vecs
    Hd, ntok = 4 * Cn, shape[0] * shape[1] * shape[2]
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    x = r(8, ntok, Cn)
    ws = pack_pair_block(r(3 * Cn, Cn, sc=Cn ** -0.5), r(Cn, Cn, sc=Cn ** -0.5), r(Hd, Cn, sc=Cn ** -0.5), r(Cn, Hd, sc=Hd ** -0.5))
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol = tabs["vol"]
    vecs = pack_pair_vecs(1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), 1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), r(Hd, sc=.1), r(4, vol, vol, sc=.5))
    tok = tabs["tok_index"].to(DEV)

    def run(xx, opts=None):
        o = torch.full_like(xx, float("nan"))
        L.attn_ffn_pair(xx, o, ws, vecs, tok, xx.shape[0], ntok, tabs["nc"], vol, (Cn // 4) ** -0.5, tok_affine=tabs["affine"], units=Cn, opts=opts)
        torch.cuda.synchronize()
        return o
    o8 = run(x)
    assert bool(torch.isfinite(o8).all())
    assert torch.equal(o8[:4], run(x[:4].contiguous())) and torch.equal(o8[4:], run(x[4:].contiguous()))
    assert torch.equal(o8[5:6], run(x[5:6].contiguous()))
    if Cn == 256:                                       # the three forms of the units-256 kernel, forced
        for form in (1, 2, 8):
            assert torch.equal(run(x, L.CallOpts(pair_form=form)), o8), f"form {form} differs"


@pytest.mark.parametrize("operand", ["bf16", "fp16", "fp16x2"])
@pytest.mark.parametrize("name,B", [("L1t13", 2), ("L1h8", 1), ("L1w8", 4), ("L1odd", 3), ("L1h8", 9)])
def test_attn_ffn_pair_split_vs_oracle(name, B, operand):
    """pd_attn_ffn_pair_split -- the units-512 pair for small grids as (tile, head) + (tile, hidden quarter) workgroups and an ordered sum
    of partial slabs (csrc/pair_block.hip MODE 1 / 2) -- against the oracle's statement of the pair (reference cuboid_transformer.py:
    1147-1156), against the one-launch kernel (same operands; partial sums instead of one running accumulator: fp32 round-off amplified
    by the 16-bit roundings downstream), in place, twice (bit-equal: the slab sums have a fixed order), and with more tiles than the
    chip has room for four workgroups each (B = 9 at 104 groups per sample: the persistent path)."""
    from _weights import seeded_input
    from oracle import unet as OU
    from prediff_amd.cuboid_geometry import attention_tables, relative_position_bias
    from prediff_amd.packing import pack_pair_block, pack_pair_ffn_split, pack_pair_vecs
    opts, odt, fold = _pair_opts(operand)
    tol = {"bf16": 6e-3, "fp16": 1e-3, "fp16x2": 7e-4}[operand]
    shape, cuboid, _, Cn, heads, Hd, sd_a, sd_f, _ = _pair_case(name)
    x = seeded_input("pairsplit" + name, (B,) + shape + (Cn,), 1)
    y1 = x + OU.cuboid_self_attention(sd_a, "", x, heads, cuboid, (0, 0, 0), LLL, "zeros")
    y_ref = OU.positionwise_ffn(sd_f, "", y1, "gelu")
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol, nc = tabs["vol"], tabs["nc"]
    d = lambda t: t.to(DEV)
    bias = relative_position_bias(sd_a["relative_position_bias_table"], sd_a["relative_position_index"], vol).to(DEV)
    ws_full = pack_pair_block(d(sd_a["qkv.weight"]), d(sd_a["proj.weight"]), d(sd_f["ffn_1.weight"]), d(sd_f["ffn_2.weight"]), dtype=odt, fold=fold)
    ws_ffn = pack_pair_ffn_split(d(sd_f["ffn_1.weight"]), d(sd_f["ffn_2.weight"]), dtype=odt, fold=fold)
    vecs = pack_pair_vecs(d(sd_a["norm.weight"]), d(sd_a["norm.bias"]), d(sd_a["proj.bias"]), d(sd_f["layer_norm.weight"]),
                          d(sd_f["layer_norm.bias"]), d(sd_f["ffn_2.bias"]), d(sd_f["ffn_1.bias"]), bias)
    ntok = shape[0] * shape[1] * shape[2]
    xd = x.reshape(B, ntok, Cn).to(DEV).contiguous()
    tok = tabs["tok_index"].to(DEV)
    scale = (Cn // heads) ** -0.5
    wsp = torch.full((L.attn_ffn_pair_split_ws_floats(B, ntok, Cn),), float("nan"), device=DEV)
    assert wsp.numel() == L.lib().pd_attn_ffn_pair_split_ws_floats(B, ntok, Cn)
    out = torch.full_like(xd, float("nan"))
    L.attn_ffn_pair_split(xd, out, ws_full, ws_ffn, vecs, tok, B, ntok, nc, vol, scale, wsp, tok_affine=tabs["affine"], units=Cn, opts=opts)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()), "a row was not written (or written with garbage)"
    e = rel_l2((out - xd).reshape(x.shape).cpu(), y_ref - x)
    full = torch.full_like(xd, float("nan"))
    L.attn_ffn_pair(xd, full, ws_full, vecs, tok, B, ntok, nc, vol, scale, tok_affine=tabs["affine"], units=Cn, opts=opts)
    torch.cuda.synchronize()
    e_full = rel_l2(out - xd, full - xd)
    print(f"[attn_ffn_pair_split {name} B={B} {operand}] update rel-L2 vs oracle {e:.3e}, vs the one-launch kernel {e_full:.3e}")
    assert e < tol and e_full < tol / 4
    # in place, token ids from the table instead of the affine form, and a repeat: bit-identical
    for aff in (None, tabs["affine"], tabs["affine"]):
        t = xd.clone()
        wsp.fill_(float("nan"))
        L.attn_ffn_pair_split(t, t, ws_full, ws_ffn, vecs, tok, B, ntok, nc, vol, scale, wsp, tok_affine=aff, units=Cn, opts=opts)
        torch.cuda.synchronize()
        assert torch.equal(t, out)
    # a trajectory's rows do not depend on the launch they ride in (groups never straddle samples, the slab order is fixed)
    if B > 1:
        t = torch.full_like(xd[:1], float("nan"))
        L.attn_ffn_pair_split(xd[B - 1:].contiguous(), t, ws_full, ws_ffn, vecs, tok, 1, ntok, nc, vol, scale, wsp, tok_affine=tabs["affine"], units=Cn,
                              opts=opts)
        torch.cuda.synchronize()
        assert torch.equal(t[0], out[B - 1])


@pytest.mark.parametrize("operand", ["bf16", "fp16"])
@pytest.mark.parametrize("Cn,rows", [(256, 3328 * 2), (256, 57600), (256, 1000), (512, 832 * 2), (512, 14400 * 2), (512, 77)])
def test_ffn_rows_vs_oracle(Cn, rows, operand):
    """pd_ffn_rows -- PositionwiseFFN.forward (cuboid_transformer.py:182-208) on the pair kernel's FFN half alone (MODE 2 with one hidden slice),
    for blocks whose attention the pair kernel cannot take -- against the oracle's statement of the layer, against the launches it replaces
    (pd_ffn_fused at units 256), in place, twice (bit-equal), with row counts that are / are not multiples of the 16-row groups and of the
    64- / 128-row tiles (the last partial group is masked), and at the full-resolution row counts (57 600 / 2 x 14 400)."""
    from oracle import unet as OU
    from prediff_amd.packing import pack_pair_ffn_split, pack_pair_vecs
    opts = L.CallOpts(operand)
    odt, tol = opts.dtype, {"bf16": 6e-3, "fp16": 1e-3}[operand]
    Hd = 4 * Cn
    g = torch.Generator(device="cpu").manual_seed(Cn + rows)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {"layer_norm.weight": 1 + r(Cn, sc=.1), "layer_norm.bias": r(Cn, sc=.1), "ffn_1.weight": r(Hd, Cn, sc=Cn ** -0.5), "ffn_1.bias": r(Hd, sc=.1),
          "ffn_2.weight": r(Cn, Hd, sc=Hd ** -0.5), "ffn_2.bias": r(Cn, sc=.1)}
    x = r(rows, Cn)
    assert L.ffn_rows_supported(Cn, Hd) and not L.ffn_rows_supported(128, 512) and not L.ffn_rows_supported(256, 1024, act="leaky")
    y_ref = OU.positionwise_ffn(sd, "", x[None], "gelu")[0]
    d = lambda t: t.to(DEV)
    wf = pack_pair_ffn_split(d(sd["ffn_1.weight"]), d(sd["ffn_2.weight"]), dtype=odt, nsplit=1)
    zc = torch.zeros(Cn, device=DEV)
    vecs = pack_pair_vecs(zc, zc, None, d(sd["layer_norm.weight"]), d(sd["layer_norm.bias"]), d(sd["ffn_2.bias"]), d(sd["ffn_1.bias"]), torch.zeros(4, 16, 16, device=DEV))
    xd = d(x).contiguous()
    out = torch.full_like(xd, float("nan"))
    L.ffn_rows(xd, out, wf, vecs, rows, Cn, 1e-5, opts=opts)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()), "a row was not written (or written with garbage)"
    e = rel_l2((out - xd).cpu(), y_ref - x)
    print(f"[ffn_rows units {Cn} rows {rows} {operand}] update rel-L2 vs oracle {e:.3e}")
    assert e < tol and rel_l2(out.cpu(), y_ref) < tol
    for _ in range(2):                                   # in place, twice: bit-identical
        t = xd.clone()
        L.ffn_rows(t, t, wf, vecs, rows, Cn, 1e-5, opts=opts)
        torch.cuda.synchronize()
        assert torch.equal(t, out)
    # rows are independent: a prefix alone gives the same rows
    k = min(rows, 48)
    t = torch.full((k, Cn), float("nan"), device=DEV)
    L.ffn_rows(xd[:k].contiguous(), t, wf, vecs, k, Cn, 1e-5, opts=opts)
    torch.cuda.synchronize()
    assert torch.equal(t, out[:k])
    if Cn == 256:
        w1_p, _ = pack_linear(d(sd["ffn_1.weight"]), False, dtype=odt)
        w2_p, _ = pack_linear(d(sd["ffn_2.weight"]), False, dtype=odt)
        t = xd.clone()
        L.ffn_fused(t, t, d(sd["layer_norm.weight"]), d(sd["layer_norm.bias"]), w1_p, d(sd["ffn_1.bias"]), w2_p, d(sd["ffn_2.bias"]), rows, Cn, Hd, act="gelu", opts=opts)
        torch.cuda.synchronize()
        assert rel_l2(out - xd, t - xd) < 1.5e-3
    with pytest.raises(L.PrediffHipError):
        L.ffn_rows(xd, out, wf, vecs, rows, Cn, 1e-5, opts=L.CallOpts(operand, w_fold=1))


def test_attn_ffn_pair_rejects_what_it_does_not_run():
    assert not L.attn_ffn_pair_supported(128, 2, 512, 16)
    assert not L.attn_ffn_pair_supported(256, 4, 1024, 25)
    assert not L.attn_ffn_pair_supported(256, 4, 1024, 16, act="leaky")
    assert not L.attn_ffn_pair_supported(512, 4, 1024, 16) and not L.attn_ffn_pair_supported(512, 8, 2048, 16)
    assert L.attn_ffn_pair_supported(512, 4, 2048, 13)
    x = torch.zeros(1, 32, 256, device=DEV)
    with pytest.raises(L.PrediffHipError):
        L.attn_ffn_pair(x, x, x, x, None, 1, 32, 2, 16, 0.125)        # neither a token table nor its affine form
    with pytest.raises(L.PrediffHipError):
        L.attn_ffn_pair(x, x, x, x, x.int(), 1, 32, 2, 16, 0.125, units=384)


@pytest.mark.parametrize("Cn", [256, 512])
def test_attn_ffn_pair_stress_bit_equal(Cn):
    """60 launches of pd_attn_ffn_pair at the benchmark's occupancy (32 trajectories: 832 tiles, four per workgroup, the weight stream and
    the pipelined tile boundary running on across tiles) interleaved with a bandwidth-hungry copy on another stream: every result
    bit-equal to the first.  The kernel's counted `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` waits assume an exact, in-order instruction
    census; a miscount shows up as timing-dependent garbage in a few rows, which this would catch."""
    from prediff_amd.cuboid_geometry import attention_tables
    from prediff_amd.packing import pack_pair_block, pack_pair_vecs
    shape, cuboid = ((13, 16, 16), (1, 1, 16)) if Cn == 256 else ((13, 8, 8), (1, 1, 8))      # (units 512: 416 tiles, two per workgroup)
    B, heads, Hd = 32, 4, 4 * Cn
    ntok = shape[0] * shape[1] * shape[2]
    g = torch.Generator(device="cpu").manual_seed(77)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    x = r(B, ntok, Cn)
    ws = pack_pair_block(r(3 * Cn, Cn, sc=Cn ** -0.5), r(Cn, Cn, sc=Cn ** -0.5), r(Hd, Cn, sc=Cn ** -0.5), r(Cn, Hd, sc=Hd ** -0.5))
    tabs = attention_tables(shape, cuboid, (0, 0, 0), LLL, "zeros")
    vol = tabs["vol"]
    vecs = pack_pair_vecs(1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), 1 + r(Cn, sc=.1), r(Cn, sc=.1), r(Cn, sc=.1), r(Hd, sc=.1), r(4, vol, vol, sc=.5))
    tok = tabs["tok_index"].to(DEV)
    run = lambda out: L.attn_ffn_pair(x, out, ws, vecs, tok, B, ntok, tabs["nc"], vol, (Cn // 4) ** -0.5, tok_affine=tabs["affine"], units=Cn)
    ref = torch.empty_like(x)
    run(ref)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref).all())
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
    out = torch.empty_like(x)
    for it in range(60):
        if it % 2:
            with torch.cuda.stream(side):
                big_b.copy_(big_a)                       # 512 MB of HBM traffic beside the launch
        out.fill_(float("nan"))
        run(out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"launch {it}: {int((out != ref).sum())} elements differ"
