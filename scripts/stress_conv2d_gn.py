"""Stress of pd_conv2d_gn_silu at > 256 workgroups (two per CU): every launch against a torch reference, repeated launches bit-equal."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from prediff_amd import _lib as L
from prediff_amd.packing import pack_conv
DEV = "cuda"
for (N, H, W, Cin, Cout, G, res) in ((7, 128, 128, 128, 128, 32, False), (7, 128, 128, 128, 128, 32, True), (4, 64, 64, 256, 256, 32, True),
                                     (8, 32, 32, 512, 512, 32, True), (3, 64, 64, 128, 256, 32, False), (10, 128, 128, 128, 128, 32, True)):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.3).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(DEV)
    bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    r = torch.randn(N, H, W, Cout, generator=g).to(DEV) if res else None
    w_p, _ = pack_conv(w, False)
    S = H * W
    part = torch.empty(N * L.groupnorm_nchunk(S, Cin) * G * 2, dtype=torch.float64, device=DEV)
    stats = torch.empty(N, G, 2, device=DEV)
    L.groupnorm_stats(x, part, stats, N, S, Cin, G, 1e-6)
    act = F.silu(F.group_norm(x.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
    ref = F.conv2d(act.bfloat16().float(), w.bfloat16().float(), bias, padding=1).permute(0, 2, 3, 1).contiguous()
    if res:
        ref = ref + r
    outs, errs = [], []
    for k in range(16):
        out = torch.full((N, H, W, Cout), float("nan"), device=DEV)
        L.conv2d_gn_silu(x, stats, gamma, beta, w_p, bias, r, out, N, H, W, Cin, Cout, G)
        torch.cuda.synchronize()
        errs.append(float((out - ref).abs().max())); outs.append(out)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    grid = N * (H // 8) * (W // 16) * (Cout // 128)
    print(f"N{N} {H}x{W} {Cin}->{Cout} res={res} grid {grid}: 16 launches bit-equal {same}; max err vs ref {max(errs):.5f}")
