"""A/B of the fused level-0 kernels' engine switches (pd_fused_opts: bit 0 atomic in-place epilogue, bit 1 deep weight ring for small
grids, bit 2 arithmetic token ids) at the v1 level-0 shapes, in place as the denoiser calls them.  Run on the GPU box.
Interleaved rounds in ONE process (guide rule 24): median and min per variant."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prediff_amd import _lib as L
from prediff_amd.packing import pack_linear
from prediff_amd.cuboid_geometry import attention_tables

Cn, heads, shape = 256, 4, (13, 16, 16)
ntok = 13 * 16 * 16
VARIANTS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,4,7".split(","))]
ROUNDS, REPS = 7, 10


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for B in (4, 8, 32):
    for cuboid in ((13, 1, 1), (1, 16, 1)):
        tabs = attention_tables(shape, cuboid, (0, 0, 0), ("l", "l", "l"), "zeros")
        vol, nc = tabs["vol"], tabs["nc"]
        x = torch.randn(B, ntok, Cn, device="cuda")
        g, b = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
        wq, _ = pack_linear(torch.randn(3 * Cn, Cn, device="cuda") / 16, False)
        wp, _ = pack_linear(torch.randn(Cn, Cn, device="cuda") / 64, False)
        bp = torch.zeros(Cn, device="cuda")
        bias = torch.zeros(heads, vol, vol, device="cuda")
        tok = tabs["tok_index"].cuda()
        fn = lambda: L.attn_block_fused(x, x, g, b, wq, None, wp, bp, tok, bias, None, B, ntok, Cn, heads, nc, vol, 0.125, tok_affine=tabs["affine"])
        ts = {v: [] for v in VARIANTS}
        for r in range(ROUNDS):
            for v in VARIANTS:
                L.fused_opts(v)
                fn()
                ts[v].append(timed(fn))
        gf = B * (2 * ntok * 768 * 256 + 2 * ntok * 256 * 256 + 4 * ntok * vol * 256) / 1e9
        print(f"attn_block L0 B={B} cuboid {cuboid}: " + "  ".join(
            f"opts {v}: {statistics.median(t):.1f} us (min {min(t):.1f}, {gf * 1e3 / statistics.median(t):.0f} TF)" for v, t in ts.items()), flush=True)
    M, Hd = B * ntok, 1024
    x = torch.randn(M, Cn, device="cuda")
    g, b = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    w1, _ = pack_linear(torch.randn(Hd, Cn, device="cuda") / 16, False)
    w2, _ = pack_linear(torch.randn(Cn, Hd, device="cuda") / 128, False)
    b1, b2 = torch.zeros(Hd, device="cuda"), torch.zeros(Cn, device="cuda")
    fn = lambda: L.ffn_fused(x, x, g, b, w1, b1, w2, b2, M, Cn, Hd)
    ts = {v: [] for v in VARIANTS}
    for r in range(ROUNDS):
        for v in VARIANTS:
            L.fused_opts(v)
            fn()
            ts[v].append(timed(fn))
    print(f"ffn64 L0 B={B}: " + "  ".join(
        f"opts {v}: {statistics.median(t):.1f} us (min {min(t):.1f}, {4.0 * M * Cn * Hd / statistics.median(t) / 1e6:.0f} TF)" for v, t in ts.items()), flush=True)
